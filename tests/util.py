"""Shared helpers for the test-suite (fixtures loading, error metrics)."""
import json
import os

import numpy as np
import torch

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def load_npz(name):
    z = np.load(os.path.join(GOLDEN, name))
    return {k: torch.from_numpy(z[k]) for k in z.files}


def load_keys(name):
    with open(os.path.join(GOLDEN, name)) as fh:
        return {k: tuple(v) for k, v in json.load(fh).items()}


_CALIB = None


def bf16_budget(*path, factor=1.5):
    """Budget of a bf16 GPU test = `factor` x the error of the REFERENCE'S OWN bf16-autocast run against its fp32 run in the same
    metric (tests/golden/bf16_calibration.json, made by tests/golden/make_golden_r4.py by running the reference under
    torch.autocast("cpu", bfloat16)).  path: keys into that file, e.g. ("vae_probe_120x208", "decode", "rms")."""
    global _CALIB
    if _CALIB is None:
        with open(os.path.join(GOLDEN, "bf16_calibration.json")) as fh:
            _CALIB = json.load(fh)
    v = _CALIB
    for k in path:
        v = v[k]
    return factor * float(v)


def rel_err(a, b):
    """max |a-b| / max |b|  (the '1e-3 relative fp32' metric of BASELINE.json)."""
    a = a.double()
    b = b.double()
    return float((a - b).abs().max() / b.abs().max().clamp_min(1e-30))


def rms_rel_err(a, b):
    a = a.double()
    b = b.double()
    return float(((a - b).pow(2).mean().sqrt()) / b.pow(2).mean().sqrt().clamp_min(1e-30))


def grad_sample(g, n=4096):
    """Same deterministic subsample as tests/golden/make_golden.py:grad_sample."""
    flat = g.detach().reshape(-1)
    k = max(1, flat.numel() // n)
    return flat[::k][:n].clone()


def custom_mse_loss(pred, target, threshold=50.0):
    """train_wan.py:1953-1963 with unit weighting."""
    import torch.nn.functional as F
    diff = pred.float() - target.float()
    return (F.mse_loss(pred.float(), target.float(), reduction="none") * (diff.abs() <= threshold).float()).mean()


def check_grads(named_grads, z, tol, norm_tol=None):
    """named_grads: {param name: grad}; z: dit_tiny_grads.npz.  Every parameter the reference trained must match:
    error of the subsampled values relative to max|ref| of that tensor, and the gradient norm.  Gradients that are
    pure cancellation residue (e.g. the key bias of an attention without key norm: analytically zero) are measured
    against 1e-3 of the largest gradient in the model instead of their own ~0 magnitude."""
    worst = ("", 0.0)
    names = [k[5:] for k in z if k.startswith("grad/")]
    assert names
    gfloor = 1e-3 * max(float(z["grad/" + n].abs().max()) for n in names)
    nfloor = 1e-3 * max(float(z["norm/" + n]) for n in names)
    for name in names:
        assert name in named_grads and named_grads[name] is not None, f"missing gradient for {name}"
        g = named_grads[name].detach().float().cpu()
        ref = z["grad/" + name]
        e = float((grad_sample(g).double() - ref.double()).abs().max() / max(float(ref.abs().max()), gfloor))
        ne = abs(float(g.norm()) - float(z["norm/" + name])) / max(float(z["norm/" + name]), nfloor)
        if max(e, ne) > worst[1]:
            worst = (name, max(e, ne))
        assert e < tol, f"{name}: gradient rel err {e:.3e}"
        assert ne < (norm_tol or tol), f"{name}: gradient norm rel err {ne:.3e}"
    return worst


def same_grads(named_a, named_b, tol=1e-4):
    """Two gradient sets from the same kernels in different schedules: equal up to atomic-order noise, measured against
    each tensor's own magnitude or 1e-3 of the largest gradient (cancellation-dominated tensors)."""
    gmax = max(float(v.abs().max()) for v in named_b.values() if v is not None)
    for n, b in named_b.items():
        if b is None:
            assert named_a[n] is None, n
            continue
        err = float((named_a[n].double().cpu() - b.double().cpu()).abs().max() / max(float(b.abs().max()), 1e-3 * gmax))
        assert err < tol, f"{n}: {err:.3e}"
