"""Round-5 fixtures, produced by RUNNING THE REFERENCE in the build container (needs /root/reference):

    python tests/golden/make_golden_r5.py [stack40|noqknorm_grads|pertoken_grads|vae_enc_grads|block14_pertoken|all]

* dit_stack40_14b.npz + bf16_calibration.json["stack40_14b"] — FORTY stacked 14B-width WanAttentionBlocks (the depth of the
  Wan2.1-14B DiT, wan_transformer4d.py:633-688 forty times, each layer with its own weights) at L = 2080 tokens: the reference's fp32
  output (sampled rows, row norms) and how far the reference's OWN bf16-autocast run of the same stack sits from it, in the metrics of
  tests/test_round5_gpu.py::test_stack_of_forty_14b_blocks_vs_reference.  L = 2080 keeps every projection on the production 256 x 256
  GEMM tiles and the self-attention on the production long-key kernel; intermediate depths (4, 10, 20, 40) are recorded so that
  error GROWTH over depth is visible, not only its end point.
  Weights: tests/golden/weights.py:fill_hash (device-agnostic integer-hash recipe — the GPU test builds the same bits on the device).
* dit_block_noqknorm_grads.npz — gradients of the qk_norm=False block of dit_block_noqknorm.npz (same weights / inputs) for a seeded
  cotangent, by torch autograd through the reference block (wan_transformer4d.py:431-432, 633-688): every parameter + dL/dx.
* dit_tiny_pertoken_grads.npz — training with PER-TOKEN timesteps (t [B, seq_len], wan_transformer4d.py:655-657, 713-715, 1161-1167): the
  reference's loss.backward() on the dit_tiny.npz inputs with the t_tok of dit_tiny_pertoken.npz; norms + sampled values of every gradient.
* vae_train_enc.npz — the train_vae.py step of vae_train.npz case B (gradient through the encoder) with the ENCODER TRAINABLE as well
  (train_vae.py:355 freezes it; a caller may not): gradients of every encoder parameter, incl. the stride-2 Resample convs and the strided
  time_conv (wan_vae.py:96-100, 108-110).
* dit_block_14b_pertoken.npz + bf16_calibration.json["block_14b_pertoken"] — one 14B-width block at L = 2080 with a PER-TOKEN modulation
  e [1, L, 6, 5120] (:655-657): the production GEMM's gated-residual epilogue with one gate row per token.
Data only; no reference source is stored."""
import json
import os
import sys
import time

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import _ref_import  # noqa: E402
from make_golden import npz_save  # noqa: E402
from make_golden_r4 import _block_metrics, _long_inputs, ref_bf16  # noqa: E402

torch.set_grad_enabled(False)
torch.set_num_threads(8)

DEPTHS = (4, 10, 20, 40)


def _block14_hash(ref, seed):
    from weights import block_shapes, fill_hash
    blk = ref.dit.WanAttentionBlock("i2v_cross_attn", 5120, 13824, 40, (-1, -1), True, True, 1e-6, use_spatial_guidance=False).eval()
    sd = fill_hash(block_shapes(5120, 13824), 500 + seed)
    blk.load_state_dict({k[len("blocks.0."):]: v for k, v in sd.items()})
    return blk


def make_stack40(ref):
    L, grid, x, e0, ctx, freqs = _long_inputs(ref)
    args = (torch.tensor([L]), torch.tensor([list(grid)]), freqs, ctx, None)
    ys32, ys16 = x, x
    rows = torch.cat([torch.arange(0, L, 32), torch.tensor([L - 1])])
    calib, arrs = {}, {}
    t0 = time.time()
    for layer in range(max(DEPTHS)):
        blk = _block14_hash(ref, layer)
        ys32 = blk(ys32, e0, *args, dtype=torch.float32, t=0, dino_features=None)
        with ref_bf16(ref):
            ys16 = blk(ys16, e0, *args, dtype=torch.bfloat16, t=0, dino_features=None).float()
        del blk
        if layer + 1 in DEPTHS:
            d = layer + 1
            calib[f"depth{d}"] = _block_metrics(ys16, ys32, x)
            if d == max(DEPTHS):      # sampled rows of the end point only (fixture size); intermediate depths keep the norms
                arrs[f"out_rows_{d}"] = ys32[0, rows].clone()
                arrs[f"delta_rows_{d}"] = (ys32 - x)[0, rows].clone()
            arrs[f"row_norm_{d}"] = ys32[0].norm(dim=-1)
            arrs[f"delta_norm_{d}"] = (ys32 - x)[0].norm(dim=-1)
            print(f"depth {d}: {calib[f'depth{d}']}  |x| rms {float(ys32.pow(2).mean().sqrt()):.3f}  ({time.time() - t0:.0f} s)", flush=True)
    npz_save("dit_stack40_14b.npz", grid=np.array(grid), rows=rows, depths=np.array(DEPTHS), **arrs)
    path = os.path.join(HERE, "bf16_calibration.json")
    out = json.load(open(path))
    out["stack40_14b"] = calib
    with open(path, "w") as fh:
        json.dump(out, fh, indent=1, sort_keys=True)


def make_noqknorm_grads(ref):
    d = ref.dit
    z = np.load(os.path.join(HERE, "dit_block_noqknorm.npz"))
    blk = d.WanAttentionBlock("i2v_cross_attn", 128, 512, 4, (-1, -1), False, True, 1e-6, use_spatial_guidance=False).eval()
    blk.load_state_dict({k[2:]: torch.from_numpy(z[k]) for k in z.files if k.startswith("w/")})
    x = torch.from_numpy(z["x"]).clone().requires_grad_(True)
    e0, ctx, grid = torch.from_numpy(z["e0"]), torch.from_numpy(z["ctx"]), tuple(int(v) for v in z["grid"])
    L, hd = x.shape[1], 32
    freqs = torch.cat([d.rope_params(1024, hd - 4 * (hd // 6)), d.rope_params(1024, 2 * (hd // 6)), d.rope_params(1024, 2 * (hd // 6))], dim=1)
    r = torch.randn(x.shape, generator=torch.Generator().manual_seed(77))
    with torch.enable_grad():
        y = blk(x, e0, torch.tensor([L]), torch.tensor([list(grid)]), freqs, ctx, None, dtype=torch.float32, t=0, dino_features=None)
        (y * r).sum().backward()
    arrs = {"cot": r, "grad/x": x.grad}
    for n, p_ in blk.named_parameters():
        arrs["grad/" + n] = p_.grad
    assert not any("norm_q" in k or "norm_k" in k for k in arrs), "qk_norm=False: no norm weights"
    npz_save("dit_block_noqknorm_grads.npz", **arrs)


def make_pertoken_grads(ref):
    from make_golden import TINY_DIT, grad_sample, load_recipe
    z = {k: torch.from_numpy(v) for k, v in dict(np.load(os.path.join(HERE, "dit_tiny.npz"))).items()}
    pz = np.load(os.path.join(HERE, "dit_tiny_pertoken.npz"))
    t_tok = torch.from_numpy(pz["t_tok"])
    with torch.enable_grad():
        m = ref.dit.WanTransformer4DModel(**TINY_DIT).train()
        load_recipe(m, None, seed=1234)
        target = torch.randn(z["out_ref"].shape, generator=torch.Generator().manual_seed(23))
        pred = m(x=z["x"], t=t_tok, context=[z["ctx0"], z["ctx1"]], seq_len=int(z["seq_len_pad"]), clip_fea=z["clip"], y=z["y"],
                 full_ref=z["full_ref"])
        diff = pred.float() - target
        loss = (torch.nn.functional.mse_loss(pred.float(), target, reduction="none") * (diff.abs() <= 50).float()).mean()
        loss.backward()
    assert float((pred.detach() - torch.from_numpy(pz["out_ref"])).abs().max()) < 1e-5
    out = {"target": target, "loss": loss.detach(), "pred": pred.detach()}
    for name, p_ in m.named_parameters():
        if p_.grad is None:
            continue
        out["norm/" + name] = p_.grad.norm()
        out["grad/" + name] = grad_sample(p_.grad)
    npz_save("dit_tiny_pertoken_grads.npz", **out)


def make_vae_enc_grads(ref):
    import types
    from make_golden import _reference_functions, grad_sample
    from weights import fill
    tv = _reference_functions(os.path.join(_ref_import.REF_ROOT, "scripts/4D_STraG_training/train_vae.py"), ["compute_loss"])
    args = types.SimpleNamespace(rec_loss="l1", kl_scale=1e-6)
    z = np.load(os.path.join(HERE, "vae_train.npz"))
    torch.set_grad_enabled(True)
    try:
        vae = ref.vae.AutoencoderKLWan()
        vae.load_state_dict(fill({k: list(t.shape) for k, t in vae.state_dict().items()}, seed=2024))
        ea, da = ref.traj.VAEEncoderadaptor(), ref.traj.VAEDecoderadaptor()
        ea.load_state_dict(fill({k: list(t.shape) for k, t in ea.state_dict().items()}, seed=78))
        da.load_state_dict(fill({k: list(t.shape) for k, t in da.state_dict().items()}, seed=77))
        ea.requires_grad_(True).train()
        da.requires_grad_(True).train()
        vae.model.encoder.requires_grad_(True).train()
        vae.model.decoder.requires_grad_(True).train()
        targets = torch.from_numpy(z["targets"])
        pseudo = ea(targets) * 2 - 1
        posterior = vae.encode_memory_saver(pseudo).latent_dist
        latents = posterior.sample(generator=torch.Generator().manual_seed(5))
        rec2 = da(vae.decode_memory_saver(latents).sample)
        loss, nll, kl = tv["compute_loss"](rec2, targets, posterior, args)
        loss.backward()
        assert abs(float(loss) - float(z["B/loss"])) < 1e-4 * float(z["B/loss"])
        out = {"loss": loss.detach()}
        for n, p_ in vae.model.encoder.named_parameters():
            assert p_.grad is not None, n
            out["grad/vae.model.encoder." + n] = grad_sample(p_.grad)
            out["norm/vae.model.encoder." + n] = p_.grad.norm()
        print("encoder parameters with gradients:", len(out) // 2)
        npz_save("vae_train_enc.npz", **out)
    finally:
        torch.set_grad_enabled(False)


def make_block14_pertoken(ref):
    from weights import randn_named
    L, grid, x, e0, ctx, freqs = _long_inputs(ref)
    e_tok = randn_named("in.e0tok", (1, L, 6, 5120), 6, 0.2)
    args = (torch.tensor([L]), torch.tensor([list(grid)]), freqs, ctx, None)
    blk = _block14_hash(ref, 0)
    y32 = blk(x, e_tok, *args, dtype=torch.float32, t=0, dino_features=None)
    with ref_bf16(ref):
        y16 = blk(x, e_tok, *args, dtype=torch.bfloat16, t=0, dino_features=None).float()
    rows = torch.cat([torch.arange(0, L, 32), torch.tensor([L - 1])])
    cal = _block_metrics(y16, y32, x)
    print("block_14b_pertoken", cal)
    npz_save("dit_block_14b_pertoken.npz", grid=np.array(grid), rows=rows, out_rows=y32[0, rows], row_norm=y32[0].norm(dim=-1),
             delta_rows=(y32 - x)[0, rows], delta_norm=(y32 - x)[0].norm(dim=-1))
    path = os.path.join(HERE, "bf16_calibration.json")
    out = json.load(open(path))
    out["block_14b_pertoken"] = cal
    with open(path, "w") as fh:
        json.dump(out, fh, indent=1, sort_keys=True)


if __name__ == "__main__":
    what = sys.argv[1] if len(sys.argv) > 1 else "all"
    ref = _ref_import.load_reference()
    if what in ("stack40", "all"):
        make_stack40(ref)
    if what in ("noqknorm_grads", "all"):
        make_noqknorm_grads(ref)
    if what in ("pertoken_grads", "all"):
        make_pertoken_grads(ref)
    if what in ("vae_enc_grads", "all"):
        make_vae_enc_grads(ref)
    if what in ("block14_pertoken", "all"):
        make_block14_pertoken(ref)
