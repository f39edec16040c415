"""GPU parity of the Motion-Sensitive VAE path: every VAE kernel against the torch stand-ins of tests/cpu_ops.py
(themselves pinned to the reference by tests/test_vae_host_logic.py), then the full encode/decode and the two
adaptors against the fixtures produced by the reference itself."""
import pytest
import torch

import cpu_ops
from util import load_keys, load_npz, rel_err, rms_rel_err
from weights import fill

pytestmark = pytest.mark.gpu
DEV = "cuda"


def rnd(*shape, seed=0, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    return torch.randn(*shape, generator=g) * scale


CONV_CASES = [
    # name, Tin,Hin,Win,Cin,Cout, k, stride, pad, out_thw, ups, tsplit
    ("c333_tail", 6, 9, 11, 16, 24, (3, 3, 3), (1, 1, 1), (0, 1, 1), (4, 9, 11), False, False),
    ("c333_cold", 3, 8, 8, 8, 96, (3, 3, 3), (1, 1, 1), (2, 1, 1), (3, 8, 8), False, False),
    ("c133", 2, 10, 12, 96, 96, (1, 3, 3), (1, 1, 1), (0, 1, 1), (2, 10, 12), False, False),
    ("down2d", 2, 10, 12, 32, 32, (1, 3, 3), (1, 2, 2), (0, 0, 0), (2, 5, 6), False, False),
    ("up2d", 2, 5, 6, 64, 32, (1, 3, 3), (1, 1, 1), (0, 1, 1), (2, 10, 12), True, False),
    ("up3d_tsplit", 2, 5, 6, 32, 16, (1, 3, 3), (1, 1, 1), (0, 1, 1), (4, 10, 12), True, True),
    ("time_s2", 5, 6, 7, 48, 48, (3, 1, 1), (2, 1, 1), (0, 0, 0), (2, 6, 7), False, False),
    ("c111_big", 1, 20, 30, 384, 192, (1, 1, 1), (1, 1, 1), (0, 0, 0), (1, 20, 30), False, False),
    ("cout4", 2, 9, 9, 96, 4, (3, 3, 3), (1, 1, 1), (2, 1, 1), (2, 9, 9), False, False),
]


@pytest.mark.parametrize("dt", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("case", CONV_CASES, ids=[c[0] for c in CONV_CASES])
def test_conv_cl(dt, case):
    from more4d_amd import ops
    _, Tin, Hin, Win, Cin, Cout, k, stride, pad, out_thw, ups, tsplit = case
    cw = Cin * (2 if tsplit else 1)
    x = rnd(Tin, Hin, Win, cw, seed=1).to(dt)
    K = k[0] * k[1] * k[2] * Cin
    w = rnd(Cout, K, seed=2, scale=K ** -0.5).to(dt)
    b = rnd(Cout, seed=3).to(dt)
    M = out_thw[0] * out_thw[1] * out_thw[2]
    r = rnd(M, Cout, seed=4).to(dt)
    kw = dict(Tin=Tin, Hin=Hin, Win=Win, Cin=Cin, k=k, stride=stride, pad=pad, out_thw=out_thw, ups=ups, tsplit=tsplit)
    tol = 1e-4 if dt == torch.float32 else 2.5e-2
    ref = cpu_ops.conv_cl(x.float(), w.float(), b.float(), **kw)
    out = ops.conv_cl(x.to(DEV), w.to(DEV), b.to(DEV), **kw)
    assert rel_err(out.float().cpu(), ref) < tol
    ref = cpu_ops.conv_cl(x.float(), w.float(), None, resid=r.float(), **kw)
    out = ops.conv_cl(x.to(DEV), w.to(DEV), None, resid=r.to(DEV), **kw)
    assert rel_err(out.float().cpu(), ref) < tol


@pytest.mark.parametrize("dt", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("C", [16, 96, 192, 384])
def test_rmsnorm_silu_cl(dt, C):
    from more4d_amd import ops
    x = rnd(301, C, seed=1, scale=2.0).to(dt)
    g = rnd(C, seed=2) * 0.1 + 1
    tol = 2e-5 if dt == torch.float32 else 1.5e-2
    for silu in (True, False):
        ref = cpu_ops.rmsnorm_silu_cl(x.float(), g, silu=silu)
        out = ops.rmsnorm_silu_cl(x.to(DEV), g.to(DEV), silu=silu)
        assert rel_err(out.float().cpu(), ref) < tol
    wide = torch.zeros(301, C + 16, device=DEV, dtype=dt)
    ops.rmsnorm_silu_cl(x.to(DEV), g.to(DEV), out=wide[:, 8:8 + C])
    assert rel_err(wide[:, 8:8 + C].float().cpu(), cpu_ops.rmsnorm_silu_cl(x.float(), g)) < tol


@pytest.mark.parametrize("dt", [torch.float32, torch.bfloat16])
def test_groupnorm_softmax_layouts(dt):
    from more4d_amd import ops
    F, HW, C = 3, 5000, 128
    x = (rnd(F, HW, C, seed=1) * 1.5 + 0.3).to(dt)
    w, b = rnd(C, seed=2) * 0.1 + 1, rnd(C, seed=3) * 0.1
    ref = cpu_ops.groupnorm_cl(x.float(), w, b, F=F, HW=HW)
    out = ops.groupnorm_cl(x.to(DEV), w.to(DEV), b.to(DEV), F=F, HW=HW)
    assert rel_err(out.float().cpu(), ref) < (2e-5 if dt == torch.float32 else 2e-2)
    out2 = ops.groupnorm_cl(x.to(DEV), w.to(DEV), b.to(DEV), F=F, HW=HW)
    assert torch.equal(out, out2)      # deterministic two-pass reduction
    s = rnd(70, 77, seed=4, scale=3.0)
    p = ops.softmax_rows(s.to(DEV), dt, C=75, Cpad=80, scale=0.3)
    refp = cpu_ops.softmax_rows(s, torch.float32, C=75, Cpad=80, scale=0.3)
    assert rel_err(p.float().cpu(), refp) < (1e-5 if dt == torch.float32 else 8e-3)
    v = rnd(3, 4, 6, 10, seed=5)
    cs, sh = rnd(3, seed=6), rnd(3, seed=7)
    a = ops.ncthw_to_cl(v.to(DEV), dt, Cp=8, scale=2.0, shift=-1.0, ch_scale=cs.to(DEV), ch_shift=sh.to(DEV))
    refa = cpu_ops.ncthw_to_cl(v, torch.float32, Cp=8, scale=2.0, shift=-1.0, ch_scale=cs, ch_shift=sh)
    assert rel_err(a.float().cpu(), refa) < (1e-6 if dt == torch.float32 else 8e-3)
    for act, aux in ((0, None), (1, None), (2, rnd(3, 4, 6, 10, seed=8).to(dt))):
        o = ops.cl_to_ncthw(a, dt, C=3, T=4, H=6, W=10, pixel_stride=8, ch_scale=cs.to(DEV), ch_shift=sh.to(DEV), act=act,
                            aux=None if aux is None else aux.to(DEV))
        refo = cpu_ops.cl_to_ncthw(a.float().cpu(), torch.float32, C=3, T=4, H=6, W=10, pixel_stride=8, ch_scale=cs,
                                   ch_shift=sh, act=act, aux=None if aux is None else aux.float())
        assert rel_err(o.float().cpu(), refo) < (1e-5 if dt == torch.float32 else 1e-2)


def make_vae(dtype):
    from more4d_amd.models.wan_vae import AutoencoderKLWan
    vae = AutoencoderKLWan().eval()
    vae.load_state_dict(fill(load_keys("vae_keys.json"), 2024))
    return vae.to(DEV, dtype)


def test_vae_roundtrip_fp32():
    """Full-size network on [1,3,9,32,32] (3 chunks) vs the reference's own encode / decode outputs."""
    z = load_npz("vae_roundtrip.npz")
    vae = make_vae(torch.float32)
    with torch.no_grad():
        enc = vae._encode(z["x"].to(DEV))
        assert rel_err(enc.cpu(), z["enc"]) < 1e-3
        dec = vae.decode(z["enc"][:, :16].to(DEV)).sample
        assert rel_err(dec.cpu(), z["dec"]) < 1e-3
        assert torch.equal(vae.encode(z["x"].to(DEV))[0].mode(), enc[:, :16])


def test_vae_roundtrip_bf16_budget():
    z = load_npz("vae_roundtrip.npz")
    vae = make_vae(torch.bfloat16)
    with torch.no_grad():
        enc = vae._encode(z["x"].to(DEV, torch.bfloat16))
        dec = vae.decode(z["enc"][:, :16].to(DEV, torch.bfloat16)).sample
    assert rms_rel_err(enc.float().cpu(), z["enc"]) < 4e-2
    assert rms_rel_err(dec.float().cpu(), z["dec"]) < 6e-2


@pytest.mark.parametrize("which", ["enc", "dec"])
def test_adaptors_fp32(which):
    from more4d_amd.models.trajectory_module import VAEDecoderadaptor, VAEEncoderadaptor
    z = load_npz(f"adaptor_{which}.npz")
    m = (VAEEncoderadaptor if which == "enc" else VAEDecoderadaptor)().eval()
    m.load_state_dict({k[3:]: v for k, v in z.items() if k.startswith("sd.")}, strict=True)
    m = m.to(DEV)
    with torch.no_grad():
        out = m(z["x"].to(DEV))
    assert rel_err(out.cpu(), z["out"]) < 1e-3


@pytest.mark.parametrize("which", ["enc", "dec"])
def test_adaptor_planar_groups_equal_channels_last_bf16(which):
    """bf16 adaptors on a 40x64 map: GroupNorm writing planar-16 frame groups + m4d_conv_cl_planar (default at >= 1024 pixels; here with
    the group size forced down to 2 and 1 frames so that several groups and a short last group occur) give the same bits as the
    channels-last GroupNorm + m4d_conv_cl pair."""
    from more4d_amd.models import trajectory_module
    from more4d_amd.models.trajectory_module import VAEDecoderadaptor, VAEEncoderadaptor
    z = load_npz(f"adaptor_{which}.npz")
    m = (VAEEncoderadaptor if which == "enc" else VAEDecoderadaptor)().eval()
    m.load_state_dict({k[3:]: v for k, v in z.items() if k.startswith("sd.")}, strict=True)
    m = m.to(DEV, torch.bfloat16)
    x = (torch.randn(1, 3, 5, 40, 64, generator=torch.Generator().manual_seed(3)) * 0.5).to(DEV, torch.bfloat16)
    base = trajectory_module._AdaptorBase
    saved = base.PLANAR, base.PLANAR_MAX_BYTES
    try:
        base.PLANAR = False
        with torch.no_grad():
            ref = m(x)
        outs = []
        for limit in ((1 << 31) - (1 << 20), 2 * 8 * 40 * 64 * 32, 1):      # one group of 5 | groups of 2, 2, 1 | 5 groups of 1
            base.PLANAR, base.PLANAR_MAX_BYTES = True, limit
            with torch.no_grad():
                outs.append(m(x))
    finally:
        base.PLANAR, base.PLANAR_MAX_BYTES = saved
    assert torch.isfinite(ref.float()).all()
    # the planar path takes the GroupNorm statistics from the producing conv's epilogue (per-patch sums, another summation order than
    # the statistics kernel's): mean / rstd agree to fp32 rounding, the bf16 results to one ulp (on ~10 % of the elements after six layers)
    for o in outs:
        d = (o.float() - ref.float()).abs()
        assert float(d.max()) <= 2.0 ** -7 * float(ref.float().abs().max()) and float((d > 0).float().mean()) < 0.25
    assert torch.equal(outs[0], outs[1])        # the frame grouping itself changes nothing (at limit 1 conv_in runs channels-last: its
    #                                             statistics come from the statistics kernel again, covered by the bound above)


def test_conv_gnstats_are_the_group_sums_of_the_result():
    """m4d_conv_cl_planar_gnstats: the per-patch (sum, sum of squares) rows add up to the per-frame GroupNorm(32 x 4) sums of the stored
    bf16 result, and m4d_groupnorm_cl_planar_apply on them equals the two-pass GroupNorm to fp32 rounding."""
    from more4d_amd import ops as o
    F, H, W, C = 3, 40, 72, 128
    g = torch.Generator().manual_seed(11)
    x = torch.randn(F * H * W, C, generator=g).bfloat16().to(DEV)
    w = (torch.randn(C, 9 * C, generator=g) * (9 * C) ** -0.5).bfloat16().to(DEV)
    b = torch.randn(C, generator=g).bfloat16().to(DEV)
    res = torch.randn(F * H * W, C, generator=g).bfloat16().to(DEV)
    gw, gb = (1 + 0.1 * torch.randn(C, generator=g)).to(DEV), (0.1 * torch.randn(C, generator=g)).to(DEV)
    xin = o.groupnorm_cl_planar(x.view(F, H * W, C), gw, gb, F=F, HW=H * W, frames_per_group=F)[0]
    st = torch.full((F, o.gnstats_blocks(H, W), 32, 2), float("nan"), device=DEV)
    y = o.conv_cl_planar(xin, w, b, Tin=F, Hin=H, Win=W, kt=1, resid=res, gn_stats=st)
    assert torch.equal(y, o.conv_cl_planar(xin, w, b, Tin=F, Hin=H, Win=W, kt=1, resid=res))
    v = y.float().view(F, H * W, 32, 4)
    tot = st.sum(dim=1)
    assert torch.allclose(tot[..., 0], v.sum(dim=(1, 3)), rtol=1e-4, atol=0.05)
    assert torch.allclose(tot[..., 1], (v * v).sum(dim=(1, 3)), rtol=1e-4, atol=0.05)
    a = o.groupnorm_cl_planar(y.view(F, H * W, C), gw, gb, F=F, HW=H * W, frames_per_group=2, stats=st)
    r = o.groupnorm_cl_planar(y.view(F, H * W, C), gw, gb, F=F, HW=H * W, frames_per_group=2)
    for pa, pr in zip(a, r):
        d = (pa.t.float() - pr.t.float()).abs()
        assert float(d.max()) <= 2.0 ** -7 * float(pr.t.float().abs().max()) and float((d > 0).float().mean()) < 0.02
