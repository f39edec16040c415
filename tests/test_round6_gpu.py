"""Round-6 GPU parity (VERDICT r5 "next round" item 1): the one-wave-per-SIMD attention backward.

* attn_bwd_dq64_kernel + attn_bwd_kv64_kernel (csrc/attention_bwd64.h, generated streams of tools/gen_attn_bwd64*.py) against fp32
  autograd on the bf16-rounded operands — ragged query / key tails, NaN-poisoned padding rows, strided q / k / v slices of one buffer,
  score spikes, zero rows in the key padding, bit-identical reruns — with the launch counters proving that the new kernels ran;
* the same inputs through the two-waves-per-SIMD kernels of rounds 3-4 (M4D_ATTN_BWD64=0) in a child process;
* the training shape L = 21 840 on sampled queries / keys against fp32 formulas over the FULL other axis.
The kernels take the softmax scale folded into q (scale = ln 2: models/wan_transformer4d.py:_FOLD_QSCALE), which is how the training
step calls them (autograd.py: block_backward); any other scale stays on the older kernels (covered by test_train_gpu.py)."""
import math
import os
import subprocess
import sys

import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda"
BF = torch.bfloat16
D = 128
LN2 = math.log(2.0)
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _inputs(B, n, Lq, Lk, seed, spikes, strided):
    """q is drawn at 1/sqrt(d) / ln 2 so that q k^T * ln 2 has the usual spread; rows of k / v beyond Lk are NaN (never consumed)"""
    C = n * D
    g = torch.Generator().manual_seed(seed)
    qq = torch.randn(B, Lq, n, D, generator=g) * (D ** -0.5 / LN2)
    kk = torch.randn(B, Lk, n, D, generator=g)
    vv = torch.randn(B, Lk, n, D, generator=g)
    dd = torch.randn(B, Lq, n, D, generator=g)
    if spikes:
        kk[0, Lk // 2 + 3, 0] = qq[0, 5, 0] * 60.0
        kk[0, Lk - 1, n - 1] = qq[0, Lq - 1, n - 1] * 50.0
    Lkp = (Lk + 7) // 8 * 8
    if strided:
        assert Lq == Lkp
        buf = torch.full((B * Lq, 3 * C), float("nan"), dtype=BF)
        buf[:, :C] = qq.reshape(B * Lq, C).to(BF)
        buf.view(B, Lq, 3 * C)[:, :Lk, C:2 * C] = kk.reshape(B, Lk, C).to(BF)
        buf.view(B, Lq, 3 * C)[:, :Lk, 2 * C:] = vv.reshape(B, Lk, C).to(BF)
        buf = buf.to(DEV)
        q, k, v = buf[:, :C], buf[:, C:2 * C], buf[:, 2 * C:]
    else:
        q = qq.reshape(B * Lq, C).to(BF).to(DEV)
        k = torch.full((B, Lkp, C), float("nan"), dtype=BF)
        v = torch.full((B, Lkp, C), float("nan"), dtype=BF)
        k[:, :Lk] = kk.reshape(B, Lk, C).to(BF)
        v[:, :Lk] = vv.reshape(B, Lk, C).to(BF)
        k, v = k.reshape(B * Lkp, C).to(DEV), v.reshape(B * Lkp, C).to(DEV)
    d_o = dd.reshape(B * Lq, C).to(BF).to(DEV)
    return q, k, v, d_o, Lkp


def _forward(q, k, v, B, n, Lq, Lk, Lkp):
    from more4d_amd import ops
    C = n * D
    vt = torch.nan_to_num(v.float()).to(BF).t().contiguous()
    lse = torch.empty(B, n, Lq, device=DEV)
    o = ops.attention(q, [ops.KV(k, vt, Lkp * k.stride(0), k.stride(0), Lkp, B * Lkp, Lk)], B=B, Lq=Lq, heads=n, head_dim=D,
                      q_bs=Lq * q.stride(0), q_ls=q.stride(0), lse=lse, scale=LN2).view(B * Lq, C)
    return o, lse


def _reference(q, k, v, d_o, o, B, n, Lq, Lk, Lkp):
    """fp32 gradients of softmax(q k^T ln 2) v on the bf16-rounded operands — and the same formulas with the bf16 roundings the kernels
    make (P and dS rounded to bf16 in front of their MFMAs, delta from the bf16 forward output, bf16 results): the calibration of the
    budget, as for every bf16 bound of this suite (tests/util.py: 1.5 x the error of the rounded model)"""
    C = n * D
    perm = lambda x, L: x.float().reshape(B, L, n, D).permute(0, 2, 1, 3)
    qf, kf, vf = perm(q, Lq), perm(k, Lkp)[:, :, :Lk], perm(v, Lkp)[:, :, :Lk]
    gf, of = perm(d_o, Lq), perm(o, Lq)
    s = (qf @ kf.transpose(-1, -2)) * LN2
    P = torch.softmax(s, -1)
    G = gf @ vf.transpose(-1, -2)
    r = lambda x: x.to(BF).float()

    def grads(P_mm, dS_q, dS_k):
        return r(dS_q @ kf * LN2), r(dS_k.transpose(-1, -2) @ qf * LN2), r(P_mm.transpose(-1, -2) @ gf)
    dS = P * (G - (gf * (P @ vf)).sum(-1, keepdim=True))
    exact = (dS @ kf * LN2, dS.transpose(-1, -2) @ qf * LN2, P.transpose(-1, -2) @ gf)
    dlt = (gf * of).sum(-1, keepdim=True)
    model = grads(r(P), r(P * (G - dlt)), r(r(P) * (G - dlt)))
    back = lambda x, L: x.permute(0, 2, 1, 3).reshape(B, L, C)
    return [back(x, L) for x, L in zip(exact, (Lq, Lk, Lk))], [back(x, L) for x, L in zip(model, (Lq, Lk, Lk))]


CASES = {
    "strided_qkv_2304": (1, 8, 2304, 2304, 0, False, True),        # the training layout: column slices of one [rows, 3C] buffer
    "ragged_q52_k60_pad4": (2, 3, 2100, 2300, 1, True, False),     # 2100 = 32 * 64 + 52 queries, 2300 = 35 * 64 + 60 keys, 4 padding rows
    "ragged_q4_k3_spikes": (1, 16, 4100, 4099, 2, True, False),    # one-row tails, keys 4099 of 4104 rows
    "key_tile_boundary": (1, 4, 2048, 2176, 3, False, False),      # 2176 = 17 * 128: every key workgroup full
}


@pytest.mark.parametrize("name", list(CASES))
def test_attention_bwd64_kernels_vs_fp32(name):
    """dQ, dK, dV of the one-wave-per-SIMD passes against the fp32 gradient formulas on the same bf16 operands: max and rms error within
    1.5 x the error of the same formulas with the kernels' bf16 roundings (P, dS, delta from the bf16 output, bf16 results); padding rows
    of dK / dV exactly zero; a second launch reproduces every bit (no atomics, fixed summation order)."""
    from more4d_amd import ops
    B, n, Lq, Lk, seed, spikes, strided = CASES[name]
    C = n * D
    q, k, v, d_o, Lkp = _inputs(B, n, Lq, Lk, seed, spikes, strided)
    o, lse = _forward(q, k, v, B, n, Lq, Lk, Lkp)
    outs = []
    for rep in range(2):
        if strided:
            g = torch.full((B * Lq, 3 * C), float("nan"), dtype=BF, device=DEV)
            dq, dk, dv = g[:, :C], g[:, C:2 * C], g[:, 2 * C:]
        else:
            dq, dk, dv = (torch.full_like(x, float("nan")) for x in (q, k, v))
        ops.launch_counts(reset=True)
        ops.attention_bwd(q, k, v, o, d_o, lse, B=B, Lq=Lq, Lk=Lk, Lk_rows=Lkp, heads=n, head_dim=D, dq=dq, dk=dk, dv=dv, scale=LN2)
        torch.cuda.synchronize()
        cnt = ops.launch_counts()
        assert cnt["attn_bwd64"] == 2 and cnt["attn_bwd128"] == 0 and cnt["attn_bwd_generic"] == 0, cnt
        outs.append((dq.clone(), dk.clone(), dv.clone()))
    for a, b in zip(*outs):
        assert torch.equal(a.view(torch.int16), b.view(torch.int16))
    dq, dk, dv = outs[0]
    exact, model = _reference(q, torch.nan_to_num(k.float()), torch.nan_to_num(v.float()), d_o, o, B, n, Lq, Lk, Lkp)
    dkv, dvv = dk.reshape(B, Lkp, C), dv.reshape(B, Lkp, C)

    def errs(a, b):
        return float((a - b).abs().max() / b.abs().max()), float((a - b).pow(2).mean().sqrt() / b.pow(2).mean().sqrt())
    for nm, got, want, mod in zip(("dq", "dk", "dv"), (dq.reshape(B, Lq, C), dkv[:, :Lk], dvv[:, :Lk]), exact, model):
        got = got.float()
        assert bool(torch.isfinite(got).all()), nm
        (e_max, e_rms), (m_max, m_rms) = errs(got, want), errs(mod, want)
        print(f"{name} {nm}: max {e_max:.3e} rms {e_rms:.3e}   rounded model: max {m_max:.3e} rms {m_rms:.3e}")
        assert e_max < 1.5 * m_max + 1e-3 and e_rms < 1.5 * m_rms + 2e-4, (nm, e_max, e_rms, m_max, m_rms)
    if Lkp > Lk:
        assert float(dkv[:, Lk:].float().abs().sum()) == 0.0 and float(dvv[:, Lk:].float().abs().sum()) == 0.0


def test_attention_bwd64_other_scales_stay_on_the_two_wave_kernels():
    """scale != ln 2 (no folded softmax scale): the forward that produced lse did not round Q * sc to bf16, so the backward may not
    either — such calls keep the rounds 3-4 kernels"""
    from more4d_amd import ops
    B, n, Lq, Lk = 1, 2, 2048, 2048
    C = n * D
    g = torch.Generator(device=DEV).manual_seed(0)
    q, k, v, d_o = ((torch.randn(B * Lq, C, device=DEV, generator=g)).to(BF) for _ in range(4))
    lse = torch.empty(B, n, Lq, device=DEV)
    o = ops.attention(q, [ops.KV(k, ops.transpose(v), Lk * C, C, Lk, B * Lk, Lk)], B=B, Lq=Lq, heads=n, head_dim=D, q_bs=Lq * C, q_ls=C,
                      lse=lse).view(B * Lq, C)
    dq, dk, dv = (torch.empty_like(x) for x in (q, k, v))
    ops.launch_counts(reset=True)
    ops.attention_bwd(q, k, v, o, d_o, lse, B=B, Lq=Lq, Lk=Lk, Lk_rows=Lk, heads=n, head_dim=D, dq=dq, dk=dk, dv=dv)
    cnt = ops.launch_counts()
    assert cnt["attn_bwd64"] == 0 and cnt["attn_bwd128"] == 2, cnt


def test_attention_bwd64_vs_two_wave_kernels_same_inputs():
    """A/B in child processes (the switch is read once per process): the same inputs through attn_bwd_dqp / attn_bwd_kvp (M4D_ATTN_BWD64=0)
    and through the one-wave-per-SIMD kernels (default): sums and sampled entries agree to the bf16 rounding of dS and of the outputs."""
    code = r"""
import sys, math, torch
sys.path.insert(0, %r)
from more4d_amd import ops
g = torch.Generator(device="cuda").manual_seed(5)
B, n, L = 1, 8, 4160 + 24
C = n * 128
qkv = torch.randn(B * L, 3 * C, device="cuda", generator=g)
qkv[:, :C] *= 128 ** -0.5 / math.log(2.0)
qkv = qkv.bfloat16()
q, k, v = qkv[:, :C], qkv[:, C:2 * C], qkv[:, 2 * C:]
d_o = torch.randn(B * L, C, device="cuda", generator=g).bfloat16()
lse = torch.empty(B, n, L, device="cuda")
o = ops.attention(q, [ops.KV(k, ops.transpose(v.contiguous()), L * 3 * C, 3 * C, L, B * L, L)], B=B, Lq=L, heads=n, head_dim=128, q_bs=L * 3 * C,
                  q_ls=3 * C, lse=lse, scale=math.log(2.0)).view(B * L, C)
g3 = torch.empty_like(qkv)
ops.launch_counts(reset=True)
ops.attention_bwd(q, k, v, o, d_o, lse, B=B, Lq=L, Lk=L, Lk_rows=L, heads=n, head_dim=128, dq=g3[:, :C], dk=g3[:, C:2 * C], dv=g3[:, 2 * C:],
                  scale=math.log(2.0))
torch.cuda.synchronize()
c = ops.launch_counts()
f = g3.float()
print("RES", c["attn_bwd64"], c["attn_bwd128"], *[float(f[:, j * C:(j + 1) * C].abs().sum()) for j in range(3)],
      *[float(f[r, cc]) for r, cc in ((77, 5), (4183, 1000), (13, C + 300), (4100, 2 * C + 7), (4183, 3 * C - 1))])
""" % ROOT
    res = {}
    for mode in ("0", "3"):
        r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, env={**os.environ, "M4D_ATTN_BWD64": mode}, timeout=300)
        line = [l for l in r.stdout.splitlines() if l.startswith("RES")]
        assert line, r.stdout + r.stderr
        res[mode] = [float(x) for x in line[0].split()[1:]]
    assert res["0"][:2] == [0.0, 2.0] and res["3"][:2] == [2.0, 0.0], res
    for i in (2, 3, 4):
        assert abs(res["0"][i] - res["3"][i]) < 2e-3 * abs(res["0"][i]), res
    for i in range(5, 10):
        assert abs(res["0"][i] - res["3"][i]) < 3e-2 * max(abs(res["0"][i]), 0.02), res


def test_attention_bwd64_full_length_sampled():
    """The training shape (L = 21 840 = 341 * 64 + 16, B = 1, 8 of the 40 heads to bound memory; q / k / v strided as in block_backward):
    dQ on sampled queries, dK / dV on sampled keys against fp32 formulas over the FULL other axis (P from the forward's log-sum-exp)."""
    from more4d_amd import ops
    L, heads = 21840, 8
    Cc = heads * D
    g = torch.Generator(device=DEV).manual_seed(11)
    qkv = torch.randn(L, 3 * Cc, generator=g, device=DEV)
    qkv[:, :Cc] *= D ** -0.5 / LN2
    qkv = qkv.to(BF)
    q, k, v = qkv[:, :Cc], qkv[:, Cc:2 * Cc], qkv[:, 2 * Cc:]
    d_o = (torch.randn(L, Cc, generator=g, device=DEV) * 0.1).to(BF)
    lse = torch.empty(1, heads, L, device=DEV)
    o = ops.attention(q, [ops.KV(k, ops.transpose(v.contiguous()), L * 3 * Cc, 3 * Cc, L, L, L)], B=1, Lq=L, heads=heads, head_dim=D,
                      q_bs=L * 3 * Cc, q_ls=3 * Cc, lse=lse, scale=LN2).view(L, Cc)
    g3 = torch.empty_like(qkv)
    dq, dk, dv = g3[:, :Cc], g3[:, Cc:2 * Cc], g3[:, 2 * Cc:]
    ops.launch_counts(reset=True)
    ops.attention_bwd(q, k, v, o, d_o, lse, B=1, Lq=L, Lk=L, Lk_rows=L, heads=heads, head_dim=D, dq=dq, dk=dk, dv=dv, scale=LN2)
    assert ops.launch_counts()["attn_bwd64"] == 2

    def rel(a, b):
        return float((a - b).abs().max() / b.abs().max())
    idx = torch.tensor([0, 63, 64, 9999, 21776, 21823, 21824, 21839], device=DEV)
    for h in (0, 5):
        sl = slice(h * D, (h + 1) * D)
        qf, kf, vf, of, gf = (t[:, sl].float() for t in (q, k, v, o, d_o))
        lse_nat = lse[0, h] * LN2
        delta = (gf * of).sum(-1)
        P = torch.exp(qf[idx] @ kf.t() * LN2 - lse_nat[idx, None])
        dS = P * (gf[idx] @ vf.t() - delta[idx, None]) * LN2
        assert rel(dq[idx, sl].float(), dS @ kf) < 1e-2
        Pc = torch.exp(qf @ kf[idx].t() * LN2 - lse_nat[:, None])
        dSc = Pc * (gf @ vf[idx].t() - delta[:, None]) * LN2
        assert rel(dv[idx, sl].float(), Pc.t() @ gf) < 1e-2
        assert rel(dk[idx, sl].float(), dSc.t() @ qf) < 1e-2


def test_per_row_modulation_backward_beyond_65535_rows():
    """per-token modulation maps every row to its own modulation group; at B * Lp > 65 535 (a 720p clip, B >= 4) the
    grouped column sum and the LayerNorm backward overflowed grid.y.  rows_per_group == 1 is now elementwise and the LayerNorm backward
    walks rows with a persistent grid: 70 001 rows against fp32 torch, and the grouped forms with more than 65 535 groups."""
    from more4d_amd import ops
    R, C = 70001, 256
    g = torch.Generator(device=DEV).manual_seed(3)
    a = torch.randn(R, C, device=DEV, generator=g)
    b = torch.randn(R, C, device=DEV, generator=g).to(BF)
    out = ops.colsum(a, b, rows_per_group=1)
    assert out.shape == (R, C) and float((out - a * b.float()).abs().max()) < 1e-6
    out2 = ops.colsum(a, b, rows_per_group=1, out=out.clone())            # accumulates
    assert float((out2 - 2 * a * b.float()).abs().max()) < 1e-5
    grp = ops.colsum(a[:70000], None, rows_per_group=1)                   # (no second operand)
    assert torch.equal(grp, a[:70000])
    # grouped form with > 65 535 groups (two rows per group): the (y, z) grid
    big = torch.randn(2 * 66000, C, device=DEV, generator=g)
    s2 = ops.colsum(big, None, rows_per_group=2)
    assert float((s2 - big.view(66000, 2, C).sum(1)).abs().max()) < 1e-5
    # LayerNorm (+ per-row modulation) backward, one modulation vector per row
    x = torch.randn(R, C, device=DEV, generator=g)
    dy = torch.randn(R, C, device=DEV, generator=g).to(BF)
    sc = torch.randn(R, C, device=DEV, generator=g) * 0.2
    dx0 = torch.randn(R, C, device=DEV, generator=g)
    dx, dsh, dsc = dx0.clone(), torch.zeros(R, C, device=DEV), torch.zeros(R, C, device=DEV)
    ops.ln_modulate_bwd(x, dy, dx, B=R, rows_per_sample=1, scale=sc, mod_stride=C, eps=1e-6, dshift=dsh, dscale=dsc, red_stride=C)
    xr = x.clone().requires_grad_(True)
    scr = sc.clone().requires_grad_(True)
    xh = torch.nn.functional.layer_norm(xr, (C,), eps=1e-6)
    y = xh * (1 + scr)
    (gx, gs) = torch.autograd.grad(y, (xr, scr), dy.float())
    assert float((dx - dx0 - gx).abs().max()) < 2e-4 * float(gx.abs().max()) + 1e-5
    assert float((dsc - gs).abs().max()) < 1e-4 * float(gs.abs().max()) + 1e-6
    assert float((dsh - dy.float()).abs().max()) == 0.0


def test_tiny_dit_subject_ref_vs_reference():
    """`subject_ref` of WanTransformer4DModel.forward (reference wan_transformer4d.py:1092-1097, 1328-1331; VERDICT r5 missing #2): two extra
    frames through the patch embedding, appended behind the video tokens, cut off after the head — against the reference's own output
    (tests/golden/make_golden_r6.py), fp32, 1e-3, with and without the reference row; and the same call through the training forward."""
    from test_dit_gpu import tiny_model
    from util import load_npz, rel_err
    z, s = load_npz("dit_tiny.npz"), load_npz("dit_tiny_subject_ref.npz")
    m = tiny_model()
    ctx = [z["ctx0"].to(DEV), z["ctx1"].to(DEV)]
    kw = dict(x=z["x"].to(DEV), t=z["t"].to(DEV), context=ctx, clip_fea=z["clip"].to(DEV), y=z["y"].to(DEV),
              subject_ref=s["subject_ref"].to(DEV))
    with torch.no_grad():
        out = m(seq_len=int(s["seq_len_pad"]), full_ref=z["full_ref"].to(DEV), **kw)
        assert rel_err(out.cpu(), s["out_ref"]) < 1e-3
        assert rel_err(out.cpu(), z["out_ref"]) > 5e-3            # (the extra tokens do change the result)
        out = m(seq_len=int(s["seq_len"]), full_ref=None, **kw)
        assert rel_err(out.cpu(), s["out_noref"]) < 1e-3
    m.train()
    for p in m.parameters():
        p.requires_grad_(True)
    out = m(seq_len=int(s["seq_len_pad"]), full_ref=z["full_ref"].to(DEV), **kw)
    assert out.requires_grad and rel_err(out.detach().cpu(), s["out_ref"]) < 1e-3
    out.float().square().mean().backward()
    gpe = m.patch_embedding.weight.grad
    assert gpe is not None and bool(torch.isfinite(gpe).all()) and float(gpe.abs().sum()) > 0


def test_conv_halo64_bit_identical_to_the_two_wave_kernel():
    """conv_halo64_kernel (csrc/conv_halo64.h: the VAE's 3 x 3 x 3 conv of 96-channel tiles as one wave per SIMD, generated main loop,
    tiled weights) against conv_halo_kernel<3, 3, 12, 32, 3, 3> on plain AND on tiled weights, same inputs, in child processes (the
    switch is read once per process) — same accumulation order, the SAME epilogue source: every output (raw, + shortcut, fused RMS_norm +
    SiLU into planar-16, 96 / 192 / 384 channels, ragged right edge, a 208-column map, planar-16 and channels-last inputs) must agree bit for bit — and
    against 27 shifted fp32 GEMMs; m4d_conv_pack_weights against a torch restatement of the tiled order."""
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "check_conv64.py")], capture_output=True, text=True, timeout=900)
    out = r.stdout + r.stderr
    assert r.returncode == 0, out[-3000:]
    assert out.count("bit-identical") == 16 and "DIFFERENT" not in out and "differs from" not in out, out[-3000:]
    assert "RESULT halo64 2 tiled 1 PASS" in out and "RESULT halo64 0 tiled 1 PASS" in out and "RESULT halo64 0 tiled 0 PASS" in out, out[-3000:]
    assert out.count("'conv_halo64': 1") == 8, out[-3000:]


def test_vae_residual_block_runs_on_conv_halo64():
    """the production path takes the new kernel: a 96-channel conv at 120 x 416 with tiled weights launches conv_halo64 (planar-16 input
    by default); without tiled weights (the kernel has no gather form) and on channels-last inputs the 12 x 32 kernel runs"""
    from more4d_amd import ops
    T, H, W, C = 4, 120, 416, 96
    g = torch.Generator(device=DEV).manual_seed(0)
    x = torch.randn(T * H * W, C, device=DEV, generator=g).to(BF)
    w = (torch.randn(C, 27 * C, device=DEV, generator=g) * (27 * C) ** -0.5).to(BF)
    wt = ops.conv_pack_weights(w, C)
    xp = ops.Planar16(x.view(T, H * W, C // 16, 16).permute(2, 0, 1, 3).contiguous())
    ops.launch_counts(reset=True)
    a = ops.conv_cl_planar(xp, w, None, Tin=T, Hin=H, Win=W, kt=3, w_tiled=wt)
    cnt = ops.launch_counts()
    assert cnt["conv_halo64"] == 1 and cnt["conv_halo_mt3_12x32"] == 0, cnt
    ops.launch_counts(reset=True)
    b = ops.conv_cl_planar(xp, w, None, Tin=T, Hin=H, Win=W, kt=3)
    cnt = ops.launch_counts()
    assert cnt["conv_halo64"] == 0 and cnt["conv_halo_mt3_12x32"] == 1, cnt
    assert torch.equal(a.view(torch.int16), b.view(torch.int16))
    with pytest.raises(ValueError):
        ops.conv_cl_planar(xp, w, None, Tin=T, Hin=H, Win=W, kt=3, w_tiled=wt[:-8])


@pytest.mark.parametrize("name,kw", [
    ("3x3x3 96->32 (NT=1)", dict(T=3, H=64, W=64, Cin=96, Cout=32, kt=3)),
    ("3x3x3 32->64 (NT=2)", dict(T=3, H=64, W=96, Cin=32, Cout=64, kt=3)),
    ("3x3x3 192->128 (NT=4), narrow map", dict(T=3, H=60, W=104, Cin=192, Cout=128, kt=3)),
    ("3x3x3 384->384 small map", dict(T=3, H=30, W=52, Cin=384, Cout=384, kt=3)),
    ("3x3 192->96 up-sampled view", dict(T=2, H=60, W=104, Cin=192, Cout=96, kt=1, ups=True)),
    ("3x3 96->12 (head, Cout padded to 32)", dict(T=2, H=64, W=64, Cin=96, Cout=12, kt=1)),
    ("3x3 stride 2 96->96", dict(T=2, H=120, W=208, Cin=96, Cout=96, kt=1, stride=2)),
    ("3x3 128->128 adaptor shape", dict(T=2, H=120, W=208, Cin=128, Cout=128, kt=1)),
])
def test_tiled_weights_same_bits_on_every_halo_kernel(name, kw):
    """m4d_conv_cl_tw / m4d_conv_cl_planar_tw (weights in the tiled order of m4d_conv_pack_weights: one contiguous KiB per DMA request)
    against the plain-weight entries on the LDS-halo kernel's other instantiations — channel tiles of 32 / 64 / 96 / 128, 16 x 16 and
    8 x 32 patches, up-sampled input view, stride 2, Cout below a row block: bit-identical, and an LDS-halo kernel actually ran."""
    from more4d_amd import ops
    T, H, W, Cin, Cout, kt = (kw[k] for k in ("T", "H", "W", "Cin", "Cout", "kt"))
    ups, stride = kw.get("ups", False), kw.get("stride", 1)
    g = torch.Generator(device=DEV).manual_seed(5)
    x = torch.randn(T * H * W, Cin, device=DEV, generator=g).to(BF)
    w = (torch.randn(Cout, kt * 9 * Cin, device=DEV, generator=g) * (kt * 9 * Cin) ** -0.5).to(BF)
    b = torch.randn(Cout, device=DEV, generator=g).to(BF)
    wt = ops.conv_pack_weights(w, Cin)
    assert wt is not None and wt.numel() == -(-Cout // 32) * 32 * kt * 9 * Cin
    To = T - kt + 1
    Hl, Wl = H * (2 if ups else 1), W * (2 if ups else 1)
    if stride == 2:
        args = dict(Tin=T, Hin=H, Win=W, Cin=Cin, k=(1, 3, 3), stride=(1, 2, 2), pad=(0, 0, 0), out_thw=(T, H // 2, W // 2))
    else:
        args = dict(Tin=T, Hin=H, Win=W, Cin=Cin, k=(kt, 3, 3), pad=(0, 1, 1), out_thw=(To, Hl, Wl), ups=ups)
    ops.launch_counts(reset=True)
    a = ops.conv_cl(x, w, b, **args)
    c = ops.conv_cl(x, w, b, w_tiled=wt, **args)
    cnt = ops.launch_counts()
    assert sum(v for k, v in cnt.items() if k.startswith("conv_halo")) == 2, cnt
    assert torch.equal(a.view(torch.int16), c.view(torch.int16)), name
    if stride == 1 and not ups:
        xp = ops.Planar16(x.view(T, H * W, Cin // 16, 16).permute(2, 0, 1, 3).contiguous())
        d = ops.conv_cl_planar(xp, w, b, Tin=T, Hin=H, Win=W, kt=kt, w_tiled=wt)
        assert torch.equal(a.view(torch.int16), d.view(torch.int16)), name


def test_vae_decode_uses_tiled_weights_and_keeps_its_bits(monkeypatch):
    """the VAE's decode with the tiled copies (default) and without them (WanVAE views' tiled() -> None): same bits, and the tiled run
    launches conv_halo64"""
    import more4d_amd.models.wan_vae as wv
    from more4d_amd import ops
    from more4d_amd.models.wan_vae import AutoencoderKLWan
    from util import load_keys
    from weights import fill
    vae = AutoencoderKLWan().eval()
    vae.load_state_dict(fill(load_keys("vae_keys.json"), 2024))
    vae = vae.to(DEV, BF)
    z = torch.randn(1, 16, 2, 60, 104, device=DEV, generator=torch.Generator(device=DEV).manual_seed(3)).to(BF)
    ops.launch_counts(reset=True)
    with torch.no_grad():
        a = vae.decode(z)
    a = a.sample if hasattr(a, "sample") else a
    cnt = ops.launch_counts()
    assert cnt["conv_halo64"] > 0, cnt
    cls = next(c for c in vars(wv).values() if isinstance(c, type) and hasattr(c, "tiled") and hasattr(c, "packed"))
    monkeypatch.setattr(cls, "tiled", lambda self, conv: None)
    ops.launch_counts(reset=True)
    with torch.no_grad():
        b = vae.decode(z)
    b = b.sample if hasattr(b, "sample") else b
    assert ops.launch_counts()["conv_halo64"] == 0
    assert torch.equal(a.view(torch.int16), b.view(torch.int16))


def test_guided_dit_per_token_timesteps_gradients_vs_reference():
    """Spatial guidance (reference wan_transformer4d.py:757-783) TOGETHER with per-token timesteps (:655-657) in training — the last
    NotImplementedError of the training path (VERDICT r5 missing #3): m4d_ln_modulate_g indexes the modulation per row and the guidance
    table per position, m4d_guidance_bwd_m the same in the backward.  Prediction, loss and every gradient against the reference's
    (tests/golden/make_golden_r6.py: dit_tiny_guid_pertoken_grads.npz), fp32 1e-3; inference forward == training forward."""
    from more4d_amd.models import WanTransformer4DModel
    from test_dit_gpu import TINY
    from util import check_grads, custom_mse_loss, load_keys, load_npz, rel_err
    from weights import fill
    z, pz, zg = load_npz("dit_tiny.npz"), load_npz("dit_tiny_pertoken.npz"), load_npz("dit_tiny_guid_pertoken_grads.npz")
    m = WanTransformer4DModel(**dict(TINY, use_omnimae_guidance=True))
    missing = m.load_state_dict(fill(load_keys("dit_tiny_guid_keys.json"), 4321), strict=False)
    assert all(k.startswith("omnimae_extractor.") for k in missing.missing_keys)
    m = m.to(DEV, torch.float32).train()
    kw = dict(x=z["x"].to(DEV), t=pz["t_tok"].to(DEV), context=[z["ctx0"].to(DEV), z["ctx1"].to(DEV)], seq_len=int(z["seq_len_pad"]),
              clip_fea=z["clip"].to(DEV), y=z["y"].to(DEV), full_ref=z["full_ref"].to(DEV),
              first_frame_features=(zg["patch"].to(DEV), zg["cls"].to(DEV)))
    for budget in (None, 0):
        m.zero_grad(set_to_none=True)
        m.activation_budget_gb = budget
        pred = m(**kw)
        custom_mse_loss(pred, zg["target"].to(DEV)).backward()
        assert rel_err(pred.detach().cpu(), zg["pred"]) < 1e-3
        grads = {n: p.grad for n, p in m.named_parameters()}
        assert grads["blocks.1.spatial_guidance_ffn.gate"] is not None and grads["feature_adapter.0.weight"] is not None
        print("worst guided per-token gradient error", check_grads(grads, zg, 1e-3))
    with torch.no_grad():
        out = m(**kw)
    assert rel_err(out.float().cpu(), zg["pred"]) < 1e-3


def test_ln_modulate_per_row_modulation_with_guidance_kernel():
    """m4d_ln_modulate_g: one (shift, scale) vector per ROW (rows_per_sample = 1) and the guidance table by the row's position inside its
    sample (g_rows = Lp): (LN(x) (1 + sc) + sh)(1 + gs gate) + gh gate for the guided rows l < g_len, plain LN-modulate beyond
    (replaces the torch-side fold of round 4); and its backward m4d_guidance_bwd_m with mod_rows = 1."""
    from more4d_amd import ops
    g = torch.Generator(device=DEV).manual_seed(0)
    B, Lp, C, P, glen = 2, 24, 256, 8, 20
    x = torch.randn(B, Lp, C, device=DEV, generator=g)
    e = torch.randn(B * Lp, 2, C, device=DEV, generator=g) * 0.3
    gss = torch.randn(B, P, 2 * C, device=DEV, generator=g) * 0.5
    gate = torch.randn(C, device=DEV, generator=g)
    out = ops.ln_modulate(x, torch.float32, shift=e[:, 0], scale=e[:, 1], mod_stride=2 * C, rows_per_sample=1, eps=1e-6, g_ss=gss, g_gate=gate,
                          g_period=P, g_len=glen, g_rows=Lp)
    ev = e.view(B, Lp, 2, C)
    u = torch.nn.functional.layer_norm(x, (C,), eps=1e-6) * (1 + ev[:, :, 1]) + ev[:, :, 0]
    want = u.clone()
    idx = torch.arange(glen, device=DEV) % P
    want[:, :glen] = u[:, :glen] * (1 + gss[:, idx, :C] * gate) + gss[:, idx, C:] * gate
    assert float((out - want).abs().max()) < 2e-5
    # backward: dz -> du in place, ab = (sum_f dz u | sum_f dz) per position
    dz = torch.randn(B, Lp, C, device=DEV, generator=g)
    du = dz.clone()
    ab = ops.guidance_bwd_(x, du, B=B, rows_per_sample=Lp, shift=e[:, 0], scale=e[:, 1], mod_stride=2 * C, g_ss=gss, g_gate=gate, g_period=P,
                           g_len=glen, mod_rows=1)
    want_du = dz.clone()
    want_du[:, :glen] = dz[:, :glen] * (1 + gss[:, idx, :C] * gate)
    assert float((du - want_du).abs().max()) < 2e-5
    A = torch.zeros(B, P, C, device=DEV)
    Bm = torch.zeros(B, P, C, device=DEV)
    for l in range(glen):
        A[:, l % P] += dz[:, l] * u[:, l]
        Bm[:, l % P] += dz[:, l]
    assert float((ab[..., :C] - A).abs().max()) < 1e-4 and float((ab[..., C:] - Bm).abs().max()) < 1e-4


def test_sp_remote_calls_on_a_second_stream(monkeypatch):
    """M4D_SP_OVERLAP=1 (remote-shard attention calls on a side stream beside the local call, merges in the sequential order) gives
    every emulated rank the rows of the unsharded result (A/B with clocks: profiles/r06_ab_sp_overlap.log — same bits, no gain)."""
    import more4d_amd.models.wan_transformer4d as wt
    from test_dit_gpu import test_token_sharded_local_first_schedule_single_gpu as run
    monkeypatch.setattr(wt, "_SP_OVERLAP", True)
    run()
    torch.cuda.synchronize()


def test_conv_halo64_k1_bit_identical_to_the_two_wave_kernel():
    """conv_halo64_kernel<1, 4, 4> (the adaptors' 3 x 3 convs on 128-channel tiles as one wave per SIMD, generated with
    tools/gen_conv_halo64.py --shape 1x4x4; measured no faster and therefore off by default, M4D_CONV_HALO64K1=1) against
    conv_halo_kernel<1, 3, 8, 32, 4, 2>: raw, shortcut, ragged patches, fused norm bit for bit, GroupNorm sums to 1e-5."""
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "check_conv64k1.py")], capture_output=True, text=True, timeout=900)
    out = r.stdout + r.stderr
    assert r.returncode == 0, out[-3000:]
    assert out.count("bit-identical") == 6 and "DIFFERENT" not in out and "RESULT k1 1 PASS" in out and "RESULT k1 0 PASS" in out, out[-3000:]
    assert out.count("'conv_halo64': 1") >= 5, out[-3000:]
