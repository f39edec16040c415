"""Per-kernel parity on the MI355X: every C-ABI entry point (through more4d_amd.ops) against the CPU oracle /
plain fp32 torch on the same seeded inputs.  fp32 mode must meet the path's 1e-3 bar with a wide margin
(1e-4 here: exact-fp32 MFMA, only the summation order differs); bf16 mode is held to a bf16 budget."""
import math

import pytest
import torch

from util import load_npz, rel_err

pytestmark = pytest.mark.gpu

DEV = "cuda"
F32_TOL = 1e-4
BF16_TOL = 8e-3   # bf16 OUTPUT rounding bounds max|err|/max|ref| at 2^-8 = 3.9e-3; operands are compared after the same bf16 quantisation


def ops():
    from more4d_amd import ops as o
    return o


def rnd(*shape, seed=0, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    return torch.randn(*shape, generator=g) * scale


def q(x, dt):
    """round to the compute dtype and back (inputs the kernel really sees)."""
    return x.to(dt).float()


@pytest.mark.parametrize("dt", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("M,N,K", [(128, 128, 128), (300, 192, 264), (2, 64, 256), (1000, 520, 72), (130, 4, 8),
                                   (1000, 520, 192), (515, 1028, 64), (768, 512, 320)])  # last three: 256x256 DMA kernel
def test_gemm_store_and_activations(dt, M, N, K):
    o = ops()
    a, w, b = rnd(M, K, seed=1), rnd(N, K, seed=2, scale=K ** -0.5), rnd(N, seed=3)
    ref = q(a, dt) @ q(w, dt).t() + q(b, dt)
    tol = F32_TOL if dt == torch.float32 else BF16_TOL
    ad, wd, bd = a.to(DEV, dt), w.to(DEV, dt), b.to(DEV, dt)
    out = o.gemm_bt(ad, wd, bd)
    assert rel_err(out.float().cpu(), ref) < tol
    out = o.gemm_bt(ad, wd, None)
    assert rel_err(out.float().cpu(), ref - q(b, dt)) < tol
    for epi, fn in ((o.EPI_GELU_TANH, lambda x: torch.nn.functional.gelu(x, approximate="tanh")),
                    (o.EPI_GELU_ERF, torch.nn.functional.gelu), (o.EPI_SILU, torch.nn.functional.silu)):
        out = o.gemm_bt(ad, wd, bd, epilogue=epi)
        assert rel_err(out.float().cpu(), fn(ref)) < tol
    out = o.gemm_bt(ad, wd, bd, epilogue=o.EPI_STORE_F32)
    assert out.dtype == torch.float32
    assert rel_err(out.cpu(), ref) < tol


@pytest.mark.parametrize("dt", [torch.float32, torch.bfloat16])
def test_gemm_resid_gate_and_bias_on_m(dt):
    o = ops()
    B, L, N, K = (2, 77, 136, 96) if dt == torch.float32 else (2, 300, 520, 128)   # bf16: 256x256 DMA kernel
    M = B * L
    a, w, b = rnd(M, K, seed=1), rnd(N, K, seed=2, scale=K ** -0.5), rnd(N, seed=3)
    gate = rnd(B, 6, N, seed=4)
    resid = rnd(M, N, seed=5)
    y = (q(a, dt) @ q(w, dt).t() + q(b, dt)).to(dt).float()
    ref = resid + y * gate[:, 2].repeat_interleave(L, dim=0)
    tol = F32_TOL if dt == torch.float32 else BF16_TOL
    r = resid.to(DEV).clone()
    gd = gate.to(DEV)
    o.gemm_bt(a.to(DEV, dt), w.to(DEV, dt), b.to(DEV, dt), out=r, epilogue=o.EPI_RESID_GATE, gate=gd[:, 2],
              gate_stride=6 * N, rows_per_sample=L)
    assert rel_err(r.cpu(), ref) < tol
    r = resid.to(DEV).clone()
    o.gemm_bt(a.to(DEV, dt), w.to(DEV, dt), b.to(DEV, dt), out=r, epilogue=o.EPI_RESID_GATE, gate=None)
    assert rel_err(r.cpu(), resid + y) < tol
    # transposed product with a per-row bias: V^T = W_v x^T + b[:, None]
    bm = rnd(M, seed=6)
    out = o.gemm_bt(a.to(DEV, dt), w.to(DEV, dt)[:N // 8 * 8], bm.to(DEV, dt), bias_on_m=True)
    ref2 = q(a, dt) @ q(w, dt)[:N // 8 * 8].t() + q(bm, dt)[:, None]
    assert rel_err(out.float().cpu(), ref2) < tol
    # strided output rows (writing into a slice of a wider buffer)
    wide = torch.zeros(M, N + 24, device=DEV, dtype=dt)
    o.gemm_bt(a.to(DEV, dt), w.to(DEV, dt), b.to(DEV, dt), out=wide[:, 8:8 + N])
    assert rel_err(wide[:, 8:8 + N].float().cpu(), q(a, dt) @ q(w, dt).t() + q(b, dt)) < tol
    assert float(wide[:, :8].abs().sum()) == 0 and float(wide[:, 8 + N:].abs().sum()) == 0


def test_gemm_large_k_accumulation():
    """K = 13824 (the FFN down projection) in bf16: fp32 accumulation must hold."""
    o = ops()
    M, N, K = 256, 256, 13824
    a, w = rnd(M, K, seed=1), rnd(N, K, seed=2, scale=K ** -0.5)
    ref = q(a, torch.bfloat16).double() @ q(w, torch.bfloat16).double().t()
    out = o.gemm_bt(a.to(DEV, torch.bfloat16), w.to(DEV, torch.bfloat16), None, epilogue=o.EPI_STORE_F32)
    assert rel_err(out.cpu(), ref.float()) < BF16_TOL   # bf16 rounding of the result only


def test_gemm_rejects_bad_args():
    o = ops()
    from more4d_amd._lib import More4DHipError
    a = torch.zeros(8, 12, device=DEV, dtype=torch.bfloat16)     # K*2 = 24 bytes: not a multiple of 16
    with pytest.raises(More4DHipError):
        o.gemm_bt(a, a, None)
    with pytest.raises(More4DHipError):
        o.gemm_bt(torch.zeros(4, 8), torch.zeros(4, 8))          # CPU tensors


@pytest.mark.parametrize("C", [128, 1536, 5120])
@pytest.mark.parametrize("odt", [torch.float32, torch.bfloat16])
def test_ln_modulate(C, odt):
    o = ops()
    B, L = 2, 37
    x = rnd(B, L, C, seed=1, scale=2.0) + 0.5
    e = rnd(B, 6, C, seed=2, scale=0.3)
    xf = x.float()
    mu = xf.mean(-1, keepdim=True)
    ln = (xf - mu) * torch.rsqrt((xf - mu).pow(2).mean(-1, keepdim=True) + 1e-6)
    ref = ln * (1 + e[:, 1:2]) + e[:, 0:1]
    tol = 1e-5 if odt == torch.float32 else 8e-3
    ed = e.to(DEV)
    out = o.ln_modulate(x.to(DEV), odt, shift=ed[:, 0], scale=ed[:, 1], mod_stride=6 * C, rows_per_sample=L, eps=1e-6)
    assert rel_err(out.float().cpu(), ref) < tol
    w, b = rnd(C, seed=3) * 0.1 + 1, rnd(C, seed=4) * 0.1
    out = o.ln_modulate(x.to(DEV), odt, ln_w=w.to(DEV), ln_b=b.to(DEV), eps=1e-6)
    assert rel_err(out.float().cpu(), ln * w + b) < tol
    # spatial guidance: periodic table over P positions, zero past g_len
    P, glen = 5, 30
    gss = rnd(B, P, 2 * C, seed=5, scale=0.2)
    gate = rnd(C, seed=6, scale=0.5)
    sc = torch.zeros(B, L, C)
    sh = torch.zeros(B, L, C)
    for l in range(glen):
        sc[:, l] = gss[:, l % P, :C]
        sh[:, l] = gss[:, l % P, C:]
    refg = ref * (1 + sc * gate) + sh * gate
    out = o.ln_modulate(x.to(DEV), odt, shift=ed[:, 0], scale=ed[:, 1], mod_stride=6 * C, rows_per_sample=L, eps=1e-6,
                        g_ss=gss.to(DEV), g_gate=gate.to(DEV), g_period=P, g_len=glen)
    assert rel_err(out.float().cpu(), refg) < tol
    # bf16 input (MLPProj's second LayerNorm)
    xb = x.to(torch.bfloat16)
    xbf = xb.float()
    mu = xbf.mean(-1, keepdim=True)
    lnb = (xbf - mu) * torch.rsqrt((xbf - mu).pow(2).mean(-1, keepdim=True) + 1e-5)
    out = o.ln_modulate(xb.to(DEV), odt, ln_w=w.to(DEV), ln_b=b.to(DEV), eps=1e-5)
    assert rel_err(out.float().cpu(), lnb * w + b) < tol


@pytest.mark.parametrize("dt", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("C,hd", [(256, 128), (5120, 128), (128, 32)])
def test_rmsnorm_rope(dt, C, hd):
    from oracle import dit as odit
    o = ops()
    B, grid, pad = 2, (2, 3, 4), 3
    S = grid[0] * grid[1] * grid[2]
    L = S + pad
    xq, xk = rnd(B, L, C, seed=1, scale=2.0), rnd(B, L, C, seed=2)
    wq, wk = rnd(C, seed=3) * 0.1 + 1, rnd(C, seed=4) * 0.1 + 1
    n = C // hd

    def ref(x, w):
        xr = q(x, dt)
        y = odit.rms_norm(xr, w, 1e-6)
        if dt == torch.bfloat16:   # (x*rsqrt).to(bf16) * w
            inv = torch.rsqrt(xr.pow(2).mean(-1, keepdim=True) + 1e-6)
            y = (xr * inv).to(dt).float() * w
        return odit.rope_apply(y.view(B, L, n, hd), grid).reshape(B, L, C)

    cos, sin = odit.rope_token_table(hd, grid)
    cos, sin = cos.float().contiguous().to(DEV), sin.float().contiguous().to(DEV)
    dq, dk = xq.to(DEV, dt).clone(), xk.to(DEV, dt).clone()
    o.rmsnorm_rope(dq, wq.to(DEV), dk, wk.to(DEV), head_dim=hd, eps=1e-6, cos=cos, sin=sin, rows_per_sample=L,
                   rope_len=S)
    tol = 2e-5 if dt == torch.float32 else BF16_TOL
    assert rel_err(dq.float().cpu(), ref(xq, wq)) < tol
    assert rel_err(dk.float().cpu(), ref(xk, wk)) < tol
    # norm only (cross-attention), single tensor, strided rows
    wide = torch.zeros(B * L, C + 16, device=DEV, dtype=dt)
    wide[:, :C] = xq.view(B * L, C).to(DEV, dt)
    o.rmsnorm_rope(wide[:, :C], wq.to(DEV), head_dim=hd, eps=1e-6)
    xr = q(xq, dt)
    inv = torch.rsqrt(xr.pow(2).mean(-1, keepdim=True) + 1e-6)
    refn = (xr * inv).to(dt).float() * wq
    assert rel_err(wide[:, :C].float().cpu().view(B, L, C), refn) < tol
    # shard semantics: rows [4, 4+8) of the sequence with pos_offset
    sub = xq[:, 4:12].contiguous().to(DEV, dt)
    o.rmsnorm_rope(sub, wq.to(DEV), head_dim=hd, eps=1e-6, cos=cos, sin=sin, rows_per_sample=8, rope_len=8,
                   pos_offset=4)
    assert rel_err(sub.float().cpu(), ref(xq, wq)[:, 4:12]) < tol


def _attn_ref(qq, kk, vv, klen=None):
    from oracle import dit as odit
    return odit.sdpa(qq, kk, vv, klen)


@pytest.mark.parametrize("dt", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("D", [32, 64, 128])
@pytest.mark.parametrize("Lq,Lk", [(64, 80), (200, 257), (128, 64), (5, 9)])
def test_attention(dt, D, Lq, Lk):
    o = ops()
    B, n = 2, 3
    C = n * D
    qq, kk, vv = rnd(B, Lq, n, D, seed=1), rnd(B, Lk, n, D, seed=2), rnd(B, Lk, n, D, seed=3)
    ref = _attn_ref(q(qq, dt), q(kk, dt), q(vv, dt)).reshape(B, Lq, C)
    Lkp = (Lk + 7) // 8 * 8
    kd = torch.zeros(B, Lkp, C, dtype=dt)
    kd[:, :Lk] = kk.reshape(B, Lk, C).to(dt)
    vt = torch.full((C, B * Lkp), float("nan"), dtype=dt)     # padding columns poisoned: must never be read into the sum
    for b in range(B):
        vt[:, b * Lkp:b * Lkp + Lk] = vv[b].reshape(Lk, C).t().to(dt)
    kd, vt = kd.to(DEV), vt.to(DEV)
    seg = o.KV(kd, vt, Lkp * C, C, Lkp, B * Lkp, Lk)
    out = o.attention(qq.reshape(B, Lq, C).to(DEV, dt).contiguous(), [seg], B=B, Lq=Lq, heads=n, head_dim=D)
    tol = F32_TOL if dt == torch.float32 else BF16_TOL
    assert rel_err(out.float().cpu(), ref) < tol
    # accumulate: out += second attention (x + img_x)
    out2 = o.attention(qq.reshape(B, Lq, C).to(DEV, dt).contiguous(), [seg], B=B, Lq=Lq, heads=n, head_dim=D,
                       out=out.clone(), accumulate=True)
    assert rel_err(out2.float().cpu(), 2 * ref) < 2 * tol


@pytest.mark.parametrize("dt", [torch.float32, torch.bfloat16])
def test_attention_segments_equal_concat(dt):
    """K/V handed over as 3 ragged segments (what the T-sharded loop does) == one concatenated K/V."""
    o = ops()
    B, n, D, Lq = 2, 2, 128, 70
    C = n * D
    lens = [40, 24, 33]
    qq = rnd(B, Lq, n, D, seed=1)
    ks = [rnd(B, l, n, D, seed=10 + i) for i, l in enumerate(lens)]
    vs = [rnd(B, l, n, D, seed=20 + i) for i, l in enumerate(lens)]
    ref = _attn_ref(q(qq, dt), q(torch.cat(ks, 1), dt), q(torch.cat(vs, 1), dt)).reshape(B, Lq, C)
    segs = []
    for k_, v_ in zip(ks, vs):
        l = k_.shape[1]
        lp = 48   # every shard buffer is 48 rows, only `l` valid (padding rows hold garbage)
        kd = torch.full((B, lp, C), 7.0, dtype=dt)
        kd[:, :l] = k_.reshape(B, l, C).to(dt)
        vt = torch.full((C, B * lp), -3.0, dtype=dt)
        for b in range(B):
            vt[:, b * lp:b * lp + l] = v_[b].reshape(l, C).t().to(dt)
        segs.append(o.KV(kd.to(DEV), vt.to(DEV), lp * C, C, lp, B * lp, l))
    out = o.attention(qq.reshape(B, Lq, C).to(DEV, dt).contiguous(), segs, B=B, Lq=Lq, heads=n, head_dim=D)
    assert rel_err(out.float().cpu(), ref) < (F32_TOL if dt == torch.float32 else BF16_TOL)


def test_attention_online_softmax_rescale():
    """Force the running-max rescale: one key far above the rest in a late tile (cdna guide §5.4 rule 26)."""
    o = ops()
    B, n, D, Lq, Lk = 1, 1, 128, 32, 256
    qq, kk, vv = rnd(B, Lq, n, D, seed=1), rnd(B, Lk, n, D, seed=2), rnd(B, Lk, n, D, seed=3)
    kk[0, 200] = qq[0, 7] * 3.0          # spike: q7 . k200 >> everything else, arrives in the 4th tile
    kk[0, 3] = qq[0, 20] * 2.0           # and an early spike for another row
    ref = _attn_ref(qq, kk, vv).reshape(B, Lq, D)
    kd = kk.reshape(B, Lk, D).to(DEV)
    vt = vv[0].reshape(Lk, D).t().contiguous().to(DEV)
    out = o.attention(qq.reshape(B, Lq, D).to(DEV), [o.KV(kd, vt, Lk * D, D, Lk, Lk, Lk)], B=B, Lq=Lq, heads=n,
                      head_dim=D)
    assert rel_err(out.cpu(), ref) < F32_TOL


def test_golden_sdpa_fixture():
    o = ops()
    z = load_npz("dit_ops.npz")
    qq, kk, vv = z["att_q"], z["att_k"], z["att_v"]
    B, Lq, n, D = qq.shape
    Lk = kk.shape[1]
    C = n * D
    vt = vv[0].reshape(Lk, C).t().contiguous().to(DEV)
    out = o.attention(qq.reshape(B, Lq, C).to(DEV), [o.KV(kk.reshape(B, Lk, C).to(DEV), vt, Lk * C, C, Lk, Lk, Lk)],
                      B=B, Lq=Lq, heads=n, head_dim=D)
    assert rel_err(out.cpu().view(B, Lq, n, D), z["att_out"]) < F32_TOL


@pytest.mark.parametrize("sdt,odt", [(torch.float32, torch.float32), (torch.bfloat16, torch.bfloat16),
                                     (torch.float32, torch.bfloat16)])
def test_patchify_unpatchify(sdt, odt):
    from oracle import dit as odit
    o = ops()
    B, F, H, W = 2, 3, 8, 12
    x, y = rnd(B, 16, F, H, W, seed=1), rnd(B, 48, F, H, W, seed=2)
    out = o.patchify(x.to(DEV, sdt), y.to(DEV, sdt), (1, 2, 2), odt)
    xc = torch.cat([x, y], 1).to(sdt).float()
    ref = xc.view(B, 64, F, 1, H // 2, 2, W // 2, 2).permute(0, 2, 4, 6, 1, 3, 5, 7).reshape(B, -1, 256)
    assert torch.equal(out.float().cpu(), ref.to(odt).float())
    tok = rnd(B, 5 + F * (H // 2) * (W // 2), 64, seed=3)
    res = o.unpatchify(tok.to(DEV), 5, (F, H // 2, W // 2), (1, 2, 2), 16, odt)
    refu = odit.unpatchify(tok[:, 5:], (F, H // 2, W // 2), (1, 2, 2), 16)
    assert torch.equal(res.float().cpu(), refu.to(odt).float())


def test_cfg_euler_unary_add_bcast():
    from oracle import sched
    o = ops()
    x, v = rnd(1, 16, 3, 8, 8, seed=1), rnd(2, 16, 3, 8, 8, seed=2)
    ref = sched.euler_step(x, sched.cfg_combine(v[0:1], v[1:2], 6.0), 0.9, 0.85)
    xd = x.to(DEV).clone()
    o.cfg_euler_(xd, v.to(DEV), 6.0, 0.85 - 0.9)
    assert rel_err(xd.cpu(), ref) < 1e-6
    a = rnd(1000, seed=3)
    assert rel_err(o.unary(a.to(DEV), torch.float32, act=1).cpu(), torch.nn.functional.silu(a)) < 1e-6
    assert torch.equal(o.unary(a.to(DEV), torch.bfloat16).cpu(), a.to(torch.bfloat16))
    e0, m = rnd(2, 6, 128, seed=4), rnd(1, 6, 128, seed=5)
    assert torch.equal(o.add_bcast(e0.to(DEV), m.to(DEV)).cpu(), e0 + m)


@pytest.mark.parametrize("M,N,K", [(1000, 520, 192), (515, 1028, 64), (768, 512, 320), (600, 2048, 1024)])
def test_gemm_packed_weights(M, N, K):
    """Production kernel with the weight pre-shuffled into MFMA fragment order (packed side 0 = W, 1 = A)."""
    o = ops()
    dt = torch.bfloat16
    a, w, b = rnd(M, K, seed=1), rnd(N, K, seed=2, scale=K ** -0.5), rnd(N, seed=3)
    ref = q(a, dt) @ q(w, dt).t() + q(b, dt)
    ad, wd, bd = a.to(DEV, dt), w.to(DEV, dt), b.to(DEV, dt)
    wp = o.pack_frag(wd)
    out = o.gemm_bt(ad, wp, bd)
    assert rel_err(out.float().cpu(), ref) < BF16_TOL
    assert torch.equal(out, o.gemm_bt(ad, wd, bd))          # same products, same fp32 accumulation order per element? (K order)
    out = o.gemm_bt(ad, wp, bd, epilogue=o.EPI_GELU_TANH)
    assert rel_err(out.float().cpu(), torch.nn.functional.gelu(ref, approximate="tanh")) < BF16_TOL
    # residual + gate epilogue
    B, L = 2, M // 2
    gate = rnd(B, 6, N, seed=4)
    resid = rnd(B * L, N, seed=5)
    r = resid.to(DEV).clone()
    gd = gate.to(DEV)
    o.gemm_bt(ad[:B * L], wp, bd, out=r, epilogue=o.EPI_RESID_GATE, gate=gd[:, 2], gate_stride=6 * N, rows_per_sample=L)
    y = ref[:B * L].to(dt).float()
    assert rel_err(r.cpu(), resid + y * gate[:, 2].repeat_interleave(L, dim=0)) < BF16_TOL
    # packed operand on the M side with a per-row bias: V^T = W x^T + b[:, None]
    outT = o.gemm_bt(wp, ad, bd, bias_on_m=True) if M % 4 == 0 else None
    if outT is not None:
        assert rel_err(outT.float().cpu(), (q(w, dt) @ q(a, dt).t()) + q(b, dt)[:, None]) < BF16_TOL


def test_attention_phased_kernel_segments_and_rescale():
    """The production long-range kernel (attention_phased.h: Lq > 1024, >= 2048 keys, bf16, D = 128): three ragged
    segments with garbage padding (the T-sharded call), plus late score spikes far beyond the lazy 2^8 rescale threshold."""
    o = ops()
    dt = torch.bfloat16
    B, n, D, Lq = 1, 2, 128, 1280
    C = n * D
    lens, lp = [2730, 2736, 1000], 2736
    qq = rnd(B, Lq, n, D, seed=1)
    ks = [rnd(B, l, n, D, seed=10 + i) for i, l in enumerate(lens)]
    vs = [rnd(B, l, n, D, seed=20 + i) for i, l in enumerate(lens)]
    ks[1][0, 2000, 0] = qq[0, 5, 0] * 6.0          # row 5 / head 0: a spike ~ 2^60 over the running reference, late in segment 1
    ks[2][0, 900, 1] = qq[0, 700, 1] * 5.0         # and one inside the last segment's full tiles
    ks[2][0, 999, 1] = qq[0, 701, 1] * 5.0         # and one in its peeled ragged tail
    ref = _attn_ref(q(qq, dt), q(torch.cat(ks, 1), dt), q(torch.cat(vs, 1), dt)).reshape(B, Lq, C)
    segs = []
    for k_, v_ in zip(ks, vs):
        l = k_.shape[1]
        kd = torch.full((B, lp, C), 7.0, dtype=dt)
        kd[:, :l] = k_.reshape(B, l, C).to(dt)
        vt = torch.full((C, B * lp), float("nan"), dtype=dt)
        for b in range(B):
            vt[:, b * lp:b * lp + l] = v_[b].reshape(l, C).t().to(dt)
        segs.append(o.KV(kd.to(DEV), vt.to(DEV), lp * C, C, lp, B * lp, l))
    lse = torch.empty(B, n, Lq, device=DEV)
    out = o.attention(qq.reshape(B, Lq, C).to(DEV, dt).contiguous(), segs, B=B, Lq=Lq, heads=n, head_dim=D, lse=lse)
    assert torch.isfinite(out).all()
    assert rel_err(out.float().cpu(), ref) < BF16_TOL
    s = torch.einsum("bqhd,bkhd->bhqk", q(qq, dt), q(torch.cat(ks, 1), dt)) / D ** 0.5
    lse_ref = torch.logsumexp(s, -1) * 1.4426950408889634
    assert float((lse.cpu() - lse_ref).abs().max()) < 5e-2


@pytest.mark.parametrize("cin,cout,k,pad", [(96, 96, (3, 3, 3), (0, 1, 1)), (192, 384, (3, 3, 3), (0, 1, 1)),
                                            (384, 384, (1, 1, 1), (0, 0, 0)), (128, 128, (1, 3, 3), (0, 1, 1)),
                                            (96, 4, (3, 3, 3), (0, 1, 1)), (16, 96, (3, 3, 3), (0, 1, 1)),      # the VAE's head / conv1
                                            (128, 4, (1, 3, 3), (0, 1, 1)), (16, 128, (1, 3, 3), (0, 1, 1)),   # the adaptors' conv_out / conv_in
                                            (48, 64, (3, 3, 3), (0, 1, 1)), (32, 20, (1, 3, 3), (0, 1, 1))])
def test_conv_cl_production_kernel(cin, cout, k, pad):
    _conv_cl_production_case(cin, cout, k, pad, 4, 20, 24)


@pytest.mark.parametrize("cin,cout,k,pad,thw", [(96, 96, (3, 3, 3), (0, 1, 1), (4, 128, 160)),      # 8 x 32 patches, 3 column tiles per wave
                                                (128, 128, (1, 3, 3), (0, 1, 1), (4, 128, 160)),   # 4 column tiles (the adaptors)
                                                (48, 192, (3, 3, 3), (0, 1, 1), (3, 120, 104))])   # 16 x 16 patches (narrow maps)
def test_conv_cl_production_kernel_large_maps(cin, cout, k, pad, thw):
    """Maps with more than 256 patches: the 96 / 128-channel tiles of conv_halo_kernel (smaller maps take 32-channel tiles)."""
    _conv_cl_production_case(cin, cout, k, pad, *thw)


def _conv_cl_production_case(cin, cout, k, pad, To, H, W):
    """conv_cl256_kernel (bf16, unit stride, M >= 1024: DMA gather with a zero page for the padding taps) against fp32 torch
    on bf16-rounded operands: borders in H and W, the causal 2-frame tail in T, K not a multiple of 64, Cout not of 128."""
    import torch.nn.functional as F
    o = ops()
    Tin = To + k[0] - 1
    g = torch.Generator().manual_seed(0)
    x = torch.randn(Tin, H, W, cin, generator=g).bfloat16()
    w = (torch.randn(cout, cin, *k, generator=g) * (cin * k[0] * k[1] * k[2]) ** -0.5).bfloat16()
    b = torch.randn(cout, generator=g).bfloat16()
    res = torch.randn(To * H * W, cout, generator=g).bfloat16()
    ref = F.conv3d(x.float().permute(3, 0, 1, 2)[None], w.float(), b.float(), padding=pad)[0].permute(1, 2, 3, 0).reshape(-1, cout)
    wp = w.permute(0, 2, 3, 4, 1).reshape(cout, -1).contiguous()
    out = o.conv_cl(x.to(DEV), wp.to(DEV), b.to(DEV), Tin=Tin, Hin=H, Win=W, Cin=cin, k=k, pad=pad, out_thw=(To, H, W))
    assert rel_err(out.float().cpu(), ref) < BF16_TOL
    out2 = o.conv_cl(x.to(DEV), wp.to(DEV), b.to(DEV), Tin=Tin, Hin=H, Win=W, Cin=cin, k=k, pad=pad, out_thw=(To, H, W),
                     resid=res.to(DEV))
    assert rel_err(out2.float().cpu(), ref.bfloat16().float() + res.float()) < BF16_TOL


@pytest.mark.parametrize("cin,cout,kt,thw", [(96, 96, 3, (4, 128, 160)), (96, 4, 3, (2, 40, 64)), (384, 384, 3, (2, 60, 104)),
                                             (128, 128, 1, (3, 64, 96)), (192, 96, 3, (2, 30, 52)), (32, 64, 3, (1, 33, 47))])
def test_planar16_norm_and_conv_equal_channels_last(cin, cout, kt, thw):
    """m4d_rmsnorm_silu_cl_planar + m4d_conv_cl_planar (the norm -> conv staging path of the VAE) against the channels-last pair:
    the same arithmetic on another layout, so bit-identical; the planar buffer is a window of a larger ring with a plane stride."""
    o = ops()
    To, H, W = thw
    Tin = To + kt - 1
    g = torch.Generator().manual_seed(5)
    x = torch.randn(Tin * H * W, cin, generator=g).bfloat16().to(DEV)
    gamma = (1 + 0.1 * torch.randn(cin, generator=g)).to(DEV)
    w = (torch.randn(cout, kt * 9 * cin, generator=g) * (kt * 9 * cin) ** -0.5).bfloat16().to(DEV)
    b = torch.randn(cout, generator=g).bfloat16().to(DEV)
    res = torch.randn(To * H * W, cout, generator=g).bfloat16().to(DEV)
    xn = o.rmsnorm_silu_cl(x, gamma, silu=True)
    ref = o.conv_cl(xn, w, b, Tin=Tin, Hin=H, Win=W, Cin=cin, k=(kt, 3, 3), pad=(0, 1, 1), out_thw=(To, H, W), resid=res)
    ring = torch.full((cin // 16, Tin + 5, H * W, 16), float("nan"), dtype=torch.bfloat16, device=DEV)
    win = o.Planar16(ring[:, 3:3 + Tin])
    o.rmsnorm_silu_cl_planar(x, gamma, win, silu=True)
    assert torch.equal(win.t.permute(1, 2, 0, 3).reshape(Tin * H * W, cin), xn)
    assert torch.isnan(ring[:, :3].float()).all() and torch.isnan(ring[:, 3 + Tin:].float()).all()
    out = o.conv_cl_planar(win, w, b, Tin=Tin, Hin=H, Win=W, kt=kt, resid=res)
    assert torch.equal(out, ref)
    if cout in (32, 64, 96, 128):
        # the next layer's RMS_norm+SiLU in the conv epilogue == the separate kernel on the conv's result, bit for bit
        gamma2 = (1 + 0.1 * torch.randn(cout, generator=g)).to(DEV)
        want = o.rmsnorm_silu_cl(ref, gamma2, silu=True)
        for keep in (True, False):
            ring2 = torch.full((cout // 16, To + 4, H * W, 16), float("nan"), dtype=torch.bfloat16, device=DEV)
            dst = o.Planar16(ring2[:, 2:2 + To])
            raw = o.conv_cl_planar(win, w, b, Tin=Tin, Hin=H, Win=W, kt=kt, resid=res, norm=(gamma2, dst, True), keep_raw=keep)
            assert (raw is None) == (not keep) and (raw is None or torch.equal(raw, ref))
            assert torch.equal(dst.t.permute(1, 2, 0, 3).reshape(To * H * W, cout), want)
            assert torch.isnan(ring2[:, :2].float()).all() and torch.isnan(ring2[:, 2 + To:].float()).all()


@pytest.mark.parametrize("cin,cout,thw", [(96, 96, (3, 64, 96)), (192, 192, (2, 40, 104)), (48, 128, (2, 34, 70)), (32, 64, (1, 33, 65))])
def test_conv_cl_production_kernel_stride2(cin, cout, thw):
    """The Resample down-sampling conv (ZeroPad2d((0,1,0,1)) + Conv2d(3, stride 2), wan_vae.py:96-100) on conv_halo_kernel's stride-2
    variant (halo rows stored as even | odd pixels): even and odd map sizes, wide and narrow patches, 96- and 128-channel tiles."""
    import torch.nn.functional as F
    o = ops()
    T, H, W = thw
    Ho, Wo = (H + 1) // 2, (W + 1) // 2
    g = torch.Generator().manual_seed(2)
    x = torch.randn(T, H, W, cin, generator=g).bfloat16()
    w = (torch.randn(cout, cin, 3, 3, generator=g) * (cin * 9) ** -0.5).bfloat16()
    b = torch.randn(cout, generator=g).bfloat16()
    xp = F.pad(x.float().permute(0, 3, 1, 2), (0, 2 * Wo + 1 - W, 0, 2 * Ho + 1 - H))
    ref = F.conv2d(xp, w.float(), b.float(), stride=2).permute(0, 2, 3, 1).reshape(-1, cout)
    wp = w.permute(0, 2, 3, 1).reshape(cout, -1).contiguous()
    out = o.conv_cl(x.to(DEV), wp.to(DEV), b.to(DEV), Tin=T, Hin=H, Win=W, Cin=cin, k=(1, 3, 3), stride=(1, 2, 2), pad=(0, 0, 0),
                    out_thw=(T, Ho, Wo))
    assert ref.shape[0] == T * Ho * Wo and rel_err(out.float().cpu(), ref) < BF16_TOL


@pytest.mark.parametrize("tsplit", [False, True])
@pytest.mark.parametrize("cin,cout", [(192, 96), (384, 192), (32, 40)])
def test_conv_cl_production_kernel_upsampled(cin, cout, tsplit):
    """The Resample up-sampling conv (wan_vae.py:61-67, 81-90, 138-141) on conv_halo_kernel: the conv reads a nearest-exact 2x view
    of x (ups) and, after the time_conv of upsample3d, frame f from channels (f & 1) * Cin of physical frame f >> 1 (tsplit)."""
    import torch.nn.functional as F
    o = ops()
    T, H, W = 3, 12, 20
    g = torch.Generator().manual_seed(1)
    x = torch.randn(T, H, W, cin * (2 if tsplit else 1), generator=g).bfloat16()
    w = (torch.randn(cout, cin, 3, 3, generator=g) * (cin * 9) ** -0.5).bfloat16()
    b = torch.randn(cout, generator=g).bfloat16()
    frames = x.float().view(T, H, W, 2, cin).permute(0, 3, 1, 2, 4).reshape(2 * T, H, W, cin) if tsplit else x.float()
    up = F.interpolate(frames.permute(0, 3, 1, 2), scale_factor=2.0, mode="nearest-exact")
    ref = F.conv2d(up, w.float(), b.float(), padding=1).permute(0, 2, 3, 1).reshape(-1, cout)
    wp = w.permute(0, 2, 3, 1).reshape(cout, -1).contiguous()
    out = o.conv_cl(x.to(DEV), wp.to(DEV), b.to(DEV), Tin=T, Hin=H, Win=W, Cin=cin, k=(1, 3, 3), pad=(0, 1, 1),
                    out_thw=(T * (2 if tsplit else 1), 2 * H, 2 * W), ups=True, tsplit=tsplit,
                    x_pixel_stride=cin * (2 if tsplit else 1))
    assert rel_err(out.float().cpu(), ref) < BF16_TOL


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("hd,heads,Lq,La,Lb", [(128, 2, 264, 200, 328), (64, 3, 40, 0, 72), (128, 4, 2304, 2304, 4608)])
def test_attn_merge_equals_joint_softmax(dtype, hd, heads, Lq, La, Lb):
    """m4d_attn_merge: attention over key set A merged with attention over key set B == one attention over A u B (the local-
    first schedule of the T-sharded self-attention); an empty side (lse = -inf) contributes nothing."""
    from more4d_amd import ops
    from more4d_amd.ops import KV
    B, C = 2, heads * hd
    g = torch.Generator().manual_seed(3)
    q = torch.randn(B * Lq, C, generator=g).to(dtype).to(DEV)
    rows_a, rows_b = max(La, 8), Lb
    ka, kb = (torch.randn(B * r, C, generator=g).to(dtype).to(DEV) for r in (rows_a, rows_b))
    va, vb = (torch.randn(C, B * r, generator=g).to(dtype).to(DEV) for r in (rows_a, rows_b))
    sa = KV(ka, va, rows_a * C, C, rows_a, B * rows_a, La)
    sb = KV(kb, vb, rows_b * C, C, rows_b, B * rows_b, Lb)
    kw = dict(B=B, Lq=Lq, heads=heads, head_dim=hd, q_bs=Lq * C, q_ls=C)
    lse_j = torch.empty(B, heads, Lq, device=DEV)
    joint = ops.attention(q, [sa, sb], lse=lse_j, **kw)
    la, lb = torch.empty_like(lse_j), torch.empty_like(lse_j)
    if La == 0:        # an empty side: what a rank without valid local keys contributes (the ABI rejects key-less calls)
        oa, la = torch.zeros_like(joint), torch.full_like(lse_j, -float("inf"))
    else:
        oa = ops.attention(q, [sa], lse=la, **kw)
    ob = ops.attention(q, [sb], lse=lb, **kw)
    ops.attn_merge_(oa, la, ob, lb, B=B, L=Lq, heads=heads, head_dim=hd)
    tol = 1e-5 if dtype == torch.float32 else BF16_TOL
    assert rel_err(oa.float().cpu(), joint.float().cpu()) < tol
    assert rel_err(la.cpu(), lse_j.cpu()) < 1e-5
