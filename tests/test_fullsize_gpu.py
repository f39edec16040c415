"""BASELINE.json's FULL sizes (49x480x832: L = 21 840 tokens, d = 5120, 40 heads x 128; VAE 49x480x832x3) on the MI355X.
The CPU oracle needs minutes to hours at these sizes, so parity is checked (a) exactly on SAMPLED rows / queries /
pixels against fp32 torch arithmetic over the full reduction length, and (b) through size-independent properties of the
domain: key-permutation and segment-split invariance of attention, the log-sum-exp merge, linearity of the GEMM, causality
of the chunked VAE (a prefix of the frames encodes / decodes to the prefix of the result), determinism, batch consistency.
All calls go through the C ABI (ops.*) with the production bf16 kernels."""
import math

import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda"
BF = torch.bfloat16
L, C, HEADS, D, FFN = 21840, 5120, 40, 128, 13824


def gen(seed):
    return torch.Generator(device=DEV).manual_seed(seed)


def rel(a, b):
    return float((a.double() - b.double()).abs().max() / b.double().abs().max().clamp_min(1e-30))


def test_gemm_full_shape_sampled_rows_and_linearity():
    """gemm_bt256p_kernel at the bench shape of a CFG pair (M = 43 680, ragged last row tile) for N x K = 5120 x 5120 and
    13824 x 5120: sampled rows (first / middle / last tiles) against fp32 matmul over the full K; additivity in A."""
    from more4d_amd import ops
    M = 2 * L
    for N, K in ((C, C), (FFN, C), (C, FFN)):
        a = (torch.randn(M, K, generator=gen(1), device=DEV) * 0.5).to(BF)
        w = (torch.randn(N, K, generator=gen(2), device=DEV) * K ** -0.5).to(BF)
        b = torch.randn(N, generator=gen(3), device=DEV).to(BF)
        out = ops.gemm_bt(a, w, b)
        rows = torch.tensor([0, 1, 255, 256, 257, 21839, 21840, 43519, 43520, 43679], device=DEV)
        ref = a[rows].float() @ w.float().t() + b.float()
        assert rel(out[rows].float(), ref) < 8e-3                      # bf16 output rounding (2^-8 of the largest value)
        a2 = (torch.randn(M, K, generator=gen(4), device=DEV) * 0.5).to(BF)
        s = ops.gemm_bt((a.float() + a2.float()).to(BF), w)
        parts = ops.gemm_bt(a, w, epilogue=ops.EPI_STORE_F32) + ops.gemm_bt(a2, w, epilogue=ops.EPI_STORE_F32)
        assert rel(s[rows].float(), parts[rows]) < 2e-2                # (a + a2) is rounded to bf16 before the product
        assert torch.equal(out, ops.gemm_bt(a, w, b))                  # deterministic
        del a, a2, w, out, s, parts


def test_attention_full_length_sampled_queries_and_invariances():
    """attn128p_kernel over all 21 840 keys, 40 heads: sampled queries against an fp32 softmax over the full key axis;
    permuting the keys, splitting them into 4 segments (the T-sharded hand-over) and merging partial softmaxes change nothing
    beyond bf16 rounding; the LSE matches."""
    from more4d_amd import ops
    from more4d_amd.ops import KV
    B = 1
    q = torch.randn(B * L, C, generator=gen(1), device=DEV).to(BF)
    k = torch.randn(B * L, C, generator=gen(2), device=DEV).to(BF)
    vt = torch.randn(C, B * L, generator=gen(3), device=DEV).to(BF)
    kw = dict(B=B, Lq=L, heads=HEADS, head_dim=D, q_bs=L * C, q_ls=C)
    lse = torch.empty(B, HEADS, L, device=DEV)
    out = ops.attention(q, [KV(k, vt, L * C, C, L, B * L, L)], lse=lse, **kw).view(L, C)
    # (a) exact on samples: 3 heads x 6 queries over ALL keys
    qs = torch.tensor([0, 127, 128, 10000, 21712, 21839], device=DEV)
    for h in (0, 17, 39):
        sl = slice(h * D, (h + 1) * D)
        s = (q[qs, sl].float() @ k[:, sl].float().t()) / math.sqrt(D)
        ref = torch.softmax(s, -1) @ vt[sl].float().t()
        assert rel(out[qs, sl].float(), ref) < 8e-3
        assert rel(lse[0, h, qs], torch.logsumexp(s, -1) * 1.4426950408889634) < 1e-4
    # (b) key permutation
    perm = torch.randperm(L, generator=gen(4), device=DEV)
    outp = ops.attention(q, [KV(k[perm].contiguous(), vt[:, perm].contiguous(), L * C, C, L, B * L, L)], **kw).view(L, C)
    assert rel(outp.float(), out.float()) < 8e-3
    # (c) four ragged segments == one segment
    Ls = 5464
    segs = []
    for r in range(4):
        n = min(Ls, L - r * Ls)
        kk = torch.zeros(Ls, C, device=DEV, dtype=BF)
        vv = torch.zeros(C, Ls, device=DEV, dtype=BF)
        kk[:n] = k[r * Ls:r * Ls + n]
        vv[:, :n] = vt[:, r * Ls:r * Ls + n]
        segs.append(KV(kk, vv, Ls * C, C, Ls, Ls, n))
    outs = ops.attention(q, segs, **kw).view(L, C)
    assert rel(outs.float(), out.float()) < 8e-3
    # (d) local shard + remote shards merged through the LSEs
    la, lb = torch.empty_like(lse), torch.empty_like(lse)
    oa = ops.attention(q, segs[:1], lse=la, **kw)
    ob = ops.attention(q, segs[1:], lse=lb, **kw)
    ops.attn_merge_(oa, la, ob, lb, B=B, L=L, heads=HEADS, head_dim=D)
    assert rel(oa.view(L, C).float(), out.float()) < 8e-3 and rel(la, lse) < 1e-4


def test_norm_kernels_full_rows_sampled():
    """LayerNorm+modulate and RMSNorm+RoPE at [2, 21 840, 5120]: sampled rows against fp32 torch; idempotent shapes."""
    from more4d_amd import ops
    B = 2
    x = torch.randn(B, L, C, generator=gen(1), device=DEV) * 3 + 0.5
    e = torch.randn(B, 6, C, generator=gen(2), device=DEV) * 0.2
    y = ops.ln_modulate(x, BF, shift=e[:, 0], scale=e[:, 1], mod_stride=6 * C, rows_per_sample=L, eps=1e-6)
    rows = torch.tensor([0, 1, 4095, 21839], device=DEV)
    for b in range(B):
        xr = x[b, rows]
        ref = torch.nn.functional.layer_norm(xr, (C,), eps=1e-6) * (1 + e[b, 1]) + e[b, 0]
        assert rel(y[b, rows].float(), ref) < 8e-3
    qk = torch.randn(B * L, C, generator=gen(3), device=DEV).to(BF)
    w = 1 + 0.1 * torch.randn(C, generator=gen(4), device=DEV)
    got = qk.clone()
    ops.rmsnorm_rope(got, w, head_dim=D, eps=1e-6)           # no rope tables: pure WanRMSNorm over the full 5120 row
    xr = qk[rows].float()
    ref = (xr * torch.rsqrt(xr.pow(2).mean(-1, keepdim=True) + 1e-6)).to(BF).float() * w
    assert rel(got[rows].float(), ref) < 8e-3


def test_block_full_size_batch_consistency_and_determinism():
    """One 14B-width WanAttentionBlock at L = 21 840 (bf16): two identical samples in a batch give identical rows, a second run
    is bit-identical, and the output is finite (the numerics themselves are pinned at L = 260 against the reference)."""
    from more4d_amd.models import WanAttentionBlock
    torch.manual_seed(0)
    with torch.device(DEV):
        blk = WanAttentionBlock("i2v_cross_attn", C, FFN, HEADS, (-1, -1), True, True, 1e-6, use_spatial_guidance=False)
    with torch.no_grad():
        for n, p_ in blk.named_parameters():
            if n.endswith("weight") and p_.dim() == 1:
                p_.fill_(1.0)
            elif n.endswith("bias"):
                p_.zero_()
            else:
                p_.normal_(0, 0.02)
    blk = blk.to(BF).eval()
    grid = (14, 30, 52)
    from more4d_amd.models.wan_transformer4d import rope_params
    freqs = torch.cat([rope_params(1024, D - 4 * (D // 6)), rope_params(1024, 2 * (D // 6)), rope_params(1024, 2 * (D // 6))], dim=1)
    x1 = torch.randn(1, L, C, generator=gen(1), device=DEV)
    x = torch.cat([x1, x1])
    e = (torch.randn(1, 6, C, generator=gen(2), device=DEV) * 0.1).expand(2, -1, -1).contiguous()
    ctx1 = torch.randn(1, 257 + 512, C, generator=gen(3), device=DEV).to(BF)
    ctx = torch.cat([ctx1, ctx1])
    args = dict(seq_lens=torch.tensor([L, L]), grid_sizes=torch.tensor([grid, grid]), freqs=freqs, context=ctx, context_lens=None)
    with torch.no_grad():
        a = blk(x, e, **args)
        b = blk(x, e, **args)
    assert bool(torch.isfinite(a).all()) and torch.equal(a, b)
    assert torch.equal(a[0], a[1])
    assert float((a[0] - x1[0]).abs().mean()) > 1e-3          # the block did something


def test_vae_full_size_causality():
    """Motion-Sensitive VAE at 480x832: the chunked causal encoder / decoder are prefix-consistent — the first 9 frames alone
    encode to the first 3 latent frames of the 17-frame encode, and 3 latent frames decode to the first 9 frames of the 5-frame
    decode (wan_vae.py:520-547, 678-703: every conv is causal in T, chunks only see cached past frames)."""
    from more4d_amd.models.wan_vae import AutoencoderKLWan
    torch.manual_seed(0)
    vae = AutoencoderKLWan().eval()
    with torch.no_grad():
        for n, p_ in vae.named_parameters():
            if n.endswith("gamma"):
                p_.fill_(1.0)
            elif p_.dim() > 1:
                p_.normal_(0, (p_[0].numel()) ** -0.5)
            else:
                p_.zero_()
    vae = vae.to(DEV, BF)
    x = (torch.randn(1, 3, 17, 480, 832, generator=gen(1), device=DEV) * 0.3).to(BF)
    with torch.no_grad():
        full = vae.encode(x)[0].mode()
        part = vae.encode(x[:, :, :9].contiguous())[0].mode()
        assert full.shape == (1, 16, 5, 60, 104) and part.shape == (1, 16, 3, 60, 104)
        assert torch.equal(part, full[:, :, :3])
        dec_full = vae.decode(full).sample
        dec_part = vae.decode(full[:, :, :3].contiguous()).sample
    assert dec_full.shape == (1, 3, 17, 480, 832) and torch.equal(dec_part, dec_full[:, :, :9])
    assert bool(torch.isfinite(dec_full.float()).all()) and float(dec_full.float().abs().max()) <= 1.0
    # the planar-16 staging layout (default for these maps) is only a layout: channels-last staging gives the same bits
    from more4d_amd.models import wan_vae
    was = wan_vae._Runner.PLANAR
    wan_vae._Runner.PLANAR = False
    try:
        with torch.no_grad():
            assert torch.equal(vae.encode(x)[0].mode(), full)
            assert torch.equal(vae.decode(full).sample, dec_full)
    finally:
        wan_vae._Runner.PLANAR = was


def test_attention_backward_full_length_sampled():
    """attn_bwd128 passes at L = 21 840 (B = 1, 8 of the 40 heads to bound memory): dQ on sampled queries, dK / dV on sampled
    keys against fp32 formulas over the FULL other axis (P from the forward's LSE; dS = P (dP - delta) scale)."""
    from more4d_amd import ops
    from more4d_amd.ops import KV
    heads, Cc = 8, 8 * D
    q = torch.randn(L, Cc, generator=gen(1), device=DEV).to(BF)
    k = torch.randn(L, Cc, generator=gen(2), device=DEV).to(BF)
    v = torch.randn(L, Cc, generator=gen(3), device=DEV).to(BF)
    d_o = (torch.randn(L, Cc, generator=gen(4), device=DEV) * 0.1).to(BF)
    lse = torch.empty(1, heads, L, device=DEV)
    o = ops.attention(q, [KV(k, ops.transpose(v), L * Cc, Cc, L, L, L)], B=1, Lq=L, heads=heads, head_dim=D, q_bs=L * Cc,
                      q_ls=Cc, lse=lse).view(L, Cc)
    dq, dk, dv = torch.empty_like(q), torch.empty_like(k), torch.empty_like(v)
    ops.attention_bwd(q, k, v, o, d_o, lse, B=1, Lq=L, Lk=L, Lk_rows=L, heads=heads, head_dim=D, dq=dq, dk=dk, dv=dv)
    scale = 1.0 / math.sqrt(D)
    ln2 = math.log(2.0)
    idx = torch.tensor([0, 63, 64, 9999, 21776, 21839], device=DEV)
    for h in (0, 5):
        sl = slice(h * D, (h + 1) * D)
        qf, kf, vf, of, gf = (t[:, sl].float() for t in (q, k, v, o, d_o))
        lse_nat = lse[0, h] * ln2                                      # [L]
        delta = (gf * of).sum(-1)                                      # [L]
        # sampled queries: full rows of P and dS
        P = torch.exp(qf[idx] @ kf.t() * scale - lse_nat[idx, None])   # [6, L]
        dS = P * (gf[idx] @ vf.t() - delta[idx, None]) * scale
        assert rel(dq[idx, sl].float(), dS @ kf) < 1e-2
        # sampled keys: full columns
        Pc = torch.exp(qf @ kf[idx].t() * scale - lse_nat[:, None])    # [L, 6]
        dSc = Pc * (gf @ vf[idx].t() - delta[:, None]) * scale
        assert rel(dv[idx, sl].float(), Pc.t() @ gf) < 1e-2
        assert rel(dk[idx, sl].float(), dSc.t() @ qf) < 1e-2


def test_linear_backward_full_token_axis_sampled():
    """wgrad / dgrad / bias gradient of a Linear over all 21 840 tokens (the token axis is the GEMM K of the wgrad, zero-padded to
    a multiple of 64): sampled entries against fp32 over the full token axis."""
    from more4d_amd.autograd import linear_bwd
    x = (torch.randn(L, C, generator=gen(1), device=DEV) * 0.5).to(BF)
    w = (torch.randn(C, C, generator=gen(2), device=DEV) * C ** -0.5).to(BF)
    dy = (torch.randn(L, C, generator=gen(3), device=DEV) * 0.1).to(BF)
    dx, dw, db = linear_bwd(x, w, dy)
    r = torch.tensor([0, 255, 256, 5119], device=DEV)
    assert rel(dw[r].float(), dy[:, r].float().t() @ x.float()) < 8e-3
    t = torch.tensor([0, 21583, 21584, 21839], device=DEV)
    assert rel(dx[t].float(), dy[t].float() @ w.float()) < 8e-3
    assert rel(db, dy.float().sum(0)) < 1e-4
