"""Training path of the DiT (reference: scripts/4D_STraG_training/train_wan.py:1939-1988 calls
`transformer3d(...)` under autocast and `accelerator.backward(loss)`).

torch.autograd is used ONLY as the tape: every node below computes its forward and its gradients with the HIP
kernels of libmore4d_hip.so (ops.*), so `loss.backward()`, DDP gradient hooks and torch optimizers keep working on
the reference-named parameters.  One `BlockFn` node per WanAttentionBlock implements the reference's per-block
gradient checkpointing (`enable_gradient_checkpointing`, wan_transformer4d.py:1273-1291): the forward keeps only the
block input (float32 residual stream), the backward recomputes the block and walks it in reverse.  Parameter
gradients leave each node as soon as that block is done, so DDP's bucketed all-reduce (RCCL) overlaps the
remaining backward.
"""
import torch
from torch.autograd import Function

from . import ops
from .ops import (ACT_GELU_TANH, EPI_STORE, EPI_STORE_F32, KV)


# ------------------------------------------------------------------ GEMM-shaped gradients
def _tpad(x):
    """x [R, C] -> x^T [C, Rp] with Rp = R rounded up (zero filled) so that the token axis can be the K of the
    production GEMM kernel (K % 64 == 0; small problems only need the 16-byte row alignment)."""
    R, C = x.shape
    Rp = -(-R // 64) * 64 if R >= 512 else -(-R // 8) * 8
    out = torch.empty((C, Rp), device=x.device, dtype=x.dtype)
    if Rp != R:
        out[:, R:].zero_()
    ops.transpose(x, out=out[:, :R] if Rp != R else out)
    return out


def linear_bwd(x, w, dy, need_dx=True):
    """y = x w^T + b.  x [R, K], w [N, K], dy [R, N] (all T) -> (dx [R, K] | None, dw [N, K], db float32 [N])."""
    db = ops.colsum(dy)[0]
    dx = ops.gemm_bt(dy, ops.transpose(w)) if need_dx else None          # dy [R,N] . (w^T)[K,N]^T
    dw = ops.gemm_bt(_tpad(dy), _tpad(x))                               # dy^T [N,Rp] . (x^T)[K,Rp]^T
    return dx, dw, db


class LinearFn(Function):
    """act(x w^T + b) computed in `cdt` (the kernels' T); output float32 when out_f32 (EPI_STORE_F32)."""

    @staticmethod
    def forward(ctx, x, weight, bias, act, cdt, out_f32):
        K = x.shape[-1]
        x2 = x.reshape(-1, K)
        x2 = x2.contiguous() if x2.dtype == cdt else ops.unary(x2.contiguous(), cdt)
        w = weight.detach().reshape(weight.shape[0], -1).to(cdt)
        b = bias.detach().to(cdt)
        pre = None
        if act:
            pre = ops.gemm_bt(x2, w, b)
            y = ops.unary(pre, torch.float32 if out_f32 else cdt, act=act)
        else:
            y = ops.gemm_bt(x2, w, b, epilogue=EPI_STORE_F32 if out_f32 else EPI_STORE)
        ctx.save_for_backward(x2, w, pre)
        ctx.meta = (act, cdt, x.shape, x.dtype, weight.shape, weight.dtype, bias.dtype)
        return y.view(*x.shape[:-1], w.shape[0])

    @staticmethod
    def backward(ctx, dy):
        x2, w, pre = ctx.saved_tensors
        act, cdt, xshape, xdt, wshape, wdt, bdt = ctx.meta
        N = w.shape[0]
        dy2 = dy.reshape(-1, N).contiguous()
        if dy2.dtype != cdt:
            dy2 = ops.unary(dy2, cdt)
        elif act and dy2.data_ptr() == dy.data_ptr():
            dy2 = dy2.clone()
        if act:
            ops.act_bwd_(dy2, pre, act)
        dx, dw, db = linear_bwd(x2, w, dy2, need_dx=ctx.needs_input_grad[0])
        if dx is not None:
            dx = dx.view(xshape)
            if dx.dtype != xdt:
                dx = ops.unary(dx, xdt)
        return dx, dw.view(wshape).to(wdt), db.to(bdt), None, None, None


class ActFn(Function):
    """act(x) -> out_dtype (1 silu, 2 gelu_tanh, 3 gelu_erf)."""

    @staticmethod
    def forward(ctx, x, act, out_dtype):
        x = x.contiguous()
        ctx.save_for_backward(x)
        ctx.act = act
        return ops.unary(x, out_dtype, act=act)

    @staticmethod
    def backward(ctx, dy):
        (x,) = ctx.saved_tensors
        d = ops.unary(dy.contiguous(), x.dtype) if dy.dtype != x.dtype else dy.contiguous().clone()
        ops.act_bwd_(d, x, ctx.act)
        return d, None, None


class LayerNormFn(Function):
    """ln_modulate on a float32 [B, n, C] input: affine (ln_w, ln_b [C]) or modulated (shift, scale [B, C]; [B, n, C] = one modulation
    vector per TOKEN: per-token timesteps, reference wan_transformer4d.py:713-715)."""

    @staticmethod
    def forward(ctx, x, ln_w, ln_b, shift, scale, eps, out_dtype):
        x = x.contiguous()
        B, n, C = x.shape
        w32 = ln_w.detach().float().contiguous() if ln_w is not None else None
        b32 = ln_b.detach().float().contiguous() if ln_b is not None else None
        sh = shift.detach().contiguous() if shift is not None else None
        sc = scale.detach().contiguous() if scale is not None else None
        per_token = sh is not None and sh.dim() == 3
        y = ops.ln_modulate(x, out_dtype, shift=sh, scale=sc, mod_stride=C, rows_per_sample=1 if per_token else n, ln_w=w32, ln_b=b32,
                            eps=eps)
        ctx.save_for_backward(x, w32, sc)
        ctx.meta = (eps, ln_w.dtype if ln_w is not None else None)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, w32, sc = ctx.saved_tensors
        eps, wdt = ctx.meta
        B, n, C = x.shape
        dy = dy.contiguous()
        dx = torch.zeros_like(x)
        per_token = sc is not None and sc.dim() == 3
        G = 1 if w32 is not None else (B * n if per_token else B)
        d1 = torch.zeros((G, C), device=x.device, dtype=torch.float32)
        d2 = torch.zeros((G, C), device=x.device, dtype=torch.float32)
        ops.ln_modulate_bwd(x, dy, dx, B=B * n if per_token else B, rows_per_sample=1 if per_token else n, scale=sc, mod_stride=C,
                            ln_w=w32, eps=eps, dshift=d1, dscale=d2, red_stride=0 if w32 is not None else C)
        if w32 is not None:
            return dx, d2[0].to(wdt), d1[0].to(wdt), None, None, None, None
        if per_token:
            return dx, None, None, d1.view(B, n, C), d2.view(B, n, C), None, None
        return dx, None, None, d1, d2, None, None


# ------------------------------------------------------------------ spatial-guidance feature adapter
_BILINEAR_T = {}


def _bilinear_matrix_t(hw, dtype, device):
    """Transpose [196, Pp] (token axis zero-padded like _tpad) of the matrix of the 14x14 -> hw bilinear resize, obtained by
    resizing the identity with the product's own kernel (so forward and backward use the same weights)."""
    key = (tuple(hw), dtype, str(device))
    m = _BILINEAR_T.get(key)
    if m is None:
        eye = torch.eye(196, device=device, dtype=torch.float32).view(1, 14, 14, 196)
        m = _tpad(ops.bilinear_cl(eye, hw).view(hw[0] * hw[1], 196).to(dtype))      # [196, Pp]
        _BILINEAR_T[key] = m
    return m


def _im2col3x3(x, B, D):
    """x [B*196, D] channels-last 14x14 maps -> [B*196, 9*D] rows in the packed-weight order (kh, kw, c); data movement only."""
    xp = torch.nn.functional.pad(x.view(B, 14, 14, D), (0, 0, 1, 1, 1, 1))
    return torch.cat([xp[:, i:i + 14, j:j + 14] for i in range(3) for j in range(3)], dim=-1).reshape(B * 196, 9 * D)


def _pack3x3(w, cdt):
    return w.detach().to(cdt).permute(0, 2, 3, 1).contiguous().view(w.shape[0], -1)


class GuidanceAdapterFn(Function):
    """SiLU(bilinear(feature_adapter(patch))) -> T [B, h*w, 768]: Conv3x3 - SiLU - Conv3x3 on the 14x14 OmniMAE map, bilinear
    resize to the token grid (reference wan_transformer4d.py:889-893, :1150-1152) and the SiLU every SpatialGuidanceModule
    applies first (:746).  apply(patch [B,196,768], hw, w0, b0, w2, b2, cdt); the features themselves get no gradient
    (the extractor is frozen, train_wan.py:952)."""

    @staticmethod
    def forward(ctx, patch, hw, w0, b0, w2, b2, cdt):
        B, D = patch.shape[0], patch.shape[-1]
        x0 = patch.detach().to(cdt).contiguous().view(B * 196, D)
        kw = dict(Tin=B, Hin=14, Win=14, Cin=D, k=(1, 3, 3), pad=(0, 1, 1), out_thw=(B, 14, 14))
        a1 = ops.conv_cl(x0, _pack3x3(w0, cdt), b0.detach().to(cdt).contiguous(), **kw)
        s1 = ops.unary(a1, cdt, act=1)
        a2 = ops.conv_cl(s1, _pack3x3(w2, cdt), b2.detach().to(cdt).contiguous(), **kw)
        y = ops.bilinear_cl(a2.view(B, 14, 14, D), hw).view(B, hw[0] * hw[1], D)
        ctx.save_for_backward(x0, a1, s1, y, w2)
        ctx.meta = (B, D, tuple(hw), cdt, w0.dtype, w0.shape)
        return ops.unary(y, cdt, act=1)

    @staticmethod
    def backward(ctx, dout):
        x0, a1, s1, y, w2 = ctx.saved_tensors
        B, D, hw, cdt, wdt, wshape = ctx.meta
        P = hw[0] * hw[1]
        d = dout.to(cdt).contiguous().clone().view(B * P, D)
        ops.act_bwd_(d, y.view(B * P, D), 1)                         # last SiLU
        mt = _bilinear_matrix_t(hw, cdt, d.device)                   # resize^T: [196, Pp] x [Pp, D]
        da2 = torch.empty((B * 196, D), device=d.device, dtype=cdt)
        for b in range(B):
            ops.gemm_bt(mt, _tpad(d[b * P:(b + 1) * P]), out=da2[b * 196:(b + 1) * 196])
        kw = dict(Tin=B, Hin=14, Win=14, Cin=D, k=(1, 3, 3), pad=(0, 1, 1), out_thw=(B, 14, 14))

        def wgrad(dy, x):                                            # dW [Cout, kh, kw, Cin] -> reference layout [Cout, Cin, kh, kw]
            dw = ops.gemm_bt(_tpad(dy), _tpad(_im2col3x3(x, B, D)))
            return dw.view(D, 3, 3, D).permute(0, 3, 1, 2).to(wdt), ops.colsum(dy)[0].to(wdt)

        dw2, db2 = wgrad(da2, s1)
        # data gradient of a stride-1 pad-1 conv = the same conv with the taps flipped and in/out channels swapped
        w2t = _pack3x3(w2.flip(2, 3).transpose(0, 1), cdt)
        ds1 = ops.conv_cl(da2, w2t, None, **kw)
        ops.act_bwd_(ds1, a1, 1)
        dw0, db0 = wgrad(ds1, x0)
        return None, None, dw0, db0, dw2, db2, None


# ------------------------------------------------------------------ one DiT block: recompute + backward
def _block_param_names(blk):
    return [n for n, _ in blk.named_parameters()]


STORE_KEYS = ("qkv_pre", "o", "lse1", "y1", "qc_pre", "yc", "pre", "y2")
STORE_LITE = ("o", "lse1", "y2")     # best recompute time saved per stored byte: self-attention output + ffn_down output


def _guidance_site_bwd(x, dz, shift, scale, st, Lp, eps, g, mod, gfeat2, G, prefix, mod_rows=0):
    """Gradients of one SpatialGuidanceModule application (reference :757-783) given dz = dL/d(guided LN output):
    dz becomes dL/d(plain LN-modulate output) in place; fills G[prefix.gate / .spatial_guide.1.*]; returns the gradient
    w.r.t. the SiLU'd guidance features [B*P, 768] (T)."""
    B, C = x.shape[0], x.shape[-1]
    T = gfeat2.dtype
    ab = ops.guidance_bwd_(x, dz, B=B, rows_per_sample=Lp, shift=shift, scale=scale, mod_stride=st, eps=eps, mod_rows=mod_rows, **g).view(-1, 2 * C)
    dg = ops.colsum(ab, g["g_ss"].view(-1, 2 * C))[0]                # sum (A*S | Bm*H)
    G[prefix + ".gate"] = ops.add(dg[:C].contiguous(), dg[C:].contiguous())
    dss = ops.scale_cast(ab, T, gate=torch.cat([g["g_gate"], g["g_gate"]]), gate_stride=0, rows_per_sample=ab.shape[0])
    lin = mod.spatial_guide[1]
    dfeat, G[prefix + ".spatial_guide.1.weight"], G[prefix + ".spatial_guide.1.bias"] = linear_bwd(gfeat2, lin.weight, dss)
    return dfeat


def block_backward(blk, x0, e0, c, txt, txt_len, img, img_len, dres, saved=None, forward_only=False, guid=None):
    """Recompute WanAttentionBlock.run on x0 keeping the intermediates, then back-propagate `dres`
    (float32 [B, Lp, C], gradient w.r.t. the block output; overwritten with the gradient w.r.t. x0).
    Returns (de0 [B,6,C] float32, dtxt, dimg, {param name: grad}).

    `saved` (STORE_KEYS -> tensor): outputs of the six big GEMMs and of the self-attention that the forward kept in
    HBM ("stored" blocks, see BlockFn); they are used instead of being recomputed — only the HBM-bound LayerNorm /
    RMSNorm / GELU passes and the small cross-attention run again.  forward_only=True runs just the forward half and
    returns (x3, {STORE_KEYS}) — the same code path produces the stored tensors, so both modes are bit-identical."""
    saved = saved or {}
    from .models.wan_transformer4d import _f32
    B, Lp, C = x0.shape
    R = B * Lp
    sa, ca = blk.self_attn, blk.cross_attn
    n, d = sa.num_heads, sa.head_dim
    T = blk.ffn[0].weight.dtype
    dev = x0.device
    st = 6 * C
    eps = blk.eps
    f32 = lambda p: _f32(p, c.f32cache)  # noqa: E731
    G = {}
    # qk_norm=False (reference wan_transformer4d.py:431-432): norm_q / norm_k are nn.Identity — the self-attention's norm + RoPE kernel
    # then only rotates (NULL weights, forward and backward), the cross-attention's norms vanish, and there are no norm-weight gradients
    nw = lambda mod: f32(mod.weight) if getattr(mod, "weight", None) is not None else None  # noqa: E731

    def zeros(*shape):
        return torch.zeros(shape, device=dev, dtype=torch.float32)

    # per-token timesteps (reference :655-657, `e.dim() > 3`): e0 [B, Lp, 6, C] — one modulation / gate vector per ROW.  The kernels take
    # "rows per modulation vector" (rps) and the number of vectors (nG), so the same calls serve both forms; the gradients of the vectors
    # come back per row (the reductions over a sample's rows degenerate) and are summed over ALL rows for the shared `modulation`
    per_token = e0.dim() == 4
    rps, nG = (1, R) if per_token else (Lp, B)
    e = ops.add_bcast(e0, f32(blk.modulation)).view(nG, 6, C)      # shift1 scale1 gate1 shift2 scale2 gate2
    de = zeros(nG, 6, C)
    g1, g2, gfeat2 = {}, {}, None
    if guid is not None and blk.spatial_guidance_self is not None:  # (SiLU'd features T [B, P, 768], period, length)
        gfeat, period, glen = guid
        gfeat2 = gfeat.reshape(-1, gfeat.shape[-1])
        for gd, mod in ((g1, blk.spatial_guidance_self), (g2, blk.spatial_guidance_ffn)):
            gd.update(g_ss=mod.table(gfeat, c.f32cache), g_gate=f32(mod.gate), g_period=period, g_len=glen)
            if per_token:      # modulation per row, guidance by the row's position inside its sample (m4d_ln_modulate_g / m4d_guidance_bwd_m)
                gd.update(g_rows=Lp)
    dres2 = dres.view(R, C) if dres is not None else None

    # ================= recompute (reference :659-684) =================
    xn1 = ops.ln_modulate(x0, T, shift=e[:, 0], scale=e[:, 1], mod_stride=st, rows_per_sample=rps, eps=eps, **g1).view(R, C)
    qkv_pre = saved.get("qkv_pre")
    if qkv_pre is None:
        qkv_pre = torch.empty((R, 3 * C), device=dev, dtype=T)
        for j, lin in enumerate((sa.q, sa.k, sa.v)):
            ops.gemm_bt(xn1, lin.weight, lin.bias, out=qkv_pre[:, j * C:(j + 1) * C])
    qkv = qkv_pre.clone()
    q, k, v = qkv[:, :C], qkv[:, C:2 * C], qkv[:, 2 * C:]
    rope = dict(head_dim=d, eps=sa.eps, cos=c.cos, sin=c.sin, rows_per_sample=Lp, rope_len=c.rope_len,
                pos_offset=c.pos_offset)
    # bf16 + qk_norm: the softmax scale (x log2 e) rides in q's RMSNorm weight, as in the inference path (models/wan_transformer4d.py:
    # _FOLD_QSCALE) — the attention calls then take scale = ln 2, the forward runs on attn128q_kernel with an exact log-sum-exp, and the
    # norm-weight gradient the backward kernel returns is d/d(w c): scaled back by c below
    from .models.wan_transformer4d import LOG2E, _FOLD_QSCALE, _folded_norm_weight
    fold = _FOLD_QSCALE and sa.qk_norm and T == torch.bfloat16
    qc_ = d ** -0.5 * LOG2E
    wq_self = _folded_norm_weight(sa.norm_q.weight, d, c.f32cache) if fold else nw(sa.norm_q)
    sm = dict(scale=1.0 / LOG2E) if fold else {}
    ops.rmsnorm_rope(q, wq_self, k, nw(sa.norm_k), **rope)
    o, lse1 = saved.get("o"), saved.get("lse1")
    if o is None:
        vt = ops.transpose(v)                                        # V^T [C, R]
        lse1 = torch.empty((B, n, Lp), device=dev, dtype=torch.float32)
        o = ops.attention(q, [KV(k, vt, Lp * 3 * C, 3 * C, Lp, R, c.key_len)], B=B, Lq=Lp, heads=n, head_dim=d,
                          q_bs=Lp * 3 * C, q_ls=3 * C, lse=lse1, **sm).view(R, C)
    y1 = saved.get("y1")
    if y1 is None:
        y1 = ops.gemm_bt(o, sa.o.weight, sa.o.bias)
    x1 = ops.resid_gate(x0, y1, gate=e[:, 2], gate_stride=st, rows_per_sample=rps)
    if blk.cross_attn_norm:
        xn3 = ops.ln_modulate(x1, T, ln_w=f32(blk.norm3.weight), ln_b=f32(blk.norm3.bias), eps=eps).view(R, C)
    else:
        xn3 = ops.unary(x1, T).view(R, C)
    qc_pre = saved.get("qc_pre")
    if qc_pre is None:
        qc_pre = ops.gemm_bt(xn3, ca.q.weight, ca.q.bias)
    qc = qc_pre.clone()
    if nw(ca.norm_q) is not None:
        ops.rmsnorm_rope(qc, nw(ca.norm_q), head_dim=d, eps=ca.eps)
    srcs = [("txt", txt, txt_len, ca.k, ca.v, ca.norm_k)]
    if img is not None and getattr(ca, "has_img", False):
        srcs.append(("img", img, img_len, ca.k_img, ca.v_img, ca.norm_k_img))
    cross = []
    oc = None
    for name, src, valid, kl, vl, nk in srcs:
        Bc, Sp, _ = src.shape
        s2 = src.reshape(Bc * Sp, C)
        k_pre = ops.gemm_bt(s2, kl.weight, kl.bias)
        kk = k_pre.clone()
        if nw(nk) is not None:
            ops.rmsnorm_rope(kk, nw(nk), head_dim=d, eps=ca.eps)
        vv = ops.gemm_bt(s2, vl.weight, vl.bias)
        vvt = ops.transpose(vv)
        lse = torch.empty((B, n, Lp), device=dev, dtype=torch.float32)
        oo = ops.attention(qc, [KV(kk, vvt, Sp * C, C, Sp, Bc * Sp, valid)], B=B, Lq=Lp, heads=n, head_dim=d,
                           q_bs=Lp * C, q_ls=C, lse=lse).view(R, C)
        oc = oo if oc is None else ops.add(oc, oo)
        cross.append((name, s2, Sp, valid, kl, vl, nk, k_pre, kk, vv, oo, lse))
    yc = saved.get("yc")
    if yc is None:
        yc = ops.gemm_bt(oc, ca.o.weight, ca.o.bias)
    x2 = ops.resid_gate(x1, yc)
    xn2 = ops.ln_modulate(x2, T, shift=e[:, 3], scale=e[:, 4], mod_stride=st, rows_per_sample=rps, eps=eps, **g2).view(R, C)
    pre = saved.get("pre")
    if pre is None:
        pre = ops.gemm_bt(xn2, blk.ffn[0].weight, blk.ffn[0].bias)
    h = ops.unary(pre, T, act=ACT_GELU_TANH)
    y2 = saved.get("y2")
    if y2 is None:
        y2 = ops.gemm_bt(h, blk.ffn[2].weight, blk.ffn[2].bias)
    if forward_only:
        x3 = ops.resid_gate(x2, y2, gate=e[:, 5], gate_stride=st, rows_per_sample=rps)
        return x3, dict(qkv_pre=qkv_pre, o=o, lse1=lse1, y1=y1, qc_pre=qc_pre, yc=yc, pre=pre, y2=y2)

    # ================= backward =================
    # ---- ffn: x3 = x2 + y2 * gate2
    de[:, 5].copy_(ops.colsum(dres2, y2, rows_per_group=rps))
    dy2 = ops.scale_cast(dres2, T, gate=e[:, 5], gate_stride=st, rows_per_sample=rps)
    del y2
    dh, G["ffn.2.weight"], G["ffn.2.bias"] = linear_bwd(h, blk.ffn[2].weight, dy2)
    del h, dy2
    ops.act_bwd_(dh, pre, ACT_GELU_TANH)
    del pre
    dxn2, G["ffn.0.weight"], G["ffn.0.bias"] = linear_bwd(xn2, blk.ffn[0].weight, dh)
    del dh
    dgf = None
    if g2:
        dgf = _guidance_site_bwd(x2, dxn2, e[:, 3], e[:, 4], st, Lp, eps, g2, blk.spatial_guidance_ffn, gfeat2, G,
                                 "spatial_guidance_ffn", mod_rows=rps)
    ops.ln_modulate_bwd(x2, dxn2, dres, B=nG, rows_per_sample=rps, scale=e[:, 4], mod_stride=st, eps=eps,
                        dshift=de[:, 3], dscale=de[:, 4], red_stride=st)
    # ---- cross attention: x2 = x1 + yc
    dyc = ops.scale_cast(dres2, T)
    doc, G["cross_attn.o.weight"], G["cross_attn.o.bias"] = linear_bwd(oc, ca.o.weight, dyc)
    dqc = torch.empty((R, C), device=dev, dtype=T)
    dctx = {}
    for i, (name, s2, Sp, valid, kl, vl, nk, k_pre, kk, vv, oo, lse) in enumerate(cross):
        dk = torch.empty_like(kk)
        dv = torch.empty_like(vv)
        ops.attention_bwd(qc, kk, vv, oo, doc, lse, B=B, Lq=Lp, Lk=valid, Lk_rows=Sp, heads=n, head_dim=d, dq=dqc, dk=dk,
                          dv=dv, accumulate_dq=i > 0)
        sfx = "_img" if name == "img" else ""
        if nw(nk) is not None:
            dwn = zeros(C)
            ops.rmsnorm_rope_bwd_(dk, k_pre, nw(nk), dwn, head_dim=d, eps=ca.eps)
            G[f"cross_attn.norm_k{sfx}.weight"] = dwn
        ds_k, G[f"cross_attn.k{sfx}.weight"], G[f"cross_attn.k{sfx}.bias"] = linear_bwd(s2, kl.weight, dk)
        ds_v, G[f"cross_attn.v{sfx}.weight"], G[f"cross_attn.v{sfx}.bias"] = linear_bwd(s2, vl.weight, dv)
        dctx[name] = ops.add(ds_k, ds_v).view(-1, Sp, C)
    if nw(ca.norm_q) is not None:
        dwq = zeros(C)
        ops.rmsnorm_rope_bwd_(dqc, qc_pre, nw(ca.norm_q), dwq, head_dim=d, eps=ca.eps)
        G["cross_attn.norm_q.weight"] = dwq
    dxn3, G["cross_attn.q.weight"], G["cross_attn.q.bias"] = linear_bwd(xn3, ca.q.weight, dqc)
    if blk.cross_attn_norm:
        dw3, db3 = zeros(C), zeros(C)
        ops.ln_modulate_bwd(x1, dxn3, dres, B=B, rows_per_sample=Lp, ln_w=f32(blk.norm3.weight), eps=eps, dshift=db3,
                            dscale=dw3, red_stride=0)
        G["norm3.weight"], G["norm3.bias"] = dw3, db3
    else:
        ops.resid_gate(dres, dxn3, out=dres)
    # ---- self attention: x1 = x0 + y1 * gate1
    de[:, 2].copy_(ops.colsum(dres2, y1, rows_per_group=rps))
    dy1 = ops.scale_cast(dres2, T, gate=e[:, 2], gate_stride=st, rows_per_sample=rps)
    do, G["self_attn.o.weight"], G["self_attn.o.bias"] = linear_bwd(o, sa.o.weight, dy1)
    dqkv = torch.empty((R, 3 * C), device=dev, dtype=T)
    ops.attention_bwd(q, k, v, o, do, lse1, B=B, Lq=Lp, Lk=c.key_len, Lk_rows=Lp, heads=n, head_dim=d,
                      dq=dqkv[:, :C], dk=dqkv[:, C:2 * C], dv=dqkv[:, 2 * C:], **sm)
    if sa.qk_norm:
        dwq, dwk = zeros(C), zeros(C)
        G["self_attn.norm_q.weight"], G["self_attn.norm_k.weight"] = dwq, dwk
    else:
        dwq = dwk = None
    ops.rmsnorm_rope_bwd_(dqkv[:, :C], qkv_pre[:, :C], wq_self, dwq, dqkv[:, C:2 * C], qkv_pre[:, C:2 * C], nw(sa.norm_k), dwk,
                          **rope)
    if fold:
        dwq.mul_(qc_)
    wt = torch.empty((C, 3 * C), device=dev, dtype=T)               # [Wq^T | Wk^T | Wv^T]
    for j, lin in enumerate((sa.q, sa.k, sa.v)):
        ops.transpose(lin.weight, out=wt[:, j * C:(j + 1) * C])
    dxn1 = ops.gemm_bt(dqkv, wt)
    dwqkv = ops.gemm_bt(_tpad(dqkv), _tpad(xn1))                     # [3C, C]
    dbqkv = ops.colsum(dqkv)[0]
    for j, nm in enumerate(("q", "k", "v")):
        G[f"self_attn.{nm}.weight"] = dwqkv[j * C:(j + 1) * C]
        G[f"self_attn.{nm}.bias"] = dbqkv[j * C:(j + 1) * C]
    if g1:
        dgf = ops.add(dgf, _guidance_site_bwd(x0, dxn1, e[:, 0], e[:, 1], st, Lp, eps, g1, blk.spatial_guidance_self, gfeat2,
                                              G, "spatial_guidance_self", mod_rows=rps))
    ops.ln_modulate_bwd(x0, dxn1, dres, B=nG, rows_per_sample=rps, scale=e[:, 1], mod_stride=st, eps=eps,
                        dshift=de[:, 0], dscale=de[:, 1], red_stride=st)
    G["modulation"] = ops.colsum(de.view(nG, 6 * C))[0].view(1, 6, C)
    if per_token:
        de = de.view(e0.shape)
    if guid is not None:
        return de, dctx.get("txt"), dctx.get("img"), G, (dgf.view(guid[0].shape) if dgf is not None else None)
    return de, dctx.get("txt"), dctx.get("img"), G


def _folds_qscale(blk):
    from .models.wan_transformer4d import _FOLD_QSCALE
    return bool(_FOLD_QSCALE and blk.self_attn.qk_norm)


class BlockFn(Function):
    """WanAttentionBlock on the tape.  apply(x, e0, txt, img, gfeat, blk, c, txt_len, img_len, store, gmeta, *params);
    gfeat = SiLU'd spatial-guidance features T [B, P, 768] (None without guidance), gmeta = (period, length).

    store=0: classic per-block gradient checkpointing (keep the block input, recompute everything in backward).
    store=2: the forward additionally keeps the outputs of the six big GEMMs and of the self-attention (STORE_KEYS,
    2.4 GB per block at L = 21 840, C = 5120) and the backward skips their recompute: fwd + bwd instead of fwd +
    recompute + bwd.  store=1 keeps only STORE_LITE (0.45 GB per block: the self-attention and ffn_down outputs, the
    best recompute time per byte).  `WanTransformer4DModel.activation_budget_gb` decides how many blocks get which."""

    @staticmethod
    def forward(ctx, x, e0, txt, img, gfeat, blk, c, txt_len, img_len, store, gmeta, *params):
        from .models.wan_transformer4d import ContextCache
        if c.sp is not None and c.sp.world_size > 1:
            raise NotImplementedError("training uses data parallelism; sequence parallelism is the inference path")
        stash = None
        guid = (gfeat.detach(), gmeta[0], gmeta[1]) if gfeat is not None else None
        if store:   # 2: every GEMM / attention output, 1: STORE_LITE only
            out, stash = block_backward(blk, x.detach(), e0.detach().contiguous(), c, txt.detach(), txt_len,
                                        img.detach() if img is not None else None, img_len, None, forward_only=True,
                                        guid=guid)
            if store == 1:
                stash = {k: stash[k] for k in STORE_LITE}
        elif blk.ffn[0].weight.dtype == torch.bfloat16 and not _folds_qscale(blk):
            # bf16 without the folded softmax scale (qk_norm=False, or M4D_FOLD_QSCALE=0): blk.run's self-attention would take attn128q_kernel,
            # which rounds Q * c to bf16 once more, while the backward's recompute (log-sum-exp requested) runs on attn128p_kernel — the loss
            # would come from other bits than the ones the gradient is taken of.  Same code path as the recompute instead.
            out, _ = block_backward(blk, x.detach(), e0.detach().contiguous(), c, txt.detach(), txt_len,
                                    img.detach() if img is not None else None, img_len, None, forward_only=True, guid=guid)
        else:
            out = x.detach().clone()
            cc = ContextCache()
            cc.txt, cc.txt_len, cc.img, cc.img_len = txt.detach(), txt_len, (img.detach() if img is not None else None), img_len
            blk.run(out, e0.detach().contiguous(), c, cc, 0, guid)
        ctx.save_for_backward(x, e0, txt, img, gfeat)
        ctx.blk, ctx.c, ctx.lens, ctx.stash, ctx.gmeta = blk, c, (txt_len, img_len), stash, gmeta
        ctx.names = _block_param_names(blk)
        return out

    @staticmethod
    def backward(ctx, dout):
        x, e0, txt, img, gfeat = ctx.saved_tensors
        blk = ctx.blk
        dres = dout.contiguous().clone()
        stash, ctx.stash = ctx.stash, None
        guid = (gfeat.detach(), ctx.gmeta[0], ctx.gmeta[1]) if gfeat is not None else None
        res = block_backward(blk, x.detach(), e0.detach().contiguous(), ctx.c, txt.detach(), ctx.lens[0],
                             img.detach() if img is not None else None, ctx.lens[1], dres, saved=stash, guid=guid)
        de, dtxt, dimg, G = res[:4]
        dgf = res[4] if guid is not None else None
        del stash
        grads = []
        for name, p in zip(ctx.names, blk.parameters()):
            g = G.get(name)
            grads.append(None if g is None else g.to(p.dtype).view(p.shape))
        return (dres, de, dtxt, dimg, dgf, None, None, None, None, None, None, *grads)
