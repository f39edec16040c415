"""Tensor-level wrappers over the C ABI (include/more4d_hip.h).  torch supplies device memory and the
current HIP stream; every arithmetic op below runs in a hand-written gfx950 kernel.  No fallback."""
import math
import os

import torch

from . import _lib
from ._lib import (EPI_GELU_ERF, EPI_GELU_TANH, EPI_RESID_GATE, EPI_SILU, EPI_STORE, EPI_STORE_F32,  # noqa: F401
                   KvSegs, check)

_DT = {torch.float32: _lib.M4D_F32, torch.bfloat16: _lib.M4D_BF16}


def dt_code(dtype):
    try:
        return _DT[dtype]
    except KeyError:
        raise TypeError(f"more4d_amd kernels support float32 and bfloat16, got {dtype}") from None


def _stream():
    return torch.cuda.current_stream().cuda_stream


def _dev(*ts):
    for t in ts:
        if t is not None and not t.is_cuda:
            raise _lib.More4DHipError("more4d_amd kernels need tensors on the HIP device (got a CPU tensor); "
                                      "there is no CPU path")


def _ptr(t):
    return 0 if t is None else t.data_ptr()


def _rows2d(t):
    """(rows, row stride) of a tensor viewed as [rows, last]; last dim must be contiguous."""
    if t.stride(-1) != 1:
        raise ValueError("last dimension must be contiguous")
    if t.dim() == 1:
        return 1, t.shape[0]
    if t.dim() == 2:
        return t.shape[0], t.stride(0)
    # leading dims must collapse onto the row stride
    ld = t.stride(-2)
    rows = t.shape[-2]
    for d in range(t.dim() - 3, -1, -1):
        if t.shape[d] != 1 and t.stride(d) != rows * ld:
            raise ValueError(f"tensor of shape {tuple(t.shape)} / strides {t.stride()} is not a strided 2-D matrix")
        rows *= t.shape[d]
    return rows, ld


class Packed:
    """A weight re-laid-out by `pack_frag` (MFMA fragment order) together with its logical shape."""
    __slots__ = ("data", "rows", "K")

    def __init__(self, data, rows, K):
        self.data, self.rows, self.K = data, rows, K


KERNEL_CLASSES = ("gemm_phased", "gemm_wide", "gemm_generic", "attn_phased", "attn_other", "conv_halo_mt3_12x32",
                  "conv_halo_mt3_24x16", "conv_halo", "conv_generic", "conv_fused_norm", "conv_fused_norm_resid", "conv_gnstats",
                  "attn_bwd128", "attn_bwd_generic", "attn_xp", "attn_q64", "attn_bwd64", "conv_halo64")


def launch_counts(reset=False):
    """{kernel class: launches} of this process (m4d_launch_count; include/more4d_hip.h: m4d_kernel_class) — diagnostics for tests
    that must prove which tile path a module-level case ran."""
    lib = _lib.load()
    out = {n: int(lib.m4d_launch_count(i, 0)) for i, n in enumerate(KERNEL_CLASSES)}
    if reset:
        lib.m4d_launch_count(-1, 1)
    return out


def pack_frag(w):
    """bf16 weight [rows, K] (row-strided) -> Packed (fragment order, rows padded to 32)."""
    _dev(w)
    rows, ld = _rows2d(w)
    K = w.shape[-1]
    lib = _lib.load()
    out = torch.empty(lib.m4d_pack_frag_elems(rows, K), device=w.device, dtype=w.dtype)
    check(lib.m4d_pack_frag(dt_code(w.dtype), _ptr(w), ld, _ptr(out), rows, K, _stream()), "m4d_pack_frag")
    return Packed(out, rows, K)


def packed_ok(M, N, K, dtype):
    """Shapes the packed production kernel takes (otherwise callers use the row-major weight)."""
    return dtype == torch.bfloat16 and K % 64 == 0 and M >= 512 and N >= 512


def gemm_bt(a, w, bias=None, *, out=None, epilogue=EPI_STORE, gate=None, gate_stride=0, rows_per_sample=0,
            bias_on_m=False, out_rows_ld=None):
    """out = epilogue(a @ w^T + bias).  a [M,K] (row-strided), w [N,K] (row-strided).
    EPI_RESID_GATE / EPI_STORE_F32 write float32 `out`; others write a.dtype.
    Either operand may be a `Packed` weight (pack_frag): the production bf16 kernel streams it from VGPRs."""
    if isinstance(a, Packed) or isinstance(w, Packed):
        return _gemm_bt_packed(a, w, bias, out=out, epilogue=epilogue, gate=gate, gate_stride=gate_stride,
                               rows_per_sample=rows_per_sample, bias_on_m=bias_on_m)
    _dev(a, w, bias, out, gate)
    if a.dtype != w.dtype:
        raise TypeError(f"gemm_bt: A {a.dtype} vs W {w.dtype}")
    M, lda = _rows2d(a)
    N, ldw = _rows2d(w)
    K = a.shape[-1]
    if w.shape[-1] != K:
        raise ValueError(f"gemm_bt: K mismatch {K} vs {w.shape[-1]}")
    f32_out = epilogue in (EPI_RESID_GATE, EPI_STORE_F32)
    if out is None:
        if epilogue == EPI_RESID_GATE:
            raise ValueError("gemm_bt: EPI_RESID_GATE needs the residual tensor as `out`")
        out = torch.empty((M, N), device=a.device, dtype=torch.float32 if f32_out else a.dtype)
    want = torch.float32 if f32_out else a.dtype
    if out.dtype != want:
        raise TypeError(f"gemm_bt: out dtype {out.dtype}, expected {want}")
    om, ldc = _rows2d(out)
    if om != M or out.shape[-1] != N:
        raise ValueError(f"gemm_bt: out shape {tuple(out.shape)} vs M={M} N={N}")
    if bias is not None and (bias.dtype != a.dtype or bias.numel() != (M if bias_on_m else N)):
        raise ValueError("gemm_bt: bias must have A's dtype and N (or M) elements")
    if gate is not None and gate.dtype != torch.float32:
        raise TypeError("gemm_bt: gate must be float32")
    lib = _lib.load()
    if _gemm_tail_mode():          # opt-in split-K tail (M4D_GEMM_TAIL=1 with the phased kernel): the only caller of the workspace ABI
        nws = lib.m4d_gemm_bt_workspace_bytes(dt_code(a.dtype), M, N, K)
        if nws:
            ws = _gemm_workspace(a.device, nws)
            check(lib.m4d_gemm_bt_ws(dt_code(a.dtype), _ptr(a), lda, _ptr(w), ldw, _ptr(bias), int(bias_on_m), _ptr(out), ldc,
                                     M, N, K, epilogue, _ptr(gate), gate_stride, rows_per_sample, _ptr(ws), nws, _stream()), "m4d_gemm_bt_ws")
            return out
    check(lib.m4d_gemm_bt(dt_code(a.dtype), _ptr(a), lda, _ptr(w), ldw, _ptr(bias), int(bias_on_m), _ptr(out), ldc,
                          M, N, K, epilogue, _ptr(gate), gate_stride, rows_per_sample, _stream()), "m4d_gemm_bt")
    return out


_GEMM_WS = {}
_GEMM_TAIL = None


def _gemm_tail_mode():
    global _GEMM_TAIL
    if _GEMM_TAIL is None:
        import os
        _GEMM_TAIL = os.environ.get("M4D_GEMM_TAIL", "0") not in ("", "0")
    return _GEMM_TAIL


def _gemm_workspace(device, nbytes):
    """Per-device split-K workspace of m4d_gemm_bt_ws (grown on demand; GEMMs on one stream are ordered, so one buffer serves all)."""
    key = (device.type, device.index, torch.cuda.current_stream().cuda_stream)
    if len(_GEMM_WS) > 8 and key not in _GEMM_WS:      # streams come and go: keep the table bounded
        _GEMM_WS.clear()
    buf = _GEMM_WS.get(key)
    if buf is None or buf.numel() < nbytes:
        buf = torch.empty(nbytes, device=device, dtype=torch.uint8)
        _GEMM_WS[key] = buf
    return buf


def _gemm_bt_packed(a, w, bias, *, out, epilogue, gate, gate_stride, rows_per_sample, bias_on_m):
    side = 1 if isinstance(a, Packed) else 0
    pk, act = (a, w) if side else (w, a)
    _dev(pk.data, act, bias, out, gate)
    rows_act, ld_act = _rows2d(act)
    K = act.shape[-1]
    if pk.K != K:
        raise ValueError(f"gemm_bt(packed): K mismatch {pk.K} vs {K}")
    M, N = (pk.rows, rows_act) if side else (rows_act, pk.rows)
    f32_out = epilogue in (EPI_RESID_GATE, EPI_STORE_F32)
    if out is None:
        if epilogue == EPI_RESID_GATE:
            raise ValueError("gemm_bt: EPI_RESID_GATE needs the residual tensor as `out`")
        out = torch.empty((M, N), device=act.device, dtype=torch.float32 if f32_out else act.dtype)
    if out.dtype != (torch.float32 if f32_out else act.dtype):
        raise TypeError("gemm_bt(packed): out dtype")
    om, ldc = _rows2d(out)
    if om != M or out.shape[-1] != N:
        raise ValueError(f"gemm_bt(packed): out shape {tuple(out.shape)} vs M={M} N={N}")
    if bias is not None and (bias.dtype != act.dtype or bias.numel() != (M if bias_on_m else N)):
        raise ValueError("gemm_bt(packed): bias must have the activation dtype and N (or M) elements")
    lib = _lib.load()
    A_ptr, lda, W_ptr, ldw = (_ptr(pk.data), K, _ptr(act), ld_act) if side else (_ptr(act), ld_act, _ptr(pk.data), K)
    check(lib.m4d_gemm_bt_packed(dt_code(act.dtype), A_ptr, lda, W_ptr, ldw, side, _ptr(bias), int(bias_on_m), _ptr(out), ldc,
                                 M, N, K, epilogue, _ptr(gate), gate_stride, rows_per_sample, _stream()),
          "m4d_gemm_bt_packed")
    return out


def ln_modulate(x, out_dtype, *, shift=None, scale=None, mod_stride=0, rows_per_sample=0, ln_w=None, ln_b=None,
                eps=1e-6, g_ss=None, g_gate=None, g_period=0, g_len=0, g_rows=0, out=None):
    """LayerNorm over the last dim (+affine) (+modulate) (+spatial guidance) -> out_dtype, same shape.  g_rows: rows per guidance
    sample when it differs from rows_per_sample (per-token modulation: rows_per_sample = 1, g_rows = tokens per sample)."""
    _dev(x, shift, scale, ln_w, ln_b, g_ss, g_gate, out)
    if not x.is_contiguous():
        raise ValueError("ln_modulate: x must be contiguous")
    C = x.shape[-1]
    rows = x.numel() // C
    if out is None:
        out = torch.empty(x.shape, device=x.device, dtype=out_dtype)
    for t in (shift, scale, ln_w, ln_b, g_ss, g_gate):
        if t is not None and t.dtype != torch.float32:
            raise TypeError("ln_modulate: modulation / affine / guidance tensors must be float32")
    lib = _lib.load()
    check(lib.m4d_ln_modulate_g(dt_code(x.dtype), _ptr(x), dt_code(out.dtype), _ptr(out), rows, C, rows_per_sample,
                                _ptr(shift), _ptr(scale), mod_stride, _ptr(ln_w), _ptr(ln_b), eps, _ptr(g_ss),
                                _ptr(g_gate), g_period, g_len, g_rows, _stream()), "m4d_ln_modulate_g")
    return out


def rmsnorm_rope(x0, w0, x1=None, w1=None, *, head_dim, eps=1e-6, cos=None, sin=None, rows_per_sample=0,
                 rope_len=0, pos_offset=0):
    """In-place RMSNorm over the last dim (+RoPE) on x0 (and x1)."""
    _dev(x0, x1, w0, w1, cos, sin)
    rows, ld = _rows2d(x0)
    C = x0.shape[-1]
    if x1 is not None:
        r1, ld1 = _rows2d(x1)
        if r1 != rows or ld1 != ld or x1.dtype != x0.dtype or x1.shape[-1] != C:
            raise ValueError("rmsnorm_rope: x0/x1 layout mismatch")
    for t in (w0, w1, cos, sin):
        if t is not None and (t.dtype != torch.float32 or not t.is_contiguous()):
            raise TypeError("rmsnorm_rope: weights and rope tables must be contiguous float32")
    if cos is not None and cos.shape[-1] != head_dim // 2:
        raise ValueError("rmsnorm_rope: rope table must have head_dim/2 columns")
    if cos is not None and pos_offset + min(rope_len, rows_per_sample or rows) > cos.shape[0]:
        raise ValueError("rmsnorm_rope: rope table too short")
    lib = _lib.load()
    check(lib.m4d_rmsnorm_rope(dt_code(x0.dtype), _ptr(x0), _ptr(x1), ld, _ptr(w0), _ptr(w1), rows, C, head_dim, eps,
                               _ptr(cos), _ptr(sin), rows_per_sample, rope_len, pos_offset, _stream()),
          "m4d_rmsnorm_rope")
    return x0, x1


class KV:
    """One K/V segment for `attention`: k [B, Lk, C] (token rows strided), vt = V^T [C, >=...] with the keys of
    batch b starting at column b*vt_bs."""
    __slots__ = ("k", "vt", "k_bs", "k_ls", "vt_bs", "vt_ls", "len")

    def __init__(self, k, vt, k_bs, k_ls, vt_bs, vt_ls, length):
        self.k, self.vt, self.k_bs, self.k_ls, self.vt_bs, self.vt_ls, self.len = k, vt, k_bs, k_ls, vt_bs, vt_ls, length


def attention(q, segs, *, B, Lq, heads, head_dim, out=None, q_bs=None, q_ls=None, accumulate=False, scale=None, lse=None,
              new_softmax=0):
    """softmax(q k^T * scale) v over the concatenation of `segs` (list of KV).  q/out: [B, Lq, heads*head_dim].
    lse: optional float32 [B, heads, Lq] receiving the log2-domain log-sum-exp (training).
    new_softmax: bit s set = segment s starts a new softmax whose output is ADDED to the previous ones (one launch for the text +
    image branches of the i2v cross-attention; bf16, head_dim 128)."""
    _dev(q, out, *[s.k for s in segs], *[s.vt for s in segs])
    C = heads * head_dim
    if q_ls is None:
        q_ls = q.stride(-2)
    if q_bs is None:
        q_bs = q.stride(0) if q.dim() == 3 else Lq * q_ls
    if out is None:
        out = torch.empty((B, Lq, C), device=q.device, dtype=q.dtype)
    if not 1 <= len(segs) <= _lib.MAX_KV_SEGS:
        raise ValueError(f"attention: 1..{_lib.MAX_KV_SEGS} KV segments, got {len(segs)}")
    kv = KvSegs()
    kv.nseg = len(segs)
    kv.new_softmax = int(new_softmax)
    for i, s in enumerate(segs):
        if s.k.dtype != q.dtype or s.vt.dtype != q.dtype:
            raise TypeError("attention: q/k/v dtype mismatch")
        kv.k[i], kv.vt[i] = s.k.data_ptr(), s.vt.data_ptr()
        kv.k_bs[i], kv.k_ls[i], kv.vt_bs[i], kv.vt_ls[i], kv.len[i] = s.k_bs, s.k_ls, s.vt_bs, s.vt_ls, s.len
    if scale is None:
        scale = 1.0 / math.sqrt(head_dim)
    lib = _lib.load()
    if lse is not None:
        _dev(lse)
        if lse.dtype != torch.float32 or not lse.is_contiguous() or lse.numel() != B * heads * Lq:
            raise ValueError("attention: lse must be contiguous float32 [B, heads, Lq]")
        check(lib.m4d_attention_lse(dt_code(q.dtype), _ptr(q), q_bs, q_ls, kv, _ptr(out), out.stride(0), out.stride(1),
                                    B, Lq, heads, head_dim, scale, int(accumulate), _ptr(lse), _stream()),
              "m4d_attention_lse")
        return out
    check(lib.m4d_attention(dt_code(q.dtype), _ptr(q), q_bs, q_ls, kv, _ptr(out), out.stride(0), out.stride(1), B, Lq,
                            heads, head_dim, scale, int(accumulate), _stream()), "m4d_attention")
    return out


def attn_merge_(o_a, lse_a, o_b, lse_b, *, B, L, heads, head_dim):
    """In place: o_a, lse_a <- the attention over the union of two disjoint key sets, from the partial results (o_a, lse_a) and
    (o_b, lse_b) of `attention(..., lse=...)` on the same queries.  o_*: [B, L, heads*head_dim] (row-strided)."""
    _dev(o_a, lse_a, o_b, lse_b)
    if o_a.dtype != o_b.dtype or lse_a.dtype != torch.float32 or lse_b.dtype != torch.float32:
        raise TypeError("attn_merge_: outputs share a dtype, lse tensors are float32")
    if not (lse_a.is_contiguous() and lse_b.is_contiguous()) or lse_a.numel() != B * heads * L or lse_b.numel() != B * heads * L:
        raise ValueError("attn_merge_: lse must be contiguous [B, heads, L]")
    oa3, ob3 = o_a.view(B, L, heads * head_dim), o_b.view(B, L, heads * head_dim)
    check(_lib.load().m4d_attn_merge(dt_code(o_a.dtype), _ptr(oa3), oa3.stride(0), oa3.stride(1), _ptr(lse_a), _ptr(ob3),
                                     ob3.stride(0), ob3.stride(1), _ptr(lse_b), B, L, heads, head_dim, _stream()),
          "m4d_attn_merge")
    return o_a


def patchify(src0, src1, patch, out_dtype):
    """[B,c0,F,H,W] (+ [B,c1,F,H,W]) -> [B, f*h*w, (c0+c1)*pt*ph*pw]."""
    _dev(src0, src1)
    src0 = src0.contiguous()
    B, c0, F, H, W = src0.shape
    c1 = 0
    if src1 is not None:
        src1 = src1.contiguous()
        if src1.dtype != src0.dtype or src1.shape[0] != B or tuple(src1.shape[2:]) != (F, H, W):
            raise ValueError("patchify: src1 must match src0's dtype, batch and grid")
        c1 = src1.shape[1]
    pt, ph, pw = patch
    out = torch.empty((B, (F // pt) * (H // ph) * (W // pw), (c0 + c1) * pt * ph * pw), device=src0.device,
                      dtype=out_dtype)
    lib = _lib.load()
    check(lib.m4d_patchify(dt_code(src0.dtype), _ptr(src0), c0, _ptr(src1), c1, dt_code(out_dtype), _ptr(out), B, F, H,
                           W, pt, ph, pw, _stream()), "m4d_patchify")
    return out


def unpatchify(tok, row0, grid, patch, c, out_dtype):
    """tok float32 [B, L, pt*ph*pw*c]; rows row0.. hold the (f,h,w) grid -> [B, c, f*pt, h*ph, w*pw]."""
    _dev(tok)
    if tok.dtype != torch.float32 or not tok.is_contiguous():
        raise TypeError("unpatchify: tok must be contiguous float32")
    B = tok.shape[0]
    f, h, w = grid
    pt, ph, pw = patch
    out = torch.empty((B, c, f * pt, h * ph, w * pw), device=tok.device, dtype=out_dtype)
    lib = _lib.load()
    check(lib.m4d_unpatchify(_ptr(tok), tok.stride(0), row0, dt_code(out_dtype), _ptr(out), B, c, f, h, w, pt, ph, pw,
                             _stream()), "m4d_unpatchify")
    return out


def cfg_euler_(x, v, guidance, dsigma, round_dtype=torch.float32):
    """x (float32, in place) += dsigma * (v[0] + guidance * (v[1] - v[0])); v: [2, ...] uncond first."""
    _dev(x, v)
    if x.dtype != torch.float32 or not x.is_contiguous() or not v.is_contiguous():
        raise TypeError("cfg_euler_: x must be contiguous float32, v contiguous")
    n = x.numel()
    if v.numel() != 2 * n:
        raise ValueError("cfg_euler_: v must hold the uncond and cond halves")
    lib = _lib.load()
    check(lib.m4d_cfg_euler(_ptr(x), dt_code(v.dtype), _ptr(v), n, float(guidance), float(dsigma),
                            dt_code(round_dtype), _stream()), "m4d_cfg_euler")
    return x


def unary(x, out_dtype, act=0, out=None):
    """act: 0 cast, 1 silu, 2 gelu(tanh), 3 gelu(erf)."""
    _dev(x, out)
    x = x.contiguous()
    if out is None:
        out = torch.empty(x.shape, device=x.device, dtype=out_dtype)
    lib = _lib.load()
    check(lib.m4d_unary(dt_code(x.dtype), _ptr(x), dt_code(out.dtype), _ptr(out), x.numel(), act, _stream()),
          "m4d_unary")
    return out


def add_bcast(a, bias):
    """float32 a [B, ...] + bias [...] broadcast over the leading dim (modulation + e)."""
    _dev(a, bias)
    if a.dtype != torch.float32 or bias.dtype != torch.float32:
        raise TypeError("add_bcast: float32 only")
    a = a.contiguous()
    bias = bias.contiguous()
    n = bias.numel()
    if a.numel() % n:
        raise ValueError("add_bcast: shape mismatch")
    out = torch.empty_like(a)
    lib = _lib.load()
    check(lib.m4d_add_bcast(_ptr(a), _ptr(bias), _ptr(out), a.numel() // n, n, _stream()), "m4d_add_bcast")
    return out


# ------------------------------------------------------------------ Motion-Sensitive 3D-VAE (channels-last)

def conv_pack_weights(w, Cin):
    """The weights w [Cout, taps*Cin] (bf16, Cin % 16 == 0) once more in the tiled order the LDS-halo conv kernels stage from with
    contiguous 1 KiB requests (m4d_conv_pack_weights); pass the result as `w_tiled` to conv_cl / conv_cl_planar.  None if not tileable."""
    _dev(w)
    Cout, K = w.shape
    lib = _lib.load()
    if w.dtype != torch.bfloat16 or not w.is_contiguous() or Cin % 16 or K % Cin:
        return None
    n = lib.m4d_conv_tiled_weight_bytes(Cin, Cout, K // Cin)
    if n <= 0 or n >= 1 << 31:
        return None
    out = torch.empty(n // 2, device=w.device, dtype=w.dtype)
    check(lib.m4d_conv_pack_weights(dt_code(w.dtype), _ptr(w), _ptr(out), Cin, Cout, K // Cin, _stream()), "m4d_conv_pack_weights")
    return out


def _check_tiled(w_tiled, w, Cin):
    if w_tiled is None:
        return
    _dev(w_tiled)
    Cout, K = w.shape
    if w_tiled.dtype != w.dtype or not w_tiled.is_contiguous() or \
            w_tiled.numel() * 2 != _lib.load().m4d_conv_tiled_weight_bytes(Cin, Cout, K // Cin):
        raise ValueError("conv: w_tiled is not conv_pack_weights(w, Cin)")


def conv_cl(x, w, bias, *, Tin, Hin, Win, Cin, k, stride=(1, 1, 1), pad=(0, 0, 0), out_thw, x_pixel_stride=None,
            resid=None, out=None, ups=False, tsplit=False, w_tiled=None):
    """Implicit-GEMM conv on channels-last x (flat or [T,H,W,C]); w [Cout, kt*kh*kw*Cin] packed (dt,dh,dw,c); w_tiled: optionally
    conv_pack_weights(w, Cin) as well.  Returns out [To*Ho*Wo, Cout] (or writes `out`, a row-strided 2-D view)."""
    _dev(x, w, bias, resid, out)
    _check_tiled(w_tiled, w, Cin)
    Cout = w.shape[0]
    kt, kh, kw = k
    if w.shape[1] != kt * kh * kw * Cin:
        raise ValueError(f"conv_cl: weight K {w.shape[1]} != {kt}*{kh}*{kw}*{Cin}")
    if x.dtype != w.dtype or (bias is not None and bias.dtype != x.dtype) or (resid is not None and resid.dtype != x.dtype):
        raise TypeError("conv_cl: dtype mismatch")
    To, Ho, Wo = out_thw
    M = To * Ho * Wo
    if x_pixel_stride is None:
        x_pixel_stride = Cin * (2 if tsplit else 1)
    if out is None:
        out = torch.empty((M, Cout), device=x.device, dtype=x.dtype)
    om, ldo = _rows2d(out)
    if om != M or out.shape[-1] != Cout:
        raise ValueError(f"conv_cl: out {tuple(out.shape)} vs M={M} Cout={Cout}")
    ldr = 0
    if resid is not None:
        rm, ldr = _rows2d(resid)
        if rm != M or resid.shape[-1] != Cout:
            raise ValueError("conv_cl: resid shape mismatch")
    lib = _lib.load()
    if w_tiled is not None:
        check(lib.m4d_conv_cl_tw(dt_code(x.dtype), _ptr(x), x_pixel_stride, _ptr(w), _ptr(w_tiled), _ptr(bias), _ptr(resid), ldr, _ptr(out), ldo,
                                 Tin, Hin, Win, Cin, Cout, kt, kh, kw, stride[0], stride[1], stride[2], pad[0], pad[1], pad[2],
                                 To, Ho, Wo, int(ups), int(tsplit), _stream()), "m4d_conv_cl_tw")
        return out
    check(lib.m4d_conv_cl(dt_code(x.dtype), _ptr(x), x_pixel_stride, _ptr(w), _ptr(bias), _ptr(resid), ldr, _ptr(out), ldo,
                          Tin, Hin, Win, Cin, Cout, kt, kh, kw, stride[0], stride[1], stride[2], pad[0], pad[1], pad[2],
                          To, Ho, Wo, int(ups), int(tsplit), _stream()), "m4d_conv_cl")
    return out


class Planar16:
    """A bf16 activation in planar-16 layout: `t` is a [C/16, frames, h*w, 16] view (dims 1.. contiguous) whose dim-0 stride is the
    plane stride — what rmsnorm_silu_cl_planar writes and conv_cl_planar reads (a causal conv's staging buffer)."""

    def __init__(self, t):
        if t.dim() != 4 or t.shape[3] != 16 or t.stride(3) != 1 or t.stride(2) != 16 or t.stride(1) != t.shape[2] * 16:
            raise ValueError(f"Planar16: bad view {tuple(t.shape)} / {t.stride()}")
        self.t = t

    @property
    def plane_stride(self):
        return self.t.stride(0)

    @property
    def rows(self):
        return self.t.shape[1] * self.t.shape[2]

    @property
    def channels(self):
        return self.t.shape[0] * 16


def gnstats_blocks(Hin, Win):
    """Rows per frame of the GroupNorm statistics conv_cl_planar(gn_stats=...) writes."""
    return _lib.load().m4d_conv_cl_planar_gnstats_blocks(Hin, Win)


def conv_cl_planar(x, w, bias, *, Tin, Hin, Win, kt, resid=None, out=None, norm=None, keep_raw=True, gn_stats=None, w_tiled=None):
    """3x3(x3) stride-1 conv (pad (0,1,1), valid in T) of a Planar16 input; w / bias / resid / out as conv_cl.
    norm = (gamma float32 [Cout], dst Planar16, silu): the next layer's RMS_norm(+SiLU) fused into the epilogue, written to `dst`
    (rows == To*Hin*Win); with keep_raw=False the un-normalised result is not stored and None is returned."""
    _dev(x.t, w, bias, resid, out)
    Cin, Cout = x.channels, w.shape[0]
    if x.t.dtype != torch.bfloat16 or w.dtype != torch.bfloat16 or x.rows != Tin * Hin * Win or w.shape[1] != kt * 9 * Cin:
        raise ValueError("conv_cl_planar: bf16, rows == Tin*Hin*Win, w [Cout, kt*9*Cin]")
    To = Tin - kt + 1
    M = To * Hin * Win
    ldo = 0
    if norm is None or keep_raw:
        if out is None:
            out = torch.empty((M, Cout), device=w.device, dtype=w.dtype)
        om, ldo = _rows2d(out)
        if om != M or out.shape[-1] != Cout:
            raise ValueError(f"conv_cl_planar: out {tuple(out.shape)} vs M={M} Cout={Cout}")
    elif out is not None:
        raise ValueError("conv_cl_planar: keep_raw=False with an output buffer")
    ldr = 0
    if resid is not None:
        rm, ldr = _rows2d(resid)
        if rm != M or resid.shape[-1] != Cout:
            raise ValueError("conv_cl_planar: resid shape mismatch")
    lib = _lib.load()
    if w_tiled is not None:
        _check_tiled(w_tiled, w, Cin)
        g_ptr = d_ptr = st_ptr = None
        d_plane = silu = 0
        if gn_stats is not None:
            _dev(gn_stats)
            if norm is not None or gn_stats.dtype != torch.float32 or not gn_stats.is_contiguous() or \
                    gn_stats.numel() != To * gnstats_blocks(Hin, Win) * 64:
                raise ValueError("conv_cl_planar: gn_stats must be contiguous float32 [To, blocks, 32, 2] (and excludes norm=)")
            st_ptr = _ptr(gn_stats)
        if norm is not None:
            gamma, dst, silu = norm
            _dev(gamma, dst.t)
            if gamma.dtype != torch.float32 or gamma.numel() != Cout or dst.rows != M or dst.channels != Cout or dst.t.dtype != w.dtype:
                raise ValueError("conv_cl_planar: norm = (gamma float32 [Cout], Planar16 of To*Hin*Win rows x Cout channels, silu)")
            g_ptr, d_ptr, d_plane = _ptr(gamma), _ptr(dst.t), dst.plane_stride
        check(lib.m4d_conv_cl_planar_tw(dt_code(w.dtype), _ptr(x.t), x.plane_stride, _ptr(w), _ptr(w_tiled), _ptr(bias), _ptr(resid), ldr,
                                        _ptr(out), ldo, Tin, Hin, Win, Cin, Cout, kt, To, g_ptr, d_ptr, d_plane, int(silu), st_ptr, _stream()),
              "m4d_conv_cl_planar_tw")
        return out
    if gn_stats is not None:
        # gn_stats: float32 [To, gnstats_blocks(Hin, Win), 32, 2] view: per-patch GroupNorm(32 x 4 channels) sums of the result (Cout = 128)
        _dev(gn_stats)
        if norm is not None or gn_stats.dtype != torch.float32 or not gn_stats.is_contiguous() or \
                gn_stats.numel() != To * gnstats_blocks(Hin, Win) * 64:
            raise ValueError("conv_cl_planar: gn_stats must be contiguous float32 [To, blocks, 32, 2] (and excludes norm=)")
        check(lib.m4d_conv_cl_planar_gnstats(dt_code(w.dtype), _ptr(x.t), x.plane_stride, _ptr(w), _ptr(bias), _ptr(resid), ldr, _ptr(out), ldo,
                                             Tin, Hin, Win, Cin, Cout, kt, To, _ptr(gn_stats), _stream()), "m4d_conv_cl_planar_gnstats")
        return out
    if norm is None:
        check(lib.m4d_conv_cl_planar(dt_code(w.dtype), _ptr(x.t), x.plane_stride, _ptr(w), _ptr(bias), _ptr(resid), ldr, _ptr(out), ldo,
                                     Tin, Hin, Win, Cin, Cout, kt, To, _stream()), "m4d_conv_cl_planar")
        return out
    gamma, dst, silu = norm
    _dev(gamma, dst.t)
    if gamma.dtype != torch.float32 or gamma.numel() != Cout or dst.rows != M or dst.channels != Cout or dst.t.dtype != w.dtype:
        raise ValueError("conv_cl_planar: norm = (gamma float32 [Cout], Planar16 of To*Hin*Win rows x Cout channels, silu)")
    check(lib.m4d_conv_cl_planar_norm(dt_code(w.dtype), _ptr(x.t), x.plane_stride, _ptr(w), _ptr(bias), _ptr(resid), ldr, _ptr(out), ldo,
                                      Tin, Hin, Win, Cin, Cout, kt, To, _ptr(gamma), _ptr(dst.t), dst.plane_stride, int(silu), _stream()),
          "m4d_conv_cl_planar_norm")
    return out


def rmsnorm_silu_cl_planar(x, gamma, out, *, silu=True):
    """rmsnorm_silu_cl with the result written into the Planar16 view `out` (rows == x's rows)."""
    _dev(x, gamma, out.t)
    P, ldx = _rows2d(x)
    C = x.shape[-1]
    if x.dtype != torch.bfloat16 or out.t.dtype != torch.bfloat16 or out.rows != P or out.channels != C:
        raise ValueError("rmsnorm_silu_cl_planar: bf16, matching rows / channels")
    if gamma.dtype != torch.float32 or gamma.numel() != C:
        raise TypeError("rmsnorm_silu_cl_planar: gamma must be float32 [C]")
    check(_lib.load().m4d_rmsnorm_silu_cl_planar(dt_code(x.dtype), _ptr(x), ldx, _ptr(gamma), _ptr(out.t), out.plane_stride, P, C, int(silu),
                                                 _stream()), "m4d_rmsnorm_silu_cl_planar")
    return out


def rmsnorm_silu_cl(x, gamma, *, silu=True, out=None):
    """x [P, C] (row-strided) -> RMS_norm(x) (* SiLU) in x.dtype."""
    _dev(x, gamma, out)
    P, ldx = _rows2d(x)
    C = x.shape[-1]
    if out is None:
        out = torch.empty((P, C), device=x.device, dtype=x.dtype)
    po, ldo = _rows2d(out)
    if po != P or out.shape[-1] != C or out.dtype != x.dtype:
        raise ValueError("rmsnorm_silu_cl: out mismatch")
    if gamma.dtype != torch.float32 or gamma.numel() != C:
        raise TypeError("rmsnorm_silu_cl: gamma must be float32 [C]")
    lib = _lib.load()
    check(lib.m4d_rmsnorm_silu_cl(dt_code(x.dtype), _ptr(x), ldx, _ptr(gamma), _ptr(out), ldo, P, C, int(silu), _stream()),
          "m4d_rmsnorm_silu_cl")
    return out


def groupnorm_cl(x, weight, bias, *, F, HW, groups=32, eps=1e-6, silu=True, out=None):
    """x [F, HW, C] contiguous channels-last -> GroupNorm (+swish)."""
    _dev(x, weight, bias, out)
    if not x.is_contiguous():
        raise ValueError("groupnorm_cl: x must be contiguous")
    C = x.shape[-1]
    if out is None:
        out = torch.empty_like(x)
    lib = _lib.load()
    n = lib.m4d_groupnorm_cl_workspace(F, HW, groups)
    ws = torch.empty(n, device=x.device, dtype=torch.float32)
    check(lib.m4d_groupnorm_cl(dt_code(x.dtype), _ptr(x), _ptr(out), _ptr(ws), n, _ptr(weight), _ptr(bias), F, HW, C, groups,
                               eps, int(silu), _stream()), "m4d_groupnorm_cl")
    return out


def groupnorm_cl_planar(x, weight, bias, *, F, HW, groups=32, eps=1e-6, silu=True, frames_per_group, stats=None):
    """groupnorm_cl with the result as a list of Planar16 views, one per group of `frames_per_group` frames (the last may be shorter),
    ready for conv_cl_planar (kt = 1).  stats: float32 [F, blocks, groups, 2] per-block sums written by the producing conv
    (conv_cl_planar(gn_stats=...)): the statistics pass over x is skipped."""
    _dev(x, weight, bias)
    if not x.is_contiguous() or x.dtype != torch.bfloat16:
        raise ValueError("groupnorm_cl_planar: contiguous bf16 x")
    C = x.shape[-1]
    ng = (F + frames_per_group - 1) // frames_per_group
    buf = torch.empty((ng, C // 16, frames_per_group, HW, 16), device=x.device, dtype=x.dtype)
    lib = _lib.load()
    if stats is not None:
        _dev(stats)
        if stats.dtype != torch.float32 or not stats.is_contiguous() or stats.dim() != 4 or stats.shape[0] != F or stats.shape[2:] != (groups, 2):
            raise ValueError("groupnorm_cl_planar: stats must be contiguous float32 [F, blocks, groups, 2]")
        st = torch.empty((F, groups, 2), device=x.device, dtype=torch.float32)
        check(lib.m4d_groupnorm_cl_planar_apply(dt_code(x.dtype), _ptr(x), _ptr(buf), _ptr(stats), stats.shape[1], _ptr(st), _ptr(weight),
                                                _ptr(bias), F, HW, C, groups, eps, int(silu), frames_per_group, buf.stride(1), buf.stride(0),
                                                _stream()), "m4d_groupnorm_cl_planar_apply")
    else:
        n = lib.m4d_groupnorm_cl_workspace(F, HW, groups)
        ws = torch.empty(n, device=x.device, dtype=torch.float32)
        check(lib.m4d_groupnorm_cl_planar(dt_code(x.dtype), _ptr(x), _ptr(buf), _ptr(ws), n, _ptr(weight), _ptr(bias), F, HW, C, groups, eps,
                                          int(silu), frames_per_group, buf.stride(1), buf.stride(0), _stream()), "m4d_groupnorm_cl_planar")
    return [Planar16(buf[g, :, :min(frames_per_group, F - g * frames_per_group)]) for g in range(ng)]


def softmax_rows(x, out_dtype, *, C, Cpad, scale):
    """x [R, >=C] float32/bf16 -> softmax(x[:, :C]*scale) in out_dtype [R, Cpad] (pad columns zero)."""
    _dev(x)
    R, ldx = _rows2d(x)
    out = torch.empty((R, Cpad), device=x.device, dtype=out_dtype)
    lib = _lib.load()
    check(lib.m4d_softmax_rows(dt_code(x.dtype), _ptr(x), ldx, dt_code(out_dtype), _ptr(out), Cpad, R, C, Cpad, float(scale),
                               _stream()), "m4d_softmax_rows")
    return out


def ncthw_to_cl(src, out_dtype, *, Cp=None, scale=1.0, shift=0.0, ch_scale=None, ch_shift=None, out=None):
    """src [C,T,H,W] -> [T,H,W,Cp] channels-last (zero-padded channels), optional affine."""
    _dev(src, ch_scale, ch_shift, out)
    src = src.contiguous()
    C, T, H, W = src.shape
    Cp = Cp or C
    if out is None:
        out = torch.empty((T, H, W, Cp), device=src.device, dtype=out_dtype)
    lib = _lib.load()
    check(lib.m4d_ncthw_to_cl(dt_code(src.dtype), _ptr(src), dt_code(out.dtype), _ptr(out), out.stride(2), C, Cp, T, H, W,
                              float(scale), float(shift), _ptr(ch_scale), _ptr(ch_shift), _stream()), "m4d_ncthw_to_cl")
    return out


def cl_to_ncthw(src, out_dtype, *, C, T, H, W, pixel_stride, scale=1.0, shift=0.0, ch_scale=None, ch_shift=None, act=0,
                aux=None):
    """channels-last src (pixel_stride elements per pixel) -> [C,T,H,W]; act: 0 none, 1 clamp(-1,1), 2 sigmoid(v+aux)."""
    _dev(src, ch_scale, ch_shift, aux)
    out = torch.empty((C, T, H, W), device=src.device, dtype=out_dtype)
    if aux is not None:
        aux = aux.contiguous()
        if aux.dtype != out_dtype or aux.numel() != out.numel():
            raise ValueError("cl_to_ncthw: aux must match the output")
    lib = _lib.load()
    check(lib.m4d_cl_to_ncthw(dt_code(src.dtype), _ptr(src), pixel_stride, dt_code(out_dtype), _ptr(out), C, T, H, W,
                              float(scale), float(shift), _ptr(ch_scale), _ptr(ch_shift), act, _ptr(aux), _stream()),
          "m4d_cl_to_ncthw")
    return out


# ------------------------------------------------------------------ TeaCache / guidance helpers

def axpby(x, y, a=1.0, b=1.0, out=None):
    """float32 out = a*x + b*y."""
    _dev(x, y, out)
    if x.dtype != torch.float32 or y.dtype != torch.float32 or x.numel() != y.numel():
        raise TypeError("axpby: float32 tensors of equal size")
    x, y = x.contiguous(), y.contiguous()
    if out is None:
        out = torch.empty_like(x)
    lib = _lib.load()
    check(lib.m4d_axpby(_ptr(x), _ptr(y), _ptr(out), x.numel(), float(a), float(b), _stream()), "m4d_axpby")
    return out


def lincomb(terms, out=None):
    """out = sum a_k * x_k over up to four (a_k, x_k) float32 terms; `out` may be one of the inputs."""
    xs = [x for _, x in terms]
    _dev(*xs, out)
    if not 1 <= len(terms) <= 4 or any(x.dtype != torch.float32 or not x.is_contiguous() for x in xs):
        raise ValueError("lincomb: 1..4 contiguous float32 tensors")
    if out is None:
        out = torch.empty_like(xs[0])
    pad = list(terms) + [(0.0, None)] * (4 - len(terms))
    args = []
    for a, x in pad:
        args += [_ptr(x), float(a)]
    check(_lib.load().m4d_lincomb(*args, _ptr(out), xs[0].numel(), _stream()), "m4d_lincomb")
    return out


def rel_l1(prev, cur):
    """(|cur-prev|.mean() / |prev|.mean()) as a python float (host sync, like the reference's .cpu().item())."""
    _dev(prev, cur)
    prev, cur = prev.contiguous().float(), cur.contiguous().float()
    out = torch.empty(2, device=prev.device, dtype=torch.float32)
    lib = _lib.load()
    check(lib.m4d_rel_l1(_ptr(prev), _ptr(cur), _ptr(out), prev.numel(), _stream()), "m4d_rel_l1")
    d, p_ = out.cpu().tolist()
    return d / p_


# ------------------------------------------------------------------ VAE / adaptor training (vae_autograd.py)
def pad_transpose(src, pixel_stride, C, T, H, W, Hp, Wp, pad_top, pad_left, nshift, cols, rows=None):
    """Channels-last frames src [T, H, W] x C (pixel stride `pixel_stride`) -> pixel-major panels [nshift*C (or rows), cols]:
    out[s*C + c, q] = P[q + s][c] where P is the stack of frames zero-padded to [T, Hp, Wp] (image at (pad_top, pad_left)) and
    flattened; columns beyond the last frame are zero.  The operand layout of the conv weight-gradient GEMMs."""
    _dev(src)
    rows = rows or nshift * C
    out = torch.empty((rows, cols), device=src.device, dtype=src.dtype)
    if rows > nshift * C:
        out[nshift * C:].zero_()
    check(_lib.load().m4d_pad_transpose(dt_code(src.dtype), _ptr(src), pixel_stride, C, T, H, W, Hp, Wp, pad_top, pad_left, nshift,
                                        _ptr(out), cols, _stream()), "m4d_pad_transpose")
    return out


def gemm_bt_batched(a, w, *, M, N, K, nb1, a_bs1, w_bs1, nb2=1, a_bs2=0, w_bs2=0):
    """float32 out[i2, i1] = A_(i1,i2) [M, K] . W_(i1,i2) [N, K]^T with A_(i1,i2) = a + i1*a_bs1 + i2*a_bs2 (elements; row stride
    a.stride(0)), likewise W.  Split-K partial products of the conv weight gradient: K-slices as batch 1, taps as batch 2."""
    _dev(a, w)
    if a.dtype != w.dtype or a.stride(-1) != 1 or w.stride(-1) != 1:
        raise TypeError("gemm_bt_batched: operands of one dtype with contiguous rows")
    out = torch.empty((nb2, nb1, M, N), device=a.device, dtype=torch.float32)
    check(_lib.load().m4d_gemm_bt_batched(dt_code(a.dtype), _ptr(a), a.stride(0), a_bs1, a_bs2, _ptr(w), w.stride(0), w_bs1, w_bs2,
                                          _ptr(out), M, N, K, nb1, nb2, _stream()), "m4d_gemm_bt_batched")
    return out


def wgrad_reduce(part, dw, dt, M):
    """dw[co, dt, dh, dw_, ci] += sum_s part[dh, s, co, dw_*cip + ci]  (part float32 [kh, S, M, kw*cip], dw float32
    [cop, kt, kh, kw, cip])."""
    _dev(part, dw)
    kh, S, Mp, N = part.shape
    cop, kt, kh2, kw, cip = dw.shape
    if kh != kh2 or N != kw * cip or Mp < cop or part.dtype != torch.float32 or dw.dtype != torch.float32:
        raise ValueError("wgrad_reduce: shape mismatch")
    check(_lib.load().m4d_wgrad_reduce(_ptr(part), _ptr(dw), S, Mp, cop, kt, kh, kw, cip, dt, _stream()), "m4d_wgrad_reduce")
    return dw


def gemm_bt_taps(a, w, *, M, N, K, nb1, a_bs1, w_bs1, tap_rows, tap_kh, tap_s1, tap_s2, a_off=0):
    """float32 out[s] [M, N] = A_s . W_s^T on the production 256 x 256 kernel with the rows of A as stacked taps (m4d_gemm_bt_taps);
    a_off: element offset of tap (0, 0) inside `a`'s rows."""
    _dev(a, w)
    if a.dtype != torch.bfloat16 or w.dtype != torch.bfloat16 or a.stride(-1) != 1 or w.stride(-1) != 1:
        raise TypeError("gemm_bt_taps: bf16 operands with contiguous rows")
    out = torch.empty((nb1, M, N), device=a.device, dtype=torch.float32)
    check(_lib.load().m4d_gemm_bt_taps(dt_code(a.dtype), a.data_ptr() + 2 * a_off, a.stride(0), a_bs1, _ptr(w), w.stride(0), w_bs1, _ptr(out),
                                       M, N, K, nb1, tap_rows, tap_kh, tap_s1, tap_s2, _stream()), "m4d_gemm_bt_taps")
    return out


def wgrad_reduce_taps(part, dw):
    """dw[co, dt, dh, dw_, ci] += sum_s part[s, dt*kh + dh, co, dw_*cip + ci]  (part float32 [S, kt*kh*cop, kw*cip])."""
    _dev(part, dw)
    cop, kt, kh, kw, cip = dw.shape
    S = part.shape[0]
    if part.shape[1:] != (kt * kh * cop, kw * cip) or part.dtype != torch.float32 or dw.dtype != torch.float32 or not part.is_contiguous():
        raise ValueError("wgrad_reduce_taps: shape mismatch")
    check(_lib.load().m4d_wgrad_reduce_taps(_ptr(part), _ptr(dw), S, cop, kt, kh, kw, cip, _stream()), "m4d_wgrad_reduce_taps")
    return dw


def rmsnorm_silu_cl_bwd(x, gamma, dy, *, silu=True):
    """Backward of rmsnorm_silu_cl: x, dy [P, C] (row-strided) -> (dx [P, C] in x.dtype, dgamma float32 [C])."""
    _dev(x, gamma, dy)
    P, ldx = _rows2d(x)
    Pd, ldd = _rows2d(dy)
    C = x.shape[-1]
    if Pd != P or dy.shape[-1] != C or dy.dtype != x.dtype:
        raise ValueError("rmsnorm_silu_cl_bwd: dy mismatch")
    dx = torch.empty((P, C), device=x.device, dtype=x.dtype)
    dg = torch.zeros(C, device=x.device, dtype=torch.float32)
    check(_lib.load().m4d_rmsnorm_silu_cl_bwd(dt_code(x.dtype), _ptr(x), ldx, _ptr(gamma), _ptr(dy), ldd, _ptr(dx), C, _ptr(dg), P, C,
                                              int(silu), _stream()), "m4d_rmsnorm_silu_cl_bwd")
    # (argument order of the ABI: x, x_ld, gamma, dy, dy_ld, dx, dx_ld, dgamma, P, C, silu)
    return dx, dg


def softmax_rows_bwd(p, dp, *, scale, C):
    """p T [R, Cpad] (softmax rows, columns >= C zero), dp float32 [R, Cpad] -> dS T [R, Cpad] = scale * p * (dp - rowsum(p*dp))."""
    _dev(p, dp)
    R, Cpad = p.shape
    if dp.shape != p.shape or dp.dtype != torch.float32 or not p.is_contiguous() or not dp.is_contiguous():
        raise ValueError("softmax_rows_bwd: contiguous p (T) and dp (float32) of one shape")
    out = torch.empty_like(p)
    check(_lib.load().m4d_softmax_rows_bwd(dt_code(p.dtype), _ptr(p), _ptr(dp), _ptr(out), R, C, Cpad, float(scale), _stream()),
          "m4d_softmax_rows_bwd")
    return out


def upsample2x_cl(x, t, h, w, c, *, tsplit=False):
    """Nearest-exact 2x of channels-last frames [t, h, w, c] -> [t*2h*2w, c]; tsplit: the input holds 2c channels per pixel and
    frame 2i / 2i+1 of the result comes from the first / second channel half (wan_vae.py:138-141)."""
    _dev(x)
    tt = t * (2 if tsplit else 1)
    out = torch.empty((tt * 4 * h * w, c), device=x.device, dtype=x.dtype)
    check(_lib.load().m4d_upsample2x_cl(dt_code(x.dtype), _ptr(x), _ptr(out), t, h, w, c, int(tsplit), 0, _stream()), "m4d_upsample2x_cl")
    return out


def upsample2x_cl_bwd(du, t, h, w, c, *, tsplit=False):
    """Transpose of upsample2x_cl: du [t'*2h*2w, c] -> [t*h*w, c (2c with tsplit)] (sum over each 2x2 block)."""
    _dev(du)
    du = du.contiguous()
    out = torch.empty((t * h * w, c * (2 if tsplit else 1)), device=du.device, dtype=du.dtype)
    check(_lib.load().m4d_upsample2x_cl(dt_code(du.dtype), _ptr(du), _ptr(out), t, h, w, c, int(tsplit), 1, _stream()),
          "m4d_upsample2x_cl(bwd)")
    return out


def groupnorm_cl_bwd(x, weight, bias, dy, *, F, HW, groups=32, eps=1e-6, silu=True):
    """Backward of groupnorm_cl(+swish): x, dy [F, HW, C] -> (dx [F, HW, C], dweight float32 [C], dbias float32 [C])."""
    _dev(x, weight, bias, dy)
    C = x.shape[-1]
    if not x.is_contiguous() or not dy.is_contiguous() or dy.dtype != x.dtype:
        raise ValueError("groupnorm_cl_bwd: contiguous x, dy of one dtype")
    dx = torch.empty_like(x)
    dwt = torch.zeros(C, device=x.device, dtype=torch.float32)
    dbs = torch.zeros(C, device=x.device, dtype=torch.float32)
    lib = _lib.load()
    n = lib.m4d_groupnorm_cl_bwd_workspace(F, HW, groups)
    ws = torch.empty((n,), device=x.device, dtype=torch.float32)
    check(lib.m4d_groupnorm_cl_bwd(dt_code(x.dtype), _ptr(x), _ptr(weight), _ptr(bias), _ptr(dy), _ptr(dx), _ptr(dwt), _ptr(dbs),
                                   _ptr(ws), n, F, HW, C, groups, float(eps), int(silu), _stream()), "m4d_groupnorm_cl_bwd")
    return dx, dwt, dbs


def minmax(x, n_groups):
    """x float32, contiguous, viewed as [n_groups, -1] -> float32 [n_groups, 2] = (min, max) of every group."""
    _dev(x)
    if x.dtype != torch.float32 or not x.is_contiguous() or x.numel() % n_groups:
        raise TypeError("minmax: contiguous float32 input divisible into the groups")
    out = torch.empty((n_groups, 2), device=x.device, dtype=torch.float32)
    check(_lib.load().m4d_minmax(_ptr(x), n_groups, x.numel() // n_groups, _ptr(out), _stream()), "m4d_minmax")
    return out


def backproject(depth, inv_fx, inv_fy):
    """depth float32 [H, W] -> (coords float32 [3, H, W], zclean float32 [H, W]) (infer.py:179-195, :823-825)."""
    _dev(depth)
    if depth.dtype != torch.float32 or depth.dim() != 2 or not depth.is_contiguous():
        raise TypeError("backproject: contiguous float32 [H, W] depth")
    H, W = depth.shape
    coords = torch.empty((3, H, W), device=depth.device, dtype=torch.float32)
    zc = torch.empty((H, W), device=depth.device, dtype=torch.float32)
    check(_lib.load().m4d_backproject(_ptr(depth), H, W, float(inv_fx), float(inv_fy), _ptr(coords), _ptr(zc), _stream()),
          "m4d_backproject")
    return coords, zc


def depth_control(zclean, mm, out_dtype):
    """zclean float32 [H, W], mm float32 [1, 2] -> out_dtype [3, H, W] in [-1, 1] (infer.py:826-828)."""
    _dev(zclean, mm)
    H, W = zclean.shape
    out = torch.empty((3, H, W), device=zclean.device, dtype=out_dtype)
    check(_lib.load().m4d_depth_control(dt_code(out_dtype), _ptr(zclean), _ptr(mm), _ptr(out), H * W, _stream()), "m4d_depth_control")
    return out


def flow_recover(rel, frame0, mm=None, track_z=False, first_frame="coords"):
    """rel [B,3,F,H,W] (T), frame0 float32 [B,3,H,W], mm float32 [B*3,2] -> float32 [B,3,F,H,W] point trajectories.
    first_frame: "coords" = frame 0 of the result is frame0 itself (the stored cloud, infer.py:870), "recovered" = computed like every
    other frame (what the reference's inverse_flow_norm_transform_no_diff returns)."""
    if first_frame not in ("coords", "recovered"):
        raise ValueError("flow_recover: first_frame must be 'coords' or 'recovered'")
    _dev(rel, frame0, mm)
    rel = rel.contiguous()
    B, C, F, H, W = rel.shape
    if C != 3 or tuple(frame0.shape) != (B, 3, H, W) or frame0.dtype != torch.float32 or not frame0.is_contiguous():
        raise ValueError("flow_recover: rel [B,3,F,H,W], frame0 float32 contiguous [B,3,H,W]")
    out = torch.empty((B, 3, F, H, W), device=rel.device, dtype=torch.float32)
    check(_lib.load().m4d_flow_recover(dt_code(rel.dtype), _ptr(rel), _ptr(frame0), _ptr(mm), _ptr(out), B, F, H * W,
                                       (1 if track_z else 0) | (2 if first_frame == "recovered" else 0), _stream()), "m4d_flow_recover")
    return out


def bilinear_cl(x, out_hw):
    """x [B, Hi, Wi, C] channels-last -> [B, Ho, Wo, C] (bilinear, align_corners=False)."""
    _dev(x)
    x = x.contiguous()
    B, Hi, Wi, C = x.shape
    out = torch.empty((B, out_hw[0], out_hw[1], C), device=x.device, dtype=x.dtype)
    lib = _lib.load()
    check(lib.m4d_bilinear_cl(dt_code(x.dtype), _ptr(x), _ptr(out), B, Hi, Wi, out_hw[0], out_hw[1], C, _stream()),
          "m4d_bilinear_cl")
    return out


# ---------------------------------------------------------------------------------------------- training step

def transpose(x, out=None):
    """out[c, r] = x[r, c] for a 2-D matrix (row stride = x.stride(0))."""
    _dev(x, out)
    R, C = x.shape
    if out is None:
        out = torch.empty((C, R), device=x.device, dtype=x.dtype)
    check(_lib.load().m4d_transpose(dt_code(x.dtype), _ptr(x), x.stride(0), _ptr(out), out.stride(0), R, C, _stream()),
          "m4d_transpose")
    return out


def colsum(a, b=None, *, rows_per_group=None, out=None):
    """float32 [G, C]: sum over the rows of each group of a (* b).  a, b: [R, C] (row strides honoured)."""
    _dev(a, b, out)
    R, C = a.shape
    if rows_per_group is None:
        rows_per_group = R
    G = (R + rows_per_group - 1) // rows_per_group
    if out is None:
        out = torch.zeros((G, C), device=a.device, dtype=torch.float32)
    check(_lib.load().m4d_colsum(dt_code(a.dtype), _ptr(a), a.stride(0), dt_code(b.dtype) if b is not None else 0, _ptr(b),
                                 b.stride(0) if b is not None else 0, _ptr(out), R, C, rows_per_group, _stream()),
          "m4d_colsum")
    return out


def scale_cast(x, out_dtype, *, gate=None, gate_stride=0, rows_per_sample=0, out=None):
    """out_dtype [R, C] = float32 x [R, C] * gate[sample, :]."""
    _dev(x, gate, out)
    C = x.shape[-1]
    R = x.numel() // C
    if out is None:
        out = torch.empty(x.shape, device=x.device, dtype=out_dtype)
    check(_lib.load().m4d_scale_cast(_ptr(x), _ptr(gate), gate_stride, rows_per_sample or R, dt_code(out_dtype), _ptr(out),
                                     R, C, _stream()), "m4d_scale_cast")
    return out


def resid_gate(x, y, *, gate=None, gate_stride=0, rows_per_sample=0, out=None):
    """float32 out = x + y * gate[sample, :]  (x float32 [.., C], y T of the same shape)."""
    _dev(x, y, gate, out)
    C = x.shape[-1]
    R = x.numel() // C
    if out is None:
        out = torch.empty_like(x)
    check(_lib.load().m4d_resid_gate(_ptr(x), dt_code(y.dtype), _ptr(y), _ptr(gate), gate_stride, rows_per_sample or R,
                                     _ptr(out), R, C, _stream()), "m4d_resid_gate")
    return out


def add(a, b, out=None):
    """out = a + b for contiguous tensors of one dtype (numel % 4 == 0)."""
    _dev(a, b, out)
    if out is None:
        out = torch.empty_like(a)
    check(_lib.load().m4d_add(dt_code(a.dtype), _ptr(a), _ptr(b), _ptr(out), a.numel(), _stream()), "m4d_add")
    return out


ACT_SILU, ACT_GELU_TANH, ACT_GELU_ERF = 1, 2, 3
ACT_SIGMOID, ACT_SIGMOID_OUT, ACT_CLAMP1 = 4, 5, 6      # sigmoid(pre); sigmoid given its OUTPUT; clamp(pre, -1, 1)


def act_bwd_(dy, pre, act):
    """dy *= act'(pre) in place."""
    _dev(dy, pre)
    if dy.dtype != pre.dtype or not dy.is_contiguous() or not pre.is_contiguous():
        raise ValueError("act_bwd_: contiguous tensors of one dtype")
    check(_lib.load().m4d_act_bwd(dt_code(dy.dtype), _ptr(dy), _ptr(pre), dy.numel(), act, _stream()), "m4d_act_bwd")
    return dy


def guidance_bwd_(x, dz, *, B, rows_per_sample, shift, scale, mod_stride, g_ss, g_gate, g_period, g_len, eps=1e-6, mod_rows=0, g_rows=0):
    """Spatial-guidance tail of ln_modulate, backward: dz (T, in place) becomes the gradient w.r.t. the un-guided
    LN-modulate output; returns float32 [B, g_period, 2C] = (sum_f dz*u | sum_f dz) per spatial position."""
    _dev(x, dz, shift, scale, g_ss, g_gate)
    C = x.shape[-1]
    if not dz.is_contiguous() or not x.is_contiguous():
        raise ValueError("guidance_bwd_: contiguous tensors")
    ab = torch.empty((B, g_period, 2 * C), device=x.device, dtype=torch.float32)
    check(_lib.load().m4d_guidance_bwd_m(_ptr(x), dt_code(dz.dtype), _ptr(dz), B, rows_per_sample, C, _ptr(shift), _ptr(scale),
                                         mod_stride, mod_rows, eps, _ptr(g_ss), _ptr(g_gate), g_period, g_len, _ptr(ab), _stream()),
          "m4d_guidance_bwd_m")
    return ab


def ln_modulate_bwd(x, dy, dx, *, B, rows_per_sample, scale=None, mod_stride=0, ln_w=None, eps=1e-6, dshift=None,
                    dscale=None, red_stride=0):
    """dx (float32, accumulated in place) += LayerNorm-input gradient; dshift/dscale accumulate sum(dy), sum(dy*xhat)."""
    _dev(x, dy, dx, scale, ln_w, dshift, dscale)
    C = x.shape[-1]
    check(_lib.load().m4d_ln_modulate_bwd(_ptr(x), dt_code(dy.dtype), _ptr(dy), _ptr(dx), B, rows_per_sample, C, _ptr(scale),
                                          mod_stride, _ptr(ln_w), eps, _ptr(dshift), _ptr(dscale), red_stride, _stream()),
          "m4d_ln_modulate_bwd")
    return dx


def rmsnorm_rope_bwd_(dy0, x0, w0, dw0, dy1=None, x1=None, w1=None, dw1=None, *, head_dim, eps=1e-6, cos=None, sin=None,
                      rows_per_sample=0, rope_len=0, pos_offset=0):
    """In place: dy{0,1} [rows, C] become the gradients w.r.t. the pre-norm inputs x{0,1}; dw{0,1} accumulate."""
    _dev(dy0, x0, w0, dw0, dy1, x1, w1, dw1, cos, sin)
    rows, ld = _rows2d(dy0)
    _, ldx = _rows2d(x0)
    if dy1 is not None and (_rows2d(dy1) != (rows, ld) or _rows2d(x1)[1] != ldx):
        raise ValueError("rmsnorm_rope_bwd_: both tensors must share shape and strides")
    C = dy0.shape[-1]
    check(_lib.load().m4d_rmsnorm_rope_bwd(dt_code(dy0.dtype), _ptr(dy0), _ptr(dy1), ld, _ptr(x0), _ptr(x1), ldx, _ptr(w0),
                                           _ptr(w1), _ptr(dw0), _ptr(dw1), rows, C, head_dim, eps, _ptr(cos), _ptr(sin),
                                           rows_per_sample or rows, rope_len, pos_offset, _stream()),
          "m4d_rmsnorm_rope_bwd")


def attention_bwd(q, k, v, o, d_o, lse, *, B, Lq, Lk, Lk_rows, heads, head_dim, dq, dk, dv, scale=None,
                  accumulate_dq=False, accumulate_dkv=False):
    """Gradients of attention(): q, o, d_o, dq are [B*Lq, C]-shaped matrices (row stride honoured), k, v, dk, dv
    [B*Lk_rows, C]; only the first Lk keys of each sample are real.  The transposed operands are built here."""
    _dev(q, k, v, o, d_o, lse, dq, dk, dv)
    C = heads * head_dim
    for t in (q, k, v, o, d_o, dq, dk, dv):
        if t.dtype != q.dtype or t.stride(-1) != 1:
            raise TypeError("attention_bwd: operands must share a dtype and be row-major")
        if t.dim() != 2 or t.shape[1] != C:
            raise ValueError("attention_bwd: operands are 2-D [rows, heads*head_dim] matrices")
    # The bf16 / head_dim 128 passes (csrc/attention_bwd128.h, attention_bwd_kvp.h) read their transposed operands out of the row-major
    # tiles; only the generic kernels (fp32 parity mode, other head dims, M4D_ATTN_BWD_GENERIC=1) want transposed copies of Q, K, dO.
    need_t = not (q.dtype == torch.bfloat16 and head_dim == 128) or os.environ.get("M4D_ATTN_BWD_GENERIC", "0") not in ("", "0")
    qt, kt, dot = (transpose(q), transpose(k), transpose(d_o)) if need_t else (None, None, None)
    delta = torch.empty((B, heads, Lq), device=q.device, dtype=torch.float32)
    a = _lib.AttnBwdArgs()
    a.q, a.k, a.v, a.o, a.d_o = q.data_ptr(), k.data_ptr(), v.data_ptr(), o.data_ptr(), d_o.data_ptr()
    a.qt, a.kt, a.dot = (qt.data_ptr(), kt.data_ptr(), dot.data_ptr()) if need_t else (0, 0, 0)
    a.lse, a.delta = lse.data_ptr(), delta.data_ptr()
    a.dq, a.dk, a.dv = dq.data_ptr(), dk.data_ptr(), dv.data_ptr()

    def st(t, L):
        return L * t.stride(0), t.stride(0)

    a.q_bs, a.q_ls = st(q, Lq)
    a.k_bs, a.k_ls = st(k, Lk_rows)
    a.v_bs, a.v_ls = st(v, Lk_rows)
    a.o_bs, a.o_ls = st(o, Lq)
    a.do_bs, a.do_ls = st(d_o, Lq)
    a.dq_bs, a.dq_ls = st(dq, Lq)
    a.dk_bs, a.dk_ls = st(dk, Lk_rows)
    a.dv_bs, a.dv_ls = st(dv, Lk_rows)
    a.qt_bs, a.qt_ls = Lq, B * Lq
    a.kt_bs, a.kt_ls = Lk_rows, B * Lk_rows
    a.dot_bs, a.dot_ls = Lq, B * Lq
    a.Lq, a.Lk, a.Lk_rows = Lq, Lk, Lk_rows
    a.B, a.heads, a.head_dim = B, heads, head_dim
    a.accumulate_dq, a.accumulate_dkv = int(accumulate_dq), int(accumulate_dkv)
    a.scale = scale if scale is not None else 1.0 / math.sqrt(head_dim)
    ws = None
    if Lk_rows * 8 <= Lq:     # few keys against many queries: let the dk/dv passes split the query loop
        ws = torch.empty(B * Lk_rows * C, device=q.device, dtype=torch.float32)
        a.ws, a.ws_elems = ws.data_ptr(), ws.numel()
    check(_lib.load().m4d_attention_bwd(dt_code(q.dtype), a, _stream()), "m4d_attention_bwd")


def sumsq(x, out):
    """out (float32 scalar tensor) += sum(x^2)."""
    _dev(x, out)
    check(_lib.load().m4d_sumsq(dt_code(x.dtype), _ptr(x), x.numel(), _ptr(out), _stream()), "m4d_sumsq")
    return out


def adamw_(p, g, m, v, *, lr, beta1, beta2, eps, weight_decay, step, grad_scale=None):
    _dev(p, g, m, v, grad_scale)
    if not (p.is_contiguous() and g.is_contiguous() and m.is_contiguous() and v.is_contiguous()):
        raise ValueError("adamw_: contiguous tensors required")
    if g.dtype != p.dtype or m.dtype != v.dtype:
        raise TypeError("adamw_: dtype mismatch")
    check(_lib.load().m4d_adamw(dt_code(p.dtype), _ptr(p), _ptr(g), dt_code(m.dtype), _ptr(m), _ptr(v), p.numel(), lr, beta1,
                                beta2, eps, weight_decay, step, _ptr(grad_scale), _stream()), "m4d_adamw")
