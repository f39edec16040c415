"""Wan2.1-DiT with MoRe4D spatial guidance — MI355X-native host side.

Drop-in for the reference's `MoRe4D/models/wan_transformer4d.py` (same class names, ctor kwargs,
forward kwargs, parameter names/shapes — SURVEY.md §8b, Appendix A), but every arithmetic op is a
hand-written gfx950 kernel reached through the C ABI (`more4d_amd.ops` -> libmore4d_hip.so).  The
nn.Linear / nn.Conv3d / nn.LayerNorm objects below are parameter CONTAINERS only (so checkpoints,
`named_parameters()`, `.to(dtype)`, optimizers keep working); their torch forward is never called.

Compute dtype T = the parameters' dtype: bfloat16 is the production path (bf16 MFMA operands, fp32
accumulation, fp32 residual stream, casts where the reference's autocast places them — SURVEY.md
Appendix C); float32 is the parity path (exact-fp32 MFMA), checked against the CPU oracle to 1e-3.

Internal layout: tokens are rows of [B, Lp, C] buffers with Lp = seq_len rounded up to 8 (16-byte
rows for V^T); rows >= seq_len are never used as keys.  V is produced TRANSPOSED by its projection
GEMM (V^T [C, B*Lp]) because the attention kernel consumes V^T tiles (csrc/attention.hip).
"""
import math
import os
import types
from typing import Optional

import torch
import torch.nn as nn

from .. import ops
from ..ops import EPI_GELU_ERF, EPI_GELU_TANH, EPI_RESID_GATE, EPI_SILU, EPI_STORE, EPI_STORE_F32, KV
from ..utils.cfg_optimization import cfg_skip
from .cache_utils import TeaCache

__all__ = ["WanTransformer4DModel", "WanAttentionBlock", "WanSelfAttention", "WanI2VCrossAttention",
           "WanT2VCrossAttention", "WanRMSNorm", "WanLayerNorm", "Head", "MLPProj", "SpatialGuidanceModule",
           "sinusoidal_embedding_1d", "rope_params", "ContextCache"]


def _round8(n):
    return (n + 7) // 8 * 8


def sinusoidal_embedding_1d(dim, position):
    """float64 cat(cos, sin) table, computed on the host side of the boundary (reference :239-249)."""
    assert dim % 2 == 0
    half = dim // 2
    pos = position.to(torch.float64)
    inv = torch.pow(10000, -torch.arange(half, dtype=torch.float64, device=pos.device).div(half))
    s = torch.outer(pos, inv)
    return torch.cat([torch.cos(s), torch.sin(s)], dim=1)


def rope_params(max_seq_len, dim, theta=10000):
    """complex128 unit phasors [max_seq_len, dim/2] (reference :252-260)."""
    assert dim % 2 == 0
    ang = torch.outer(torch.arange(max_seq_len, dtype=torch.float64),
                      1.0 / torch.pow(theta, torch.arange(0, dim, 2, dtype=torch.float64).div(dim)))
    return torch.polar(torch.ones_like(ang), ang)


def build_rope_tables(freqs, grid, head_dim, device):
    """Per-token cos/sin float32 [f*h*w, head_dim/2] from the complex128 `freqs` attribute: the first
    c-2(c//3) pairs rotate with the frame index, the next c//3 with the row, the last c//3 with the column
    (reference rope_apply :346-361).  Built in float64 on the host, cast once."""
    f, h, w = grid
    c = head_dim // 2
    fr = freqs.cpu().split([c - 2 * (c // 3), c // 3, c // 3], dim=1)
    z = torch.cat([fr[0][:f].view(f, 1, 1, -1).expand(f, h, w, -1),
                   fr[1][:h].view(1, h, 1, -1).expand(f, h, w, -1),
                   fr[2][:w].view(1, 1, w, -1).expand(f, h, w, -1)], dim=-1).reshape(f * h * w, -1)
    return (z.real.to(torch.float32).contiguous().to(device), z.imag.to(torch.float32).contiguous().to(device))


_XATTN_FUSED = os.environ.get("M4D_XATTN_FUSED", "1") != "0"      # A/B switch: 0 = text and image branches as two launches
# fused q+k projection of the self-attention (one N = 2C GEMM launch); M4D_FUSE_QK=0 = two launches (A/B)
_FUSE_QK = os.environ.get("M4D_FUSE_QK", "1") != "0"


def _f32(p, cache):
    """float32 view/copy of a (possibly bf16) parameter, cached per (storage, version)."""
    if p.dtype == torch.float32:
        return p.detach()
    key = (p.data_ptr(), p._version, tuple(p.shape))
    hit = cache.get(id(p))
    if hit is None or hit[0] != key:
        hit = (key, p.detach().float().contiguous())
        cache[id(p)] = hit
    return hit[1]


LOG2E = 1.4426950408889634
# The softmax scale of the DiT's self-attention, folded into q where q is made: q' = rope(rmsnorm(x) * (w * c)), c = head_dim^-0.5 *
# log2(e) — RoPE is linear, so q' = c * q exactly in fp32, and the bf16 store of q' is the ONE rounding q gets either way.  The attention
# kernels are then called with scale = ln 2 (scale * log2(e) == 1): attn128q_kernel takes q' as its Q~ operand bit for bit instead of
# rounding Q * c to bf16 a second time (csrc/attention_q64.h).  bf16 inference with qk_norm only; M4D_FOLD_QSCALE=0 = A/B.
_FOLD_QSCALE = os.environ.get("M4D_FOLD_QSCALE", "1") != "0"
# T-sharded self-attention, A/B (tools/bench_shard.py): M4D_SP_PER_SEGMENT=1 = the gathered remote shards one call per shard + LSE merges
# instead of one multi-segment call (attn128q_kernel takes up to 8 segments with up to 5 ragged tails itself, so this is off)
_SP_PER_SEGMENT = os.environ.get("M4D_SP_PER_SEGMENT", "0") != "0"
# M4D_SP_OVERLAP=1: the remote shards' attention calls beside the local call on a second stream (default: behind it; same-box A/B
# profiles/r06_ab_sp_overlap.log: same bits, 1 % slower — the two launches' partial rounds do not fill each other)
_SP_OVERLAP = os.environ.get("M4D_SP_OVERLAP", "0") != "0"
_SIDE_STREAMS = {}


def _side_stream(device):
    """One extra HIP stream per device for the remote-shard attention calls of the T-sharded path."""
    key = torch.device(device).index if torch.device(device).index is not None else torch.cuda.current_device()
    st = _SIDE_STREAMS.get(key)
    if st is None:
        st = _SIDE_STREAMS[key] = torch.cuda.Stream(device=device)
    return st


def _folded_norm_weight(p, head_dim, cache):
    """float32 norm_q weight times head_dim^-0.5 * log2(e), cached per (storage, version)."""
    key = (p.data_ptr(), p._version, tuple(p.shape), head_dim)
    hit = cache.get(("fold", id(p)))
    if hit is None or hit[0] != key:
        hit = (key, (p.detach().float() * (head_dim ** -0.5 * LOG2E)).contiguous())
        cache[("fold", id(p))] = hit
    return hit[1]


def _ragged_chunks(segs, max_ragged):
    """consecutive runs of K / V^T segments with at most `max_ragged` lengths that are not a multiple of the 64-key tile"""
    runs, cur, n = [], [], 0
    for sg in segs:
        rag = 1 if sg.len % 64 else 0
        if cur and n + rag > max_ragged:
            runs.append(cur)
            cur, n = [], 0
        cur.append(sg)
        n += rag
    if cur:
        runs.append(cur)
    return runs


class WanRMSNorm(nn.Module):
    def __init__(self, dim, eps=1e-5):
        super().__init__()
        self.dim, self.eps = dim, eps
        self.weight = nn.Parameter(torch.ones(dim))


class WanLayerNorm(nn.LayerNorm):
    def __init__(self, dim, eps=1e-6, elementwise_affine=False):
        super().__init__(dim, elementwise_affine=elementwise_affine, eps=eps)


class _Ctx:
    """Per-forward shared state handed to the blocks: shapes, rope tables, guidance tables, SP group."""

    def __init__(self, B, L, Lp, grid, cos, sin, rope_len, f32cache, key_len, sp=None, pos_offset=0):
        self.B, self.L, self.Lp, self.grid = B, L, Lp, grid
        self.cos, self.sin, self.rope_len = cos, sin, rope_len
        self.f32cache = f32cache
        self.key_len = key_len          # valid keys in the full (unsharded) sequence
        self.sp = sp                    # sequence-parallel group wrapper or None
        self.pos_offset = pos_offset    # first global token index of this rank's shard
        self.guid = None                # (g_ss_self[layer], g_ss_ffn[layer], period, len) when guidance is on


class WanSelfAttention(nn.Module):
    def __init__(self, dim, num_heads, window_size=(-1, -1), qk_norm=True, eps=1e-6):
        assert dim % num_heads == 0
        super().__init__()
        self.dim, self.num_heads, self.head_dim = dim, num_heads, dim // num_heads
        self.window_size, self.qk_norm, self.eps = window_size, qk_norm, eps
        self.q = nn.Linear(dim, dim)
        self.k = nn.Linear(dim, dim)
        self.v = nn.Linear(dim, dim)
        self.o = nn.Linear(dim, dim)
        self.norm_q = WanRMSNorm(dim, eps=eps) if qk_norm else nn.Identity()
        self.norm_k = WanRMSNorm(dim, eps=eps) if qk_norm else nn.Identity()

    def _ones(self, ref):
        return torch.ones(self.dim, device=ref.device, dtype=torch.float32)

    def _qk_weights(self):
        """[Wq; Wk] [2C, C] and [bq; bk] for the fused q+k projection.
        No second resident copy (105 MB per layer at 14B width): the first call allocates ONE [2C, C] / [2C] storage and re-points
        `q.weight` / `k.weight` / the biases at its two halves (`param.data = view`), so in-place updates of either parameter ARE updates of
        the fused operand and nothing can go stale; a `.to()` / `load_state_dict(assign=True)` that gives the parameters new storage is
        seen through the data pointers and re-fuses.  Parameters that already live inside someone else's buffer (the flat buckets of
        dist/data_parallel.py) are left where they are: (None, None) — the caller then projects q and k with two launches (a cached
        copy could go stale under raw-pointer writers that do not bump the version counter)."""
        qw, kw, qb, kb = self.q.weight, self.k.weight, self.q.bias, self.k.bias
        C, es = self.dim, qw.element_size()
        hit = self.__dict__.get("_qk_cache")
        if hit is not None and hit[0] == "view":
            w, b = hit[1], hit[2]
            if (w.dtype == qw.dtype and qw.data_ptr() == w.data_ptr() and kw.data_ptr() == w.data_ptr() + C * C * es and
                    qb.data_ptr() == b.data_ptr() and kb.data_ptr() == b.data_ptr() + C * es):
                return w, b
        owned = all(t.is_contiguous() and t.untyped_storage().nbytes() == t.numel() * t.element_size() for t in (qw, kw, qb, kb))
        if owned:
            # inference_mode(False): a first call under torch.inference_mode() would otherwise turn the four Parameters into inference
            # tensors, which a later optimizer step cannot update in place; call fuse_qk_() after loading to do this eagerly
            with torch.inference_mode(False), torch.no_grad():
                w = torch.cat([qw.detach(), kw.detach()]).contiguous()
                b = torch.cat([qb.detach(), kb.detach()]).contiguous()
                qw.data, kw.data, qb.data, kb.data = w[:C], w[C:], b[:C], b[C:]
            self.__dict__["_qk_cache"] = ("view", w, b)
            return w, b
        return None, None

    def fuse_qk_(self):
        """Explicit form of the storage fusing `_qk_weights()` otherwise does on the first forward: afterwards q.weight / k.weight (and the
        biases) are the two halves of ONE storage.  Call it after `load_state_dict` / `.to()` (WanTransformer4DModel.fuse_qk_() does it for
        every block) when something is about to capture parameter storages or data pointers (HIP graphs, flat-parameter wrappers)."""
        return self._qk_weights()[0] is not None

    def run(self, xn, xres, gate, gate_stride, c: _Ctx, gate_rows=None):
        """xn: T [B, Lp, C] modulated input; accumulates o-proj * gate into xres (float32) in place.  gate_rows: rows that share
        one gate vector (default: a sample's Lp rows; 1 = per-token gates)."""
        B, Lp, C = xn.shape
        n, d = self.num_heads, self.head_dim
        sm = {}                           # softmax-scale keyword of every attention call below
        if self.qk_norm:
            wq, wk = _f32(self.norm_q.weight, c.f32cache), _f32(self.norm_k.weight, c.f32cache)
            if _FOLD_QSCALE and xn.dtype == torch.bfloat16:
                wq = _folded_norm_weight(self.norm_q.weight, d, c.f32cache)
                sm = dict(scale=1.0 / LOG2E)
        else:       # norm_q / norm_k are nn.Identity (:431-432): the same kernel with NULL weights rotates only
            wq = wk = None
        rope = dict(head_dim=d, eps=self.eps, cos=c.cos, sin=c.sin, rows_per_sample=Lp, rope_len=c.rope_len,
                    pos_offset=c.pos_offset)
        q_ls = C                         # row stride of q (2C when q and k share one [rows, 2C] projection buffer)
        if c.sp is None or c.sp.world_size == 1:
            wqk, bqk = self._qk_weights() if _FUSE_QK else (None, None)
            if wqk is not None:
                # ONE projection launch for q and k (N = 2C = 10 240: twice the tiles per launch for the persistent GEMM, one
                # partial tile round and one launch less per layer); q / k are the two column halves of the [rows, 2C] result —
                # the norm+rope kernel and the attention kernel take row strides
                qk = ops.gemm_bt(xn, wqk, bqk).view(B * Lp, 2 * C)
                q, k = qk[:, :C], qk[:, C:]
                q_ls = 2 * C
            else:
                q = ops.gemm_bt(xn, self.q.weight, self.q.bias)
                k = ops.gemm_bt(xn, self.k.weight, self.k.bias)
            vt = ops.gemm_bt(self.v.weight, xn, self.v.bias, bias_on_m=True)          # V^T [C, B*Lp]
            ops.rmsnorm_rope(q, wq, k, wk, **rope)
            segs = [KV(k, vt, Lp * q_ls, q_ls, Lp, B * Lp, c.key_len)]
        elif c.sp.mode == "ulysses":
            o = self._ulysses(xn, wq, wk, rope, c, sm)
            segs = None
        else:
            # T-sharded: K and V^T are all-gathered over xGMI while the next projections run (async collectives)
            k = ops.gemm_bt(xn, self.k.weight, self.k.bias)
            ops.rmsnorm_rope(k, wk, **rope)
            hk = c.sp.gather_start(k)
            vt = ops.gemm_bt(self.v.weight, xn, self.v.bias, bias_on_m=True)
            hv = c.sp.gather_start(vt)
            q = ops.gemm_bt(xn, self.q.weight, self.q.bias)
            ops.rmsnorm_rope(q, wq, **rope)
            if c.sp.local_first:
                # attend the LOCAL shard while the peers' shards are still on the links, then the gathered remote shards,
                # and merge the two partial softmaxes through their log-sum-exps (== one softmax over all keys)
                r = c.sp.rank
                n_loc = max(0, min(Lp, c.key_len - r * Lp))
                kw = dict(B=B, Lq=Lp, heads=n, head_dim=d, q_bs=Lp * C, q_ls=C, **sm)
                o = None
                side, parts = None, []
                if _SP_OVERLAP and q.is_cuda and n_loc > 0 and not (_SP_PER_SEGMENT and sm):
                    # the gathered shards are attended on a second HIP stream: that stream (not this one) waits for the collectives,
                    # and its calls overlap the local call below
                    side = _side_stream(q.device)
                    side.wait_stream(torch.cuda.current_stream())                      # q, k, V^T are ready
                    with torch.cuda.stream(side):
                        rem = [s for i, s in enumerate(c.sp.gather_finish(hk, hv, B, Lp, C, c.key_len)) if i != r and s.len > 0]
                        for chunk in _ragged_chunks(rem, 5 if sm else len(rem)):
                            lse_r = torch.empty((B, n, Lp), device=q.device, dtype=torch.float32)
                            parts.append((ops.attention(q, chunk, lse=lse_r, **kw), lse_r))
                if n_loc > 0:
                    lse = torch.empty((B, n, Lp), device=q.device, dtype=torch.float32)
                    o = ops.attention(q, [KV(k, vt, Lp * C, C, Lp, B * Lp, n_loc)], lse=lse, **kw)
                if side is None:
                    rem = [s for i, s in enumerate(c.sp.gather_finish(hk, hv, B, Lp, C, c.key_len)) if i != r and s.len > 0]
                if o is None:           # this rank holds only padding rows: nothing local to attend
                    o = ops.attention(q, rem, **kw)
                elif rem and _SP_PER_SEGMENT and sm and len(rem) <= 3:      # (every merge rounds the running output to bf16 once more)
                    # one call per remote shard, merged through the log-sum-exps as it completes
                    lse_r, o_r = torch.empty_like(lse), torch.empty_like(o)
                    for seg in rem:
                        ops.attention(q, [seg], lse=lse_r, out=o_r, **kw)
                        ops.attn_merge_(o, lse, o_r, lse_r, B=B, L=Lp, heads=n, head_dim=d)
                elif rem and side is not None:
                    # the remote calls ran on the side stream beside the local call (a rank's call is 880 workgroups = 3.44 rounds of the
                    # 256 CUs: two calls back to back leave two partial rounds, side by side they share the tail); the merges
                    # follow on this stream in the order of the sequential path, so the bits are the same
                    torch.cuda.current_stream().wait_stream(side)
                    for o_r, lse_r in parts:
                        ops.attn_merge_(o, lse, o_r, lse_r, B=B, L=Lp, heads=n, head_dim=d)
                        o_r.record_stream(torch.cuda.current_stream())
                        lse_r.record_stream(torch.cuda.current_stream())
                elif rem:
                    # the remote shards in as few calls as attn128q_kernel allows (it stages at most five ragged tails per call: 7 remote
                    # shards of 2 730 keys at sp8 = two calls); every call's partial softmax is merged through the log-sum-exps
                    lse_r, o_r = torch.empty_like(lse), None
                    for chunk in _ragged_chunks(rem, 5 if sm else len(rem)):
                        o_r = ops.attention(q, chunk, lse=lse_r, out=o_r, **kw)
                        ops.attn_merge_(o, lse, o_r, lse_r, B=B, L=Lp, heads=n, head_dim=d)
                segs = None
            else:
                segs = c.sp.gather_finish(hk, hv, B, Lp, C, c.key_len)
        if segs is not None:
            o = ops.attention(q, segs, B=B, Lq=Lp, heads=n, head_dim=d, q_bs=Lp * q_ls, q_ls=q_ls, **sm)
        ops.gemm_bt(o, self.o.weight, self.o.bias, out=xres, epilogue=EPI_RESID_GATE, gate=gate,
                    gate_stride=gate_stride, rows_per_sample=gate_rows or Lp)
        return xres


    def _ulysses(self, xn, wq, wk, rope, c, sm):
        """Head-split sequence parallelism: project + norm + RoPE on the local tokens (WanRMSNorm runs over the full 5120-wide
        row, so it stays in front of the split), one all-to-all each for q, k, V^T (chunk j = the heads of rank j), attention of
        ALL tokens for the local heads with one K/V segment per source rank, one all-to-all back.  Buffers travel token-major
        ([W, Ls, B, c]) so that the gathered queries / outputs are plain strided [B, L, c] views."""
        B, Lp, C = xn.shape
        n, d, W = self.num_heads, self.head_dim, c.sp.world_size
        if n % W:
            raise ValueError(f"ulysses needs the head count ({n}) to be a multiple of the sequence-parallel world ({W})")
        nl = n // W
        cl = nl * d
        q = ops.gemm_bt(xn, self.q.weight, self.q.bias)
        k = ops.gemm_bt(xn, self.k.weight, self.k.bias)
        vt = ops.gemm_bt(self.v.weight, xn, self.v.bias, bias_on_m=True)           # [C, B*Lp]: rows already grouped by head
        ops.rmsnorm_rope(q, wq, k, wk, **rope)

        def out_layout(t):       # [B*Lp, C] -> [W, Lp, B, cl]
            return t.view(B, Lp, W, cl).permute(2, 1, 0, 3).contiguous()
        qg = c.sp.all_to_all(out_layout(q))                                        # [W(src), Lp, B, cl]
        kg = c.sp.all_to_all(out_layout(k))
        vg = c.sp.all_to_all(vt.view(W, cl, B * Lp))                               # [W(src), cl, B*Lp]
        segs = []
        for r in range(W):
            nk = max(0, min(Lp, c.key_len - r * Lp))
            segs.append(KV(kg[r].reshape(-1), vg[r], cl, B * cl, Lp, B * Lp, nk))
        L = W * Lp
        og = torch.empty((L, B, cl), device=q.device, dtype=q.dtype)
        ops.attention(qg.view(L, B, cl), [s for s in segs if s.len > 0], B=B, Lq=L, heads=nl, head_dim=d, q_bs=cl, q_ls=B * cl,
                      out=og.permute(1, 0, 2), **sm)
        ob = c.sp.all_to_all(og.view(W, Lp, B, cl))                                # [W(head group), Lp, B, cl]
        return ob.permute(2, 1, 0, 3).reshape(B, Lp, C)


class ContextCache:
    """Step-invariant tensors of one (prompt, CLIP) pair: embedded context and per-layer cross-attention
    K / V^T (reference recomputes them in every block of every step, :528-531, :1175-1184)."""

    def __init__(self):
        self.txt = None      # T [B, Tp, C]
        self.img = None      # T [B, Ip, C] or None
        self.txt_len = 0
        self.img_len = 0
        self.layers = {}     # layer idx -> {"txt": KV, "img": KV}

    def slice_batch(self, start):
        """View of the samples [start:] (cfg_skip runs only the conditional half)."""
        cc = ContextCache()
        cc.txt = self.txt[start:]
        cc.img = None if self.img is None else self.img[start:]
        cc.txt_len, cc.img_len = self.txt_len, self.img_len
        for layer, d in self.layers.items():
            cc.layers[layer] = {
                name: KV(kv.k[start * kv.k_bs:], kv.vt[:, start * kv.vt_bs:], kv.k_bs, kv.k_ls, kv.vt_bs, kv.vt_ls,
                         kv.len) for name, kv in d.items()}
        return cc


class WanT2VCrossAttention(WanSelfAttention):
    has_img = False

    def _kv(self, c: _Ctx, cc: ContextCache, layer):
        hit = cc.layers.get(layer)
        if hit is not None:
            return hit
        d, C = self.head_dim, self.dim
        out = {}

        def project(src, valid, kl, vl, nw):
            Bc, Sp, _ = src.shape
            k = ops.gemm_bt(src, kl.weight, kl.bias)                      # [Bc*Sp, C]
            if self.qk_norm:
                ops.rmsnorm_rope(k, _f32(nw.weight, c.f32cache), head_dim=d, eps=self.eps)
            vt = ops.gemm_bt(vl.weight, src, vl.bias, bias_on_m=True)     # V^T [C, Bc*Sp]
            # k is stored flat ([rows, C]) so that slice_batch can offset rows; strides are explicit
            return KV(k.view(-1), vt, Sp * C, C, Sp, Bc * Sp, valid)

        out["txt"] = project(cc.txt, cc.txt_len, self.k, self.v, self.norm_k)
        if self.has_img and cc.img is not None:
            out["img"] = project(cc.img, cc.img_len, self.k_img, self.v_img, self.norm_k_img)
        cc.layers[layer] = out
        return out

    def run(self, xn, xres, c: _Ctx, cc: ContextCache, layer):
        B, Lp, C = xn.shape
        n, d = self.num_heads, self.head_dim
        kv = self._kv(c, cc, layer)
        q = ops.gemm_bt(xn, self.q.weight, self.q.bias)
        if self.qk_norm:
            ops.rmsnorm_rope(q, _f32(self.norm_q.weight, c.f32cache), head_dim=d, eps=self.eps)
        if "img" in kv and q.dtype == torch.bfloat16 and d == 128 and _XATTN_FUSED:
            # text and image branches (:533-552: two attentions over the same queries, summed) in ONE launch: the query tile is
            # loaded once, the key-tile stream runs through both segments, and the image softmax is added on top of the text one
            o = ops.attention(q, [kv["txt"], kv["img"]], B=B, Lq=Lp, heads=n, head_dim=d, q_bs=Lp * C, q_ls=C, new_softmax=0b10)
        else:
            o = ops.attention(q, [kv["txt"]], B=B, Lq=Lp, heads=n, head_dim=d, q_bs=Lp * C, q_ls=C)
            if "img" in kv:  # x + img_x (:552)
                ops.attention(q, [kv["img"]], B=B, Lq=Lp, heads=n, head_dim=d, q_bs=Lp * C, q_ls=C, out=o,
                              accumulate=True)
        ops.gemm_bt(o, self.o.weight, self.o.bias, out=xres, epilogue=EPI_RESID_GATE, gate=None,
                    rows_per_sample=Lp)
        return xres


class WanI2VCrossAttention(WanT2VCrossAttention):
    has_img = True

    def __init__(self, dim, num_heads, window_size=(-1, -1), qk_norm=True, eps=1e-6):
        super().__init__(dim, num_heads, window_size, qk_norm, eps)
        self.k_img = nn.Linear(dim, dim)
        self.v_img = nn.Linear(dim, dim)
        self.norm_k_img = WanRMSNorm(dim, eps=eps) if qk_norm else nn.Identity()


class WanCrossAttention(WanT2VCrossAttention):
    pass


WAN_CROSSATTENTION_CLASSES = {
    't2v_cross_attn': WanT2VCrossAttention,
    'i2v_cross_attn': WanI2VCrossAttention,
    'cross_attn': WanCrossAttention,
}


class SpatialGuidanceModule(nn.Module):
    """Parameter container for the MoRe4D spatial guidance (reference :739-783).  The scale/shift table is
    computed once per forward for one frame (the feature map is T-periodic, :1153) by `table()`; the
    modulation itself is fused into the LayerNorm+modulate kernel."""

    def __init__(self, dim, dino_feature_dim=768):
        super().__init__()
        self.dim = dim
        self.spatial_guide = nn.Sequential(nn.SiLU(), nn.Linear(dino_feature_dim, dim * 2))
        nn.init.zeros_(self.spatial_guide[-1].weight)
        nn.init.zeros_(self.spatial_guide[-1].bias)
        self.gate = nn.Parameter(torch.zeros(dim))

    def table(self, feats_silu, f32cache):
        """feats_silu: T [B, P, 768] (SiLU already applied) -> float32 [B, P, 2*dim] = (scale | shift)."""
        lin = self.spatial_guide[1]
        B, P, D = feats_silu.shape
        return ops.gemm_bt(feats_silu.reshape(B * P, D), lin.weight, lin.bias, epilogue=EPI_STORE_F32).view(B, P, -1)


class WanAttentionBlock(nn.Module):
    def __init__(self, cross_attn_type, dim, ffn_dim, num_heads, window_size=(-1, -1), qk_norm=True,
                 cross_attn_norm=False, eps=1e-6, use_spatial_guidance=True):
        super().__init__()
        self.dim, self.ffn_dim, self.num_heads = dim, ffn_dim, num_heads
        self.window_size, self.qk_norm, self.cross_attn_norm, self.eps = window_size, qk_norm, cross_attn_norm, eps
        self.use_spatial_guidance = use_spatial_guidance
        self.norm1 = WanLayerNorm(dim, eps)
        self.self_attn = WanSelfAttention(dim, num_heads, window_size, qk_norm, eps)
        self.norm3 = WanLayerNorm(dim, eps, elementwise_affine=True) if cross_attn_norm else nn.Identity()
        self.cross_attn = WAN_CROSSATTENTION_CLASSES[cross_attn_type](dim, num_heads, (-1, -1), qk_norm, eps)
        self.norm2 = WanLayerNorm(dim, eps)
        self.ffn = nn.Sequential(nn.Linear(dim, ffn_dim), nn.GELU(approximate='tanh'), nn.Linear(ffn_dim, dim))
        self.modulation = nn.Parameter(torch.randn(1, 6, dim) / dim ** 0.5)
        if use_spatial_guidance:
            self.spatial_guidance_self = SpatialGuidanceModule(dim)
            self.spatial_guidance_ffn = SpatialGuidanceModule(dim)
        else:
            self.spatial_guidance_self = None
            self.spatial_guidance_ffn = None

    def run(self, xres, e0, c: _Ctx, cc: ContextCache, layer, guid=None):
        """xres: float32 [B, Lp, C] residual stream (updated in place); e0: float32 [B, 6, C], or [B, Lp, 6, C] for per-token
        timesteps (reference :655-657: `e.dim() > 3`) — the kernels index their modulation / gate vectors by
        `row / rows_per_sample`, so one vector per token is the same call with rows_per_sample = 1."""
        B, Lp, C = xres.shape
        T = self.ffn[0].weight.dtype
        per_token = e0.dim() == 4
        # e = modulation + e0 (:659): a [B,6,C] table, rows = shift1, scale1, gate1, shift2, scale2, gate2
        e = ops.add_bcast(e0, _f32(self.modulation, c.f32cache)).view(-1, 6, C)
        st = 6 * C
        rps = 1 if per_token else Lp
        g1 = dict(g_ss=None)
        g2 = dict(g_ss=None)
        if guid is not None and self.spatial_guidance_self is not None:
            feats_silu, period, glen = guid
            g1 = dict(g_ss=self.spatial_guidance_self.table(feats_silu, c.f32cache),
                      g_gate=_f32(self.spatial_guidance_self.gate, c.f32cache), g_period=period, g_len=glen)
            g2 = dict(g_ss=self.spatial_guidance_ffn.table(feats_silu, c.f32cache),
                      g_gate=_f32(self.spatial_guidance_ffn.gate, c.f32cache), g_period=period, g_len=glen)
            if per_token:
                # guidance on top of a per-token modulation: the modulation vectors are indexed per row (rows_per_sample = 1), the guidance
                # table by the row's position inside its sample (g_rows = Lp) — m4d_ln_modulate_g
                g1["g_rows"] = g2["g_rows"] = Lp
        # self-attention (:662-669)
        xn = ops.ln_modulate(xres, T, shift=e[:, 0], scale=e[:, 1], mod_stride=st, rows_per_sample=rps, eps=self.eps,
                             **g1)
        self.self_attn.run(xn, xres, e[:, 2], st, c, gate_rows=rps)
        # cross-attention (:674)
        if self.cross_attn_norm:
            xn = ops.ln_modulate(xres, T, ln_w=_f32(self.norm3.weight, c.f32cache),
                                 ln_b=_f32(self.norm3.bias, c.f32cache), eps=self.eps, out=xn)
        else:
            xn = ops.unary(xres, T, out=xn)
        self.cross_attn.run(xn, xres, c, cc, layer)
        # ffn (:677-684)
        xn = ops.ln_modulate(xres, T, shift=e[:, 3], scale=e[:, 4], mod_stride=st, rows_per_sample=rps, eps=self.eps,
                             out=xn, **g2)
        h = ops.gemm_bt(xn, self.ffn[0].weight, self.ffn[0].bias, epilogue=EPI_GELU_TANH)
        ops.gemm_bt(h, self.ffn[2].weight, self.ffn[2].bias, out=xres, epilogue=EPI_RESID_GATE, gate=e[:, 5],
                    gate_stride=st, rows_per_sample=rps)
        return xres


    def forward(self, x, e, seq_lens, grid_sizes, freqs, context, context_lens, dtype=torch.float32, t=0,
                dino_features=None, use_cls_token=False):
        """Reference signature (:633-646) for callers that drive one block: x [B, L, C], e [B, 6, C],
        grid_sizes [B, 3] (all samples share one grid), freqs complex [1024, d/2], context [B, 257+T, C]
        (already embedded; i2v layout: 257 CLIP tokens first, :522-523).  Returns float32 [B, L, C]."""
        B, L, C = x.shape
        dev = self.modulation.device
        T = self.ffn[0].weight.dtype
        grid = tuple(int(v) for v in (grid_sizes[0].tolist() if hasattr(grid_sizes, "tolist") else grid_sizes[0]))
        Lp = _round8(L)
        xres = torch.zeros((B, Lp, C), device=dev, dtype=torch.float32)
        xres[:, :L] = x.to(dev)
        cos, sin = build_rope_tables(freqs, grid, self.self_attn.head_dim, dev)
        f32cache = self.__dict__.setdefault("_f32cache", {})
        c = _Ctx(B, L, Lp, grid, cos, sin, min(L, grid[0] * grid[1] * grid[2]), f32cache, L)
        cc = ContextCache()
        context = context.to(dev)
        n_img = 257 if isinstance(self.cross_attn, WanI2VCrossAttention) else 0

        def padded(src):
            S = src.shape[1]
            out = torch.zeros((B, _round8(S), C), device=dev, dtype=T)
            out[:, :S] = src
            return out, S

        if n_img:
            cc.img, cc.img_len = padded(context[:, :n_img])
        cc.txt, cc.txt_len = padded(context[:, n_img:])
        guid = None
        if dino_features is not None and dino_features[0] is not None and self.spatial_guidance_self is not None:
            feats, cls = dino_features
            src = cls.expand(-1, feats.shape[1], -1) if (use_cls_token and cls is not None) else feats
            # the reference applies guidance per token over feats.shape[1] tokens (zero beyond, :772-776)
            guid = (ops.unary(src.to(dev).contiguous(), T, act=1), feats.shape[1], feats.shape[1])
        e = e.to(device=dev, dtype=torch.float32).contiguous()
        if e.dim() > 3:                              # per-token modulation [B, L, 6, C] (:655-657)
            ep = torch.zeros((B, Lp, 6, C), device=dev, dtype=torch.float32)
            ep[:, :L] = e
            e = ep
        self.run(xres, e, c, cc, 0, guid)
        return xres[:, :L]


class Head(nn.Module):
    def __init__(self, dim, out_dim, patch_size, eps=1e-6):
        super().__init__()
        self.dim, self.out_dim, self.patch_size, self.eps = dim, out_dim, patch_size, eps
        out_dim = math.prod(patch_size) * out_dim
        self.norm = WanLayerNorm(dim, eps)
        self.head = nn.Linear(dim, out_dim)
        self.modulation = nn.Parameter(torch.randn(1, 2, dim) / dim ** 0.5)

    def run(self, xres, e, f32cache):
        """xres float32 [B, Lp, C], e float32 [B, C] (or [B, Lp, C]: per-token timesteps, :713-715) -> float32
        [B, Lp, prod(patch)*out_dim] (:708-721)."""
        B, Lp, C = xres.shape
        T = self.head.weight.dtype
        per_token = e.dim() == 3
        m = ops.add_bcast(e.reshape(-1, 1, C).expand(-1, 2, C).contiguous(), _f32(self.modulation, f32cache))
        xn = ops.ln_modulate(xres, T, shift=m[:, 0], scale=m[:, 1], mod_stride=2 * C, rows_per_sample=1 if per_token else Lp,
                             eps=self.eps)
        out = ops.gemm_bt(xn, self.head.weight, self.head.bias, epilogue=EPI_STORE_F32)
        return out.view(B, Lp, -1)


class MLPProj(nn.Module):
    def __init__(self, in_dim, out_dim):
        super().__init__()
        self.proj = nn.Sequential(nn.LayerNorm(in_dim), nn.Linear(in_dim, in_dim), nn.GELU(),
                                  nn.Linear(in_dim, out_dim), nn.LayerNorm(out_dim))

    def run(self, clip_fea, T, f32cache):
        """clip_fea [B, 257, in_dim] -> T [B, 264, out_dim] (rows 257.. are padding)."""
        B, n, cin = clip_fea.shape
        Ip = _round8(n)
        src = torch.zeros((B, Ip, cin), device=clip_fea.device, dtype=T)
        src[:, :n] = clip_fea
        p = self.proj
        x = ops.ln_modulate(src, T, ln_w=_f32(p[0].weight, f32cache), ln_b=_f32(p[0].bias, f32cache), eps=p[0].eps)
        x = ops.gemm_bt(x, p[1].weight, p[1].bias, epilogue=EPI_GELU_ERF)
        x = ops.gemm_bt(x, p[3].weight, p[3].bias)
        x = ops.ln_modulate(x, T, ln_w=_f32(p[4].weight, f32cache), ln_b=_f32(p[4].bias, f32cache), eps=p[4].eps)
        return x.view(B, Ip, -1), n


class _Config(dict):
    """`transformer.config.patch_size`, `.config.get("add_ref_conv")` (pipeline_wan_fun_control.py:703, 737)."""
    __getattr__ = dict.__getitem__


class WanTransformer4DModel(nn.Module):
    r"""Wan diffusion backbone (t2v / i2v) with MoRe4D guidance hooks; MI355X-native forward."""

    _supports_gradient_checkpointing = True

    def __init__(self, model_type='t2v', patch_size=(1, 2, 2), text_len=512, in_dim=16, dim=2048, ffn_dim=8192,
                 freq_dim=256, text_dim=4096, out_dim=16, num_heads=16, num_layers=32, window_size=(-1, -1),
                 qk_norm=True, cross_attn_norm=True, eps=1e-6, in_channels=16, hidden_size=2048,
                 add_control_adapter=False, in_dim_control_adapter=24, add_ref_conv=False, in_dim_ref_conv=16,
                 cross_attn_type=None, use_dino_guidance=True, use_omnimae_guidance=False,
                 use_depth_guidance=False, use_cls_token=False):
        super().__init__()
        self.config = _Config(
            model_type=model_type, patch_size=patch_size, text_len=text_len, in_dim=in_dim, dim=dim, ffn_dim=ffn_dim,
            freq_dim=freq_dim, text_dim=text_dim, out_dim=out_dim, num_heads=num_heads, num_layers=num_layers,
            window_size=window_size, qk_norm=qk_norm, cross_attn_norm=cross_attn_norm, eps=eps,
            in_channels=in_channels, hidden_size=hidden_size, add_control_adapter=add_control_adapter,
            in_dim_control_adapter=in_dim_control_adapter, add_ref_conv=add_ref_conv,
            in_dim_ref_conv=in_dim_ref_conv, cross_attn_type=cross_attn_type, use_dino_guidance=use_dino_guidance,
            use_omnimae_guidance=use_omnimae_guidance, use_depth_guidance=use_depth_guidance,
            use_cls_token=use_cls_token)
        assert model_type in ['t2v', 'i2v', 'ti2v']
        self.model_type = model_type
        self.use_dino_guidance = use_dino_guidance
        self.use_omnimae_guidance = use_omnimae_guidance
        self.patch_size = tuple(patch_size)
        self.text_len = text_len
        self.in_dim = 48  # reference quirk: hard-coded (:866); the conv below uses the ctor value
        self.use_cls_token = use_cls_token
        self.dim, self.ffn_dim, self.freq_dim, self.text_dim = dim, ffn_dim, freq_dim, text_dim
        self.out_dim, self.num_heads, self.num_layers = out_dim, num_heads, num_layers
        self.window_size, self.qk_norm, self.cross_attn_norm, self.eps = window_size, qk_norm, cross_attn_norm, eps

        if use_dino_guidance:
            raise NotImplementedError("DINO guidance is not implemented")  # as in the reference (:881)
        self.dino_dim = 768
        if use_omnimae_guidance:
            # frozen OmniMAE ViT-B (reference :883-886; its weights travel inside the MoRe4D transformer checkpoints as
            # `omnimae_extractor.trunk.*`).  Callers may also hand precomputed [B,196,768] patch features in through
            # `first_frame_features` and skip the ViT (it is step-invariant at inference).
            from .omnimae import vit_base_mae_pretraining
            self.omnimae_extractor = vit_base_mae_pretraining(pretrained=False)
            for p_ in self.omnimae_extractor.parameters():
                p_.requires_grad = False
            self.feature_adapter = nn.Sequential(nn.Conv2d(self.dino_dim, self.dino_dim, 3, padding=1), nn.SiLU(),
                                                 nn.Conv2d(self.dino_dim, self.dino_dim, 3, padding=1))
        self.dino_extractor = None

        self.patch_embedding = nn.Conv3d(in_dim, dim, kernel_size=self.patch_size, stride=self.patch_size)
        self.text_embedding = nn.Sequential(nn.Linear(text_dim, dim), nn.GELU(approximate='tanh'), nn.Linear(dim, dim))
        self.time_embedding = nn.Sequential(nn.Linear(freq_dim, dim), nn.SiLU(), nn.Linear(dim, dim))
        self.time_projection = nn.Sequential(nn.SiLU(), nn.Linear(dim, dim * 6))

        if cross_attn_type is None:
            cross_attn_type = 't2v_cross_attn' if model_type == 't2v' else 'i2v_cross_attn'
        self.blocks = nn.ModuleList([
            WanAttentionBlock(cross_attn_type, dim, ffn_dim, num_heads, window_size, qk_norm, cross_attn_norm, eps,
                              use_spatial_guidance=(use_dino_guidance or use_omnimae_guidance))
            for _ in range(num_layers)])
        for layer_idx, block in enumerate(self.blocks):
            block.self_attn.layer_idx = layer_idx
            block.self_attn.num_layers = self.num_layers
        self.head = Head(dim, out_dim, self.patch_size, eps)

        assert (dim % num_heads) == 0 and (dim // num_heads) % 2 == 0
        d = dim // num_heads
        self.d = d
        # plain attribute, not a buffer, as in the reference (:923-935)
        with torch.device("cpu"):   # host-side table even when the module is built under a meta/device context
            self.freqs = torch.cat([rope_params(1024, d - 4 * (d // 6)), rope_params(1024, 2 * (d // 6)),
                                    rope_params(1024, 2 * (d // 6))], dim=1)
        if model_type == 'i2v':
            self.img_emb = MLPProj(1280, dim)
        if add_control_adapter:
            raise NotImplementedError("add_control_adapter: SimpleAdapter is undefined in the reference too (:941)")
        self.control_adapter = None
        self.ref_conv = nn.Conv2d(in_dim_ref_conv, dim, kernel_size=self.patch_size[1:],
                                  stride=self.patch_size[1:]) if add_ref_conv else None

        self.teacache = None
        self.cfg_skip_ratio = None
        self.current_steps = 0
        self.num_inference_steps = None
        self.gradient_checkpointing = False
        self.sp_world_size = 1
        self.sp_world_rank = 0
        self._sp = None
        self.mask_padding_keys = False   # False = the reference's SDPA branch (k_lens ignored, :222-226)
        # training: GB of HBM the forward may spend on stored activations (None = whatever is free minus the head-room
        # below; 0 = plain per-block gradient checkpointing like the reference)
        self.activation_budget_gb = None
        self.activation_headroom_gb = 95.0     # backward workspace + allocator slack (measured: beyond ~230 GB live the step time gets erratic)
        self.last_stored_blocks = 0
        self.last_full_blocks = 0
        self._f32cache = {}
        self._rope_cache = {}
        self.init_weights()

    # ------------------------------------------------------------------ reference API surface
    @property
    def dtype(self):
        return self.patch_embedding.weight.dtype

    @property
    def device(self):
        return self.patch_embedding.weight.device

    def _set_gradient_checkpointing(self, *args, **kwargs):
        if "value" in kwargs:
            self.gradient_checkpointing = kwargs["value"]
        elif "enable" in kwargs:
            self.gradient_checkpointing = kwargs["enable"]
        else:
            raise ValueError("Invalid set gradient checkpointing")

    def enable_gradient_checkpointing(self):
        self.gradient_checkpointing = True

    def fuse_qk_(self):
        """q / k projection parameters of every block become halves of one storage NOW instead of on the first forward
        (WanSelfAttention.fuse_qk_); returns the number of blocks fused."""
        return sum(int(blk.self_attn.fuse_qk_()) for blk in self.blocks)

    def enable_teacache(self, coefficients, num_steps: int, rel_l1_thresh: float, num_skip_start_steps: int = 0,
                        offload: bool = True):
        self.teacache = TeaCache(coefficients, num_steps, rel_l1_thresh=rel_l1_thresh,
                                 num_skip_start_steps=num_skip_start_steps, offload=offload)

    def share_teacache(self, transformer=None):
        self.teacache = transformer.teacache

    def disable_teacache(self):
        self.teacache = None

    def enable_cfg_skip(self, cfg_skip_ratio, num_steps):
        if cfg_skip_ratio != 0:
            self.cfg_skip_ratio, self.current_steps, self.num_inference_steps = cfg_skip_ratio, 0, num_steps
        else:
            self.disable_cfg_skip()

    def share_cfg_skip(self, transformer=None):
        self.cfg_skip_ratio = transformer.cfg_skip_ratio
        self.current_steps = transformer.current_steps
        self.num_inference_steps = transformer.num_inference_steps

    def disable_cfg_skip(self):
        self.cfg_skip_ratio, self.current_steps, self.num_inference_steps = None, 0, None

    def enable_riflex(self, k=6, L_test=66, L_test_scale=4.886):
        d = self.d
        da = d - 4 * (d // 6)
        inv = 1.0 / torch.pow(10000.0, torch.arange(0, da, 2, dtype=torch.float64).div(da))
        inv[k - 1] = 0.9 * 2 * torch.pi / L_test          # RIFLEx intrinsic-frequency edit (reference :306-310)
        if L_test_scale is not None:
            inv[k - 1] = inv[k - 1] / L_test_scale
        ang = torch.outer(torch.arange(1024, dtype=torch.float64), inv)
        self.freqs = torch.cat([torch.polar(torch.ones_like(ang), ang), rope_params(1024, 2 * (d // 6)),
                                rope_params(1024, 2 * (d // 6))], dim=1)
        self._rope_cache.clear()

    def disable_riflex(self):
        d = self.d
        self.freqs = torch.cat([rope_params(1024, d - 4 * (d // 6)), rope_params(1024, 2 * (d // 6)),
                                rope_params(1024, 2 * (d // 6))], dim=1)
        self._rope_cache.clear()

    def enable_multi_gpus_inference(self):
        """T-(token-)sharded denoising across the ranks of the sequence-parallel group (more4d_amd.dist)."""
        from ..dist import get_sequence_parallel_rank, get_sequence_parallel_world_size, get_sp_group
        self.sp_world_size = get_sequence_parallel_world_size()
        self.sp_world_rank = get_sequence_parallel_rank()
        self._sp = get_sp_group()
        self.all_gather = self._sp.all_gather

    # ------------------------------------------------------------------ host-side tables
    def _rope_tables(self, grid, device):
        key = (tuple(grid), str(device))
        hit = self._rope_cache.get(key)
        if hit is None:
            hit = build_rope_tables(self.freqs, grid, self.d, device)
            self._rope_cache[key] = hit
        return hit

    # ------------------------------------------------------------------ embeddings (step-invariant part)
    def prepare_context(self, context, clip_fea=None) -> ContextCache:
        """text_embedding / img_emb once per prompt (:1174-1184); per-layer K,V^T are filled lazily."""
        T, dev = self.dtype, self.device
        B = len(context)
        cc = ContextCache()
        Tp = _round8(self.text_len)
        txt = torch.zeros((B, Tp, self.text_dim), device=dev, dtype=T)
        for i, u in enumerate(context):
            txt[i, :u.size(0)] = u.to(device=dev, dtype=T)
        h = ops.gemm_bt(txt, self.text_embedding[0].weight, self.text_embedding[0].bias, epilogue=EPI_GELU_TANH)
        cc.txt = ops.gemm_bt(h, self.text_embedding[2].weight, self.text_embedding[2].bias).view(B, Tp, self.dim)
        cc.txt_len = self.text_len
        if clip_fea is not None and self.model_type == 'i2v':
            cc.img, cc.img_len = self.img_emb.run(clip_fea.to(device=dev), T, self._f32cache)
        return cc

    def _time_embed(self, t):
        """e [B, C], e0 [B, 6, C] float32 (:1160-1171): fp32 GEMMs regardless of T.  t may hold any number of timesteps (per-token
        calls pass the flattened [B * seq_len] vector, :1161-1167)."""
        dev = self.device
        s = sinusoidal_embedding_1d(self.freq_dim, t.to(dev)).float().contiguous()
        fc = self._f32cache
        te0, te2, tp = self.time_embedding[0], self.time_embedding[2], self.time_projection[1]
        h = ops.gemm_bt(s, _f32(te0.weight, fc), _f32(te0.bias, fc), epilogue=EPI_SILU)
        e = ops.gemm_bt(h, _f32(te2.weight, fc), _f32(te2.bias, fc))
        e0 = ops.gemm_bt(ops.unary(e, torch.float32, act=1), _f32(tp.weight, fc), _f32(tp.bias, fc))
        return e, e0.view(-1, 6, self.dim)

    # ------------------------------------------------------------------ forward
    @cfg_skip()
    def forward(self, x, t, context, seq_len, clip_fea=None, y=None, y_camera=None, full_ref=None, subject_ref=None,
                cond_flag=True, first_frame=None, first_frame_features=None):
        """Same contract as the reference forward (:1047-1340).

        x [B,16,F,H,W]; t [B]; context: list of [Li, text_dim] tensors OR a ContextCache from
        `prepare_context`; clip_fea [B,257,1280]; y [B,48,F,H,W]; full_ref [B,16,H,W].
        first_frame: [B,3,H,W] in [0,1] -> OmniMAE ViT-B patch features (reference :1126-1146); or pass the
        features directly as first_frame_features = (patch_feats [B,196,768], cls [B,768]).  Returns
        [B, out_dim, F, H, W] in T.
        """
        if y_camera is not None:      # (the reference's camera adapter class is undefined, :941 — SURVEY appendix E.2)
            raise NotImplementedError("y_camera: the reference's control adapter (SimpleAdapter) does not exist")
        if first_frame is not None and self.use_omnimae_guidance and first_frame_features is None:
            first_frame_features = self.omnimae_extractor.trunk.forward_patch_features(first_frame, None, normalize=True)
        if torch.is_grad_enabled() and any(p.requires_grad for p in self.parameters()):
            return self._forward_train(x, t, context, seq_len, clip_fea, y, full_ref, first_frame_features, subject_ref)
        T, dev = self.dtype, self.device
        B = x.shape[0]
        x = x.to(dev)
        if y is not None:
            y = y.to(device=dev, dtype=x.dtype)
        pt, ph, pw = self.patch_size
        f, h, w = x.shape[2] // pt, x.shape[3] // ph, x.shape[4] // pw
        Lv = f * h * w
        n_ref = 0
        grid = (f, h, w)
        if self.ref_conv is not None and full_ref is not None:
            n_ref = h * w
            grid = (f + 1, h, w)
            seq_len = seq_len + n_ref
        # subject_ref (:1092-1097): extra frames [B, in_dim, Fs, H, W] through the SAME patch embedding, appended BEHIND the video tokens (their
        # RoPE frame index continues) and cut off again after the head (:1328-1331).  (The reference cuts `subject_ref[0].size(1)` tokens
        # there, which is the model width, not the token count; its unpatchify then takes the first prod(grid) tokens, so its result is
        # the one below whenever it runs at all — tests/golden/make_golden_r6.py.)
        n_sub = 0
        if subject_ref is not None:
            if subject_ref.shape[3] // ph != h or subject_ref.shape[4] // pw != w:
                raise ValueError("subject_ref must have the video's spatial size")
            fs = subject_ref.shape[2] // pt
            n_sub = fs * h * w
            grid = (grid[0] + fs, h, w)
            seq_len = seq_len + n_sub
        L = Lv + n_ref + n_sub
        sp = self._sp if self.sp_world_size > 1 else None
        # keys: the reference's SDPA branch attends the zero rows up to seq_len (:222-226); rows added only to
        # make the sequence divisible by the SP world (:1100-1101) are masked so N ranks == 1 rank.
        key_len = L if self.mask_padding_keys else seq_len
        if sp is not None:
            seq_len = int(math.ceil(seq_len / self.sp_world_size)) * self.sp_world_size
        seq_len_tok = seq_len                     # what a per-token t is unflattened to (:1166, after the SP round-up of :1100)
        assert L <= seq_len, f"sequence of {L} tokens exceeds seq_len={seq_len}"
        # ---- tokens: patch gather + GEMM straight into the fp32 residual stream
        Lp = _round8(seq_len) if sp is None else _round8(seq_len // self.sp_world_size) * self.sp_world_size
        xres = torch.zeros((B, Lp, self.dim), device=dev, dtype=torch.float32)
        tok = ops.patchify(x, y, self.patch_size, T)
        wpe = self.patch_embedding.weight.view(self.dim, -1)
        for b in range(B):
            ops.gemm_bt(tok[b], wpe, self.patch_embedding.bias, out=xres[b, n_ref:n_ref + Lv], epilogue=EPI_STORE_F32)
        if n_ref:
            rt = ops.patchify(full_ref.to(dev).unsqueeze(2), None, (1, ph, pw), T)
            wr = self.ref_conv.weight.view(self.dim, -1)
            for b in range(B):
                ops.gemm_bt(rt[b], wr, self.ref_conv.bias, out=xres[b, :n_ref], epilogue=EPI_STORE_F32)
        if n_sub:
            stok = ops.patchify(subject_ref.to(device=dev, dtype=x.dtype), None, self.patch_size, T)
            for b in range(B):
                ops.gemm_bt(stok[b], wpe, self.patch_embedding.bias, out=xres[b, n_ref + Lv:n_ref + Lv + n_sub], epilogue=EPI_STORE_F32)
        # ---- conditioning
        if t.dim() != 1:
            # per-token timesteps (:1161-1167): t [B, seq_len] (seq_len AFTER the ref row was added, :1088) -> e [B, Lp, C],
            # e0 [B, Lp, 6, C]; rows beyond seq_len exist only as padding and get a zero modulation
            if t.shape[0] != B or t.shape[1] != seq_len_tok:
                raise ValueError(f"per-token t must be [B, seq_len] = [{B}, {seq_len_tok}], got {tuple(t.shape)}")
            ef, e0f = self._time_embed(t.reshape(-1))
            e = torch.zeros((B, Lp, self.dim), device=dev, dtype=torch.float32)
            e0 = torch.zeros((B, Lp, 6, self.dim), device=dev, dtype=torch.float32)
            e[:, :seq_len_tok] = ef.view(B, seq_len_tok, self.dim)
            e0[:, :seq_len_tok] = e0f.view(B, seq_len_tok, 6, self.dim)
        else:
            e, e0 = self._time_embed(t)
        cc = context if isinstance(context, ContextCache) else self.prepare_context(context, clip_fea)
        cos, sin = self._rope_tables(grid, dev)
        guid = None
        if self.use_omnimae_guidance and first_frame_features is not None:
            guid = self._guidance_tables(first_frame_features, (h, w), x.shape[2] // pt)
        # ---- shard for sequence parallelism (token axis == f-major == T axis, reference :1187-1198)
        pos_offset = 0
        if sp is not None:
            Ls = Lp // self.sp_world_size
            pos_offset = self.sp_world_rank * Ls
            xres = xres[:, pos_offset:pos_offset + Ls].contiguous()
            if t.dim() != 1:                      # :1190-1192: the per-token tables are chunked like the tokens
                e = e[:, pos_offset:pos_offset + Ls].contiguous()
                e0 = e0[:, pos_offset:pos_offset + Ls].contiguous()
            c = _Ctx(B, L, Ls, grid, cos, sin, max(0, min(Ls, L - pos_offset)), self._f32cache, key_len, sp, pos_offset)
        else:
            c = _Ctx(B, L, Lp, grid, cos, sin, L, self._f32cache, key_len)
        if guid is not None and sp is not None:
            # token l of this rank is global token pos_offset + l: the T-periodic guidance table is rotated to start at this
            # rank's first position and its length is what remains of the guided range (one roll of [B, P, 768] per forward)
            feats_silu, period, glen = guid
            guid = (torch.roll(feats_silu, shifts=-(c.pos_offset % period), dims=1).contiguous(), period,
                    max(0, min(c.Lp, glen - c.pos_offset)))
        # ---- TeaCache (reference :1201-1270): skip the blocks when the accumulated, rescaled relative-L1 change of the
        # modulated timestep embedding stays under the threshold, re-using the previous residual of the token stream
        should_calc = True
        tc = self.teacache
        if tc is not None:
            if cond_flag:
                if tc.cnt < tc.num_skip_start_steps or tc.previous_modulated_input is None:
                    should_calc = True
                    tc.accumulated_rel_l1_distance = 0
                else:
                    rel = ops.rel_l1(tc.previous_modulated_input, e0)
                    tc.accumulated_rel_l1_distance += float(tc.rescale_func(rel))
                    if tc.accumulated_rel_l1_distance < tc.rel_l1_thresh:
                        should_calc = False
                    else:
                        tc.accumulated_rel_l1_distance = 0
                tc.previous_modulated_input = e0.clone()
                tc.should_calc = should_calc
            else:
                should_calc = tc.should_calc
            prev = tc.previous_residual_cond if cond_flag else tc.previous_residual_uncond
            if prev is None:
                should_calc = True
            self.should_calc = should_calc       # the reference keeps the step's decision on the model too (:1206-1220)
        if should_calc:
            ori = xres.clone() if tc is not None else None
            for i, block in enumerate(self.blocks):
                block.run(xres, e0, c, cc, i, guid)
            if tc is not None:      # residual stays in HBM (288 GB) — the reference's `offload` flag is accepted and ignored
                res_ = ops.axpby(xres, ori, 1.0, -1.0)
                if cond_flag:
                    tc.previous_residual_cond = res_
                else:
                    tc.previous_residual_uncond = res_
        else:
            xres = ops.axpby(xres, prev[-xres.size(0):].contiguous(), 1.0, 1.0)
        if tc is not None and cond_flag:
            tc.cnt += 1
            if tc.cnt == tc.num_steps:
                tc.reset()
        out = self.head.run(xres, e, self._f32cache)          # float32 [B, Lp or Ls, 64]
        if sp is not None:
            out = sp.all_gather(out, dim=1)
        res = ops.unpatchify(out.contiguous(), n_ref, (f, h, w), self.patch_size, self.out_dim, T)
        return res

    # ------------------------------------------------------------------ training forward (autograd tape over HIP kernels)
    def _forward_train(self, x, t, context, seq_len, clip_fea=None, y=None, full_ref=None, first_frame_features=None, subject_ref=None):
        """Differentiable forward for `train_wan.py:1939-1951` (same arguments, same result as `forward`): every node
        is a `more4d_amd.autograd` Function whose forward AND backward run in the HIP kernels; one recomputing node per
        block (the reference trains with gradient checkpointing, :1273-1291).  Data parallel only."""
        from ..autograd import ActFn, BlockFn, GuidanceAdapterFn, LayerNormFn, LinearFn
        from ..ops import ACT_GELU_ERF, ACT_GELU_TANH, ACT_SILU
        if self.sp_world_size > 1:
            raise NotImplementedError("training is data parallel (DDP); sequence parallelism is the inference path")
        if self.teacache is not None:
            raise NotImplementedError("TeaCache is an inference-time approximation")
        if isinstance(context, ContextCache):
            raise ValueError("training needs the raw text embeddings (the context projections are trainable)")
        T, dev, C = self.dtype, self.device, self.dim
        f32 = torch.float32
        B = x.shape[0]
        x = x.to(dev)
        if y is not None:
            y = y.to(device=dev, dtype=x.dtype)
        pt, ph, pw = self.patch_size
        f, h, w = x.shape[2] // pt, x.shape[3] // ph, x.shape[4] // pw
        Lv, n_ref, grid = f * h * w, 0, (f, h, w)
        if self.ref_conv is not None and full_ref is not None:
            n_ref, grid = h * w, (f + 1, h, w)
            seq_len = seq_len + n_ref
        n_sub = 0
        if subject_ref is not None:      # (:1092-1097, see forward)
            if subject_ref.shape[3] // ph != h or subject_ref.shape[4] // pw != w:
                raise ValueError("subject_ref must have the video's spatial size")
            fs = subject_ref.shape[2] // pt
            n_sub, grid = fs * h * w, (grid[0] + fs, h, w)
            seq_len = seq_len + n_sub
        L = Lv + n_ref + n_sub
        key_len = L if self.mask_padding_keys else seq_len
        assert L <= seq_len, f"sequence of {L} tokens exceeds seq_len={seq_len}"
        Lp = _round8(seq_len)
        # ---- tokens
        parts = []
        if n_ref:
            rt = ops.patchify(full_ref.to(dev).unsqueeze(2), None, (1, ph, pw), T)
            parts.append(LinearFn.apply(rt, self.ref_conv.weight, self.ref_conv.bias, 0, T, True))
        parts.append(LinearFn.apply(ops.patchify(x, y, self.patch_size, T), self.patch_embedding.weight,
                                    self.patch_embedding.bias, 0, T, True))
        if n_sub:
            parts.append(LinearFn.apply(ops.patchify(subject_ref.to(device=dev, dtype=x.dtype), None, self.patch_size, T),
                                        self.patch_embedding.weight, self.patch_embedding.bias, 0, T, True))
        if Lp > L:
            parts.append(torch.zeros((B, Lp - L, C), device=dev, dtype=f32))
        xres = torch.cat(parts, dim=1) if len(parts) > 1 else parts[0]
        # ---- conditioning (float32 like the inference path, :1160-1171)
        te0, te2, tp = self.time_embedding[0], self.time_embedding[2], self.time_projection[1]
        per_token = t.dim() != 1
        if per_token and (t.shape[0] != B or t.shape[1] != seq_len):
            # (:1161-1167) t [B, seq_len], seq_len AFTER the reference row was added (:1088)
            raise ValueError(f"per-token t must be [B, seq_len] = [{B}, {seq_len}], got {tuple(t.shape)}")
        s = sinusoidal_embedding_1d(self.freq_dim, t.reshape(-1).to(dev)).float().contiguous()
        e = LinearFn.apply(LinearFn.apply(s, te0.weight, te0.bias, ACT_SILU, f32, True), te2.weight, te2.bias, 0, f32, True)
        e0 = LinearFn.apply(ActFn.apply(e, ACT_SILU, f32), tp.weight, tp.bias, 0, f32, True)
        if per_token:      # e [B, Lp, C], e0 [B, Lp, 6, C]: one vector per token, rows beyond seq_len exist only as padding (zero modulation)
            e, e0 = e.view(B, seq_len, C), e0.view(B, seq_len, 6, C)
            if Lp > seq_len:
                e = torch.cat([e, e.new_zeros(B, Lp - seq_len, C)], dim=1)
                e0 = torch.cat([e0, e0.new_zeros(B, Lp - seq_len, 6, C)], dim=1)
            e0 = e0.contiguous()
        else:
            e0 = e0.view(B, 6, C)
        Tp = _round8(self.text_len)
        txt = torch.zeros((B, Tp, self.text_dim), device=dev, dtype=T)
        for i, u in enumerate(context):
            txt[i, :u.size(0)] = u.to(device=dev, dtype=T)
        te = self.text_embedding
        ctx_txt = LinearFn.apply(LinearFn.apply(txt, te[0].weight, te[0].bias, ACT_GELU_TANH, T, False),
                                 te[2].weight, te[2].bias, 0, T, False)
        ctx_img, img_len = None, 0
        if clip_fea is not None and self.model_type == 'i2v':
            n = clip_fea.shape[1]
            src = torch.zeros((B, _round8(n), clip_fea.shape[2]), device=dev, dtype=f32)
            src[:, :n] = clip_fea.to(dev)
            p = self.img_emb.proj
            a = LayerNormFn.apply(src, p[0].weight, p[0].bias, None, None, p[0].eps, T)
            a = LinearFn.apply(a, p[1].weight, p[1].bias, ACT_GELU_ERF, T, False)
            a = LinearFn.apply(a, p[3].weight, p[3].bias, 0, T, True)
            ctx_img, img_len = LayerNormFn.apply(a, p[4].weight, p[4].bias, None, None, p[4].eps, T), n
        cos, sin = self._rope_tables(grid, dev)
        c = _Ctx(B, L, Lp, grid, cos, sin, L, self._f32cache, key_len)
        # activation policy (288 GB of HBM3E per GPU): within the budget, every block first keeps its STORE_LITE tensors
        # (self-attention + ffn_down outputs), then blocks are upgraded to keeping all GEMM outputs; the rest recompute
        es = xres.new_empty(0, dtype=T).element_size()
        lite = B * Lp * 2 * C * es + B * self.num_heads * Lp * 4
        full = B * Lp * (8 * C + self.ffn_dim) * es + B * self.num_heads * Lp * 4
        budget = self.activation_budget_gb
        if budget is None:
            if dev.type == "cuda":
                total = torch.cuda.get_device_properties(dev).total_memory
                budget = max(0.0, (total - torch.cuda.memory_allocated(dev)) / 2 ** 30 - self.activation_headroom_gb)
            else:
                budget = 0.0
        budget = max(0.0, budget) * 2 ** 30
        nb = len(self.blocks)
        n_lite = int(min(nb, budget // max(lite, 1)))
        n_full = int(min(n_lite, max(0.0, budget - n_lite * lite) // max(full - lite, 1))) if n_lite == nb else 0
        self.last_stored_blocks = n_lite
        self.last_full_blocks = n_full
        # ---- spatial guidance (:1126-1156): SiLU'd [B, h*w, 768] table shared by every block, T-periodic over the tokens
        gfeat, gmeta = None, None
        if self.use_omnimae_guidance and first_frame_features is not None:
            patch, cls = first_frame_features
            if not self.use_cls_token and patch.dim() == 3 and patch.shape[1] == 196 and hasattr(self, "feature_adapter"):
                fa = self.feature_adapter
                gfeat = GuidanceAdapterFn.apply(patch.to(dev), (h, w), fa[0].weight, fa[0].bias, fa[2].weight, fa[2].bias, T)
                gmeta = (h * w, (x.shape[2] // pt) * h * w)
            else:
                gfeat, period, glen = self._guidance_tables(first_frame_features, (h, w), x.shape[2] // pt)
                gmeta = (period, glen)
        for i, blk in enumerate(self.blocks):
            store = 2 if i < n_full else (1 if i < n_lite else 0)
            xres = BlockFn.apply(xres, e0, ctx_txt, ctx_img, gfeat, blk, c, self.text_len, img_len, store, gmeta,
                                 *blk.parameters())
        # ---- head (:708-721) + unpatchify (:1343-1366)
        if per_token:      # (:713-715) [B, Lp, 2, C]
            m = e.unsqueeze(2) + self.head.modulation.float().unsqueeze(0)
            xn = LayerNormFn.apply(xres, None, None, m[:, :, 0], m[:, :, 1], self.head.eps, T)
        else:
            m = e.view(B, 1, C) + self.head.modulation.float()
            xn = LayerNormFn.apply(xres, None, None, m[:, 0], m[:, 1], self.head.eps, T)
        out = LinearFn.apply(xn, self.head.head.weight, self.head.head.bias, 0, T, True)
        u = out[:, n_ref:n_ref + Lv].reshape(B, f, h, w, pt, ph, pw, self.out_dim).permute(0, 7, 1, 4, 2, 5, 3, 6)
        return u.reshape(B, self.out_dim, f * pt, h * ph, w * pw).to(T)

    def _guidance_tables(self, feats, hw, latent_T):
        """OmniMAE patch features -> SiLU'd, adapter-convolved, resized [B, h*w, 768] table (reference
        :1149-1156).  Raw 14x14 OmniMAE patch features go through `_adapt_features` (the 3x3 adapter convs on the VAE conv
        kernel + bilinear resize, once per forward); features already adapted to (h, w) are taken as they are."""
        patch, cls = feats
        T = self.dtype
        h, w = hw
        if patch.dim() == 3 and patch.shape[1] == 196 and hasattr(self, "feature_adapter"):
            patch = self._adapt_features(patch, hw)          # raw OmniMAE 14x14 patch features (:1150-1152)
        if patch.dim() != 3 or patch.shape[1] != h * w or patch.shape[2] != self.dino_dim:
            raise ValueError("first_frame_features: expected OmniMAE patch features [B,196,768] or adapted [B,h*w,768]")
        src = cls.view(cls.shape[0], 1, -1).expand(-1, h * w, -1) if self.use_cls_token else patch
        return ops.unary(src.to(self.device).contiguous(), T, act=1), h * w, latent_T * h * w

    def _adapt_features(self, patch, hw):
        """feature_adapter (Conv3x3 - SiLU - Conv3x3 on the 14x14 map) + bilinear resize to the token grid (reference
        :889-893, :1150-1152).  [B,196,768] is already channels-last [B,14,14,768]."""
        T, dev = self.dtype, self.device
        B = patch.shape[0]
        x = patch.to(device=dev, dtype=T).contiguous().view(B * 196, self.dino_dim)

        def packed(conv):
            key = (conv.weight._version, conv.weight.data_ptr(), T)
            hit = self._f32cache.get(("pk", id(conv)))
            if hit is None or hit[0] != key:
                wp = conv.weight.detach().to(T).permute(0, 2, 3, 1).contiguous().view(conv.weight.shape[0], -1)
                hit = (key, wp, conv.bias.detach().to(T).contiguous())
                self._f32cache[("pk", id(conv))] = hit
            return hit[1], hit[2]

        for j, conv in ((0, self.feature_adapter[0]), (2, self.feature_adapter[2])):
            wp, bp = packed(conv)
            x = ops.conv_cl(x, wp, bp, Tin=B, Hin=14, Win=14, Cin=self.dino_dim, k=(1, 3, 3), pad=(0, 1, 1),
                            out_thw=(B, 14, 14))
            if j == 0:
                x = ops.unary(x, T, act=1)
        y = ops.bilinear_cl(x.view(B, 14, 14, self.dino_dim), hw)
        return y.view(B, hw[0] * hw[1], self.dino_dim)

    def save_pretrained(self, save_directory, max_shard_size_gb=10.0):
        """diffusers-layout checkpoint directory (what `models[0].save_pretrained(...)` writes in the reference's save hook,
        train_wan.py:1011, and what `from_pretrained` reads back): config.json + diffusion_pytorch_model[-0000i-of-0000n]
        .safetensors shards (+ the shard index)."""
        import json
        import os
        from safetensors.torch import save_file
        os.makedirs(save_directory, exist_ok=True)
        cfg = {k: (list(v) if isinstance(v, tuple) else v) for k, v in self.config.items()}
        cfg["_class_name"] = type(self).__name__
        with open(os.path.join(save_directory, "config.json"), "w") as fh:
            json.dump(cfg, fh, indent=2, sort_keys=True)
        sd = {k: v.detach().cpu().contiguous() for k, v in self.state_dict().items()}
        limit = int(max_shard_size_gb * 2 ** 30)
        shards, cur, size = [], {}, 0
        for k, v in sd.items():
            nbytes = v.numel() * v.element_size()
            if cur and size + nbytes > limit:
                shards.append(cur)
                cur, size = {}, 0
            cur[k] = v
            size += nbytes
        shards.append(cur)
        if len(shards) == 1:
            save_file(shards[0], os.path.join(save_directory, "diffusion_pytorch_model.safetensors"), metadata={"format": "pt"})
            return
        index = {"metadata": {"total_size": sum(v.numel() * v.element_size() for v in sd.values())}, "weight_map": {}}
        for i, sh in enumerate(shards):
            name = f"diffusion_pytorch_model-{i + 1:05d}-of-{len(shards):05d}.safetensors"
            save_file(sh, os.path.join(save_directory, name), metadata={"format": "pt"})
            index["weight_map"].update({k: name for k in sh})
        with open(os.path.join(save_directory, "diffusion_pytorch_model.safetensors.index.json"), "w") as fh:
            json.dump(index, fh, indent=2, sort_keys=True)

    def unpatchify(self, x, grid_sizes):
        """Reference-compatible helper (:1343-1366) for callers that hold head outputs: x list of [L, 64]."""
        outs = []
        for u, v in zip(x, grid_sizes.tolist() if hasattr(grid_sizes, "tolist") else grid_sizes):
            tok = u[:math.prod(v)].float().contiguous().unsqueeze(0)
            outs.append(ops.unpatchify(tok, 0, tuple(v), self.patch_size, self.out_dim, u.dtype)[0])
        return outs

    def init_weights(self):
        """Same initialisation scheme as the reference (:1368-1390)."""
        for m in self.modules():
            if isinstance(m, nn.Linear):
                nn.init.xavier_uniform_(m.weight)
                if m.bias is not None:
                    nn.init.zeros_(m.bias)
        nn.init.xavier_uniform_(self.patch_embedding.weight.flatten(1))
        for m in self.text_embedding.modules():
            if isinstance(m, nn.Linear):
                nn.init.normal_(m.weight, std=.02)
        for m in self.time_embedding.modules():
            if isinstance(m, nn.Linear):
                nn.init.normal_(m.weight, std=.02)
        nn.init.zeros_(self.head.head.weight)
        # NB: like in the reference, this pass also re-initialises the Linear of every SpatialGuidanceModule that its own
        # constructor had zeroed (:750-751): a fresh model starts with Xavier scale/shift projections behind zero gates —
        # with both at zero neither would ever receive a gradient.

    @classmethod
    def from_pretrained(cls, pretrained_model_path, subfolder=None, transformer_additional_kwargs={},
                        low_cpu_mem_usage=False, torch_dtype=torch.bfloat16):
        """Load a Wan / VideoX-Fun / MoRe4D checkpoint directory: config.json + *.safetensors shards
        (reference :1392-1523).  Size-mismatched keys are skipped, like the reference does."""
        import glob
        import json
        import os
        if subfolder is not None:
            pretrained_model_path = os.path.join(pretrained_model_path, subfolder)
        config_file = os.path.join(pretrained_model_path, 'config.json')
        if not os.path.isfile(config_file):
            raise RuntimeError(f"{config_file} does not exist")
        with open(config_file, "r") as fh:
            config = json.load(fh)
        import inspect
        valid = set(inspect.signature(cls.__init__).parameters) - {"self"}
        kwargs = {k: v for k, v in {**config, **transformer_additional_kwargs}.items() if k in valid}
        model = cls(**kwargs)
        files = sorted(glob.glob(os.path.join(pretrained_model_path, "*.safetensors")))
        state = {}
        if files:
            from safetensors.torch import load_file
            for fpath in files:
                state.update(load_file(fpath))
        else:
            bin_file = os.path.join(pretrained_model_path, "diffusion_pytorch_model.bin")
            if not os.path.exists(bin_file):
                raise RuntimeError(f"no weights found under {pretrained_model_path}")
            state = torch.load(bin_file, map_location="cpu", weights_only=True)
        own = model.state_dict()
        filtered = {k: v for k, v in state.items() if k in own and own[k].shape == v.shape}
        skipped = [k for k in state if k not in filtered]
        m, u = model.load_state_dict(filtered, strict=False)
        print(f"### missing keys: {len(m)}; ### unexpected/size-mismatched keys skipped: {len(skipped)}")
        return model.to(torch_dtype)
