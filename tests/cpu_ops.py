"""TEST-ONLY stand-in for `more4d_amd.ops` on machines without a GPU.

The product never falls back to this: tests monkeypatch `more4d_amd.ops` functions with these so that the
HOST LOGIC (token sharding, RoPE offsets, K/V segment bookkeeping, all-gather plumbing, caches, cfg-skip,
the pipeline loop) can be exercised in this container and under `gloo` with world_size 2.  Arithmetic here
is plain fp32 torch / the oracle; it says nothing about the kernels (those are tested with -m gpu)."""
import math

import torch
import torch.nn.functional as F

from more4d_amd.ops import (EPI_GELU_ERF, EPI_GELU_TANH, EPI_RESID_GATE, EPI_SILU, EPI_STORE,  # noqa: F401
                            EPI_STORE_F32, KV, _rows2d)

NAMES = ["gemm_bt", "ln_modulate", "rmsnorm_rope", "attention", "attn_merge_", "patchify", "unpatchify", "cfg_euler_", "unary",
         "add_bcast"]


def install(monkeypatch):
    import more4d_amd.ops as real
    import sys
    me = sys.modules[__name__]
    for n in NAMES:
        monkeypatch.setattr(real, n, getattr(me, n))


def gemm_bt(a, w, bias=None, *, out=None, epilogue=EPI_STORE, gate=None, gate_stride=0, rows_per_sample=0,
            bias_on_m=False, out_rows_ld=None):
    M, _ = _rows2d(a)
    N, _ = _rows2d(w)
    a2 = a.reshape(M, a.shape[-1]).float()
    y = a2 @ w.reshape(N, w.shape[-1]).float().t()
    if bias is not None:
        y = y + (bias.float()[:, None] if bias_on_m else bias.float())
    if epilogue == EPI_GELU_TANH:
        y = F.gelu(y, approximate="tanh")
    elif epilogue == EPI_GELU_ERF:
        y = F.gelu(y)
    elif epilogue == EPI_SILU:
        y = F.silu(y)
    if epilogue == EPI_RESID_GATE:
        o2 = out.view(M, N)
        if gate is not None:
            rps = rows_per_sample or M
            B = M // rps
            # gate is a view into a [B, k, N] table: element (b, n) at gate.data + b*gate_stride + n
            g = torch.stack([torch.as_strided(gate, (N,), (1,), gate.storage_offset() + b * gate_stride)
                             for b in range(B)])
            y = y.to(a.dtype).float() * g.repeat_interleave(rps, dim=0)
        else:
            y = y.to(a.dtype).float()
        o2 += y
        return out
    if epilogue == EPI_STORE_F32:
        y = y.to(a.dtype).float()
        if out is None:
            return y
        out.copy_(y.view(out.shape))
        return out
    y = y.to(a.dtype)
    if out is None:
        return y
    out.copy_(y.view(out.shape))
    return out


def _strided_rows(t, B, stride, C):
    return torch.stack([torch.as_strided(t, (C,), (1,), t.storage_offset() + b * stride) for b in range(B)])


def ln_modulate(x, out_dtype, *, shift=None, scale=None, mod_stride=0, rows_per_sample=0, ln_w=None, ln_b=None,
                eps=1e-6, g_ss=None, g_gate=None, g_period=0, g_len=0, g_rows=0, out=None):
    C = x.shape[-1]
    rows = x.numel() // C
    rps = rows_per_sample or rows
    B = rows // rps
    xf = x.reshape(B, rps, C).float()
    mu = xf.mean(-1, keepdim=True)
    y = (xf - mu) * torch.rsqrt((xf - mu).pow(2).mean(-1, keepdim=True) + eps)
    if ln_w is not None:
        y = y * ln_w + ln_b
    if scale is not None:
        y = y * (1 + _strided_rows(scale, B, mod_stride, C)[:, None]) + _strided_rows(shift, B, mod_stride, C)[:, None]
    if g_ss is not None:      # guidance samples of g_rows rows (default: the modulation's rows_per_sample)
        gr = g_rows or rps
        Bg = rows // gr
        y = y.reshape(Bg, gr, C)
        sc = torch.zeros(Bg, gr, C)
        sh = torch.zeros(Bg, gr, C)
        n = min(g_len, gr)
        idx = torch.arange(n) % g_period
        sc[:, :n] = g_ss[:, idx, :C]
        sh[:, :n] = g_ss[:, idx, C:]
        y = y * (1 + sc * g_gate) + sh * g_gate
    y = y.reshape(x.shape).to(out_dtype)
    if out is not None:
        out.copy_(y)
        return out
    return y


def rmsnorm_rope(x0, w0, x1=None, w1=None, *, head_dim, eps=1e-6, cos=None, sin=None, rows_per_sample=0,
                 rope_len=0, pos_offset=0):
    for x, w in ((x0, w0), (x1, w1)):
        if x is None:
            continue
        C = x.shape[-1]
        rows = x.numel() // C
        rps = rows_per_sample or rows
        xf = x.reshape(rows, C).float()
        if w is None:      # RoPE only (qk_norm=False)
            y = xf.clone()
        else:
            y = (xf * torch.rsqrt(xf.pow(2).mean(-1, keepdim=True) + eps)).to(x.dtype).float() * w
        if cos is not None:
            y = y.view(rows // rps, rps, C // head_dim, head_dim // 2, 2)
            n = min(rope_len, rps)
            c = cos[pos_offset:pos_offset + n].view(1, n, 1, -1)
            s = sin[pos_offset:pos_offset + n].view(1, n, 1, -1)
            a, b = y[:, :n, :, :, 0].clone(), y[:, :n, :, :, 1].clone()
            y[:, :n, :, :, 0] = a * c - b * s
            y[:, :n, :, :, 1] = a * s + b * c
        x.copy_(y.reshape(x.shape).to(x.dtype))
    return x0, x1


def attention(q, segs, *, B, Lq, heads, head_dim, out=None, q_bs=None, q_ls=None, accumulate=False, scale=None, lse=None,
              new_softmax=0):
    if new_softmax:      # groups of segments, each its own softmax, outputs summed
        groups, cur = [], []
        for i, sg in enumerate(segs):
            if (new_softmax >> i) & 1 and cur:
                groups.append(cur)
                cur = []
            cur.append(sg)
        groups.append(cur)
        res = None
        for gi, gsegs in enumerate(groups):
            res = attention(q, gsegs, B=B, Lq=Lq, heads=heads, head_dim=head_dim, out=res if gi else out, q_bs=q_bs, q_ls=q_ls,
                            accumulate=accumulate if gi == 0 else True, scale=scale)
        return res
    C = heads * head_dim
    if q_ls is None:
        q_ls = C
    if q_bs is None:
        q_bs = Lq * q_ls
    qf = torch.as_strided(q, (B, Lq, C), (q_bs, q_ls, 1), q.storage_offset()).float().view(B, Lq, heads, head_dim)
    ks, vs = [], []
    for s in segs:
        if s.len <= 0:
            continue
        k = torch.as_strided(s.k, (B, s.len, C), (s.k_bs, s.k_ls, 1), s.k.storage_offset()).float()
        vt = torch.as_strided(s.vt, (B, C, s.len), (s.vt_bs, s.vt_ls, 1), s.vt.storage_offset()).float()
        ks.append(k)
        vs.append(vt.transpose(1, 2))
    k = torch.cat(ks, 1).view(B, -1, heads, head_dim)
    v = torch.cat(vs, 1).view(B, -1, heads, head_dim)
    sc = scale if scale is not None else 1.0 / math.sqrt(head_dim)
    s_ = torch.einsum("bqhd,bkhd->bhqk", qf, k) * sc
    if lse is not None:
        lse.copy_(torch.logsumexp(s_, -1) * 1.4426950408889634)
    o = torch.einsum("bhqk,bkhd->bqhd", torch.softmax(s_, -1), v).reshape(B, Lq, C).to(q.dtype)
    if out is None:
        return o
    if accumulate:
        out.copy_((out.float() + o.float()).to(out.dtype))
    else:
        out.copy_(o)
    return out


def attn_merge_(o_a, lse_a, o_b, lse_b, *, B, L, heads, head_dim):
    la, lb = lse_a.view(B, heads, L), lse_b.view(B, heads, L)
    lt = torch.logaddexp(la * math.log(2.0), lb * math.log(2.0)) / math.log(2.0)
    wa = torch.where(torch.isinf(lt), torch.zeros_like(lt), torch.exp2(la - lt)).permute(0, 2, 1).unsqueeze(-1)
    wb = torch.where(torch.isinf(lt), torch.zeros_like(lt), torch.exp2(lb - lt)).permute(0, 2, 1).unsqueeze(-1)
    a = o_a.view(B, L, heads, head_dim).float()
    b = o_b.view(B, L, heads, head_dim).float()
    o_a.view(B, L, heads, head_dim).copy_((wa * a + wb * b).to(o_a.dtype))
    lse_a.view(B, heads, L).copy_(lt)
    return o_a


def patchify(src0, src1, patch, out_dtype):
    x = src0 if src1 is None else torch.cat([src0, src1], 1)
    B, C, Fr, H, W = x.shape
    pt, ph, pw = patch
    u = x.reshape(B, C, Fr // pt, pt, H // ph, ph, W // pw, pw).permute(0, 2, 4, 6, 1, 3, 5, 7)
    return u.reshape(B, -1, C * pt * ph * pw).to(out_dtype)


def unpatchify(tok, row0, grid, patch, c, out_dtype):
    f, h, w = grid
    pt, ph, pw = patch
    B = tok.shape[0]
    u = tok[:, row0:row0 + f * h * w].reshape(B, f, h, w, pt, ph, pw, c).permute(0, 7, 1, 4, 2, 5, 3, 6)
    return u.reshape(B, c, f * pt, h * ph, w * pw).to(out_dtype)


def cfg_euler_(x, v, guidance, dsigma, round_dtype=torch.float32):
    vu, vc = v.float().reshape(2, -1)
    npred = (vu + guidance * (vc - vu)).to(v.dtype).float()
    x.copy_((x.reshape(-1) + dsigma * npred).to(round_dtype).float().view(x.shape))
    return x


def unary(x, out_dtype, act=0, out=None):
    y = x.float()
    if act == 1:
        y = F.silu(y)
    elif act == 2:
        y = F.gelu(y, approximate="tanh")
    elif act == 3:
        y = F.gelu(y)
    y = y.to(out_dtype)
    if out is not None:
        out.copy_(y)
        return out
    return y


def add_bcast(a, bias):
    n = bias.numel()      # like the kernel: `bias` broadcast over every leading group of bias.numel() elements
    return (a.reshape(-1, n) + bias.reshape(1, n)).reshape(a.shape)


# ------------------------------------------------------------------ VAE ops (channels-last)
NAMES += ["conv_cl", "rmsnorm_silu_cl", "groupnorm_cl", "softmax_rows", "ncthw_to_cl", "cl_to_ncthw"]


def conv_cl(x, w, bias, *, Tin, Hin, Win, Cin, k, stride=(1, 1, 1), pad=(0, 0, 0), out_thw, x_pixel_stride=None,
            resid=None, out=None, ups=False, tsplit=False, w_tiled=None):
    kt, kh, kw = k
    Cout = w.shape[0]
    xs = x_pixel_stride or Cin * (2 if tsplit else 1)
    flat = x.reshape(-1)
    cw = Cin * (2 if tsplit else 1)
    xv = torch.as_strided(flat, (Tin, Hin, Win, cw), (Hin * Win * xs, Win * xs, xs, 1), flat.storage_offset()).float()
    if tsplit:
        xv = torch.stack([xv[..., :Cin], xv[..., Cin:]], dim=1).reshape(2 * Tin, Hin, Win, Cin)
    if ups:
        xv = xv.repeat_interleave(2, dim=1).repeat_interleave(2, dim=2)
    To, Ho, Wo = out_thw
    Tl, Hl, Wl = xv.shape[:3]
    need_t = (To - 1) * stride[0] + kt - Tl - pad[0]
    need_h = (Ho - 1) * stride[1] + kh - Hl - pad[1]
    need_w = (Wo - 1) * stride[2] + kw - Wl - pad[2]
    xin = xv.permute(3, 0, 1, 2).unsqueeze(0)
    xin = F.pad(xin, (pad[2], max(need_w, 0), pad[1], max(need_h, 0), pad[0], max(need_t, 0)))
    wt = w.float().view(Cout, kt, kh, kw, Cin).permute(0, 4, 1, 2, 3)
    y = F.conv3d(xin, wt, None if bias is None else bias.float(), stride=stride)[0]
    y = y[:, :To, :Ho, :Wo].permute(1, 2, 3, 0).reshape(To * Ho * Wo, Cout).contiguous()
    if resid is not None:
        y = y.to(x.dtype).float() + resid.reshape(To * Ho * Wo, Cout).float()
    y = y.to(x.dtype)
    if out is None:
        return y
    out.copy_(y.view(out.shape))
    return out


def rmsnorm_silu_cl(x, gamma, *, silu=True, out=None):
    xf = x.float()
    C = x.shape[-1]
    y = xf / xf.norm(dim=-1, keepdim=True).clamp_min(1e-12) * math.sqrt(C) * gamma
    if silu:
        y = F.silu(y.to(x.dtype).float())
    y = y.to(x.dtype)
    if out is None:
        return y
    out.copy_(y.view(out.shape))
    return out


def _planar_to_cl(x):
    """Planar16 view [C/16, frames, hw, 16] -> channels-last [frames*hw, C]."""
    c16, fr, hw, _ = x.t.shape
    return x.t.permute(1, 2, 0, 3).reshape(fr * hw, c16 * 16)


def gnstats_blocks(Hin, Win):
    return 1


def conv_cl_planar(x, w, bias, *, Tin, Hin, Win, kt, resid=None, out=None, norm=None, keep_raw=True, gn_stats=None, w_tiled=None):
    if gn_stats is not None:      # one block per frame: (sum, sum of squares) of every group of 4 channels
        y = conv_cl_planar(x, w, bias, Tin=Tin, Hin=Hin, Win=Win, kt=kt, resid=resid, out=out)
        v = y.float().view(Tin - kt + 1, Hin * Win, -1, 4)
        gn_stats.view(Tin - kt + 1, -1, 2)[..., 0] = v.sum(dim=(1, 3))
        gn_stats.view(Tin - kt + 1, -1, 2)[..., 1] = (v * v).sum(dim=(1, 3))
        return y
    y = conv_cl(_planar_to_cl(x).contiguous(), w, bias, Tin=Tin, Hin=Hin, Win=Win, Cin=x.channels, k=(kt, 3, 3), pad=(0, 1, 1),
                out_thw=(Tin - kt + 1, Hin, Win), resid=resid, out=out if (norm is None or keep_raw) else None)
    if norm is None:
        return y
    gamma, dst, silu = norm
    rmsnorm_silu_cl_planar(y, gamma, dst, silu=silu)
    return y if keep_raw else None


def rmsnorm_silu_cl_planar(x, gamma, out, *, silu=True):
    y = rmsnorm_silu_cl(x, gamma, silu=silu)
    c16, fr, hw, _ = out.t.shape
    out.t.copy_(y.view(fr, hw, c16, 16).permute(2, 0, 1, 3))
    return out


def groupnorm_cl_planar(x, weight, bias, *, F, HW, groups=32, eps=1e-6, silu=True, frames_per_group, stats=None):
    import more4d_amd.ops as real
    if stats is not None:         # the producer's sums must be the statistics of x
        C_ = x.shape[-1]
        tot = stats.sum(dim=1)
        v = x.float().view(F, HW, groups, C_ // groups)
        assert torch.allclose(tot[..., 0], v.sum(dim=(1, 3)), rtol=1e-4, atol=1e-3) and \
            torch.allclose(tot[..., 1], (v * v).sum(dim=(1, 3)), rtol=1e-4, atol=1e-3)
    y = groupnorm_cl(x, weight, bias, F=F, HW=HW, groups=groups, eps=eps, silu=silu).view(F, HW, -1)
    C = y.shape[-1]
    out = []
    for f0 in range(0, F, frames_per_group):
        n = min(frames_per_group, F - f0)
        out.append(real.Planar16(y[f0:f0 + n].reshape(n, HW, C // 16, 16).permute(2, 0, 1, 3).contiguous()))
    return out


NAMES += ["conv_cl_planar", "rmsnorm_silu_cl_planar", "groupnorm_cl_planar", "gnstats_blocks"]


def groupnorm_cl(x, weight, bias, *, F, HW, groups=32, eps=1e-6, silu=True, out=None):
    C = x.shape[-1]
    y = torch.nn.functional.group_norm(x.float().view(F, HW, C).permute(0, 2, 1), groups, weight, bias, eps)
    if silu:
        y = y.to(x.dtype).float()
        y = y * torch.sigmoid(y)
    y = y.permute(0, 2, 1).reshape(x.shape).to(x.dtype).contiguous()
    if out is None:
        return y
    out.copy_(y)
    return out


def softmax_rows(x, out_dtype, *, C, Cpad, scale):
    R = x.shape[0]
    out = torch.zeros((R, Cpad), dtype=out_dtype)
    out[:, :C] = torch.softmax(x[:, :C].float() * scale, dim=-1).to(out_dtype)
    return out


def ncthw_to_cl(src, out_dtype, *, Cp=None, scale=1.0, shift=0.0, ch_scale=None, ch_shift=None, out=None):
    C, T, H, W = src.shape
    Cp = Cp or C
    v = src.float() * scale + shift
    if ch_scale is not None:
        v = v * ch_scale.view(C, 1, 1, 1) + ch_shift.view(C, 1, 1, 1)
    res = torch.zeros((T, H, W, Cp), dtype=out_dtype)
    res[..., :C] = v.permute(1, 2, 3, 0).to(out_dtype)
    if out is None:
        return res
    out.copy_(res.view(out.shape))
    return out


def cl_to_ncthw(src, out_dtype, *, C, T, H, W, pixel_stride, scale=1.0, shift=0.0, ch_scale=None, ch_shift=None, act=0,
                aux=None):
    flat = src.reshape(-1)
    v = torch.as_strided(flat, (T, H, W, C), (H * W * pixel_stride, W * pixel_stride, pixel_stride, 1),
                         flat.storage_offset()).float()
    v = v.permute(3, 0, 1, 2) * scale + shift
    if ch_scale is not None:
        v = v * ch_scale.view(C, 1, 1, 1) + ch_shift.view(C, 1, 1, 1)
    if act == 1:
        v = v.clamp(-1, 1)
    elif act == 2:
        v = torch.sigmoid(v + aux.float().view(C, T, H, W))
    return v.to(out_dtype).contiguous()


NAMES += ["axpby", "rel_l1", "bilinear_cl"]


def axpby(x, y, a=1.0, b=1.0, out=None):
    r = a * x + b * y
    if out is None:
        return r
    out.copy_(r)
    return out


def rel_l1(prev, cur):
    return float((cur.float() - prev.float()).abs().mean() / prev.float().abs().mean())


def bilinear_cl(x, out_hw):
    y = F.interpolate(x.float().permute(0, 3, 1, 2), size=tuple(out_hw), mode="bilinear", align_corners=False)
    return y.permute(0, 2, 3, 1).contiguous().to(x.dtype)


# ------------------------------------------------------------------ training-step ops
NAMES += ["transpose", "colsum", "scale_cast", "resid_gate", "add", "act_bwd_", "ln_modulate_bwd", "guidance_bwd_", "rmsnorm_rope_bwd_",
          "attention_bwd", "sumsq", "adamw_", "lincomb"]


def transpose(x, out=None):
    if out is None:
        return x.t().contiguous()
    out.copy_(x.t())
    return out


def colsum(a, b=None, *, rows_per_group=None, out=None):
    R, C = a.shape
    rpg = rows_per_group or R
    v = a.float() if b is None else a.float() * b.float()
    G = (R + rpg - 1) // rpg
    res = torch.stack([v[g * rpg:(g + 1) * rpg].sum(0) for g in range(G)])
    if out is None:
        return res
    out += res.view(out.shape)
    return out


def scale_cast(x, out_dtype, *, gate=None, gate_stride=0, rows_per_sample=0, out=None):
    C = x.shape[-1]
    R = x.numel() // C
    y = x.reshape(R, C).float()
    if gate is not None:
        rps = rows_per_sample or R
        y = y * _strided_rows(gate, R // rps, gate_stride, C).repeat_interleave(rps, dim=0)
    y = y.to(out_dtype).view(x.shape)
    if out is not None:
        out.copy_(y)
        return out
    return y


def resid_gate(x, y, *, gate=None, gate_stride=0, rows_per_sample=0, out=None):
    C = x.shape[-1]
    R = x.numel() // C
    v = y.reshape(R, C).float()
    if gate is not None:
        rps = rows_per_sample or R
        v = v * _strided_rows(gate, R // rps, gate_stride, C).repeat_interleave(rps, dim=0)
    res = x.reshape(R, C) + v
    if out is None:
        return res.view(x.shape)
    out.copy_(res.view(out.shape))
    return out


def add(a, b, out=None):
    res = (a.float() + b.float()).to(a.dtype)
    if out is None:
        return res
    out.copy_(res)
    return out


@torch.enable_grad()
def act_bwd_(dy, pre, act):
    if act == 5:        # sigmoid, given its OUTPUT
        s = pre.float()
        dy.copy_((dy.float() * s * (1 - s)).to(dy.dtype))
        return dy
    if act == 6:        # clamp(-1, 1), given the un-clamped value
        dy.copy_((dy.float() * ((pre.float() >= -1) & (pre.float() <= 1)).float()).to(dy.dtype))
        return dy
    x = pre.float().detach().requires_grad_(True)
    y = {1: F.silu, 2: lambda t: F.gelu(t, approximate="tanh"), 3: F.gelu, 4: torch.sigmoid}[act](x)
    (g,) = torch.autograd.grad(y, x, dy.float())
    dy.copy_(g.to(dy.dtype))
    return dy


@torch.enable_grad()
def ln_modulate_bwd(x, dy, dx, *, B, rows_per_sample, scale=None, mod_stride=0, ln_w=None, eps=1e-6, dshift=None,
                    dscale=None, red_stride=0):
    C = x.shape[-1]
    rps = rows_per_sample
    xf = x.reshape(B, rps, C).float().detach().requires_grad_(True)
    mu = xf.mean(-1, keepdim=True)
    xh = (xf - mu) * torch.rsqrt((xf - mu).pow(2).mean(-1, keepdim=True) + eps)
    if scale is not None:
        m = 1 + _strided_rows(scale, B, mod_stride, C).view(B, 1, C)
    elif ln_w is not None:
        m = ln_w.view(1, 1, C)
    else:
        m = 1.0
    g = dy.reshape(B, rps, C).float()
    (gx,) = torch.autograd.grad(xh * m, xf, g)
    dx += gx.view(dx.shape)
    if dshift is not None:
        G = B if red_stride else 1
        s1 = g.sum(1) if red_stride else g.sum((0, 1)).view(1, C)
        s2 = (g * xh.detach()).sum(1) if red_stride else (g * xh.detach()).sum((0, 1)).view(1, C)
        torch.as_strided(dshift, (G, C), (max(red_stride, 1), 1), dshift.storage_offset()).add_(s1)
        torch.as_strided(dscale, (G, C), (max(red_stride, 1), 1), dscale.storage_offset()).add_(s2)
    return dx


def guidance_bwd_(x, dz, *, B, rows_per_sample, shift, scale, mod_stride, g_ss, g_gate, g_period, g_len, eps=1e-6, mod_rows=0, g_rows=0):
    C = x.shape[-1]
    rps = rows_per_sample
    xf = x.reshape(B, rps, C).float()
    mu = xf.mean(-1, keepdim=True)
    xh = (xf - mu) * torch.rsqrt((xf - mu).pow(2).mean(-1, keepdim=True) + eps)
    mr = mod_rows or rps      # rows that share one (shift, scale) vector
    nm = B * rps // mr
    u = (xh.reshape(nm, mr, C) * (1 + _strided_rows(scale, nm, mod_stride, C)[:, None]) + _strided_rows(shift, nm, mod_stride, C)[:, None]).reshape(B, rps, C)
    g = dz.reshape(B, rps, C).float()
    n = min(g_len, rps)
    idx = torch.arange(n) % g_period
    ab = torch.zeros(B, g_period, 2 * C)
    ab[:, :, :C].index_add_(1, idx, g[:, :n] * u[:, :n])
    ab[:, :, C:].index_add_(1, idx, g[:, :n])
    du = g.clone()
    du[:, :n] = g[:, :n] * (1 + g_ss[:, idx, :C] * g_gate)
    dz.copy_(du.reshape(dz.shape).to(dz.dtype))
    return ab


@torch.enable_grad()
def rmsnorm_rope_bwd_(dy0, x0, w0, dw0, dy1=None, x1=None, w1=None, dw1=None, *, head_dim, eps=1e-6, cos=None, sin=None,
                      rows_per_sample=0, rope_len=0, pos_offset=0):
    for dy, x, w, dw in ((dy0, x0, w0, dw0), (dy1, x1, w1, dw1)):
        if dy is None:
            continue
        C = x.shape[-1]
        rows = x.numel() // C
        rps = rows_per_sample or rows
        xf = x.reshape(rows, C).float().detach().requires_grad_(True)
        wf = w.detach().clone().requires_grad_(True)
        y = xf * torch.rsqrt(xf.pow(2).mean(-1, keepdim=True) + eps) * wf
        if cos is not None:
            y5 = y.view(rows // rps, rps, C // head_dim, head_dim // 2, 2)
            n = min(rope_len, rps)
            c = cos[pos_offset:pos_offset + n].view(1, n, 1, -1)
            s = sin[pos_offset:pos_offset + n].view(1, n, 1, -1)
            a, b = y5[:, :n, :, :, 0], y5[:, :n, :, :, 1]
            rot = torch.stack([a * c - b * s, a * s + b * c], -1)
            y = torch.cat([rot, y5[:, n:]], 1).reshape(rows, C)
        gx, gw = torch.autograd.grad(y, (xf, wf), dy.reshape(rows, C).float())
        dy.copy_(gx.to(dy.dtype).view(dy.shape))
        dw += gw


@torch.enable_grad()
def attention_bwd(q, k, v, o, d_o, lse, *, B, Lq, Lk, Lk_rows, heads, head_dim, dq, dk, dv, scale=None,
                  accumulate_dq=False, accumulate_dkv=False):
    C = heads * head_dim
    qf = q.reshape(B, Lq, heads, head_dim).float().detach().requires_grad_(True)
    kf = k.reshape(B, Lk_rows, heads, head_dim).float().detach().requires_grad_(True)
    vf = v.reshape(B, Lk_rows, heads, head_dim).float().detach().requires_grad_(True)
    sc = scale if scale is not None else 1.0 / math.sqrt(head_dim)
    s_ = torch.einsum("bqhd,bkhd->bhqk", qf, kf[:, :Lk]) * sc
    out = torch.einsum("bhqk,bkhd->bqhd", torch.softmax(s_, -1), vf[:, :Lk]).reshape(B * Lq, C)
    gq, gk, gv = torch.autograd.grad(out, (qf, kf, vf), d_o.reshape(B * Lq, C).float())
    for dst, g, acc in ((dq, gq, accumulate_dq), (dk, gk, accumulate_dkv), (dv, gv, accumulate_dkv)):
        g = g.reshape(dst.shape)
        dst.copy_((dst.float() + g).to(dst.dtype) if acc else g.to(dst.dtype))


def lincomb(terms, out=None):
    res = sum(float(a) * x for a, x in terms)
    if out is None:
        return res
    out.copy_(res)
    return out


def sumsq(x, out):
    out += x.float().pow(2).sum()
    return out


def adamw_(p, g, m, v, *, lr, beta1, beta2, eps, weight_decay, step, grad_scale=None):
    gf = g.float() * (float(grad_scale) if grad_scale is not None else 1.0)
    pf, mf, vf = p.float(), m.float(), v.float()
    pf = pf * (1 - lr * weight_decay)
    mf = beta1 * mf + (1 - beta1) * gf
    vf = beta2 * vf + (1 - beta2) * gf * gf
    denom = vf.sqrt() / math.sqrt(1 - beta2 ** step) + eps
    pf = pf - lr / (1 - beta1 ** step) * mf / denom
    p.copy_(pf.to(p.dtype))
    m.copy_(mf.to(m.dtype))
    v.copy_(vf.to(v.dtype))


# ------------------------------------------------------------------ VAE training ops (vae_autograd.py)
NAMES += ["pad_transpose", "gemm_bt_batched", "wgrad_reduce", "rmsnorm_silu_cl_bwd", "softmax_rows_bwd", "upsample2x_cl",
          "upsample2x_cl_bwd", "groupnorm_cl_bwd"]


def pad_transpose(src, pixel_stride, C, T, H, W, Hp, Wp, pad_top, pad_left, nshift, cols, rows=None):
    flat = src.reshape(-1)
    v = torch.as_strided(flat, (T, H, W, C), (H * W * pixel_stride, W * pixel_stride, pixel_stride, 1), flat.storage_offset())
    padded = torch.zeros((T, Hp, Wp, C), dtype=src.dtype)
    padded[:, pad_top:pad_top + H, pad_left:pad_left + W] = v
    lin = torch.zeros((cols + nshift, C), dtype=src.dtype)
    n = min(cols + nshift, T * Hp * Wp)
    lin[:n] = padded.reshape(-1, C)[:n]
    out = torch.zeros((rows or nshift * C, cols), dtype=src.dtype)
    for s in range(nshift):
        out[s * C:(s + 1) * C] = lin[s:s + cols].t()
    return out


def gemm_bt_batched(a, w, *, M, N, K, nb1, a_bs1, w_bs1, nb2=1, a_bs2=0, w_bs2=0):
    out = torch.zeros((nb2, nb1, M, N), dtype=torch.float32)
    lda, ldw = a.stride(0), w.stride(0)
    af, wf = a.reshape(-1) if a.is_contiguous() else None, None
    for i2 in range(nb2):
        for i1 in range(nb1):
            ao = i1 * a_bs1 + i2 * a_bs2
            wo = i1 * w_bs1 + i2 * w_bs2
            A = torch.as_strided(a, (M, K), (lda, 1), a.storage_offset() + ao).float()
            Wm = torch.as_strided(w, (N, K), (ldw, 1), w.storage_offset() + wo).float()
            out[i2, i1] = A @ Wm.t()
    del af, wf
    return out


def wgrad_reduce(part, dw, dt, M):
    kh, S, Mp, N = part.shape
    cop, kt, kh2, kw, cip = dw.shape
    assert kh == kh2 and N == kw * cip
    dw[:, dt] += part.sum(dim=1)[:, :cop].view(kh, cop, kw, cip).permute(1, 0, 2, 3)
    return dw


@torch.enable_grad()
def rmsnorm_silu_cl_bwd(x, gamma, dy, *, silu=True):
    xf = x.float().detach().requires_grad_(True)
    g = gamma.float().detach().requires_grad_(True)
    C = x.shape[-1]
    u = xf / xf.norm(dim=-1, keepdim=True).clamp_min(1e-12) * math.sqrt(C) * g
    y = F.silu(u) if silu else u
    y.backward(dy.float().reshape(y.shape))
    return xf.grad.to(x.dtype), g.grad


def softmax_rows_bwd(p, dp, *, scale, C):
    pf = p.float()
    d = dp.float()
    pf = pf.clone()
    pf[:, C:] = 0
    ds = scale * pf * (d - (pf * d).sum(dim=-1, keepdim=True))
    return ds.to(p.dtype)


def upsample2x_cl(x, t, h, w, c, *, tsplit=False):
    cw = c * (2 if tsplit else 1)
    v = x.reshape(t, h, w, cw)
    if tsplit:
        v = torch.stack([v[..., :c], v[..., c:]], dim=1).reshape(2 * t, h, w, c)
    v = v.repeat_interleave(2, dim=1).repeat_interleave(2, dim=2)
    return v.reshape(-1, c).contiguous()


def upsample2x_cl_bwd(du, t, h, w, c, *, tsplit=False):
    tt = t * (2 if tsplit else 1)
    v = du.float().reshape(tt, h, 2, w, 2, c).sum(dim=(2, 4))
    if tsplit:
        v = v.reshape(t, 2, h, w, c).permute(0, 2, 3, 1, 4).reshape(t, h, w, 2 * c)
    return v.reshape(t * h * w, -1).to(du.dtype).contiguous()


@torch.enable_grad()
def groupnorm_cl_bwd(x, weight, bias, dy, *, F, HW, groups=32, eps=1e-6, silu=True):
    C = x.shape[-1]
    xf = x.float().detach().view(F, HW, C).permute(0, 2, 1).requires_grad_(True)
    wt = weight.float().detach().requires_grad_(True)
    bs = bias.float().detach().requires_grad_(True)
    y = torch.nn.functional.group_norm(xf, groups, wt, bs, eps)
    if silu:
        y = y * torch.sigmoid(y)
    y.backward(dy.float().view(F, HW, C).permute(0, 2, 1))
    return xf.grad.permute(0, 2, 1).reshape(x.shape).to(x.dtype).contiguous(), wt.grad, bs.grad
