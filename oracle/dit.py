"""Oracle: Wan2.1-DiT forward (WanTransformer4DModel) as a functional fp32 restatement.

TEST INFRASTRUCTURE — see oracle/__init__.py.  Weights come in as a flat dict using the
reference's state-dict key names; nothing here is an nn.Module.
All citations are /root/reference/MoRe4D/models/wan_transformer4d.py unless noted.
"""
import math
from dataclasses import dataclass, field
from typing import Optional, Tuple

import torch
import torch.nn.functional as F


@dataclass
class DiTConfig:
    # ctor kwargs of WanTransformer4DModel (:793-821)
    model_type: str = "i2v"
    patch_size: Tuple[int, int, int] = (1, 2, 2)
    text_len: int = 512
    in_dim: int = 64
    dim: int = 5120
    ffn_dim: int = 13824
    freq_dim: int = 256
    text_dim: int = 4096
    out_dim: int = 16
    num_heads: int = 40
    num_layers: int = 40
    qk_norm: bool = True
    cross_attn_norm: bool = True
    eps: float = 1e-6
    add_ref_conv: bool = True
    in_dim_ref_conv: int = 16
    cross_attn_type: Optional[str] = None
    use_spatial_guidance: bool = False
    use_cls_token: bool = False

    @property
    def head_dim(self):
        return self.dim // self.num_heads

    @property
    def xattn(self):
        if self.cross_attn_type is not None:
            return self.cross_attn_type
        return "t2v_cross_attn" if self.model_type == "t2v" else "i2v_cross_attn"  # :908-909


# --------------------------------------------------------------------------- embeddings

def sinusoidal_embedding_1d(dim, position):
    """cat(cos, sin)(t * 10000^(-i/half)) in float64 (:239-249)."""
    half = dim // 2
    pos = position.to(torch.float64)
    inv = torch.pow(torch.tensor(10000.0, dtype=torch.float64),
                    -torch.arange(half, dtype=torch.float64) / half)
    ang = pos[:, None] * inv[None, :]
    return torch.cat([ang.cos(), ang.sin()], dim=1)


def rope_angles(head_dim, max_len=1024, theta=10000.0):
    """Angle table [max_len, head_dim/2] float64 — the three axis tables side by side.

    dims per axis (:928-935): d-4*(d//6), 2*(d//6), 2*(d//6); angle_j = pos * theta^(-2j/dim_axis)
    (:252-260).  Returned as raw angles; cos/sin taken by the caller.
    """
    d = head_dim
    dims = [d - 4 * (d // 6), 2 * (d // 6), 2 * (d // 6)]
    pos = torch.arange(max_len, dtype=torch.float64)
    cols = []
    for da in dims:
        inv = 1.0 / torch.pow(torch.tensor(theta, dtype=torch.float64),
                              torch.arange(0, da, 2, dtype=torch.float64) / da)
        cols.append(pos[:, None] * inv[None, :])
    return torch.cat(cols, dim=1), [da // 2 for da in dims]


def rope_token_table(head_dim, grid):
    """cos, sin float64 [f*h*w, head_dim/2] for a (f,h,w) grid, tokens f-major (:356-361)."""
    f, h, w = grid
    ang, split = rope_angles(head_dim)
    af, ah, aw = torch.split(ang, split, dim=1)
    a = torch.cat([
        af[:f].view(f, 1, 1, -1).expand(f, h, w, -1),
        ah[:h].view(1, h, 1, -1).expand(f, h, w, -1),
        aw[:w].view(1, 1, w, -1).expand(f, h, w, -1),
    ], dim=-1).reshape(f * h * w, -1)
    return a.cos(), a.sin()


def rope_apply(x, grid, out_dtype=None):
    """x [B, L, n, d]; rotate adjacent pairs (2i, 2i+1) of the first f*h*w tokens (:340-369).

    The reference multiplies in complex128 and casts back to x.dtype; rows past f*h*w
    (sequence padding) pass through unrotated (:365).
    """
    B, L, n, d = x.shape
    cos, sin = rope_token_table(d, grid)
    S = cos.shape[0]
    xr = x[:, :S].to(torch.float64).reshape(B, S, n, d // 2, 2)
    a, b = xr[..., 0], xr[..., 1]
    c = cos.view(1, S, 1, -1)
    s = sin.view(1, S, 1, -1)
    rot = torch.stack([a * c - b * s, a * s + b * c], dim=-1).reshape(B, S, n, d)
    out = torch.cat([rot.to(x.dtype), x[:, S:]], dim=1)
    return out if out_dtype is None else out.to(out_dtype)


# --------------------------------------------------------------------------- norms / attention

def rms_norm(x, weight, eps):
    """WanRMSNorm over the last dim (:386-394): x*rsqrt(mean(x^2)+eps) then *weight."""
    inv = torch.rsqrt(x.float().pow(2).mean(dim=-1, keepdim=True) + eps)
    return (x.float() * inv).to(x.dtype) * weight


def layer_norm(x, eps, weight=None, bias=None):
    """WanLayerNorm (:397-407): biased variance, fp32."""
    xf = x.float()
    mu = xf.mean(dim=-1, keepdim=True)
    var = (xf - mu).pow(2).mean(dim=-1, keepdim=True)
    y = (xf - mu) * torch.rsqrt(var + eps)
    if weight is not None:
        y = y * weight + bias
    return y


def sdpa(q, k, v, k_len=None):
    """Non-causal softmax(q k^T / sqrt(d)) v; q [B,Lq,n,d], k/v [B,Lk,n,d] (:221-235).

    k_len: optional count of valid keys (flash path's k_lens semantics, :114-122); the
    reference's SDPA branch ignores it (:222-226) so parity runs pass None.
    """
    d = q.shape[-1]
    qh, kh, vh = (u.permute(0, 2, 1, 3).float() for u in (q, k, v))
    s = torch.matmul(qh, kh.transpose(-1, -2)) / math.sqrt(d)
    if k_len is not None:
        s[..., k_len:] = float("-inf")
    p = torch.softmax(s, dim=-1)
    return torch.matmul(p, vh).permute(0, 2, 1, 3).contiguous()


def linear(sd, name, x):
    return F.linear(x, sd[name + ".weight"], sd.get(name + ".bias"))


def self_attention(sd, p, cfg, x, grid, k_len=None):
    """WanSelfAttention.forward (:434-466)."""
    B, L, _ = x.shape
    n, d = cfg.num_heads, cfg.head_dim
    q = linear(sd, p + ".q", x)
    k = linear(sd, p + ".k", x)
    v = linear(sd, p + ".v", x)
    if cfg.qk_norm:
        q = rms_norm(q, sd[p + ".norm_q.weight"], cfg.eps)
        k = rms_norm(k, sd[p + ".norm_k.weight"], cfg.eps)
    q = rope_apply(q.view(B, L, n, d), grid)
    k = rope_apply(k.view(B, L, n, d), grid)
    o = sdpa(q, k, v.view(B, L, n, d), k_len)
    return linear(sd, p + ".o", o.flatten(2))


def cross_attention(sd, p, cfg, x, context):
    """WanI2VCrossAttention (:515-554) / WanT2VCrossAttention (:471-497)."""
    B = x.shape[0]
    n, d = cfg.num_heads, cfg.head_dim
    q = linear(sd, p + ".q", x)
    if cfg.qk_norm:
        q = rms_norm(q, sd[p + ".norm_q.weight"], cfg.eps)
    q = q.view(B, -1, n, d)

    def kv(ctx, kname, vname, nname):
        k = linear(sd, p + kname, ctx)
        if cfg.qk_norm:
            k = rms_norm(k, sd[p + nname], cfg.eps)
        return k.view(B, -1, n, d), linear(sd, p + vname, ctx).view(B, -1, n, d)

    if cfg.xattn == "i2v_cross_attn":
        img, txt = context[:, :257], context[:, 257:]  # :522-523
        k, v = kv(txt, ".k", ".v", ".norm_k.weight")
        ki, vi = kv(img, ".k_img", ".v_img", ".norm_k_img.weight")
        o = sdpa(q, k, v).flatten(2) + sdpa(q, ki, vi).flatten(2)  # :552
    else:
        k, v = kv(context, ".k", ".v", ".norm_k.weight")
        o = sdpa(q, k, v).flatten(2)
    return linear(sd, p + ".o", o)


def spatial_guidance(sd, p, x, feats, cls, use_cls_token):
    """SpatialGuidanceModule.forward (:757-783)."""
    if feats is None:
        return x
    src = cls if (use_cls_token and cls is not None) else feats
    sp = F.linear(F.silu(src), sd[p + ".spatial_guide.1.weight"], sd[p + ".spatial_guide.1.bias"])
    scale, shift = sp.chunk(2, dim=-1)
    if use_cls_token and cls is not None:
        scale = scale.repeat(1, feats.size(1), 1)
        shift = shift.repeat(1, feats.size(1), 1)
    if scale.size(1) < x.size(1):  # zero-pad to L (:772-776)
        pad = x.size(1) - scale.size(1)
        z = scale.new_zeros(scale.size(0), pad, scale.size(2))
        scale = torch.cat([scale, z], 1)
        shift = torch.cat([shift, z], 1)
    g = sd[p + ".gate"].view(1, 1, -1)
    return x * (1 + scale * g) + shift * g


def adapt_guidance_features(sd, patch, hw, latent_T):
    """OmniMAE patch features [B,196,768] -> per-token guidance features [B, latent_T*h*w, 768] (:1147-1154):
    view as a 14x14 map, feature_adapter = Conv3x3 - SiLU - Conv3x3 (:889-893), bilinear resize (align_corners=False) to
    the token grid, repeat over the latent frames, flatten f-major."""
    B = patch.shape[0]
    m = patch.view(B, 14, 14, -1).permute(0, 3, 1, 2)
    m = F.conv2d(m, sd["feature_adapter.0.weight"], sd["feature_adapter.0.bias"], padding=1)
    m = F.conv2d(F.silu(m), sd["feature_adapter.2.weight"], sd["feature_adapter.2.bias"], padding=1)
    m = F.interpolate(m, size=tuple(hw), mode="bilinear", align_corners=False)
    m = m.unsqueeze(2).repeat(1, 1, latent_T, 1, 1)
    return m.flatten(2).transpose(1, 2)


def block_forward(sd, i, cfg, x, e0, grid, context, guidance=None, k_len=None):
    """WanAttentionBlock.forward (:633-688); e0 [B,6,dim] fp32."""
    p = f"blocks.{i}"
    feats, cls = guidance if guidance is not None else (None, None)
    e = (sd[p + ".modulation"] + e0).chunk(6, dim=1)  # :659
    t = layer_norm(x, cfg.eps) * (1 + e[1]) + e[0]  # :662
    if cfg.use_spatial_guidance and feats is not None:
        t = spatial_guidance(sd, p + ".spatial_guidance_self", t, feats, cls, cfg.use_cls_token)
    y = self_attention(sd, p + ".self_attn", cfg, t, grid, k_len)
    x = x + y * e[2]  # :669
    if cfg.cross_attn_norm:
        xn = layer_norm(x, cfg.eps, sd[p + ".norm3.weight"], sd[p + ".norm3.bias"])
    else:
        xn = x
    x = x + cross_attention(sd, p + ".cross_attn", cfg, xn, context)  # :674
    t = layer_norm(x, cfg.eps) * (1 + e[4]) + e[3]  # :677
    if cfg.use_spatial_guidance and feats is not None:
        t = spatial_guidance(sd, p + ".spatial_guidance_ffn", t, feats, cls, cfg.use_cls_token)
    h = F.gelu(linear(sd, p + ".ffn.0", t), approximate="tanh")  # :620-622
    y = linear(sd, p + ".ffn.2", h)
    return x + y * e[5]  # :684


def head_forward(sd, cfg, x, e):
    """Head.forward (:708-721); e [B, dim]."""
    m = (sd["head.modulation"] + e.unsqueeze(1)).chunk(2, dim=1)
    return linear(sd, "head.head", layer_norm(x, cfg.eps) * (1 + m[1]) + m[0])


def patchify_tokens(sd, name, x, patch):
    """Conv with kernel=stride=patch as a matmul over flattened patches (:898-899, :1073, :1082).

    x [B, C, F, H, W] -> tokens [B, F'*H'*W', dim]; weight [dim, C, pt, ph, pw].
    """
    w = sd[name + ".weight"]
    b = sd[name + ".bias"]
    B, C, Fr, H, W = x.shape
    pt, ph, pw = patch
    f, h, ww = Fr // pt, H // ph, W // pw
    u = x.view(B, C, f, pt, h, ph, ww, pw).permute(0, 2, 4, 6, 1, 3, 5, 7)
    u = u.reshape(B, f * h * ww, C * pt * ph * pw)
    return F.linear(u, w.reshape(w.shape[0], -1), b), (f, h, ww)


def unpatchify(x, grid, patch, c):
    """[B, L, prod(patch)*c] -> [B, c, F, H, W]; einsum 'fhwpqrc->cfphqwr' (:1343-1366)."""
    f, h, w = grid
    pt, ph, pw = patch
    B = x.shape[0]
    u = x[:, : f * h * w].reshape(B, f, h, w, pt, ph, pw, c)
    u = u.permute(0, 7, 1, 4, 2, 5, 3, 6)
    return u.reshape(B, c, f * pt, h * ph, w * pw)


def embed_context(sd, cfg, context, clip_fea):
    """text_embedding on zero-padded prompts + img_emb on CLIP tokens (:1174-1184)."""
    ctx = torch.stack([
        torch.cat([u, u.new_zeros(cfg.text_len - u.size(0), u.size(1))]) for u in context])
    ctx = linear(sd, "text_embedding.2",
                 F.gelu(linear(sd, "text_embedding.0", ctx), approximate="tanh"))
    if clip_fea is not None:
        c = F.layer_norm(clip_fea, (clip_fea.shape[-1],), sd["img_emb.proj.0.weight"],
                         sd["img_emb.proj.0.bias"], 1e-5)  # MLPProj (:724-736), nn.LayerNorm default eps
        c = F.gelu(linear(sd, "img_emb.proj.1", c))  # erf GELU
        c = linear(sd, "img_emb.proj.3", c)
        c = F.layer_norm(c, (c.shape[-1],), sd["img_emb.proj.4.weight"], sd["img_emb.proj.4.bias"], 1e-5)
        ctx = torch.cat([c, ctx], dim=1)
    return ctx


def time_embed(sd, cfg, t):
    """e [B, dim], e0 [B, 6, dim] in fp32 (:1160-1171)."""
    s = sinusoidal_embedding_1d(cfg.freq_dim, t).float()
    e = linear(sd, "time_embedding.2", F.silu(linear(sd, "time_embedding.0", s)))
    e0 = linear(sd, "time_projection.1", F.silu(e)).unflatten(1, (6, cfg.dim))
    return e, e0


def dit_forward(sd, cfg, x, t, context, seq_len, clip_fea=None, y=None, full_ref=None,
                guidance=None, k_len=None, return_tokens=False):
    """WanTransformer4DModel.forward (:1047-1340), single-process, no TeaCache / cfg-skip.

    x [B,16,F,H,W]; y [B,48,F,H,W] or None; t [B]; context: list of [Li, text_dim];
    clip_fea [B,257,1280]; full_ref [B,16,H,W]; guidance: (feats [B,Lg,768], cls [B,1,768]).
    """
    if y is not None:
        x = torch.cat([x, y], dim=1)  # :1069-1070
    tok, grid = patchify_tokens(sd, "patch_embedding", x, cfg.patch_size)
    n_ref = 0
    if cfg.add_ref_conv and full_ref is not None:  # :1086-1090
        r, _ = patchify_tokens(sd, "ref_conv", full_ref.unsqueeze(2), (1,) + tuple(cfg.patch_size[1:]))
        n_ref = r.shape[1]
        tok = torch.cat([r, tok], dim=1)
        grid = (grid[0] + 1, grid[1], grid[2])
        seq_len = seq_len + n_ref
    assert tok.shape[1] <= seq_len  # :1102
    tok = torch.cat([tok, tok.new_zeros(tok.shape[0], seq_len - tok.shape[1], tok.shape[2])], dim=1)
    e, e0 = time_embed(sd, cfg, t)
    ctx = embed_context(sd, cfg, context, clip_fea)
    h = tok
    for i in range(cfg.num_layers):
        h = block_forward(sd, i, cfg, h, e0, grid, ctx, guidance, k_len)
    out = head_forward(sd, cfg, h, e)
    if return_tokens:
        return out
    if n_ref:
        out = out[:, n_ref:]  # :1323-1326
        grid = (grid[0] - 1, grid[1], grid[2])
    return unpatchify(out, grid, cfg.patch_size, cfg.out_dim)
