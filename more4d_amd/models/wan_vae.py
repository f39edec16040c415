"""Wan2.1 causal 3-D VAE (the core of MoRe4D's Motion-Sensitive VAE) — MI355X-native host side.

Drop-in for the reference's `MoRe4D/models/wan_vae.py`: same class names and module tree, so the state-dict keys
(`model.encoder.downsamples.K.residual.{0,3}.gamma`, `...resample.1.weight`, `...time_conv.weight`, ...; SURVEY.md
Appendix A) and `encode(x).latent_dist.sample()/.mode()`, `decode(z).sample`, `.config.*`, `.latent_channels`,
`.spatial/temporal_compression_ratio` behave like the reference's.  The torch modules are parameter CONTAINERS; the
forward pass is `_Runner` below, which drives the HIP kernels (conv as implicit GEMM on channels-last activations,
fused RMS-norm+SiLU, softmax, layout boundaries) through `more4d_amd.ops`.

Streaming (reference: feat_cache lists, wan_vae.py:105-164, :206-224, :520-547, :678-703) is restated as one
2-frame TAIL per causal conv kept IN FRONT of the chunk inside that conv's staging buffer: producers (norm kernels,
layout kernels, previous convs) write straight into the staging buffer, the conv reads tail+chunk as a plain
"valid" convolution, then the tail is refreshed from the last two frames — no torch.cat / F.pad copies of whole
activations.  First-chunk special cases of the reference fall out of zero-initialised tails and of skipping the
temporal convs of the Resample blocks on chunk 0 exactly where the reference does.
"""
import math

import os

import torch
import torch.nn as nn

from .. import ops

# M4D_CONV_TILED=0: the LDS-halo conv kernels stage their weights from the plain [Cout, taps*Cin] order (A/B; default: from the tiled copies
# of ops.conv_pack_weights — one contiguous KiB per DMA request instead of 32 cache lines)
_TILED_WEIGHTS = os.environ.get("M4D_CONV_TILED", "1") != "0"

CACHE_T = 2


class CausalConv3d(nn.Conv3d):
    def __init__(self, *args, **kwargs):
        super().__init__(*args, **kwargs)
        self._padding = (self.padding[2], self.padding[2], self.padding[1], self.padding[1], 2 * self.padding[0], 0)
        self.padding = (0, 0, 0)


class RMS_norm(nn.Module):
    def __init__(self, dim, channel_first=True, images=True, bias=False):
        super().__init__()
        broadcastable_dims = (1, 1, 1) if not images else (1, 1)
        shape = (dim, *broadcastable_dims) if channel_first else (dim,)
        self.channel_first = channel_first
        self.scale = dim ** 0.5
        self.gamma = nn.Parameter(torch.ones(shape))
        self.bias = nn.Parameter(torch.zeros(shape)) if bias else 0.


class Upsample(nn.Upsample):
    pass


class Resample(nn.Module):
    def __init__(self, dim, mode):
        assert mode in ('none', 'upsample2d', 'upsample3d', 'downsample2d', 'downsample3d')
        super().__init__()
        self.dim, self.mode = dim, mode
        if mode == 'upsample2d':
            self.resample = nn.Sequential(Upsample(scale_factor=(2., 2.), mode='nearest-exact'),
                                          nn.Conv2d(dim, dim // 2, 3, padding=1))
        elif mode == 'upsample3d':
            self.resample = nn.Sequential(Upsample(scale_factor=(2., 2.), mode='nearest-exact'),
                                          nn.Conv2d(dim, dim // 2, 3, padding=1))
            self.time_conv = CausalConv3d(dim, dim * 2, (3, 1, 1), padding=(1, 0, 0))
        elif mode == 'downsample2d':
            self.resample = nn.Sequential(nn.ZeroPad2d((0, 1, 0, 1)), nn.Conv2d(dim, dim, 3, stride=(2, 2)))
        elif mode == 'downsample3d':
            self.resample = nn.Sequential(nn.ZeroPad2d((0, 1, 0, 1)), nn.Conv2d(dim, dim, 3, stride=(2, 2)))
            self.time_conv = CausalConv3d(dim, dim, (3, 1, 1), stride=(2, 1, 1), padding=(0, 0, 0))
        else:
            self.resample = nn.Identity()


class ResidualBlock(nn.Module):
    def __init__(self, in_dim, out_dim, dropout=0.0):
        super().__init__()
        self.in_dim, self.out_dim = in_dim, out_dim
        self.residual = nn.Sequential(
            RMS_norm(in_dim, images=False), nn.SiLU(), CausalConv3d(in_dim, out_dim, 3, padding=1),
            RMS_norm(out_dim, images=False), nn.SiLU(), nn.Dropout(dropout), CausalConv3d(out_dim, out_dim, 3, padding=1))
        self.shortcut = CausalConv3d(in_dim, out_dim, 1) if in_dim != out_dim else nn.Identity()


class AttentionBlock(nn.Module):
    def __init__(self, dim):
        super().__init__()
        self.dim = dim
        self.norm = RMS_norm(dim)
        self.to_qkv = nn.Conv2d(dim, dim * 3, 1)
        self.proj = nn.Conv2d(dim, dim, 1)
        nn.init.zeros_(self.proj.weight)


class Encoder3d(nn.Module):
    def __init__(self, dim=128, z_dim=4, dim_mult=[1, 2, 4, 4], num_res_blocks=2, attn_scales=[],
                 temperal_downsample=[True, True, False], dropout=0.0):
        super().__init__()
        self.dim, self.z_dim, self.dim_mult = dim, z_dim, dim_mult
        dims = [dim * u for u in [1] + dim_mult]
        scale = 1.0
        self.conv1 = CausalConv3d(3, dims[0], 3, padding=1)
        downsamples = []
        for i, (in_dim, out_dim) in enumerate(zip(dims[:-1], dims[1:])):
            for _ in range(num_res_blocks):
                downsamples.append(ResidualBlock(in_dim, out_dim, dropout))
                if scale in attn_scales:
                    downsamples.append(AttentionBlock(out_dim))
                in_dim = out_dim
            if i != len(dim_mult) - 1:
                downsamples.append(Resample(out_dim, mode='downsample3d' if temperal_downsample[i] else 'downsample2d'))
                scale /= 2.0
        self.downsamples = nn.Sequential(*downsamples)
        self.middle = nn.Sequential(ResidualBlock(out_dim, out_dim, dropout), AttentionBlock(out_dim),
                                    ResidualBlock(out_dim, out_dim, dropout))
        self.head = nn.Sequential(RMS_norm(out_dim, images=False), nn.SiLU(), CausalConv3d(out_dim, z_dim, 3, padding=1))


class Decoder3d(nn.Module):
    def __init__(self, dim=128, z_dim=4, dim_mult=[1, 2, 4, 4], num_res_blocks=2, attn_scales=[],
                 temperal_upsample=[False, True, True], dropout=0.0):
        super().__init__()
        self.dim, self.z_dim, self.dim_mult = dim, z_dim, dim_mult
        dims = [dim * u for u in [dim_mult[-1]] + dim_mult[::-1]]
        scale = 1.0 / 2 ** (len(dim_mult) - 2)
        self.conv1 = CausalConv3d(z_dim, dims[0], 3, padding=1)
        self.middle = nn.Sequential(ResidualBlock(dims[0], dims[0], dropout), AttentionBlock(dims[0]),
                                    ResidualBlock(dims[0], dims[0], dropout))
        upsamples = []
        for i, (in_dim, out_dim) in enumerate(zip(dims[:-1], dims[1:])):
            if i == 1 or i == 2 or i == 3:
                in_dim = in_dim // 2
            for _ in range(num_res_blocks + 1):
                upsamples.append(ResidualBlock(in_dim, out_dim, dropout))
                if scale in attn_scales:
                    upsamples.append(AttentionBlock(out_dim))
                in_dim = out_dim
            if i != len(dim_mult) - 1:
                upsamples.append(Resample(out_dim, mode='upsample3d' if temperal_upsample[i] else 'upsample2d'))
                scale *= 2.0
        self.upsamples = nn.Sequential(*upsamples)
        self.head = nn.Sequential(RMS_norm(out_dim, images=False), nn.SiLU(), CausalConv3d(out_dim, 3, 3, padding=1))


class AutoencoderKLWan_(nn.Module):
    def __init__(self, dim=128, z_dim=4, dim_mult=[1, 2, 4, 4], num_res_blocks=2, attn_scales=[],
                 temperal_downsample=[True, True, False], dropout=0.0):
        super().__init__()
        self.dim, self.z_dim = dim, z_dim
        self.temperal_downsample = temperal_downsample
        self.temperal_upsample = temperal_downsample[::-1]
        self.encoder = Encoder3d(dim, z_dim * 2, dim_mult, num_res_blocks, attn_scales, self.temperal_downsample, dropout)
        self.conv1 = CausalConv3d(z_dim * 2, z_dim * 2, 1)
        self.conv2 = CausalConv3d(z_dim, z_dim, 1)
        self.decoder = Decoder3d(dim, z_dim, dim_mult, num_res_blocks, attn_scales, self.temperal_upsample, dropout)
        self.clear_cache()

    # ---- the inner model's own entry points (reference wan_vae.py:520, :549, :633, :678, :717).  The released callers go through the
    # AutoencoderKLWan wrapper (train_vae.py:444-453, the pipeline); these keep `vae.model.encode(x, vae.scale)` etc. working by name.
    # `scale` = [mean, 1 / std] (tensors of z_dim entries or two floats), as the wrapper passes it (:770-776).
    def encode(self, x, scale):
        """x [B, 3, T, H, W] -> [B, 2 z_dim, T', H/8, W/8] = (mu normalised by `scale` | logvar)  (:520-547)."""
        v = _InnerView(self, scale)
        return torch.stack([v._encode_one(u) for u in x])

    def encode_full(self, x, scale):
        """encode with the per-chunk checkpointed backward of :549-613 (forward values identical to `encode`)."""
        v = _InnerView(self, scale)
        if v._wants_grad(x, [self.encoder, self.conv1]):
            from ..vae_autograd import vae_encode_train
            return vae_encode_train(v, x)
        return torch.stack([v._encode_one(u) for u in x])

    def decode(self, z, scale):
        """z [B, z_dim, T', h, w] -> video [B, 3, T, 8h, 8w], NOT clamped (the wrapper clamps, :825-832)  (:678-703)."""
        v = _InnerView(self, scale)
        return torch.stack([v._decode_one(u, clamp=False) for u in z])

    def decode_full(self, z, scale):
        """decode with the per-latent-frame checkpointed backward of :633-676 (forward values identical to `decode`)."""
        v = _InnerView(self, scale)
        if v._wants_grad(z, [self.decoder, self.conv2]):
            from ..vae_autograd import vae_decode_train
            return vae_decode_train(v, z)
        return torch.stack([v._decode_one(u, clamp=False) for u in z])

    def clear_cache(self):
        """:717-725.  The streaming state of this implementation lives in a per-call runner (every conv's tail frames in its staging
        buffer), so nothing survives a call; the reference's bookkeeping attributes are kept for code that reads them."""
        count = lambda m: sum(isinstance(c, CausalConv3d) for c in m.modules())      # noqa: E731
        self._conv_num, self._conv_idx = count(self.decoder), [0]
        self._feat_map = [None] * self._conv_num
        self._enc_conv_num, self._enc_conv_idx = count(self.encoder), [0]
        self._enc_feat_map = [None] * self._enc_conv_num


# --------------------------------------------------------------------------------------------- runner

CIN_PAD = 16     # input channels of every conv are padded to one MFMA K step (the 3-channel video / 3-channel adaptor inputs)


def _round(n, m):
    return (n + m - 1) // m * m


class _Act:
    """A channels-last activation: `data` is a [t*h*w, C] row-strided 2-D view (`g`: its gradient in the training runner)."""
    __slots__ = ("data", "t", "h", "w", "c", "g", "prefilled_for")

    def __init__(self, data, t, h, w, c):
        self.data, self.t, self.h, self.w, self.c = data, t, h, w, c
        self.g = None
        self.prefilled_for = None   # key of the conv stage whose buffer already holds RMS_norm+SiLU of this activation (conv_causal: then=)


def _data(a):
    return a.data if isinstance(a, _Act) else a


class _Stage:
    """Staging buffer of one causal conv: [tail(n_tail) + chunk] frames of [h, w, C]; tail is zero at creation.

    With ring > 1 the buffer holds `ring` chunks: the next chunk is written right behind the current one, whose last n_tail frames
    then ARE the tail in place; only when the end of the buffer is reached are they copied back to the front (one small copy per
    `ring` chunks instead of one per chunk: 1 233 copy kernels = 19 ms of a 49x480x832 round trip at ring = 1).

    planar: the buffer is [C/16, frames, h*w, 16] (ops.Planar16) instead of [frames, h*w, C]: what the RMS-norm kernel writes for the
    LDS-halo conv kernel, whose halo DMA then uses every byte of the lines it fetches."""

    PLANAR_MAX_BYTES = (1 << 31) - (1 << 20)      # m4d_conv_cl_planar addresses its input with 32-bit offsets

    def __init__(self, n_tail, t, h, w, c, dtype, device, ring=1, planar=False):
        self.n_tail, self.h, self.w, self.c = n_tail, h, w, c
        self.cap, self.ring, self.pos, self.planar = t, ring, 0, planar
        self.buf = self._alloc(self._frames_for(t), dtype, device)

    def _frames_for(self, t):
        """Frames of a buffer for chunks of t frames: tail + ring chunks; a planar buffer must stay below 2 GiB, so large maps get
        fewer chunks per buffer, and a stage that cannot even hold one chunk that way goes (back) to channels-last."""
        if self.planar:
            most = self.PLANAR_MAX_BYTES // ((self.c // 16) * self.h * self.w * 32)
            r = min(self.ring, (most - self.n_tail) // t)
            if r >= 1:
                return self.n_tail + r * t
            self.planar = False
        return self.n_tail + self.ring * t

    def _alloc(self, frames, dtype, device):
        """Only the tail is read before it is written (every chunk is filled in full, padding channels included)."""
        shape = (self.c // 16, frames, self.h * self.w, 16) if self.planar else (frames, self.h * self.w, self.c)
        buf = torch.empty(shape, device=device, dtype=dtype)
        buf.narrow(1 if self.planar else 0, 0, self.n_tail).zero_()
        return buf

    def _frames(self, a, n):
        return self.buf.narrow(1 if self.planar else 0, a, n)

    def chunk(self, t):
        """Where the next chunk of t frames goes: a [t*h*w, C] view, or an ops.Planar16 of a planar stage."""
        total = self.buf.shape[1 if self.planar else 0]
        if t > self.cap or self.pos + self.n_tail + t > total:
            tail = self.tail()
            if t > self.cap:   # grow, keeping the tail
                old, was_planar = tail.clone(), self.planar
                self.cap = t
                self.buf = self._alloc(self._frames_for(t), self.buf.dtype, self.buf.device)
                if was_planar and not self.planar:
                    old = old.permute(1, 2, 0, 3).reshape(self.n_tail, self.h * self.w, self.c)
                self._frames(0, self.n_tail).copy_(old)
            else:              # wrap around
                self._frames(0, self.n_tail).copy_(tail.clone() if self.pos < self.n_tail else tail)
            self.pos = 0
        v = self._frames(self.pos + self.n_tail, t)
        return ops.Planar16(v) if self.planar else v.view(t * self.h * self.w, self.c)

    def window(self, t):
        """[tail + chunk of t frames]: the conv's input."""
        v = self._frames(self.pos, self.n_tail + t)
        return ops.Planar16(v) if self.planar else v

    def tail(self):
        return self._frames(self.pos, self.n_tail)

    def roll(self, t):
        """tail <- last n_tail frames of (tail + chunk of t frames)."""
        if self.ring > 1:
            self.pos += t
            return
        src = self._frames(t, self.n_tail)
        self._frames(0, self.n_tail).copy_(src.clone() if t < self.n_tail else src)


class _Runner:
    """Executes the encoder / decoder module trees with the HIP kernels.  One instance per encode()/decode() call
    (fresh streaming state = the reference's clear_cache(), wan_vae.py:717-724)."""

    PLANAR, PLANAR_MIN_PIXELS, PLANAR_DTYPES, FUSE_NORM, BATCH_ATTN = True, 1024, (torch.bfloat16,), True, True     # (class attributes: tests force either path)

    def __init__(self, vae, device, dtype):
        self.vae, self.dev, self.T = vae, device, dtype
        self.stages = {}
        self.flags = {}
        self.cin_pad = CIN_PAD
        self.ring = 4                  # chunks per staging buffer (_Stage)
        self.planar = self.PLANAR      # norm -> 3x3x3 conv staging buffers in planar-16 layout (bf16, maps of >= 1024 pixels)
        self.fuse_norm = self.FUSE_NORM   # conv1 -> norm -> conv2 of a ResidualBlock: the norm in conv1's epilogue (<= 128 channels)

    # ---- parameter views in kernel layout (cached on the owning AutoencoderKLWan)
    def packed(self, conv):
        cache = self.vae._pack_cache
        key = (id(conv), self.T, conv.weight._version, conv.weight.data_ptr())
        hit = cache.get(id(conv))
        if hit is None or hit[0] != key:
            w = conv.weight.detach()
            if w.dim() == 4:
                w = w.unsqueeze(2)
            co, ci, kt, kh, kw = w.shape
            cip, cop = _round(ci, self.cin_pad), _round(co, 4)
            wp = torch.zeros((cop, kt, kh, kw, cip), device=w.device, dtype=self.T)
            wp[:co, :, :, :, :ci] = w.permute(0, 2, 3, 4, 1)
            bp = None
            if conv.bias is not None:
                bp = torch.zeros(cop, device=w.device, dtype=self.T)
                bp[:co] = conv.bias.detach()
            hit = (key, wp.view(cop, -1), bp, (kt, kh, kw), cip, cop)
            cache[id(conv)] = hit
        return hit[1:]

    def tiled(self, conv):
        """The conv's weights once more in the tiled order of the LDS-halo kernels (ops.conv_pack_weights), or None; cached like packed()."""
        if not _TILED_WEIGHTS or self.T != torch.bfloat16 or conv.weight.shape[-1] != 3 or conv.weight.shape[-2] != 3:
            return None
        cache = self.vae._pack_cache
        key = (id(conv), self.T, conv.weight._version, conv.weight.data_ptr())
        hit = cache.get(("tiled", id(conv)))
        if hit is None or hit[0] != key:
            w, _, _, cip, _ = self.packed(conv)
            hit = (key, ops.conv_pack_weights(w, cip) if w.is_cuda else None)
            cache[("tiled", id(conv))] = hit
        return hit[1]

    def gamma(self, norm):
        cache = self.vae._pack_cache
        key = (id(norm), norm.gamma._version, norm.gamma.data_ptr())
        hit = cache.get(id(norm))
        if hit is None or hit[0] != key:
            hit = (key, norm.gamma.detach().float().reshape(-1).contiguous())
            cache[id(norm)] = hit
        return hit[1]

    def stage(self, key, n_tail, t, h, w, c, planar=False):
        st = self.stages.get(key)
        if st is None:
            st = _Stage(n_tail, t, h, w, c, self.T, self.dev, ring=self.ring, planar=planar)
            self.stages[key] = st
        return st

    # ---- primitives
    def conv_plain(self, x: _Act, conv, resid=None, out=None, stride_hw=1, ups=False, tsplit=False, x_pixel_stride=None):
        """Non-temporal conv (kt = 1): 1x1x1, Conv2d 3x3 (pad 1 | stride 2 with right/bottom zero pad), per frame."""
        w, b, (kt, kh, kw), cip, cop = self.packed(conv)
        assert kt == 1 and cip == x.c, (kt, cip, x.c)
        t = x.t * (2 if tsplit else 1)
        hl, wl = x.h * (2 if ups else 1), x.w * (2 if ups else 1)
        if stride_hw == 2:
            ho, wo, pad = hl // 2, wl // 2, 0
        else:
            ho, wo, pad = hl, wl, kh // 2
        y = ops.conv_cl(x.data, w, b, Tin=x.t, Hin=x.h, Win=x.w, Cin=x.c, k=(1, kh, kw), stride=(1, stride_hw, stride_hw),
                        pad=(0, pad, pad), out_thw=(t, ho, wo), resid=_data(resid), out=out, ups=ups, tsplit=tsplit,
                        x_pixel_stride=x_pixel_stride, w_tiled=self.tiled(conv))
        return _Act(y, t, ho, wo, cop)

    def conv_causal(self, key, conv, t, h, w, fill, resid=None, out=None, then=None, keep_raw=True):
        """k=3 causal conv with a 2-frame tail.  `fill(dst)` writes the chunk [t*h*w, Cin] into the staging buffer.
        then = (norm, next_key, next_conv): the caller will feed RMS_norm+SiLU of this conv's output into `next_conv` (conv1 -> norm ->
        SiLU -> conv2 of a ResidualBlock, a block's output -> the next block's / the head's norm); where the kernel can, that norm runs in
        this conv's epilogue straight into next_conv's staging buffer and the returned activation carries `prefilled_for = next_key`
        (with keep_raw=False its `.data` is None: nobody else reads the un-normalised result)."""
        wgt, b, (kt, kh, kw), cip, cop = self.packed(conv)
        # planar-16 staging where the producer can write it (RMS-norm) and the LDS-halo kernel reads it (the decision is per stage:
        # it must not depend on the chunk length)
        planar = self._planar_stage(fill, (kt, kh, kw), cip, h, w)
        st = self.stage(key, kt - 1, t, h, w, cip, planar=planar)
        src = getattr(fill, "src", None)
        if src is None or src.prefilled_for != key:         # (else the producing conv's epilogue already wrote this chunk)
            fill(st.chunk(t))
        fused_for = None
        if st.planar:
            norm = None
            if then is not None and self.fuse_norm and conv.weight.shape[0] in (32, 64, 96, 128) and out is None:
                nxt_norm, nxt_key, nxt_conv = then
                _, _, k2, cip2, _ = self.packed(nxt_conv)
                if cip2 == cop and self._planar_stage(self.norm_into(None, nxt_norm), k2, cip2, h, w):
                    dst2 = self.stage(nxt_key, k2[0] - 1, t, h, w, cip2, planar=True).chunk(t)
                    if isinstance(dst2, ops.Planar16):
                        norm, fused_for = (self.gamma(nxt_norm), dst2, True), nxt_key
            y = ops.conv_cl_planar(st.window(t), wgt, b, Tin=st.n_tail + t, Hin=h, Win=w, kt=kt, resid=_data(resid), out=out, norm=norm,
                                   keep_raw=keep_raw or fused_for is None, w_tiled=self.tiled(conv))
        else:
            y = ops.conv_cl(st.window(t), wgt, b, Tin=st.n_tail + t, Hin=h, Win=w, Cin=cip, k=(kt, kh, kw), pad=(0, kh // 2, kw // 2),
                            out_thw=(t, h, w), resid=_data(resid), out=out, w_tiled=self.tiled(conv))
        st.roll(t)
        a = _Act(y, t, h, w, cop)
        a.prefilled_for = fused_for
        return a

    def _planar_stage(self, fill, k, cip, h, w):
        return (self.planar and getattr(fill, "planar_ok", False) and self.T in self.PLANAR_DTYPES and tuple(k) == (3, 3, 3)
                and cip % 16 == 0 and h * w >= self.PLANAR_MIN_PIXELS)

    # fill callbacks: write a chunk into a conv's staging buffer (the training runner returns objects that also know
    # how to take the gradient of that chunk back to where it came from)
    def norm_into(self, x: _Act, norm, silu=True):
        g = self.gamma(norm)

        def fill(dst):
            if isinstance(dst, ops.Planar16):
                return ops.rmsnorm_silu_cl_planar(x.data, g, dst, silu=silu)
            return ops.rmsnorm_silu_cl(x.data, g, silu=silu, out=dst)
        fill.planar_ok = True
        fill.src = x
        return fill

    def copy_into(self, x: _Act):
        return lambda dst: dst.copy_(x.data)

    def tsplit_view(self, y: _Act, c):
        """[t, h, w, 2c] read as 2t frames of c channels (even frame = first half, odd frame = second half, :138-141)."""
        return _Act(y.data, y.t, y.h, y.w, c)

    def video_into(self, x_ncthw):
        _, t, H, W = x_ncthw.shape
        return lambda dst: ops.ncthw_to_cl(x_ncthw, self.T, Cp=self.cin_pad, out=dst.view(t, H, W, -1))

    def snapshot(self):
        """The streaming state in front of the next chunk: every conv's tail frames and the first-chunk flags (what the
        reference's `_clone_cache`, wan_vae.py:604-613, captures for the checkpointed twins)."""
        return ({k: (st.n_tail, st.h, st.w, st.c, st.planar, st.tail().clone()) for k, st in self.stages.items()}, dict(self.flags))

    def restore(self, snap):
        tails, flags = snap
        self.stages = {}
        for k, (n_tail, h, w, c, planar, tail) in tails.items():
            st = _Stage(n_tail, 1, h, w, c, self.T, self.dev, ring=self.ring, planar=planar and self.planar)
            if planar and not st.planar:         # a channels-last runner (training recompute) restoring a planar snapshot
                tail = tail.permute(1, 2, 0, 3).reshape(n_tail, h * w, c)
            st.tail().copy_(tail)
            self.stages[k] = st
        self.flags = dict(flags)

    # ---- blocks
    def residual_block(self, x: _Act, blk, key, out=None, then=None):
        """then: the (norm, key, conv) that consumes this block's output next (conv_causal)."""
        r = blk.residual
        y1 = self.conv_causal(key + ".residual.2", r[2], x.t, x.h, x.w, self.norm_into(x, r[0]), then=(r[3], key + ".residual.6", r[6]),
                              keep_raw=False)
        h = x if isinstance(blk.shortcut, nn.Identity) else self.conv_plain(x, blk.shortcut)
        return self.conv_causal(key + ".residual.6", r[6], x.t, x.h, x.w, self.norm_into(y1, r[3]), resid=h, out=out, then=then)

    @staticmethod
    def _next_norm(layers, i, prefix, tail=None):
        """What consumes layer i's output through an RMS-norm: the next ResidualBlock's conv1, or `tail` (the head) after the last."""
        if i + 1 < len(layers):
            nxt = layers[i + 1]
            if isinstance(nxt, ResidualBlock):
                return nxt.residual[0], f"{prefix}.{i + 1}.residual.2", nxt.residual[2]
            return None
        return tail

    def attention_block(self, x: _Act, blk):
        """Single-head attention over h*w tokens per frame (wan_vae.py:244-266): scores materialised per frame
        (head dim = C = 384 exceeds the flash kernel's 128), softmax rows, P.V with V^T from the projection GEMM."""
        C, hw = x.c, x.h * x.w
        hwp = _round(hw, 8)
        g = self.gamma(blk.norm)
        wq, bq, _, _, _ = self.packed(blk.to_qkv)          # [3C, C]
        wp, bp, _, _, _ = self.packed(blk.proj)
        outs = torch.empty((x.t * hw, C), device=self.dev, dtype=self.T)
        if self.BATCH_ATTN and hwp == hw and x.t > 1:
            # the per-token parts (norm, q/k/v projections, output projection + residual) over all frames of the chunk at once; only the
            # scores / softmax / P.V are per frame.  Row-wise identical to the per-frame calls (same kernels, same K order).
            xn = ops.rmsnorm_silu_cl(x.data, g, silu=False)
            qk = ops.gemm_bt(xn, wq[:2 * C], bq[:2 * C])                                   # [t*hw, 2C]
            vt = ops.gemm_bt(wq[2 * C:], xn, bq[2 * C:], bias_on_m=True)                   # V^T [C, t*hw]
            o = torch.empty((x.t * hw, C), device=self.dev, dtype=self.T)
            for f in range(x.t):
                r = slice(f * hw, (f + 1) * hw)
                s = ops.gemm_bt(qk[r, :C], qk[r, C:], None, epilogue=ops.EPI_STORE_F32)    # [hw, hw] fp32
                p = ops.softmax_rows(s, self.T, C=hw, Cpad=hw, scale=1.0 / math.sqrt(C))
                ops.gemm_bt(p, vt[:, r], None, out=o[r])
            ops.conv_cl(o, wp, bp, Tin=1, Hin=1, Win=x.t * hw, Cin=C, k=(1, 1, 1), out_thw=(1, 1, x.t * hw), resid=x.data, out=outs)
            return _Act(outs, x.t, x.h, x.w, C)
        for f in range(x.t):
            xf = x.data[f * hw:(f + 1) * hw]
            xn = torch.zeros((hwp, C), device=self.dev, dtype=self.T)
            ops.rmsnorm_silu_cl(xf, g, silu=False, out=xn[:hw])
            qk = ops.gemm_bt(xn, wq[:2 * C], bq[:2 * C])                                   # [hwp, 2C]
            vt = ops.gemm_bt(wq[2 * C:], xn, bq[2 * C:], bias_on_m=True)                   # V^T [C, hwp]
            s = ops.gemm_bt(qk[:, :C], qk[:hw, C:], None, epilogue=ops.EPI_STORE_F32) if hw % 4 == 0 else \
                ops.gemm_bt(qk[:, :C], qk[:, C:], None, epilogue=ops.EPI_STORE_F32)       # [hwp, hw(p)] fp32
            p = ops.softmax_rows(s, self.T, C=hw, Cpad=hwp, scale=1.0 / math.sqrt(C))      # [hwp, hwp]
            o = ops.gemm_bt(p, vt, None)                                                   # [hwp, C]
            ops.conv_cl(o, wp, bp, Tin=1, Hin=1, Win=hw, Cin=C, k=(1, 1, 1), out_thw=(1, 1, hw), resid=xf,
                        out=outs[f * hw:(f + 1) * hw])
        return _Act(outs, x.t, x.h, x.w, C)

    def resample(self, x: _Act, rs, key):
        mode = rs.mode
        if mode in ("upsample2d", "upsample3d"):
            conv = rs.resample[1]
            if mode == "upsample3d":
                if not self.flags.get(key):            # first chunk: 'Rep', no temporal up-sampling (:107-112)
                    self.flags[key] = True
                else:
                    # time_conv over [tail, x]; its tail starts at zero and never sees chunk 0 (:124-132)
                    y = self.conv_causal(key + ".time_conv", rs.time_conv, x.t, x.h, x.w, self.copy_into(x))
                    # channels [0,C) -> even frames, [C,2C) -> odd frames (:138-141): read through the tsplit view
                    return self.conv_plain(self.tsplit_view(y, x.c), conv, ups=True, tsplit=True, x_pixel_stride=2 * x.c)
            return self.conv_plain(x, conv, ups=True)
        if mode in ("downsample2d", "downsample3d"):
            conv = rs.resample[1]
            if mode == "downsample2d":
                return self.conv_plain(x, conv, stride_hw=2)
            wgt, b, (kt, kh, kw), cip, cop = self.packed(rs.time_conv)
            ho, wo = x.h // 2, x.w // 2
            st = self.stage(key + ".time_conv", 1, x.t, ho, wo, cop)
            if not self.flags.get(key):                # first chunk: keep the frame as the 1-frame tail (:147-152)
                self.flags[key] = True
                y = self.conv_plain(x, conv, stride_hw=2, out=st.chunk(x.t))
                st.roll(x.t)
                return y
            self.conv_plain(x, conv, stride_hw=2, out=st.chunk(x.t))
            to = x.t // 2
            y = ops.conv_cl(st.window(x.t), wgt, b, Tin=1 + x.t, Hin=ho, Win=wo, Cin=cop, k=(3, 1, 1), stride=(2, 1, 1),
                            out_thw=(to, ho, wo))
            st.roll(x.t)
            return _Act(y, to, ho, wo, cop)
        return x

    # ---- networks
    def encoder(self, x_ncthw, out_view):
        """x_ncthw [3, t, H, W] (one chunk) -> writes [t', h, w, 2z] into out_view."""
        enc = self.vae.model.encoder
        _, t, H, W = x_ncthw.shape
        a = self.conv_causal("enc.conv1", enc.conv1, t, H, W, self.video_into(x_ncthw))
        for i, layer in enumerate(enc.downsamples):
            key = f"enc.down.{i}"
            a = self.residual_block(a, layer, key, then=self._next_norm(enc.downsamples, i, "enc.down")) \
                if isinstance(layer, ResidualBlock) else self.resample(a, layer, key)
        a = self.residual_block(a, enc.middle[0], "enc.mid.0")
        a = self.attention_block(a, enc.middle[1])
        a = self.residual_block(a, enc.middle[2], "enc.mid.2")
        self.conv_causal("enc.head", enc.head[2], a.t, a.h, a.w, self.norm_into(a, enc.head[0]), out=out_view(a.t))
        return a.t

    def decoder(self, z_act: _Act):
        """z_act: one latent frame [1*h*w, z] -> _Act [t_out*H*W, 4] (3 channels + pad)."""
        dec = self.vae.model.decoder
        a = self.conv_causal("dec.conv1", dec.conv1, z_act.t, z_act.h, z_act.w, self.copy_into(z_act))
        a = self.residual_block(a, dec.middle[0], "dec.mid.0")
        a = self.attention_block(a, dec.middle[1])
        a = self.residual_block(a, dec.middle[2], "dec.mid.2")
        head = (dec.head[0], "dec.head", dec.head[2])
        for i, layer in enumerate(dec.upsamples):
            key = f"dec.up.{i}"
            a = self.residual_block(a, layer, key, then=self._next_norm(dec.upsamples, i, "dec.up", head)) \
                if isinstance(layer, ResidualBlock) else self.resample(a, layer, key)
        return self.conv_causal("dec.head", dec.head[2], a.t, a.h, a.w, self.norm_into(a, dec.head[0]))


# --------------------------------------------------------------------------------------------- public API

class DiagonalGaussianDistribution:
    """diffusers' published semantics (third-party, unpinned): chunk, clamp logvar to [-30, 20]."""

    def __init__(self, parameters):
        self.parameters = parameters
        self.mean, self.logvar = torch.chunk(parameters, 2, dim=1)
        self.logvar = torch.clamp(self.logvar, -30.0, 20.0)
        self.std = torch.exp(0.5 * self.logvar)
        self.var = torch.exp(self.logvar)

    def kl(self, other=None):
        """KL to the standard normal (or to `other`), summed over dims [1, 2, 3] exactly like the published class does —
        for the 5-D video latents this leaves the last axis, which train_vae.py:183 then sums itself."""
        if other is None:
            return 0.5 * torch.sum(self.mean.float().pow(2) + self.var.float() - 1.0 - self.logvar.float(), dim=[1, 2, 3])
        return 0.5 * torch.sum((self.mean.float() - other.mean.float()).pow(2) / other.var.float() + self.var.float() / other.var.float()
                               - 1.0 - self.logvar.float() + other.logvar.float(), dim=[1, 2, 3])

    def sample(self, generator=None):
        eps = torch.randn(self.mean.shape, generator=generator, device=self.mean.device if generator is None or
                          generator.device.type != "cpu" else "cpu", dtype=torch.float32).to(self.mean.device)
        return self.mean + self.std * eps.to(self.mean.dtype)

    def mode(self):
        return self.mean


class AutoencoderKLOutput:
    def __init__(self, latent_dist):
        self.latent_dist = latent_dist

    def __getitem__(self, i):
        return (self.latent_dist,)[i]


class DecoderOutput:
    def __init__(self, sample):
        self.sample = sample

    def __getitem__(self, i):
        return (self.sample,)[i]


class _Config(dict):
    __getattr__ = dict.__getitem__


def _video_vae(z_dim=None, **kwargs):
    cfg = dict(dim=96, z_dim=z_dim, dim_mult=[1, 2, 4, 4], num_res_blocks=2, attn_scales=[],
               temperal_downsample=[False, True, True], dropout=0.0)
    cfg.update(**kwargs)
    return AutoencoderKLWan_(**cfg)


class _VaeCore:
    """What encode / decode need from their owner: `model` (AutoencoderKLWan_), `mean` / `std` (latent normalisation), the packed-weight
    cache.  Shared by the AutoencoderKLWan wrapper and the views the inner model's own entry points build."""
    clamp_output = True      # decode: clamp(-1, 1) fused into the layout kernel (the wrapper's behaviour, :825-832)

    @property
    def _pack_cache(self):
        return self.model.__dict__.setdefault("_m4d_pack_cache", {})

    @property
    def dtype(self):
        return self.model.conv1.weight.dtype

    @property
    def device(self):
        return self.model.conv1.weight.device

    def _wants_grad(self, x, mods):
        return torch.is_grad_enabled() and (x.requires_grad or any(p.requires_grad for m in mods for p in m.parameters()))

    CHUNK_LATENT = int(os.environ.get("M4D_VAE_CHUNK", "4"))      # latent frames per streaming chunk after the first frame

    def _chunk_latent(self, H, W):
        """Latent frames per streaming chunk: CHUNK_LATENT, less on maps whose full-resolution staging buffer ([tail + 4 frames per
        latent frame] x H x W x dim channels, planar) would not fit the 2 GiB the conv kernel addresses (720p: 2, 1080p: 1)."""
        dim = self.model.encoder.dim
        most = _Stage.PLANAR_MAX_BYTES // (max(1, dim // 16) * H * W * 32)          # frames of a full-resolution planar buffer
        return max(1, min(self.CHUNK_LATENT, (most - 2) // 4))

    def _encode_one(self, x):
        """x [3, T, H, W] -> [2z, T', h, w] (mu normalised | logvar), reference encode (:520-547)."""
        dev, T = self.device, self.dtype
        z2 = 2 * self.latent_channels
        x = x.to(dev)
        t = x.shape[1]
        run = _Runner(self, dev, T)
        lat_t = 1 + (t - 1) // 4
        h, w = x.shape[2] // 8, x.shape[3] // 8
        enc_out = torch.empty((lat_t, h * w, z2), device=dev, dtype=T)
        pos = 0
        # the reference streams 1 + 4 + 4 + ... frames (:520-547); every conv is causal over the cached tail, so the chunk length
        # only changes how much work one launch carries: CHUNK_LATENT latent frames' worth per chunk after the first frame
        cl = self._chunk_latent(x.shape[2], x.shape[3])
        bounds = [0, 1] + list(range(1 + 4 * cl, 1 + 4 * (lat_t - 1), 4 * cl)) + [1 + 4 * (lat_t - 1)]
        for f0, f1 in zip(bounds[:-1], bounds[1:]):
            if f1 <= f0:
                continue
            def view(tt, pos=pos):
                return enc_out[pos:pos + tt].view(tt * h * w, z2)
            pos += run.encoder(x[:, f0:f1], view)
        a = _Act(enc_out.view(lat_t * h * w, z2), lat_t, h, w, z2)
        y = run.conv_plain(a, self.model.conv1)
        ch_scale, ch_shift = self._latent_affine(dev)
        return ops.cl_to_ncthw(y.data, T, C=z2, T=lat_t, H=h, W=w, pixel_stride=y.data.stride(0), ch_scale=ch_scale,
                               ch_shift=ch_shift)

    def _latent_affine(self, dev):
        """Per-channel (scale, shift) of the encoder output [mu | logvar]: mu <- (mu - mean) / std, logvar untouched (:539-545)."""
        zc = self.latent_channels
        inv = (1.0 / self.std).to(dev)
        return (torch.cat([inv, torch.ones(zc, device=dev)]),
                torch.cat([-self.mean.to(dev) * inv, torch.zeros(zc, device=dev)]))

    def _decode_one(self, z, clamp=None):
        """z [zc, T', h, w] -> clamp(-1,1) video [3, T, 8h, 8w], reference decode (:678-703, :825-832)."""
        dev, T = self.device, self.dtype
        clamp = self.clamp_output if clamp is None else clamp
        zc, lt, h, w = z.shape
        run = _Runner(self, dev, T)
        zin = ops.ncthw_to_cl(z.to(dev), T, ch_scale=self.std.to(dev), ch_shift=self.mean.to(dev))      # z/(1/std)+mean
        a = run.conv_plain(_Act(zin.view(lt * h * w, zc), lt, h, w, zc), self.model.conv2)
        frames = []
        cl = self._chunk_latent(8 * h, 8 * w)
        bounds = [0, 1] + list(range(1 + cl, lt, cl)) + [lt]      # reference: one latent frame per chunk (:678-703)
        for i0, i1 in zip(bounds[:-1], bounds[1:]):
            if i1 <= i0:
                continue
            o = run.decoder(_Act(a.data[i0 * h * w:i1 * h * w], i1 - i0, h, w, zc))
            frames.append(ops.cl_to_ncthw(o.data, T, C=3, T=o.t, H=o.h, W=o.w, pixel_stride=o.data.stride(0), act=1 if clamp else 0))
        return torch.cat(frames, dim=1)



class _InnerView(_VaeCore):
    """AutoencoderKLWan_.encode / decode (x, scale): the inner model + the caller's `scale` = [mean, 1 / std]."""
    clamp_output = False

    def __init__(self, model, scale):
        self.model = model
        zc = model.z_dim
        self.latent_channels = zc
        mean, inv = scale[0], scale[1]
        as_vec = lambda v: (v.detach().float().reshape(-1).cpu() if isinstance(v, torch.Tensor) else torch.tensor([float(v)])).expand(zc).contiguous()      # noqa: E731
        self.mean, self.std = as_vec(mean), 1.0 / as_vec(inv)


class AutoencoderKLWan(_VaeCore, nn.Module):
    def __init__(self, latent_channels=16, temporal_compression_ratio=4, spatial_compression_ratio=8, **vae_kwargs):
        super().__init__()
        self.config = _Config(latent_channels=latent_channels, temporal_compression_ratio=temporal_compression_ratio,
                              spatial_compression_ratio=spatial_compression_ratio)
        self.latent_channels = latent_channels                       # read as plain attributes by the pipeline
        self.temporal_compression_ratio = temporal_compression_ratio  # (pipeline_wan_fun_control.py:185-186, 736)
        self.spatial_compression_ratio = spatial_compression_ratio
        mean = [-0.7571, -0.7089, -0.9113, 0.1075, -0.1745, 0.9653, -0.1517, 1.5508,
                0.4134, -0.0715, 0.5517, -0.3632, -0.1922, -0.9497, 0.2503, -0.2921]
        std = [2.8184, 1.4541, 2.3275, 2.6558, 1.2196, 1.7708, 2.6052, 2.0743,
               3.2687, 2.1526, 2.8652, 1.5579, 1.6382, 1.1253, 2.8251, 1.9160]
        self.mean = torch.tensor(mean[:latent_channels], dtype=torch.float32)
        self.std = torch.tensor(std[:latent_channels], dtype=torch.float32)
        self.scale = [self.mean, 1.0 / self.std]
        self.model = _video_vae(z_dim=latent_channels, **vae_kwargs)

    def _encode(self, x):
        return torch.stack([self._encode_one(u) for u in x])

    def encode(self, x, return_dict=True):
        dist = DiagonalGaussianDistribution(self._encode(x))
        return AutoencoderKLOutput(latent_dist=dist) if return_dict else (dist,)

    def encode_memory_saver(self, x, return_dict=True):
        """Training-time twin (reference `encode_full`, wan_vae.py:549-613, wrapper :783-812): same forward values; under
        autograd the gradient is computed chunk by chunk with the streaming cache cut between chunks, like the reference's
        per-chunk checkpoint + `_detach_cache`.  Without grad it IS `encode`."""
        if self._wants_grad(x, [self.model.encoder, self.model.conv1]):
            from ..vae_autograd import vae_encode_train
            params = vae_encode_train(self, x)
        else:
            params = self._encode(x)
        dist = DiagonalGaussianDistribution(params)
        return AutoencoderKLOutput(latent_dist=dist) if return_dict else (dist,)

    def _decode(self, zs):
        return DecoderOutput(sample=torch.stack([self._decode_one(u) for u in zs]))

    def decode(self, z, return_dict=True):
        out = self._decode(z).sample
        return DecoderOutput(sample=out) if return_dict else (out,)

    def decode_memory_saver(self, z, return_dict=True):
        """Training-time twin (reference `decode_full`, wan_vae.py:633-676, wrapper :815-843)."""
        if self._wants_grad(z, [self.model.decoder, self.model.conv2]):
            from ..vae_autograd import vae_decode_train
            out = vae_decode_train(self, z)
        else:
            out = self._decode(z).sample
        return DecoderOutput(sample=out) if return_dict else (out,)

    @classmethod
    def from_pretrained(cls, pretrained_model_path, additional_kwargs={}):
        import inspect
        valid = set(inspect.signature(cls.__init__).parameters) - {"self", "vae_kwargs"}
        model = cls(**{k: v for k, v in additional_kwargs.items() if k in valid})
        if pretrained_model_path.endswith(".safetensors"):
            from safetensors.torch import load_file
            state_dict = load_file(pretrained_model_path)
        else:
            state_dict = torch.load(pretrained_model_path, map_location="cpu", weights_only=True)
        state_dict = {"model." + k: v for k, v in state_dict.items()}     # reference :864-868
        m, u = model.load_state_dict(state_dict, strict=False)
        print(f"### missing keys: {len(m)}; \n### unexpected keys: {len(u)};")
        return model
