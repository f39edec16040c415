"""Trajectory adaptors of the Motion-Sensitive VAE — MI355X-native host side.

Drop-in for `MoRe4D/models/trajectory_module.py` (VAEEncoderadaptor :125-196, VAEDecoderadaptor :200-279): per-frame
2-D ResNets that map XYZ trajectory fields <-> pseudo-RGB video around the Wan VAE.  Same parameter names
(`conv_in`, `down.0.block.0.{norm1,conv1,norm2,conv2}` / `up.0.block.{0,1}.*`, `norm_out`, `conv_out`); the torch
modules are parameter containers, the forward runs on channels-last activations with the HIP kernels:
Conv2d 3x3 = implicit GEMM (`ops.conv_cl`, kt = 1), GroupNorm(32)+swish = `ops.groupnorm_cl`, and the boundary
kernels fuse the final `sigmoid(h + x)` of the encoder adaptor.
"""
import torch
import torch.nn as nn

from .. import ops
from .wan_vae import CIN_PAD, _TILED_WEIGHTS, _round


def zero_module(module):
    for p in module.parameters():
        p.detach().zero_()
    return module


def Normalize(in_channels, num_groups=32):
    return torch.nn.GroupNorm(num_groups=num_groups, num_channels=in_channels, eps=1e-6, affine=True)


class ResnetBlock(nn.Module):
    def __init__(self, *, in_channels, out_channels=None, conv_shortcut=False, dropout, temb_channels=512):
        super().__init__()
        self.in_channels = in_channels
        out_channels = in_channels if out_channels is None else out_channels
        self.out_channels = out_channels
        self.use_conv_shortcut = conv_shortcut
        self.norm1 = Normalize(in_channels)
        self.conv1 = torch.nn.Conv2d(in_channels, out_channels, kernel_size=3, stride=1, padding=1)
        if temb_channels > 0:
            self.temb_proj = torch.nn.Linear(temb_channels, out_channels)
        self.norm2 = Normalize(out_channels)
        self.dropout = torch.nn.Dropout(dropout)
        self.conv2 = torch.nn.Conv2d(out_channels, out_channels, kernel_size=3, stride=1, padding=1)
        if self.in_channels != self.out_channels:
            raise NotImplementedError("the adaptors only use in_channels == out_channels blocks")


class _AdaptorBase(nn.Module):
    """Shared runner: channels-last frames [F, H*W, C]."""

    def _setup(self):
        self._pack_cache = {}

    @property
    def dtype(self):
        return self.conv_in.weight.dtype

    @property
    def device(self):
        return self.conv_in.weight.device

    def _packed(self, conv):
        T = self.dtype
        key = (conv.weight._version, conv.weight.data_ptr(), T)
        hit = self._pack_cache.get(id(conv))
        if hit is None or hit[0] != key:
            w = conv.weight.detach()
            co, ci, kh, kw = w.shape
            cip, cop = _round(ci, CIN_PAD), _round(co, 4)
            wp = torch.zeros((cop, kh, kw, cip), device=w.device, dtype=T)
            wp[:co, :, :, :ci] = w.permute(0, 2, 3, 1)
            bp = torch.zeros(cop, device=w.device, dtype=T)
            bp[:co] = conv.bias.detach()
            hit = (key, wp.view(cop, -1), bp, cip, cop)
            self._pack_cache[id(conv)] = hit
        return hit[1:]

    def _tiled(self, conv):
        """the 3x3 conv's weights in the tiled order of the LDS-halo kernels (ops.conv_pack_weights), cached like _packed()"""
        if not _TILED_WEIGHTS or self.dtype != torch.bfloat16:
            return None
        key = (conv.weight._version, conv.weight.data_ptr(), self.dtype)
        hit = self._pack_cache.get(("tiled", id(conv)))
        if hit is None or hit[0] != key:
            w, _, cip, _ = self._packed(conv)
            hit = (key, ops.conv_pack_weights(w, cip) if w.is_cuda else None)
            self._pack_cache[("tiled", id(conv))] = hit
        return hit[1]

    def _f32(self, p):
        key = (p._version, p.data_ptr())
        hit = self._pack_cache.get(id(p))
        if hit is None or hit[0] != key:
            hit = (key, p.detach().float().contiguous())
            self._pack_cache[id(p)] = hit
        return hit[1]

    def _conv(self, x, conv, F, H, W, cin, resid=None):
        w, b, cip, cop = self._packed(conv)
        assert cip == cin
        return ops.conv_cl(x, w, b, Tin=F, Hin=H, Win=W, Cin=cin, k=(1, 3, 3), pad=(0, 1, 1), out_thw=(F, H, W),
                           resid=resid, w_tiled=self._tiled(conv)), cop

    def _gn_swish(self, x, norm, F, HW):
        return ops.groupnorm_cl(x.view(F, HW, -1), self._f32(norm.weight), self._f32(norm.bias), F=F, HW=HW,
                                groups=norm.num_groups, eps=norm.eps, silu=True).view(F * HW, -1)

    PLANAR, PLANAR_MIN_PIXELS, PLANAR_MAX_BYTES, PLANAR_DTYPES = True, 1024, (1 << 31) - (1 << 20), (torch.bfloat16,)    # (tests force either path)

    def _planar_ok(self, cin, H, W):
        return self.PLANAR and self.dtype in self.PLANAR_DTYPES and cin % 16 == 0 and H * W >= self.PLANAR_MIN_PIXELS

    def _stats_buf(self, cout, F, H, W):
        """Where a conv writes the next GroupNorm's per-patch sums (32 groups of 4 channels = the adaptors' 128 channels), or None."""
        if cout != 128:
            return None
        return torch.empty((F, ops.gnstats_blocks(H, W), 32, 2), device=self.device, dtype=torch.float32)

    def _conv_groups(self, groups, conv, F, H, W, resid=None, stats=False):
        """3x3 conv over planar-16 frame groups -> ([F*H*W, cop], cop, stats or None); stats = per-patch GroupNorm sums of the result."""
        w, b, cip, cop = self._packed(conv)
        out = torch.empty((F * H * W, cop), device=w.device, dtype=w.dtype)
        wt = self._tiled(conv)
        st = self._stats_buf(conv.weight.shape[0], F, H, W) if stats else None
        f0 = 0
        for g in groups:
            n = g.t.shape[1]
            rows = slice(f0 * H * W, (f0 + n) * H * W)
            ops.conv_cl_planar(g, w, b, Tin=n, Hin=H, Win=W, kt=1, resid=None if resid is None else resid[rows], out=out[rows],
                               gn_stats=None if st is None else st[f0:f0 + n], w_tiled=wt)
            f0 += n
        return out, cop, st

    def _norm_conv(self, h, norm, conv, F, H, W, cin, resid=None, stats_in=None, stats_out=False):
        """GroupNorm + swish + 3x3 conv (:54-71) -> (out, cop, stats).  bf16 inference on real maps: the norm writes planar-16 frame groups
        (each below the 2 GiB the conv kernel addresses) and the LDS-halo conv reads them, so its halo DMA uses the lines it fetches;
        stats_in = the producing conv's per-patch sums of h (no statistics pass), stats_out = have this conv write them for the next norm."""
        if self._planar_ok(cin, H, W):
            most = max(1, self.PLANAR_MAX_BYTES // ((cin // 16) * H * W * 32))
            ng = -(-F // most)
            groups = ops.groupnorm_cl_planar(h.view(F, H * W, -1), self._f32(norm.weight), self._f32(norm.bias), F=F, HW=H * W,
                                             groups=norm.num_groups, eps=norm.eps, silu=True, frames_per_group=-(-F // ng),
                                             stats=stats_in if norm.num_groups == 32 else None)
            assert self._packed(conv)[2] == cin
            return self._conv_groups(groups, conv, F, H, W, resid=resid, stats=stats_out)
        y, cop = self._conv(self._gn_swish(h, norm, F, H * W), conv, F, H, W, cin, resid=resid)
        return y, cop, None

    def _conv_first(self, h16, conv, F, H, W):
        """conv_in on the 16-channel padded input -> (out, stats): a [rows, 16] tensor IS planar-16 with one plane."""
        if self._planar_ok(CIN_PAD, H, W) and F * H * W * 32 < self.PLANAR_MAX_BYTES:
            out, _, st = self._conv_groups([ops.Planar16(h16.view(1, F, H * W, 16))], conv, F, H, W, stats=True)
            return out, st
        return self._conv(h16, conv, F, H, W, CIN_PAD)[0], None

    def _run(self, x):
        """Per-sample forward; under autograd (trainable adaptor or an input that needs its gradient, train_vae.py:438-455) every
        sample is one `vae_autograd.AdaptorFn` node whose backward recomputes groups of frames with the HIP kernels."""
        if torch.is_grad_enabled() and (x.requires_grad or any(p.requires_grad for p in self.parameters())):
            from ..vae_autograd import adaptor_train
            return adaptor_train(self, x)
        return torch.stack([self._forward_one(u) for u in x])

    def _resnet(self, h, blk, F, H, W, stats=None):
        """-> (block output, per-patch GroupNorm sums of it for the next norm, or None)."""
        c = blk.in_channels
        y, _, st = self._norm_conv(h, blk.norm1, blk.conv1, F, H, W, c, stats_in=stats, stats_out=True)
        y, _, st = self._norm_conv(y, blk.norm2, blk.conv2, F, H, W, c, resid=h, stats_in=st, stats_out=True)
        return y, st


class VAEEncoderadaptor(_AdaptorBase):
    def __init__(self, *, ch=128, out_ch=1, ch_mult=(1,), num_res_blocks=1, attn_resolutions=[], dropout=0.0,
                 resamp_with_conv=True, in_channels=3, resolution=256, z_channels=4, double_z=True,
                 use_linear_attn=False, attn_type="vanilla", **ignore_kwargs):
        super().__init__()
        self.ch, self.temb_ch = ch, 0
        self.num_resolutions = len(ch_mult)
        assert self.num_resolutions == 1
        self.num_res_blocks, self.resolution, self.in_channels = num_res_blocks, resolution, in_channels
        self.final_activation = nn.Sigmoid()
        self.conv_in = torch.nn.Conv2d(in_channels, self.ch, kernel_size=3, stride=1, padding=1)
        in_ch_mult = (1,) + tuple(ch_mult)
        self.in_ch_mult = in_ch_mult
        self.down = nn.ModuleList()
        for i_level in range(self.num_resolutions):
            block = nn.ModuleList()
            attn = nn.ModuleList()
            block_in = ch * in_ch_mult[i_level]
            block_out = ch * ch_mult[i_level]
            for _ in range(self.num_res_blocks):
                block.append(ResnetBlock(in_channels=block_in, out_channels=block_out, temb_channels=self.temb_ch,
                                         dropout=dropout))
                block_in = block_out
            down = nn.Module()
            down.block = block
            down.attn = attn
            self.down.append(down)
        self.norm_out = Normalize(block_out)
        self.conv_out = zero_module(torch.nn.Conv2d(block_out, in_channels, kernel_size=3, stride=1, padding=1))
        self._setup()

    def _forward_one(self, x):
        """x [3, F, H, W] -> sigmoid(net(x) + x).  The skip and the sigmoid run in promote(x.dtype, T), like the reference's
        `h + x` under autocast (trajectory_module.py:194-196: the conv stack computes in T, the fp32 coordinates join at full
        precision and the result is fp32); a caller that hands in T-typed coordinates (whole-model cast, infer.py) gets T back."""
        C, F, H, W = x.shape
        T, dev = self.dtype, self.device
        odt = torch.promote_types(x.dtype, T) if x.dtype.is_floating_point else T
        xs = x.to(device=dev, dtype=odt).contiguous()
        h = ops.ncthw_to_cl(xs, T, Cp=CIN_PAD).view(F * H * W, CIN_PAD)
        h, st = self._conv_first(h, self.conv_in, F, H, W)
        for blk in self.down[0].block:
            h, st = self._resnet(h, blk, F, H, W, st)
        h, cop, _ = self._norm_conv(h, self.norm_out, self.conv_out, F, H, W, self.ch, stats_in=st)
        return ops.cl_to_ncthw(h, odt, C=C, T=F, H=H, W=W, pixel_stride=cop, act=2, aux=xs)

    def forward(self, x):
        """x [B, 3, F, H, W] -> sigmoid(net(x) + x), same shape (reference :177-196)."""
        return self._run(x)


class VAEDecoderadaptor(_AdaptorBase):
    def __init__(self, *, ch=128, out_ch=3, ch_mult=(1,), num_res_blocks=1, attn_resolutions=[], dropout=0.0,
                 resamp_with_conv=True, in_channels=3, resolution=256, z_channels=4, give_pre_end=False, tanh_out=False,
                 use_linear_attn=False, attn_type="vanilla", **ignorekwargs):
        super().__init__()
        self.ch, self.temb_ch = ch, 0
        self.num_resolutions = len(ch_mult)
        assert self.num_resolutions == 1
        self.num_res_blocks, self.resolution, self.in_channels = num_res_blocks, resolution, in_channels
        self.give_pre_end, self.tanh_out, self.out_ch = give_pre_end, tanh_out, out_ch
        if tanh_out:
            raise NotImplementedError("tanh_out is not used by MoRe4D (trajectory_module.py:203)")
        self.conv_in = torch.nn.Conv2d(in_channels, self.ch, kernel_size=3, stride=1, padding=1)
        block_in = ch * ch_mult[self.num_resolutions - 1]
        self.up = nn.ModuleList()
        for i_level in reversed(range(self.num_resolutions)):
            block = nn.ModuleList()
            attn = nn.ModuleList()
            block_out = ch * ch_mult[i_level]
            for _ in range(self.num_res_blocks + 1):
                block.append(ResnetBlock(in_channels=block_in, out_channels=block_out, temb_channels=self.temb_ch,
                                         dropout=dropout))
                block_in = block_out
            up = nn.Module()
            up.block = block
            up.attn = attn
            self.up.insert(0, up)
        self.final_activation = None
        self.norm_out = Normalize(block_in)
        self.conv_out = torch.nn.Conv2d(block_in, out_ch, kernel_size=3, stride=1, padding=1)
        self._setup()

    def _forward_one(self, z):
        C, F, H, W = z.shape
        T, dev = self.dtype, self.device
        zb = z.to(device=dev, dtype=T).contiguous()
        h = ops.ncthw_to_cl(zb, T, Cp=CIN_PAD).view(F * H * W, CIN_PAD)
        h, st = self._conv_first(h, self.conv_in, F, H, W)
        for blk in self.up[0].block:
            h, st = self._resnet(h, blk, F, H, W, st)
        h, cop, _ = self._norm_conv(h, self.norm_out, self.conv_out, F, H, W, self.ch, stats_in=st)
        return ops.cl_to_ncthw(h, T, C=self.out_ch, T=F, H=H, W=W, pixel_stride=cop)

    def forward(self, z):
        """z [B, 3, F, H, W] -> [B, out_ch, F, H, W] (reference :260-279)."""
        return self._run(z)
