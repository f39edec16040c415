"""Oracle: Wan2.1 causal 3-D VAE (the "Motion-Sensitive VAE" core) and the two trajectory adaptors, as a
functional fp32 restatement with an EXPLICIT streaming state.  TEST INFRASTRUCTURE — see oracle/__init__.py.

Reference: /root/reference/MoRe4D/models/wan_vae.py (CausalConv3d :21-40, RMS_norm :43-58, Resample :70-164,
ResidualBlock :190-224, AttentionBlock :227-266, Encoder3d :269-370, Decoder3d :373-476, encode :520-547,
decode :678-703, AutoencoderKLWan :748-847) and MoRe4D/models/trajectory_module.py (ResnetBlock :63-122,
VAEEncoderadaptor :125-196, VAEDecoderadaptor :200-279).

Streaming model used here (equivalent to the reference's feat_cache list, restated): every causal conv owns a
`Tail` = the last two input frames it has seen; a conv over a chunk x reads cat(tail, x) with the missing
leading frames zero.  The reference's special cases fall out of three rules:
  * a conv that has seen only ONE frame so far has tail = [that frame] and pads one zero frame in front
    (CausalConv3d.forward with a 1-frame cache, :33-38);
  * `upsample3d.time_conv` is skipped on the first chunk ('Rep' sentinel, :107-112) and, on the second chunk,
    starts from an all-zero tail — it never sees the first chunk's frame (:124-132);
  * `downsample3d.time_conv` is skipped on the first chunk, whose (spatially down-sampled) frame becomes the
    one-frame tail (:147-152); later chunks run the stride-2 conv over cat(tail[-1:], x) (:160-161).
"""
import math

import torch
import torch.nn.functional as F

MEAN = [-0.7571, -0.7089, -0.9113, 0.1075, -0.1745, 0.9653, -0.1517, 1.5508,
        0.4134, -0.0715, 0.5517, -0.3632, -0.1922, -0.9497, 0.2503, -0.2921]
STD = [2.8184, 1.4541, 2.3275, 2.6558, 1.2196, 1.7708, 2.6052, 2.0743,
       3.2687, 2.1526, 2.8652, 1.5579, 1.6382, 1.1253, 2.8251, 1.9160]


class Stream:
    """Per-encode / per-decode streaming state: key -> tensor of the last <=2 input frames (or a sentinel)."""

    def __init__(self):
        self.tails = {}


def causal_conv3d(sd, name, x, tail=None, stride=(1, 1, 1)):
    """Conv3d over cat(tail, x) with zero frames in front so that 2*pad_t frames precede x (:21-40)."""
    w, b = sd[name + ".weight"], sd.get(name + ".bias")
    kt, kh, kw = w.shape[2:]
    need = kt - 1  # = 2 * padding[0] for the k=3,p=1 convs; the strided time_conv passes its own context
    if tail is not None and need > 0:
        x = torch.cat([tail, x], dim=2)
        need -= tail.shape[2]
    x = F.pad(x, (kw // 2, kw // 2, kh // 2, kh // 2, max(need, 0), 0))
    return F.conv3d(x, w, b, stride=stride)


def _stream_conv(sd, name, x, st, key):
    """A k=3 causal conv inside a streaming pass: use and update this conv's tail (:209-221)."""
    if st is None:
        return causal_conv3d(sd, name, x)
    old = st.tails.get(key)
    new = x[:, :, -2:]
    if new.shape[2] < 2 and old is not None:
        new = torch.cat([old[:, :, -1:], new], dim=2)
    y = causal_conv3d(sd, name, x, old)
    st.tails[key] = new
    return y


def rms_norm(sd, name, x):
    """F.normalize over channels * sqrt(C) * gamma (:55-58); x [B,C,...]."""
    g = sd[name + ".gamma"]
    c = x.shape[1]
    n = x.norm(dim=1, keepdim=True).clamp_min(1e-12)
    return x / n * math.sqrt(c) * g.view(1, c, *([1] * (x.dim() - 2)))


def residual_block(sd, p, x, st):
    """ResidualBlock.forward (:206-224)."""
    h = causal_conv3d(sd, p + ".shortcut", x) if (p + ".shortcut.weight") in sd else x
    y = F.silu(rms_norm(sd, p + ".residual.0", x))
    y = _stream_conv(sd, p + ".residual.2", y, st, p + ".residual.2")
    y = F.silu(rms_norm(sd, p + ".residual.3", y))
    y = _stream_conv(sd, p + ".residual.6", y, st, p + ".residual.6")
    return y + h


def attention_block(sd, p, x):
    """Per-frame single-head attention over h*w tokens (:244-266)."""
    b, c, t, h, w = x.shape
    u = x.permute(0, 2, 1, 3, 4).reshape(b * t, c, h, w)
    u = rms_norm(sd, p + ".norm", u)
    qkv = F.conv2d(u, sd[p + ".to_qkv.weight"], sd[p + ".to_qkv.bias"]).reshape(b * t, 3, c, h * w)
    q, k, v = (qkv[:, i].transpose(1, 2) for i in range(3))          # [bt, hw, c]
    a = torch.softmax(q @ k.transpose(1, 2) / math.sqrt(c), dim=-1) @ v
    a = a.transpose(1, 2).reshape(b * t, c, h, w)
    a = F.conv2d(a, sd[p + ".proj.weight"], sd[p + ".proj.bias"])
    return a.reshape(b, t, c, h, w).permute(0, 2, 1, 3, 4) + x


def _per_frame(fn, x):
    b, c, t, h, w = x.shape
    y = fn(x.permute(0, 2, 1, 3, 4).reshape(b * t, c, h, w))
    return y.reshape(b, t, *y.shape[1:]).permute(0, 2, 1, 3, 4)


def resample(sd, p, mode, x, st):
    """Resample.forward (:105-164)."""
    if mode == "upsample3d" and st is not None:
        key = p + ".time_conv"
        state = st.tails.get(key)
        if state is None:
            st.tails[key] = "Rep"           # first chunk: no temporal up-sampling
        else:
            new = x[:, :, -2:]
            if new.shape[2] < 2:
                prev = torch.zeros_like(new) if isinstance(state, str) else state[:, :, -1:]
                new = torch.cat([prev, new], dim=2)
            y = causal_conv3d(sd, key, x, None if isinstance(state, str) else state)
            st.tails[key] = new
            b, c2, t, h, w = y.shape
            c = c2 // 2
            y = y.reshape(b, 2, c, t, h, w)
            x = torch.stack((y[:, 0], y[:, 1]), 3).reshape(b, c, t * 2, h, w)   # interleave (:138-141)
    if mode in ("upsample2d", "upsample3d"):
        def up(u):
            u = F.interpolate(u.float(), scale_factor=(2.0, 2.0), mode="nearest-exact").type_as(u)
            return F.conv2d(u, sd[p + ".resample.1.weight"], sd[p + ".resample.1.bias"], padding=1)
        x = _per_frame(up, x)
    elif mode in ("downsample2d", "downsample3d"):
        def down(u):
            return F.conv2d(F.pad(u, (0, 1, 0, 1)), sd[p + ".resample.1.weight"], sd[p + ".resample.1.bias"], stride=2)
        x = _per_frame(down, x)
    if mode == "downsample3d" and st is not None:
        key = p + ".time_conv"
        state = st.tails.get(key)
        if state is None:
            st.tails[key] = x.clone()
        else:
            new = x[:, :, -1:].clone()
            x = F.conv3d(torch.cat([state[:, :, -1:], x], 2), sd[key + ".weight"], sd[key + ".bias"], stride=(2, 1, 1))
            st.tails[key] = new
    return x


ENC_LAYOUT = [("res", 96, 96), ("res", 96, 96), ("down", "downsample2d"),
              ("res", 96, 192), ("res", 192, 192), ("down", "downsample3d"),
              ("res", 192, 384), ("res", 384, 384), ("down", "downsample3d"),
              ("res", 384, 384), ("res", 384, 384)]
DEC_LAYOUT = [("res", 384, 384)] * 3 + [("up", "upsample3d")] + [("res", 192, 384), ("res", 384, 384), ("res", 384, 384),
              ("up", "upsample3d")] + [("res", 192, 192)] * 3 + [("up", "upsample2d")] + [("res", 96, 96)] * 3


def layout_from_sd(sd, prefix, n):
    """Recover (kind, mode) per index of downsamples/upsamples from the key names (works for small test configs)."""
    out = []
    for i in range(n):
        p = f"{prefix}.{i}"
        if (p + ".residual.2.weight") in sd:
            out.append(("res",))
        else:
            out.append(("resample",))
    return out


def encoder3d(sd, p, x, st, modes):
    """Encoder3d.forward (:322-370); modes: resample mode per Resample module, in order."""
    x = _stream_conv(sd, p + ".conv1", x, st, p + ".conv1")
    i = 0
    mi = 0
    while (f"{p}.downsamples.{i}.residual.2.weight") in sd or (f"{p}.downsamples.{i}.resample.1.weight") in sd:
        q = f"{p}.downsamples.{i}"
        if (q + ".residual.2.weight") in sd:
            x = residual_block(sd, q, x, st)
        else:
            x = resample(sd, q, modes[mi], x, st)
            mi += 1
        i += 1
    x = residual_block(sd, p + ".middle.0", x, st)
    x = attention_block(sd, p + ".middle.1", x)
    x = residual_block(sd, p + ".middle.2", x, st)
    x = F.silu(rms_norm(sd, p + ".head.0", x))
    return _stream_conv(sd, p + ".head.2", x, st, p + ".head.2")


def decoder3d(sd, p, x, st, modes):
    """Decoder3d.forward (:427-476)."""
    x = _stream_conv(sd, p + ".conv1", x, st, p + ".conv1")
    x = residual_block(sd, p + ".middle.0", x, st)
    x = attention_block(sd, p + ".middle.1", x)
    x = residual_block(sd, p + ".middle.2", x, st)
    i = 0
    mi = 0
    while (f"{p}.upsamples.{i}.residual.2.weight") in sd or (f"{p}.upsamples.{i}.resample.1.weight") in sd:
        q = f"{p}.upsamples.{i}"
        if (q + ".residual.2.weight") in sd:
            x = residual_block(sd, q, x, st)
        else:
            x = resample(sd, q, modes[mi], x, st)
            mi += 1
        i += 1
    x = F.silu(rms_norm(sd, p + ".head.0", x))
    return _stream_conv(sd, p + ".head.2", x, st, p + ".head.2")


ENC_MODES = ["downsample2d", "downsample3d", "downsample3d"]   # temperal_downsample [F,T,T] (:727-745)
DEC_MODES = ["upsample3d", "upsample3d", "upsample2d"]


def vae_encode(sd, x, z_dim=16):
    """AutoencoderKLWan._encode for one batch (:520-547, :775-781): x [B,3,T,H,W] -> [B, 2*z_dim, T', H/8, W/8]
    = (mu normalised by mean/std | logvar)."""
    st = Stream()
    t = x.shape[2]
    outs = []
    for i in range(1 + (t - 1) // 4):
        chunk = x[:, :, :1] if i == 0 else x[:, :, 1 + 4 * (i - 1):1 + 4 * i]
        outs.append(encoder3d(sd, "model.encoder", chunk, st, ENC_MODES))
    out = torch.cat(outs, 2)
    mu, logvar = causal_conv3d(sd, "model.conv1", out).chunk(2, dim=1)
    mean = torch.tensor(MEAN[:z_dim], dtype=x.dtype).view(1, z_dim, 1, 1, 1)
    inv_std = (1.0 / torch.tensor(STD[:z_dim], dtype=x.dtype)).view(1, z_dim, 1, 1, 1)
    return torch.cat([(mu - mean) * inv_std, logvar], dim=1)


def vae_decode(sd, z, z_dim=16):
    """AutoencoderKLWan._decode (:678-703, :825-832): z [B,16,T',h,w] -> clamp(-1,1) video [B,3,T,8h,8w]."""
    st = Stream()
    mean = torch.tensor(MEAN[:z_dim], dtype=z.dtype).view(1, z_dim, 1, 1, 1)
    inv_std = (1.0 / torch.tensor(STD[:z_dim], dtype=z.dtype)).view(1, z_dim, 1, 1, 1)
    z = z / inv_std + mean
    x = causal_conv3d(sd, "model.conv2", z)
    outs = [decoder3d(sd, "model.decoder", x[:, :, i:i + 1], st, DEC_MODES) for i in range(x.shape[2])]
    return torch.cat(outs, 2).clamp(-1, 1)


def gaussian_sample(params, eps=None):
    """diffusers DiagonalGaussianDistribution (third-party, restated from its published definition; unpinned):
    mean, logvar = chunk(params, 2, dim=1); logvar clamped to [-30, 20]; sample = mean + exp(0.5*logvar)*eps."""
    mean, logvar = params.chunk(2, dim=1)
    if eps is None:
        return mean
    return mean + torch.exp(0.5 * logvar.clamp(-30.0, 20.0)) * eps


# --------------------------------------------------------------------------- trajectory adaptors

def group_norm_swish(sd, name, x, groups=32, eps=1e-6):
    y = F.group_norm(x, groups, sd[name + ".weight"], sd[name + ".bias"], eps)
    return y * torch.sigmoid(y)


def resnet_block(sd, p, x):
    """trajectory_module.ResnetBlock.forward with temb=None, in==out channels (:101-122)."""
    h = F.conv2d(group_norm_swish(sd, p + ".norm1", x), sd[p + ".conv1.weight"], sd[p + ".conv1.bias"], padding=1)
    h = F.conv2d(group_norm_swish(sd, p + ".norm2", h), sd[p + ".conv2.weight"], sd[p + ".conv2.bias"], padding=1)
    return x + h


def encoder_adaptor(sd, x):
    """VAEEncoderadaptor.forward (:177-196): x [B,3,F,H,W] -> sigmoid(h + x), same shape."""
    B, C, Fr, H, W = x.shape
    u = x.permute(0, 2, 1, 3, 4).reshape(B * Fr, C, H, W)
    h = F.conv2d(u, sd["conv_in.weight"], sd["conv_in.bias"], padding=1)
    h = resnet_block(sd, "down.0.block.0", h)
    h = F.conv2d(group_norm_swish(sd, "norm_out", h), sd["conv_out.weight"], sd["conv_out.bias"], padding=1)
    h = torch.sigmoid(h + u)
    return h.view(B, Fr, C, H, W).permute(0, 2, 1, 3, 4)


def decoder_adaptor(sd, z):
    """VAEDecoderadaptor.forward (:260-279): conv_in, 2 ResnetBlocks, GN-swish, conv_out (no activation)."""
    B, C, Fr, H, W = z.shape
    u = z.permute(0, 2, 1, 3, 4).reshape(B * Fr, C, H, W)
    h = F.conv2d(u, sd["conv_in.weight"], sd["conv_in.bias"], padding=1)
    h = resnet_block(sd, "up.0.block.0", h)
    h = resnet_block(sd, "up.0.block.1", h)
    h = F.conv2d(group_norm_swish(sd, "norm_out", h), sd["conv_out.weight"], sd["conv_out.bias"], padding=1)
    return h.view(B, Fr, -1, H, W).permute(0, 2, 1, 3, 4)
