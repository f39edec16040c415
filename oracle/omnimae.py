"""TEST INFRASTRUCTURE — CPU restatement (plain fp32 torch) of the OmniMAE ViT-B patch-feature path the reference calls:
`omnimae_extractor.trunk.forward_patch_features` (MoRe4D/models/omnivision/models/vision_transformer.py:688-703), built
by `vit_base_mae_pretraining` (MoRe4D/models/omnimae.py:77-145).  Only tests/, __graft_entry__.smoke() and bench.py's
cpu_baseline leg may import this module; the product never does.

Pinned: tests/golden/omnimae.npz holds outputs of the reference's own VisionTransformer (imported here with stand-ins for the
absent `timm` / `hydra` names only; random weights from tests/golden/weights.py — no checkpoint exists offline), see
tests/golden/make_golden.py:make_omnimae.
"""
import math

import torch
import torch.nn.functional as F

IMAGENET_MEAN = (0.485, 0.456, 0.406)
IMAGENET_STD = (0.229, 0.224, 0.225)


def normalize(first_frame):
    """torchvision Normalize as used at wan_transformer4d.py:1130-1133."""
    mean = torch.tensor(IMAGENET_MEAN).view(1, 3, 1, 1)
    std = torch.tensor(IMAGENET_STD).view(1, 3, 1, 1)
    return (first_frame - mean) / std


def sinusoid_table(n_position, d_hid):
    """get_sinusoid_encoding_table (vision_transformer.py:31-46): float64 angles, sin on even / cos on odd channels."""
    tab = torch.empty(n_position, d_hid, dtype=torch.float64)
    for j in range(d_hid):
        div = math.pow(10000, 2 * (j // 2) / d_hid)
        col = torch.arange(n_position, dtype=torch.float64) / div
        tab[:, j] = torch.sin(col) if j % 2 == 0 else torch.cos(col)
    return tab.float().unsqueeze(0)


def forward_patch_features(sd, x, prefix="trunk.", num_heads=12, depth=12, eps=1e-6):
    """x [B,3,H,W] (normalised) -> (patch features [B,196,768], token-0 features [B,768]).

    :691 bilinear resize to 224; prepare_tokens (:638-668): PadIm2Video repeats the frame twice (:58-72), Conv3d
    (2,16,16) stride (2,16,16) (omnimae.py:104-110), flatten to tokens, add the frame-0 slice of the fixed 8x14x14 position
    table (:843-877; 196 == 14*14 so no interpolation); 12 blocks x = x + attn(norm1 x); x = x + mlp(norm2 x) (:202-205)
    with softmax((q k^T) / sqrt(64)) v (:123-143) and an erf-GELU MLP (:75-98); final LayerNorm (:701); no class token
    (first_patch_idx = 0), so the second output is the feature of patch 0 (:703)."""
    p = prefix
    x = F.interpolate(x, size=(224, 224), mode="bilinear", align_corners=False)
    x = x.unsqueeze(2).repeat(1, 1, 2, 1, 1)
    x = F.conv3d(x, sd[p + "patch_embed.proj.1.weight"], sd[p + "patch_embed.proj.1.bias"], stride=(2, 16, 16))
    x = x.flatten(2).transpose(1, 2)                       # [B, 196, 768]
    B, N, C = x.shape
    x = x + sd[p + "pos_embed"][:, :N]
    hd = C // num_heads
    for i in range(depth):
        b = f"{p}blocks.{i}."
        h = F.layer_norm(x, (C,), sd[b + "norm1.weight"], sd[b + "norm1.bias"], eps)
        qkv = F.linear(h, sd[b + "attn.qkv.weight"], sd[b + "attn.qkv.bias"]).reshape(B, N, 3, num_heads, hd)
        q, k, v = qkv.permute(2, 0, 3, 1, 4)
        a = ((q @ k.transpose(-2, -1)) * hd ** -0.5).softmax(dim=-1)
        o = (a @ v).transpose(1, 2).reshape(B, N, C)
        x = x + F.linear(o, sd[b + "attn.proj.weight"], sd[b + "attn.proj.bias"])
        h = F.layer_norm(x, (C,), sd[b + "norm2.weight"], sd[b + "norm2.bias"], eps)
        h = F.gelu(F.linear(h, sd[b + "mlp.fc1.weight"], sd[b + "mlp.fc1.bias"]))
        x = x + F.linear(h, sd[b + "mlp.fc2.weight"], sd[b + "mlp.fc2.bias"])
    x = F.layer_norm(x, (C,), sd[p + "norm.weight"], sd[p + "norm.bias"], eps)
    return x, x[:, 0]
