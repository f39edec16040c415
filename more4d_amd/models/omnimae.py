"""OmniMAE ViT-B/16 patch-feature extractor (Motion-Perception front end) — MI355X-native host side.

Mirrors what the reference uses of `vit_base_mae_pretraining()` (MoRe4D/models/omnimae.py:77-145): only
`trunk.forward_patch_features` (omnivision/models/vision_transformer.py:688-703) is ever called, under `no_grad`
(wan_transformer4d.py:1136-1142), so the MAE decoder / head / mask token are not built; their checkpoint keys are skipped
at load (`strict=False`), the encoder keys keep the reference names (`trunk.patch_embed.proj.1.*`, `trunk.blocks.N.*`,
`trunk.norm.*`, `trunk.pos_embed`).

Forward = bilinear resize to 224x224 -> PadIm2Video (the frame repeated twice, vision_transformer.py:58-72) -> Conv3d
(2,16,16)/(2,16,16) patch embedding as one GEMM -> + sinusoidal position table of frame 0 (:843-877: 196 of the 8x14x14
positions, no interpolation) -> 12 pre-LN blocks (LN eps 1e-6, fused qkv Linear, 12 heads x 64, erf-GELU MLP x4) -> final
LN.  Returns (patch features [B,196,768], features of token 0 [B,768]) — the reference's "cls" output is patch 0 because
this trunk has no class token (first_patch_idx = 0, :703).  Every op is an ABI call (ops.*): LayerNorm, the GEMMs with
fused bias / GELU / residual epilogues and the flash-attention kernel (head_dim 64, keys masked past 196).
"""
import torch
import torch.nn as nn

from .. import ops
from ..ops import EPI_GELU_ERF, EPI_RESID_GATE, KV

IMAGENET_MEAN = (0.485, 0.456, 0.406)        # wan_transformer4d.py:1132 (torchvision Normalize, restated)
IMAGENET_STD = (0.229, 0.224, 0.225)


def sinusoid_table(n_position, d_hid):
    """float32 [1, n_position, d_hid]: sin on even / cos on odd channels of pos / 10000^(2*(j//2)/d) (float64 math, like the
    numpy original, vision_transformer.py:31-46)."""
    pos = torch.arange(n_position, dtype=torch.float64).unsqueeze(1)
    j = torch.arange(d_hid, dtype=torch.float64)
    ang = pos / torch.pow(torch.tensor(10000.0, dtype=torch.float64), 2 * torch.div(j, 2, rounding_mode="floor") / d_hid)
    tab = torch.empty_like(ang)
    tab[:, 0::2] = torch.sin(ang[:, 0::2])
    tab[:, 1::2] = torch.cos(ang[:, 1::2])
    return tab.float().unsqueeze(0)


class _Attention(nn.Module):
    def __init__(self, dim, num_heads):
        super().__init__()
        self.num_heads = num_heads
        self.qkv = nn.Linear(dim, dim * 3, bias=True)
        self.proj = nn.Linear(dim, dim)


class _Mlp(nn.Module):
    def __init__(self, dim, hidden):
        super().__init__()
        self.fc1 = nn.Linear(dim, hidden)
        self.fc2 = nn.Linear(hidden, dim)


class _Block(nn.Module):
    def __init__(self, dim, num_heads, mlp_ratio, eps):
        super().__init__()
        self.norm1 = nn.LayerNorm(dim, eps=eps)
        self.attn = _Attention(dim, num_heads)
        self.norm2 = nn.LayerNorm(dim, eps=eps)
        self.mlp = _Mlp(dim, int(dim * mlp_ratio))


class _PatchEmbed(nn.Module):
    def __init__(self, dim):
        super().__init__()
        # index 1 of the reference's stem (index 0 is the parameter-free PadIm2Video)
        self.proj = nn.ModuleDict({"1": nn.Conv3d(3, dim, kernel_size=(2, 16, 16), stride=(2, 16, 16))})


class VisionTransformerTrunk(nn.Module):
    def __init__(self, embed_dim=768, depth=12, num_heads=12, mlp_ratio=4, eps=1e-6):
        super().__init__()
        self.embed_dim, self.num_heads, self.eps = embed_dim, num_heads, eps
        self.patch_embed = _PatchEmbed(embed_dim)
        # img_size [3,16,224,224] / patch (2,16,16) -> 8*14*14 positions; fixed table, saved in checkpoints (:540-552)
        self.pos_embed = nn.Parameter(sinusoid_table(8 * 14 * 14, embed_dim), requires_grad=False)
        self.blocks = nn.ModuleList([_Block(embed_dim, num_heads, mlp_ratio, eps) for _ in range(depth)])
        self.norm = nn.LayerNorm(embed_dim, eps=eps)
        self.first_patch_idx = 0
        self._cache = {}

    def _f32(self, p):
        from .wan_transformer4d import _f32
        return _f32(p, self._cache)

    def _packed_stem(self, T):
        conv = self.patch_embed.proj["1"]
        key = (conv.weight._version, conv.weight.data_ptr(), T)
        hit = self._cache.get("stem")
        if hit is None or hit[0] != key:
            # [768, 3, 2, 16, 16] -> [768, (t, kh, kw, c)]: rows of the patch matrix are (kh, kw, c) of one 16x16 patch,
            # listed twice (the repeated frame)
            w = conv.weight.detach().to(T).permute(0, 2, 3, 4, 1).contiguous().view(conv.weight.shape[0], -1)
            hit = (key, w, conv.bias.detach().to(T).contiguous())
            self._cache["stem"] = hit
        return hit[1], hit[2]

    @torch.no_grad()
    def forward_patch_features(self, x, use_checkpoint=False, *, normalize=False):
        """x [B, 3, H, W] -> (float32 [B,196,768], float32 [B,768]).  The reference hands in the ImageNet-normalised
        frame (wan_transformer4d.py:1130-1133); normalize=True takes the raw [0,1] frame and fuses (x - mean) / std into
        the layout kernel."""
        dev = self.norm.weight.device
        T = self.norm.weight.dtype
        D, n = self.embed_dim, self.num_heads
        d = D // n
        B, Cin, H, W = x.shape
        if Cin != 3:
            raise ValueError("forward_patch_features: expected [B, 3, H, W]")
        # NCHW -> channels-last float32, bilinear resize (align_corners=False, :691)
        aff = {}
        if normalize:
            std = torch.tensor(IMAGENET_STD, device=dev, dtype=torch.float64)
            mean = torch.tensor(IMAGENET_MEAN, device=dev, dtype=torch.float64)
            aff = dict(ch_scale=(1.0 / std).float(), ch_shift=(-mean / std).float())
        cl = ops.ncthw_to_cl(x.to(dev).float().permute(1, 0, 2, 3).contiguous(), torch.float32, **aff)   # [B, H, W, 3]
        img = ops.bilinear_cl(cl, (224, 224)) if (H, W) != (224, 224) else cl
        # patches: [B,14,16,14,16,3] -> [B*196, 16*16*3], duplicated for the two (identical) frames: pure data movement
        tok = img.view(B, 14, 16, 14, 16, 3).permute(0, 1, 3, 2, 4, 5).reshape(B * 196, 768)
        tok = torch.cat([tok, tok], dim=1).to(T).contiguous()
        wp, bp = self._packed_stem(T)
        Lp = 200                                       # 196 tokens padded to a multiple of 8; pad keys are masked
        R = B * Lp
        xres = torch.zeros((B, Lp, D), device=dev, dtype=torch.float32)
        xres[:, :196] = self._f32(self.pos_embed)[0, :196]
        for b in range(B):                             # x = patch_embed + pos_embed (:661): residual epilogue onto the table
            ops.gemm_bt(tok[b * 196:(b + 1) * 196], wp, bp, out=xres[b, :196], epilogue=EPI_RESID_GATE, gate=None,
                        rows_per_sample=196)
        xres2 = xres.view(R, D)
        for blk in self.blocks:
            xn = ops.ln_modulate(xres, T, ln_w=self._f32(blk.norm1.weight), ln_b=self._f32(blk.norm1.bias), eps=self.eps)
            xn = xn.view(R, D)
            wqkv, bqkv = blk.attn.qkv.weight, blk.attn.qkv.bias
            qk = ops.gemm_bt(xn, wqkv[:2 * D], bqkv[:2 * D])                                  # [R, 2D] = (q | k)
            vt = ops.gemm_bt(wqkv[2 * D:], xn, bqkv[2 * D:], bias_on_m=True)                  # V^T [D, R]
            o = ops.attention(qk[:, :D], [KV(qk[:, D:], vt, Lp * 2 * D, 2 * D, Lp, R, 196)], B=B, Lq=Lp, heads=n,
                              head_dim=d, q_bs=Lp * 2 * D, q_ls=2 * D)
            ops.gemm_bt(o.view(R, D), blk.attn.proj.weight, blk.attn.proj.bias, out=xres2, epilogue=EPI_RESID_GATE,
                        gate=None, rows_per_sample=Lp)
            xn = ops.ln_modulate(xres, T, ln_w=self._f32(blk.norm2.weight), ln_b=self._f32(blk.norm2.bias), eps=self.eps)
            h = ops.gemm_bt(xn.view(R, D), blk.mlp.fc1.weight, blk.mlp.fc1.bias, epilogue=EPI_GELU_ERF)
            ops.gemm_bt(h, blk.mlp.fc2.weight, blk.mlp.fc2.bias, out=xres2, epilogue=EPI_RESID_GATE, gate=None,
                        rows_per_sample=Lp)
        y = ops.ln_modulate(xres, torch.float32, ln_w=self._f32(self.norm.weight), ln_b=self._f32(self.norm.bias),
                            eps=self.eps)
        feats = y[:, :196].contiguous()
        return feats, feats[:, self.first_patch_idx].contiguous()


class OmniMAE(nn.Module):
    """Container with the reference attribute layout (`omnimae_extractor.trunk.…`)."""

    def __init__(self):
        super().__init__()
        self.trunk = VisionTransformerTrunk()


def vit_base_mae_pretraining(pretrained=False):
    """Reference factory name (omnimae.py:77).  `pretrained=True` needs the checkpoint file the reference hard-codes; load
    it with `load_state_dict(torch.load(path), strict=False)` — the decoder / head keys are ignored."""
    if pretrained:
        raise RuntimeError("no checkpoint is bundled; build with pretrained=False and load the OmniMAE ViT-B state dict")
    return OmniMAE()


