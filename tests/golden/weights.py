"""Deterministic synthetic weights keyed by state-dict name (no checkpoints exist offline).

Large fixtures (14B-width block: 1.6 GB, full VAE: 0.5 GB) cannot be committed, so both the
fixture generator (which loads them into the REFERENCE modules) and the tests / bench (which
load them into ours) regenerate weights from this recipe: one CPU generator per key, seeded
by (seed, crc32(key)), value scale chosen from the key's role.  CPU torch.randn with a seeded
generator is bit-reproducible for a given torch build (same image here and on the GPU box).
"""
import math
import zlib

import torch


def _scale_for(key, shape):
    if key.endswith("modulation"):
        return ("randn", 1.0 / math.sqrt(shape[-1]))
    if key.endswith("gamma") or "norm" in key.rsplit(".", 2)[-2] and key.endswith("weight") and len(shape) == 1:
        return ("one_plus", 0.1)
    if key.endswith(".gate"):
        return ("randn", 0.05)
    if key.endswith("bias"):
        return ("randn", 0.02)
    if len(shape) >= 2:
        fan_in = 1
        for s in shape[1:]:
            fan_in *= s
        return ("randn", 1.0 / math.sqrt(fan_in))
    return ("randn", 0.02)


def make_tensor(key, shape, seed, dtype=torch.float32):
    g = torch.Generator().manual_seed((int(seed) * 1000003 + zlib.crc32(key.encode())) % (2 ** 63))
    kind, s = _scale_for(key, tuple(shape))
    t = torch.randn(tuple(shape), generator=g, dtype=torch.float32) * s
    if kind == "one_plus":
        t = t + 1.0
    return t.to(dtype)


def fill(shapes, seed, dtype=torch.float32):
    """shapes: {key: shape} -> {key: tensor}."""
    return {k: make_tensor(k, tuple(v), seed, dtype) for k, v in shapes.items()}


def block_shapes(dim, ffn, guidance=False, prefix="blocks.0."):
    s = {"modulation": (1, 6, dim)}
    for a in ("self_attn", "cross_attn"):
        for n in ("q", "k", "v", "o"):
            s[f"{a}.{n}.weight"] = (dim, dim)
            s[f"{a}.{n}.bias"] = (dim,)
        s[f"{a}.norm_q.weight"] = (dim,)
        s[f"{a}.norm_k.weight"] = (dim,)
    for n in ("k_img", "v_img"):
        s[f"cross_attn.{n}.weight"] = (dim, dim)
        s[f"cross_attn.{n}.bias"] = (dim,)
    s["cross_attn.norm_k_img.weight"] = (dim,)
    s["norm3.weight"] = (dim,)
    s["norm3.bias"] = (dim,)
    s["ffn.0.weight"] = (ffn, dim)
    s["ffn.0.bias"] = (ffn,)
    s["ffn.2.weight"] = (dim, ffn)
    s["ffn.2.bias"] = (dim,)
    if guidance:
        for g in ("spatial_guidance_self", "spatial_guidance_ffn"):
            s[f"{g}.spatial_guide.1.weight"] = (2 * dim, 768)
            s[f"{g}.spatial_guide.1.bias"] = (2 * dim,)
            s[f"{g}.gate"] = (dim,)
    return {prefix + k: v for k, v in s.items()}


def block_weights_14b(seed=0, dtype=torch.float32):
    return fill(block_shapes(5120, 13824), seed, dtype)


def randn_named(key, shape, seed, scale=1.0, dtype=torch.float32):
    """Seeded N(0, scale^2) tensor for fixture INPUTS too large to commit."""
    g = torch.Generator().manual_seed((int(seed) * 1000003 + zlib.crc32(key.encode())) % (2 ** 63))
    return (torch.randn(tuple(shape), generator=g, dtype=torch.float32) * scale).to(dtype)


# ---- device-agnostic recipe (round 5) ----
# torch.randn is reproducible per device only: CPU (mt19937) and GPU (Philox) streams differ, so `fill` has to run on the host — 3-4 s per
# 14B-width block, minutes for 40 of them.  `fill_hash` derives every value from integer arithmetic that is exact on both devices (a 32-bit
# mixing hash of the element index, all intermediate products < 2^63), maps the top 24 bits to an exactly representable fp32 in [-1, 1) and
# applies ONE rounding multiply: the reference (CPU, fixture generation) and the GPU test build bit-identical weights, each on its own device.
def _hash_uniform(numel, key_seed, device):
    i = torch.arange(numel, dtype=torch.int64, device=device)
    h = (i + int(key_seed)) & 0xFFFFFFFF
    h = ((h ^ (h >> 16)) * 0x45D9F3B) & 0xFFFFFFFF
    h = ((h ^ (h >> 16)) * 0x45D9F3B) & 0xFFFFFFFF
    h = h ^ (h >> 16)
    return (h >> 8).to(torch.float32) * (2.0 ** -23) - 1.0          # 24 bits -> [-1, 1), exact


def make_tensor_hash(key, shape, seed, device="cpu", dtype=torch.float32):
    kind, s = _scale_for(key, tuple(shape))
    n = 1
    for d in shape:
        n *= d
    ks = (int(seed) * 1000003 + zlib.crc32(key.encode())) & 0xFFFFFFFF
    t = _hash_uniform(n, ks, device).reshape(tuple(shape)) * float(torch.tensor(math.sqrt(3.0) * s, dtype=torch.float32))   # unit variance x s
    if kind == "one_plus":
        t = t + 1.0
    return t.to(dtype)


def fill_hash(shapes, seed, device="cpu", dtype=torch.float32):
    return {k: make_tensor_hash(k, tuple(v), seed, device, dtype) for k, v in shapes.items()}
