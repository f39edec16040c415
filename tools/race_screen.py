#!/usr/bin/env python
"""Race screen of the production kernels at the bench shapes: the same launch repeated N times must give bit-identical
results (the kernels have no atomics on these paths); a DMA / barrier ordering bug shows up as rare differing tiles."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from more4d_amd import ops  # noqa: E402
from more4d_amd.ops import KV  # noqa: E402

N = int(sys.argv[1]) if len(sys.argv) > 1 else 30
dev, bf = "cuda", torch.bfloat16
g = torch.Generator(device=dev).manual_seed(0)
L, C, H, D = 21840, 5120, 40, 128
bad = 0


def rep(name, fn):
    global bad
    ref = fn()
    diffs = sum(0 if torch.equal(fn(), ref) else 1 for _ in range(N))
    bad += diffs
    print(f"{name}: {diffs} of {N} repeats differ", flush=True)


for Nn, K in ((C, C), (13824, C), (C, 13824)):
    a = torch.randn(2 * L, K, generator=g, device=dev).to(bf)
    w = (torch.randn(Nn, K, generator=g, device=dev) * K ** -0.5).to(bf)
    b = torch.randn(Nn, generator=g, device=dev).to(bf)
    rep(f"gemm {Nn}x{K}", lambda: ops.gemm_bt(a, w, b))
    del a, w, b
q = torch.randn(2 * L, C, generator=g, device=dev).to(bf)
k = torch.randn(2 * L, C, generator=g, device=dev).to(bf)
vt = torch.randn(C, 2 * L, generator=g, device=dev).to(bf)
kw = dict(B=2, Lq=L, heads=H, head_dim=D, q_bs=L * C, q_ls=C)
rep("attention self", lambda: ops.attention(q, [KV(k, vt, L * C, C, L, 2 * L, L)], **kw))
Ls = 5464
segs = [KV(k[r * Ls:], vt[:, r * Ls:], L * C, C, L, 2 * L, min(Ls, L - r * Ls)) for r in range(4)]
rep("attention 4 segments", lambda: ops.attention(q, segs, **kw))
kc = torch.randn(2 * 512, C, generator=g, device=dev).to(bf)
vc = torch.randn(C, 2 * 512, generator=g, device=dev).to(bf)
rep("attention cross", lambda: ops.attention(q, [KV(kc, vc, 512 * C, C, 512, 1024, 512)], **kw))
# the one-wave-per-SIMD attention backward (round 6) at the training shape: B = 1, scale = ln 2 (folded)
import math  # noqa: E402
LN2 = math.log(2.0)
del q, k, vt, kc, vc
qkv = torch.randn(L, 3 * C, generator=g, device=dev)
qkv[:, :C] *= D ** -0.5 / LN2
qkv = qkv.to(bf)
q1, k1, v1 = qkv[:, :C], qkv[:, C:2 * C], qkv[:, 2 * C:]
d_o = torch.randn(L, C, generator=g, device=dev).to(bf)
lse = torch.empty(1, H, L, device=dev)
o1 = ops.attention(q1, [KV(k1, ops.transpose(v1.contiguous()), L * 3 * C, 3 * C, L, L, L)], B=1, Lq=L, heads=H, head_dim=D, q_bs=L * 3 * C,
                   q_ls=3 * C, lse=lse, scale=LN2).view(L, C)


def bwd():
    g3 = torch.empty_like(qkv)
    ops.attention_bwd(q1, k1, v1, o1, d_o, lse, B=1, Lq=L, Lk=L, Lk_rows=L, heads=H, head_dim=D, dq=g3[:, :C], dk=g3[:, C:2 * C],
                      dv=g3[:, 2 * C:], scale=LN2)
    return g3


ops.launch_counts(reset=True)
rep("attention backward (dq64 + kv64)", bwd)
assert ops.launch_counts()["attn_bwd64"] == 2 * (N + 1), ops.launch_counts()
print("RACE SCREEN", "CLEAN" if bad == 0 else f"FAILED ({bad})")
