#!/usr/bin/env python
"""Micro-benchmark of the HBM-bound DiT kernels at the bench shape (GPU box): achieved GB/s vs algorithmic bytes."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from more4d_amd import ops
dev = "cuda"
B, L, C = 2, 21840, 5120
x = torch.randn(B, L, C, device=dev)
e = torch.randn(B, 6, C, device=dev)
q = torch.randn(B * L, C, device=dev, dtype=torch.bfloat16)
k = torch.randn(B * L, C, device=dev, dtype=torch.bfloat16)
w = torch.ones(C, device=dev)
cos = torch.randn(L, 64, device=dev); sin = torch.randn(L, 64, device=dev)
out = torch.empty(B, L, C, device=dev, dtype=torch.bfloat16)

def timeit(fn, n=20):
    for _ in range(3): fn()
    s, e_ = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); s.record()
    for _ in range(n): fn()
    e_.record(); torch.cuda.synchronize()
    return s.elapsed_time(e_) / n

ms = timeit(lambda: ops.ln_modulate(x, torch.bfloat16, shift=e[:, 0], scale=e[:, 1], mod_stride=6 * C, rows_per_sample=L, out=out))
print("ln_modulate f32->bf16", round(ms, 4), "ms", round(B * L * C * 6 / ms / 1e6, 1), "GB/s")
ms = timeit(lambda: ops.rmsnorm_rope(q, w, k, w, head_dim=128, cos=cos, sin=sin, rows_per_sample=L, rope_len=L))
print("rmsnorm_rope q,k bf16", round(ms, 4), "ms", round(2 * B * L * C * 4 / ms / 1e6, 1), "GB/s")
