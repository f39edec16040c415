#!/usr/bin/env python
"""GroupNorm + swish into the planar-16 layout (the adaptors' norm in front of a 3x3 conv) at the bench shape: two-pass form and the
form that takes the producer's statistics (finalize + apply only).  python tools/bench_gn_planar.py [frames]"""
import os
import sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from more4d_amd import ops
F, H, W, C = int(sys.argv[1]) if len(sys.argv) > 1 else 8, 480, 832, 128
x = torch.randn(F, H * W, C, device="cuda").bfloat16()
w = torch.ones(C, device="cuda"); b = torch.zeros(C, device="cuda")
st = torch.randn(F, ops.gnstats_blocks(H, W), 32, 2, device="cuda").abs() + 1.0


def timed(fn, n=10):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


ms = timed(lambda: ops.groupnorm_cl_planar(x, w, b, F=F, HW=H * W, frames_per_group=F, stats=st))
print(f"finalize + apply: {ms * 1e3:.0f} us, {2 * x.numel() * 2 / ms / 1e9:.2f} TB/s (1 read + 1 write)")
ms = timed(lambda: ops.groupnorm_cl_planar(x, w, b, F=F, HW=H * W, frames_per_group=F))
print(f"stats + finalize + apply: {ms * 1e3:.0f} us, {3 * x.numel() * 2 / ms / 1e9:.2f} TB/s (2 reads + 1 write)")
out = torch.empty_like(x)
ms = timed(lambda: ops.groupnorm_cl(x, w, b, F=F, HW=H * W, out=out))
print(f"channels-last two-pass: {ms * 1e3:.0f} us, {3 * x.numel() * 2 / ms / 1e9:.2f} TB/s")
