#!/usr/bin/env python
"""LN + modulation WITH spatial guidance (the generic ln_modulate_kernel: per-row guidance vectors) at the bench shape; A/B through side
builds (M4D_LIB=<tag>)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from more4d_amd import ops
B, L, C, period = 2, 21840, 5120, 1560
g = torch.Generator(device="cuda").manual_seed(0)
x = torch.randn(B, L, C, device="cuda", generator=g)
e = torch.randn(B, 6, C, device="cuda", generator=g)
g_ss = torch.randn(B, period, 2 * C, device="cuda", generator=g)
gate = torch.randn(C, device="cuda", generator=g)
out = torch.empty(B, L, C, device="cuda", dtype=torch.bfloat16)
def run():
    ops.ln_modulate(x, torch.bfloat16, shift=e[:, 0], scale=e[:, 1], mod_stride=6 * C, rows_per_sample=L, out=out, g_ss=g_ss, g_gate=gate, g_period=period, g_len=L - period)
for _ in range(3): run()
s, t = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
torch.cuda.synchronize(); s.record()
for _ in range(20): run()
t.record(); torch.cuda.synchronize()
ms = s.elapsed_time(t) / 20
print("ln_modulate + guidance", round(ms, 4), "ms", "checksum", float(out.float().abs().sum()))
