#!/usr/bin/env python
"""One conv shape, a few launches (for PMC passes): python tools/bench_conv_one.py [index into bench_conv.SHAPES]"""
import os
import sys
import torch
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench_conv import SHAPES  # noqa: E402
from more4d_amd import ops  # noqa: E402

name, t, kt, H, W, ci, co = SHAPES[int(sys.argv[1]) if len(sys.argv) > 1 else 0]
g = torch.Generator(device="cuda").manual_seed(0)
Tin = t + kt - 1
x = torch.randn(Tin, H, W, ci, generator=g, device="cuda").bfloat16()
w = (torch.randn(co, kt * 9 * ci, generator=g, device="cuda") * (kt * 9 * ci) ** -0.5).bfloat16()
b = torch.zeros(co, device="cuda", dtype=torch.bfloat16)
for _ in range(3):
    ops.conv_cl(x, w, b, Tin=Tin, Hin=H, Win=W, Cin=ci, k=(kt, 3, 3), pad=(0, 1, 1), out_thw=(t, H, W))
torch.cuda.synchronize()
print(name)
