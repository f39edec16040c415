#!/usr/bin/env python
"""Is conv_halo_kernel power-limited?  Time single launches after an idle gap, then a back-to-back burst, launch by launch.
python tools/bench_conv_cold.py [index into bench_conv.SHAPES]"""
import os
import sys
import time
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
out = torch.empty(t * H * W, co, device="cuda", dtype=torch.bfloat16)


def launch():
    ops.conv_cl(x, w, b, Tin=Tin, Hin=H, Win=W, Cin=ci, k=(kt, 3, 3), pad=(0, 1, 1), out_thw=(t, H, W), out=out)


for _ in range(3):
    launch()
torch.cuda.synchronize()
cold = []
for _ in range(8):
    time.sleep(0.2)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); launch(); e1.record(); torch.cuda.synchronize()
    cold.append(e0.elapsed_time(e1) * 1e3)
print(name, "after 0.2 s idle, us:", " ".join(f"{v:.0f}" for v in cold))
time.sleep(0.5)
ev = [torch.cuda.Event(enable_timing=True) for _ in range(41)]
ev[0].record()
for i in range(40):
    launch(); ev[i + 1].record()
torch.cuda.synchronize()
print("burst of 40, us:", " ".join(f"{ev[i].elapsed_time(ev[i + 1]) * 1e3:.0f}" for i in range(40)))
