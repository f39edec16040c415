#!/usr/bin/env python
"""Micro-benchmark of m4d_conv_cl at the VAE's layer shapes (bf16, stride 1): TFLOP/s per shape for the selected M4D_CONV_VARIANT.
    M4D_CONV_VARIANT=2|3 python tools/bench_conv.py"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from more4d_amd import ops  # noqa: E402

SHAPES = [  # (name, t_out, kt, H, W, Cin, Cout)
    ("dec 480x832 96->96 x4", 4, 3, 480, 832, 96, 96),
    ("dec 240x416 192->192 x4", 4, 3, 240, 416, 192, 192),
    ("dec 120x208 384->384 x2", 2, 3, 120, 208, 384, 384),
    ("mid 60x104 384->384 x1", 1, 3, 60, 104, 384, 384),
    ("enc 240x416 96->192 x4", 4, 3, 240, 416, 96, 192),
    ("adaptor 2-D 480x832 128->128 x8", 8, 1, 480, 832, 128, 128),
    ("dec 480x832 96->96 x1", 1, 3, 480, 832, 96, 96),
]


def main():
    dev = "cuda"
    g = torch.Generator(device=dev).manual_seed(0)
    for name, t, kt, H, W, ci, co in SHAPES:
        Tin = t + kt - 1
        x = torch.randn(Tin, H, W, ci, generator=g, device=dev).bfloat16()
        w = (torch.randn(co, kt * 9 * ci, generator=g, device=dev) * (kt * 9 * ci) ** -0.5).bfloat16()
        b = torch.zeros(co, device=dev, dtype=torch.bfloat16)
        kw = dict(Tin=Tin, Hin=H, Win=W, Cin=ci, k=(kt, 3, 3), pad=(0, 1, 1), out_thw=(t, H, W))
        out = ops.conv_cl(x, w, b, **kw)
        # steady state: the first ~15 launches after an idle gap ride a clock transient (tools/bench_conv_cold.py), so warm up
        # with 40 launches and report the median of 30 launches timed one by one
        for _ in range(40):
            ops.conv_cl(x, w, b, out=out, **kw)
        n = 30
        ev = [torch.cuda.Event(enable_timing=True) for _ in range(n + 1)]
        ev[0].record()
        for i in range(n):
            ops.conv_cl(x, w, b, out=out, **kw)
            ev[i + 1].record()
        torch.cuda.synchronize()
        dt = sorted(ev[i].elapsed_time(ev[i + 1]) for i in range(n))[n // 2] * 1e-3
        fl = 2.0 * t * H * W * co * kt * 9 * ci
        print(f"{name:36s} {dt*1e6:9.1f} us  {fl/dt/1e12:7.1f} TF/s  ({fl/dt/1e12/25:.1f} % of 2.5 PF)", flush=True)


if __name__ == "__main__":
    main()
