#!/usr/bin/env python
"""Per-workgroup timeline of conv_halo_kernel (tool build): M4D_LIB=abl M4D_CONV_ABL=64 python tools/conv_timeline.py [shape index].
Every workgroup leaves shader-clock stamps (entry, K loop start, K loop end, stores acknowledged) + the 100 MHz wall clock."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from bench_conv import SHAPES
name, t, kt, H, W, ci, co = SHAPES[int(sys.argv[1]) if len(sys.argv) > 1 else 0]
dbg = torch.zeros(1 << 20, dtype=torch.int64, device="cuda")
os.environ["M4D_CONV_DBG_PTR"] = str(dbg.data_ptr())
from more4d_amd import ops  # noqa: E402
g = torch.Generator(device="cuda").manual_seed(0)
Tin = t + kt - 1
x = torch.randn(Tin, H, W, ci, generator=g, device="cuda").bfloat16()
w = (torch.randn(co, kt * 9 * ci, generator=g, device="cuda") * (kt * 9 * ci) ** -0.5).bfloat16()
b = torch.zeros(co, device="cuda", dtype=torch.bfloat16)
mode = sys.argv[2] if len(sys.argv) > 2 else "plain"       # plain (channels-last input) | planar | norm | normresid (production forms)
if mode == "plain":
    kw = dict(Tin=Tin, Hin=H, Win=W, Cin=ci, k=(kt, 3, 3), pad=(0, 1, 1), out_thw=(t, H, W))
    out = ops.conv_cl(x, w, b, **kw)
    run = lambda: ops.conv_cl(x, w, b, out=out, **kw)
else:
    xp = ops.Planar16(x.view(Tin, H * W, ci // 16, 16).permute(2, 0, 1, 3).contiguous())
    M = t * H * W
    out = torch.empty(M, co, device="cuda", dtype=torch.bfloat16)
    res = torch.randn(M, co, generator=g, device="cuda").bfloat16() if mode == "normresid" else None
    norm = None
    if mode in ("norm", "normresid"):
        dst = ops.Planar16(torch.empty(co // 16, t, H * W, 16, device="cuda", dtype=torch.bfloat16))
        norm = (torch.ones(co, device="cuda"), dst, True)
        if co not in (32, 64, 96, 128):
            raise SystemExit("fused norm: Cout must be 32 / 64 / 96 / 128")
    run = lambda: ops.conv_cl_planar(xp, w, b, Tin=Tin, Hin=H, Win=W, kt=kt, resid=res, out=out, norm=norm, keep_raw=True)
    run()
name = f"{name} [{mode}]"
for _ in range(40):
    run()
torch.cuda.synchronize()
r = dbg.cpu().numpy().reshape(-1, 8)
r = r[r[:, 7] == 1]
t0, t1, t2, t3, w0, w1, hw = [r[:, i] for i in range(7)]
us = (w1 - w0) / 100.0
clk = (t3 - t0) / np.maximum(us, 1e-9) / 1e3
span = (w1.max() - w0.min()) / 100.0
fl = 2.0 * t * H * W * co * kt * 9 * ci
print(f"{name}: workgroups {len(r)}  kernel span {span:.1f} us = {fl / span / 1e6:.0f} TF  shader clock {np.median(clk):.3f} GHz")
for nm, d in (("setup (addresses, first DMA issue)", t1 - t0), ("K loop", t2 - t1), ("epilogue + store ack", t3 - t2), ("total", t3 - t0)):
    print(f"  {nm:36s} mean {d.mean():9.0f} cyc  p10 {np.percentile(d, 10):9.0f}  p90 {np.percentile(d, 90):9.0f}  = {d.mean() / np.median(clk) / 1e3:7.2f} us")
nsteps = (ci // 16) * kt * 3
print(f"  K loop per (dt, dh) step: {np.mean(t2 - t1) / nsteps:.0f} cycles")
