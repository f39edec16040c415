#!/usr/bin/env python
"""Per-workgroup timeline of the wide GEMM kernel (tool build: M4D_LIB=abl M4D_GEMM_VARIANT=5 M4D_GEMM_ABL=64): every workgroup
leaves shader-clock stamps (entry, loop start, loop end, drained, stored) + the 100 MHz wall clock + its hardware id in the first
output row of its tile.  Prints the mean duration of each section, the clock, and the gap between consecutive workgroups on a CU."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from more4d_amd import ops
M, N, K = map(int, sys.argv[1:4]) if len(sys.argv) > 3 else (43680, 5120, 5120)
a = torch.randn(M, K, device="cuda", dtype=torch.bfloat16)
w = torch.randn(N, K, device="cuda", dtype=torch.bfloat16) * K ** -0.5
out = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
for _ in range(int(os.environ.get("TIMELINE_LAUNCHES", "60"))):
    ops.gemm_bt(a, w, None, out=out)
torch.cuda.synchronize()
o = out.cpu().view(torch.int16).numpy()
rows = []
for tm in range((M + 255) // 256):
    for tn in range((N + 255) // 256):
        m0, n0 = min(tm * 256, M - 256), min(tn * 256, N - 256)
        rows.append(o[m0, n0:n0 + 32].copy().view(np.uint64))
r = np.stack(rows).astype(np.int64)
# Edge tiles are shifted inwards (origin clamped to M-256 / N-256), so the first row of a clamped tile lies inside its neighbour's
# rows and its stamps may have been overwritten with that tile's OUTPUT: keep only rows whose stamps are a plausible timeline
# (monotonic shader clock, wall clock within ten seconds).  Round 3's log summed such rows into its means (int64 overflow).
n_all = len(r)
good = (r[:, 0] < r[:, 1]) & (r[:, 1] <= r[:, 2]) & (r[:, 2] <= r[:, 3]) & (r[:, 3] <= r[:, 4]) & (r[:, 4] - r[:, 0] < 10 ** 10) & \
       (r[:, 6] > r[:, 5]) & (r[:, 6] - r[:, 5] < 10 ** 9)
if good.any():
    w_med = np.median(r[good, 5])
    good &= np.abs(r[:, 5] - w_med) < 10 ** 9
r = r[good]
print(f"stamped tiles kept {len(r)} of {n_all}")
t0, t1, t2, t3, t4, w0, w1, hw = [r[:, i].astype(np.float64) if i < 7 else r[:, i] for i in range(8)]
us = (w1 - w0) / 100.0
clk = (t4 - t0) / np.maximum(us, 1e-9) / 1e3
print(f"tiles {len(r)}  kernel span {(w1.max() - w0.min()) / 100.0:.1f} us   shader clock {np.median(clk):.3f} GHz (p10 {np.percentile(clk, 10):.3f}, p90 {np.percentile(clk, 90):.3f})")
for name, d in (("prologue", t1 - t0), ("main loop", t2 - t1), ("drain + barrier", t3 - t2), ("epilogue + store ack", t4 - t3), ("total", t4 - t0)):
    print(f"  {name:22s} mean {d.mean():9.0f} cyc  p10 {np.percentile(d, 10):9.0f}  p90 {np.percentile(d, 90):9.0f}   = {d.mean() / np.median(clk) / 1e3:7.2f} us")
nk = K // 64
print(f"  effective {2.0 * M * N * K / ((w1.max() - w0.min()) / 100.0) / 1e6:.0f} TF over the kernel span (ablation {os.environ.get('M4D_GEMM_ABL')})")
print(f"  main loop per K-tile: {np.mean(t2 - t1) / nk:.0f} cycles (MFMA floor 2048)")
# gaps on the same CU: key = (xcc, se/cu bits of HW_ID)
key = (hw >> 32) * 65536 + ((hw & 0xffffffff) >> 8 & 0xfff)
gaps = []
for k in np.unique(key):
    idx = np.where(key == k)[0]
    idx = idx[np.argsort(w0[idx])]
    gaps.extend(((w0[idx][1:] - w1[idx][:-1]) / 100.0).tolist())
gaps = np.array(gaps)
print(f"  units seen {len(np.unique(key))}; gap between consecutive workgroups of a unit: mean {gaps.mean():.2f} us  p10 {np.percentile(gaps, 10):.2f}  p90 {np.percentile(gaps, 90):.2f}")
