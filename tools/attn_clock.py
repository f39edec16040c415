#!/usr/bin/env python
"""Shader clock the phased self-attention kernel runs at (tool build): M4D_LIB=abl M4D_ATTN_ABL=64 python tools/attn_clock.py
Every workgroup stamps s_memtime and the 100 MHz wall clock at entry and exit; clock = cycles / time, as tools/gemm_timeline.py."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
dbg = torch.zeros(1 << 20, dtype=torch.int64, device="cuda")
os.environ["M4D_ATTN_DBG_PTR"] = str(dbg.data_ptr())
from more4d_amd import ops  # noqa: E402
B, L, n, D = 2, 21840, 40, 128
C = n * D
q = torch.randn(B, L, C, device="cuda", dtype=torch.bfloat16)
k = torch.randn(B, L, C, device="cuda", dtype=torch.bfloat16)
vt = torch.randn(C, B * L, device="cuda", dtype=torch.bfloat16)
out = torch.empty_like(q)
s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
N = int(os.environ.get("ATTN_ITERS", "8"))
for i in range(N):
    if i == 3:
        s.record()
    ops.attention(q, [ops.KV(k, vt, L * C, C, L, B * L, L)], B=B, Lq=L, heads=n, head_dim=D, out=out)
e.record()
torch.cuda.synchronize()
ms = s.elapsed_time(e) / (N - 3)
r = dbg.cpu().numpy().reshape(-1, 4)
r = r[r[:, 1] > 0]
if len(r) == 0:          # no stamps (shipping library or M4D_ATTN_ABL without bit 64): time only
    print(f"attn128p self-attention: {ms:.3f} ms = {4.0 * B * L * L * n * D / ms / 1e9:.0f} TF")
    sys.exit(0)
cyc, us = r[:, 1] - r[:, 0], (r[:, 3] - r[:, 2]) / 100.0
clk = cyc / np.maximum(us, 1e-9) / 1e3
tf = 4.0 * B * L * L * n * D / ms / 1e9
if int(os.environ.get("M4D_ATTN_ABL", "0")) & 128:      # phase stamps of workgroup 1000 (waves 0 and 4), tiles 100..107
    st = dbg.cpu().numpy()[(1 << 19):(1 << 19) + 128].reshape(2, 8, 8)
    # one barrier per tile: the early group (0) runs V(i), M(i) inside interval i, the late group (1) M(i), V(i+1); slots: 0 V start, 1 V end,
    # 2 M start, 3 M stream issued, 4 end of interval (in front of the barrier)
    order = {0: [0, 1, 2, 3, 4], 1: [2, 3, 0, 1, 4]}
    label = {0: ["V start", "V end", "M start", "M issued", "interval end"], 1: ["M start", "M issued", "V start", "V end", "interval end"]}
    for g in range(2):
        base = st[g, 0, order[g][0]]
        print(f"group {g} ({'V then M' if g == 0 else 'M then V'}): cycles relative to the first stamp of tile 100; columns: " + ", ".join(label[g]))
        for ti in range(8):
            row = [int(st[g, ti, k] - base) for k in order[g]]
            d = [row[1] - row[0], row[2] - row[1], row[3] - row[2], row[4] - row[3]]
            nxt = int(st[g, ti + 1, order[g][0]] - base) - row[4] if ti < 7 else 0
            print(f"  tile {100 + ti}: " + " ".join(f"{x:7d}" for x in row) + f"   first phase {d[0]:5d}  gap {d[1]:5d}  second phase {d[2]:5d}  tail {d[3]:5d}  barrier -> next {nxt:5d}")
print(f"attn128p self-attention (abl {os.environ.get('M4D_ATTN_ABL')}): {ms:.2f} ms = {tf:.0f} TF; workgroups {len(r)}, lifetime {us.mean():.1f} us, shader clock "
      f"{np.median(clk):.3f} GHz (p10 {np.percentile(clk, 10):.3f}, p90 {np.percentile(clk, 90):.3f}); MFMA floor per workgroup "
      f"{(L / 64) * 32 * 32 * 2:.0f} cycles of {cyc.mean():.0f} = {(L / 64) * 32 * 32 * 2 / cyc.mean():.3f} busy")
