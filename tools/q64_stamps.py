#!/usr/bin/env python
"""Phase stamps of attn128q_kernel (side build `--stamps`: M4D_LIB=q64st): s_memtime of wave 0 of workgroup 1000 at the start of phase A,
the start of phase B and the end of phase B of tiles 100..107 at the bench shape -> cycles per phase / per MFMA."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
dbg = torch.zeros(1 << 16, dtype=torch.int64, device="cuda")
os.environ["M4D_ATTN_DBG_PTR"] = str(dbg.data_ptr())
os.environ["M4D_ATTN_Q64"] = "1"
from more4d_amd import ops  # noqa: E402

B, L, n, D = 2, 21840, 40, 128
C = n * D
g = torch.Generator(device="cuda").manual_seed(0)
q = (torch.randn(B, L, C, device="cuda", generator=g) * 0.1275).bfloat16()
k = torch.randn(B, L, C, device="cuda", generator=g).bfloat16()
vt = torch.randn(C, B * L, device="cuda", generator=g).bfloat16()
out = torch.empty_like(q)
for _ in range(int(os.environ.get("ATTN_ITERS", "6"))):
    dbg.zero_()
    ops.attention(q, [ops.KV(k, vt, L * C, C, L, B * L, L)], B=B, Lq=L, heads=n, head_dim=D, out=out, scale=0.6931471805599453)
torch.cuda.synchronize()
st = dbg[:32].view(8, 4).cpu()
print("tile  A(cycles)  B(cycles)  A+B   cycles/MFMA")
tot = []
for i in range(8):
    a, b, e = int(st[i, 0]), int(st[i, 1]), int(st[i, 2])
    if a and b and e:
        print(f"{100 + i:4d}  {b - a:8d}  {e - b:8d}  {e - a:6d}  {(e - a) / 64:6.1f}")
        tot.append(e - a)
nxt = [int(st[i + 1, 0]) - int(st[i, 0]) for i in range(7) if int(st[i + 1, 0]) and int(st[i, 0])]
print("start-to-start:", nxt, " mean per tile", sum(nxt) / max(1, len(nxt)), " per MFMA", sum(nxt) / max(1, len(nxt)) / 64,
      "(a stamp costs ~150-200 cycles: three per tile)")
