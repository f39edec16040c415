#!/usr/bin/env python
"""Sustained timing of one m4d_gemm_bt shape (N(0,1) operands): time_gemm.py M N K [launches].  Honours M4D_GEMM_VARIANT and,
with M4D_LIB=abl, M4D_GEMM_ABL (timing ablations: results wrong by design)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from more4d_amd import ops
M, N, K = map(int, sys.argv[1:4])
n = int(sys.argv[4]) if len(sys.argv) > 4 else 40
a = torch.randn(M, K, device="cuda", dtype=torch.bfloat16)
w = torch.randn(N, K, device="cuda", dtype=torch.bfloat16) * K ** -0.5
out = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
for _ in range(8):
    ops.gemm_bt(a, w, None, out=out)
s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
torch.cuda.synchronize()
s.record()
for _ in range(n):
    ops.gemm_bt(a, w, None, out=out)
e.record()
torch.cuda.synchronize()
ms = s.elapsed_time(e) / n
print(f"variant {os.environ.get('M4D_GEMM_VARIANT', '-')} abl {os.environ.get('M4D_GEMM_ABL', '-')} {M}x{N}x{K}: {ms:.4f} ms  {2 * M * N * K / ms / 1e9:.0f} TF", flush=True)
