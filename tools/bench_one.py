#!/usr/bin/env python
"""One GEMM / attention shape, few launches (for PMC runs): bench_one.py gemm M N K | attn B L heads"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from more4d_amd import ops
dev = "cuda"
kind = sys.argv[1]
if kind == "gemm":
    M, N, K = map(int, sys.argv[2:5])
    a = torch.randn(M, K, device=dev, dtype=torch.bfloat16)
    w = torch.randn(N, K, device=dev, dtype=torch.bfloat16) * K ** -0.5
    out = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
    if os.environ.get("PACKED"):
        w = ops.pack_frag(w)
    for _ in range(4):
        ops.gemm_bt(a, w, None, out=out)
else:
    B, L, n = map(int, sys.argv[2:5])
    D = 128
    C = n * D
    q = torch.randn(B, L, C, device=dev, dtype=torch.bfloat16)
    k = torch.randn(B, L, C, device=dev, dtype=torch.bfloat16)
    vt = torch.randn(C, B * L, device=dev, dtype=torch.bfloat16)
    for _ in range(3):
        ops.attention(q, [ops.KV(k, vt, L * C, C, L, B * L, L)], B=B, Lq=L, heads=n, head_dim=D)
torch.cuda.synchronize()
