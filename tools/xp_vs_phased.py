#!/usr/bin/env python
"""attn128x_kernel vs attn128p_kernel on mid-length key lists (run twice: default = phased from 2 048 keys on,
M4D_ATTN_W8_MIN_KEYS=100000000 = the persistent kernel): separates per-interval loop cost from per-item overhead."""
import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from more4d_amd import ops
def run(B, Lq, Lk, n=40, D=128, iters=10):
    C = n * D
    q = torch.randn(B, Lq, C, device="cuda", dtype=torch.bfloat16)
    k = torch.randn(B, Lk, C, device="cuda", dtype=torch.bfloat16)
    vt = torch.randn(C, B * Lk, device="cuda", dtype=torch.bfloat16)
    segs = [ops.KV(k, vt, Lk * C, C, Lk, B * Lk, Lk)]
    out = ops.attention(q, segs, B=B, Lq=Lq, heads=n, head_dim=D)
    for _ in range(3): ops.attention(q, segs, B=B, Lq=Lq, heads=n, head_dim=D, out=out)
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); s.record()
    for _ in range(iters): ops.attention(q, segs, B=B, Lq=Lq, heads=n, head_dim=D, out=out)
    e.record(); torch.cuda.synchronize()
    ms = s.elapsed_time(e) / iters
    print(dict(Lk=Lk, ms=round(ms, 4), tf=round(4 * B * Lq * Lk * n * D / ms / 1e9, 1), counts={k_: v for k_, v in ops.launch_counts(reset=True).items() if v}), flush=True)
for Lk in ([int(a) for a in sys.argv[1:]] or [2048, 4096, 8192]):
    run(2, 21840, Lk)
