#!/usr/bin/env python
"""The gated-residual GEMMs of a DiT block at the bench shape (o-projection K = 5120, ffn_down K = 13824: fp32 residual stream updated in
the epilogue), sustained: time per launch and TF/s.  A/B of epilogue variants through side builds (M4D_LIB=<tag>)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from more4d_amd import ops
M, C = 43680, 5120
g = torch.Generator(device="cuda").manual_seed(0)
res = torch.randn(M, C, device="cuda", generator=g)
gate = torch.randn(2, C, device="cuda", generator=g)
for name, K in (("o_proj", 5120), ("ffn_down", 13824)):
    a = torch.randn(M, K, device="cuda", generator=g).bfloat16()
    w = (torch.randn(C, K, device="cuda", generator=g) * K ** -0.5).bfloat16()
    b = torch.randn(C, device="cuda", generator=g).bfloat16()
    def run():
        ops.gemm_bt(a, w, b, out=res, epilogue=ops.EPI_RESID_GATE, gate=gate, gate_stride=C, rows_per_sample=M // 2)
    for _ in range(5): run()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); s.record()
    n = 40
    for _ in range(n): run()
    e.record(); torch.cuda.synchronize()
    ms = s.elapsed_time(e) / n
    print(name, round(ms, 4), "ms", round(2 * M * C * K / ms / 1e9, 1), "TF/s", "finite" if bool(torch.isfinite(res).all()) else "NOT FINITE", flush=True)
    res.normal_()
