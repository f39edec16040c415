#!/usr/bin/env python
"""The V^T projection GEMM (A = W_v [5120, 5120], W = x [43680, 5120], bias along m) next to a q projection of the same FLOPs."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, hashlib
from more4d_amd import ops
g = torch.Generator(device="cuda").manual_seed(2)
x = torch.randn(43680, 5120, device="cuda", generator=g).bfloat16()
w = (torch.randn(5120, 5120, device="cuda", generator=g) * 5120 ** -0.5).bfloat16()
b = torch.randn(5120, device="cuda", generator=g).bfloat16()
def t(f, n=40):
    for _ in range(5): f()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); s.record()
    for _ in range(n): f()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / n
vt = torch.empty(5120, 43680, device="cuda", dtype=torch.bfloat16)
q = torch.empty(43680, 5120, device="cuda", dtype=torch.bfloat16)
ops.gemm_bt(w, x, b, out=vt, bias_on_m=True)
ops.gemm_bt(x, w, b, out=q)
print("V^T == q^T bit for bit:", torch.equal(vt.t().contiguous(), q), hashlib.sha1(vt.view(torch.int16).cpu().numpy().tobytes()).hexdigest()[:12])
fl = 2 * 43680 * 5120 * 5120
m1 = t(lambda: ops.gemm_bt(w, x, b, out=vt, bias_on_m=True)); m2 = t(lambda: ops.gemm_bt(x, w, b, out=q))
print(f"V^T projection {m1:.3f} ms = {fl / m1 / 1e9:.0f} TF; q projection {m2:.3f} ms = {fl / m2 / 1e9:.0f} TF")
