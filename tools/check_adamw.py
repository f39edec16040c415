#!/usr/bin/env python
"""m4d_adamw: the eight-elements-per-thread kernel against the element-wise one (n % 8 != 0 takes the latter) — same bits — and its
bandwidth on a 5120 x 5120 parameter."""
import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from more4d_amd import ops
g = torch.Generator(device="cuda").manual_seed(0)
kw = dict(lr=1e-3, beta1=0.9, beta2=0.999, eps=1e-8, weight_decay=0.01, step=3)
for dt, sdt in ((torch.bfloat16, torch.bfloat16), (torch.bfloat16, torch.float32), (torch.float32, torch.float32)):
    n = 1 << 17
    p0 = torch.randn(n, generator=g, device="cuda").to(dt); gr = torch.randn(n, generator=g, device="cuda").to(dt)
    m0 = torch.randn(n, generator=g, device="cuda").to(sdt) * 0.1; v0 = torch.rand(n, generator=g, device="cuda").to(sdt) * 0.1
    scale = torch.tensor([0.5], device="cuda")
    a = [t.clone() for t in (p0, m0, v0)]
    ops.adamw_(a[0], gr, a[1], a[2], grad_scale=scale, **kw)                      # vectorised
    b = [t.clone() for t in (p0, m0, v0)]
    k = 4099
    for i in range(31):                                                          # element-wise path on odd-sized slices
        sl = slice(i * k, (i + 1) * k)
        ps, ms, vs = b[0][sl].clone(), b[1][sl].clone(), b[2][sl].clone()
        ops.adamw_(ps, gr[sl].clone(), ms, vs, grad_scale=scale, **kw)
        b[0][sl], b[1][sl], b[2][sl] = ps, ms, vs
    for x, y in zip(a, b):
        assert torch.equal(x[:31 * k], y[:31 * k]), (dt, sdt)
n = 5120 * 5120
p = torch.randn(n, generator=g, device="cuda").bfloat16(); gr = torch.randn(n, generator=g, device="cuda").bfloat16()
m = torch.zeros(n, device="cuda", dtype=torch.bfloat16); v = torch.zeros(n, device="cuda", dtype=torch.bfloat16)
for _ in range(3): ops.adamw_(p, gr, m, v, **kw)
s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
torch.cuda.synchronize(); s.record()
for _ in range(20): ops.adamw_(p, gr, m, v, **kw)
e.record(); torch.cuda.synchronize()
ms = s.elapsed_time(e) / 20
print("adamw bf16 / bf16 state, 26 M elements:", round(ms, 4), "ms", round(n * 14 / ms / 1e6, 1), "GB/s; same bits as the element-wise kernel")
