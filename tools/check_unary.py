import sys, os, torch
sys.path.insert(0, os.getcwd())
from more4d_amd import ops
g = torch.Generator(device="cuda").manual_seed(0)
for dt_in, dt_out in ((torch.float32, torch.bfloat16), (torch.bfloat16, torch.float32), (torch.float32, torch.float32), (torch.bfloat16, torch.bfloat16)):
    for act in (0, 1, 2, 3):
        x = torch.randn(1 << 17, generator=g, device="cuda").to(dt_in)
        big = ops.unary(x, dt_out, act=act)                       # vectorised path
        small = torch.cat([ops.unary(x[i * 4099:(i + 1) * 4099].clone(), dt_out, act=act) for i in range(31)])   # element-wise path (n % 8 != 0)
        assert torch.equal(big[:small.numel()], small), (dt_in, dt_out, act)
x = torch.randn(2 * 21840 * 5120, generator=g, device="cuda")
out = torch.empty_like(x, dtype=torch.bfloat16)
for _ in range(3): ops.unary(x, torch.bfloat16, out=out)
s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
torch.cuda.synchronize(); s.record()
for _ in range(20): ops.unary(x, torch.bfloat16, out=out)
e.record(); torch.cuda.synchronize()
ms = s.elapsed_time(e) / 20
print("unary f32->bf16", round(ms, 4), "ms", round(x.numel() * 6 / ms / 1e6, 1), "GB/s", "equal to torch:", bool(torch.equal(out, x.bfloat16())))
