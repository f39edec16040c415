import torch
x = torch.randn(43680, 5120, device="cuda")
y = torch.empty(43680, 5120, device="cuda", dtype=torch.bfloat16)
z = torch.empty_like(x)
def t(f, n=40):
    for _ in range(5): f()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); s.record()
    for _ in range(n): f()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / n * 1e3
us = t(lambda: y.copy_(x)); print("cast f32->bf16", round(us, 1), "us", round((x.numel() * 6) / us / 1e6, 2), "TB/s")
us = t(lambda: z.copy_(x)); print("copy f32", round(us, 1), "us", round((x.numel() * 8) / us / 1e6, 2), "TB/s")
us = t(lambda: x.sum()); print("sum f32 (read only)", round(us, 1), "us", round((x.numel() * 4) / us / 1e6, 2), "TB/s")
us = t(lambda: z.fill_(1.0)); print("fill f32 (write only)", round(us, 1), "us", round((x.numel() * 4) / us / 1e6, 2), "TB/s")
