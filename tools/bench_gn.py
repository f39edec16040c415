import sys, torch
sys.path.insert(0, "/root/repo")
from more4d_amd import ops
F, H, W, C = int(sys.argv[1]) if len(sys.argv) > 1 else 17, 480, 832, 128
x = torch.randn(F, H * W, C, device="cuda").bfloat16()
w = torch.ones(C, device="cuda"); b = torch.zeros(C, device="cuda")
out = torch.empty_like(x)
for _ in range(3): ops.groupnorm_cl(x, w, b, F=F, HW=H * W, groups=32, out=out)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(10): ops.groupnorm_cl(x, w, b, F=F, HW=H * W, groups=32, out=out)
e1.record(); torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / 10
print(f"groupnorm {ms*1e3:.0f} us, {3 * x.numel() * 2 / ms / 1e9:.2f} TB/s (2 reads + 1 write)")

# the same call between MFMA-heavy launches (does the power state of the conv phase slow a memory-bound kernel?)
xc = torch.randn(8, H, W, 128, device="cuda").bfloat16()
wc = (torch.randn(128, 9 * 128, device="cuda") * (9 * 128) ** -0.5).bfloat16()
bc = torch.zeros(128, device="cuda", dtype=torch.bfloat16)
oc = torch.empty(8 * H * W, 128, device="cuda", dtype=torch.bfloat16)
ts = []
for _ in range(6):
    for _ in range(6):
        ops.conv_cl(xc, wc, bc, Tin=8, Hin=H, Win=W, Cin=128, k=(1, 3, 3), pad=(0, 1, 1), out_thw=(8, H, W), out=oc)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); ops.groupnorm_cl(x, w, b, F=F, HW=H * W, groups=32, out=out); e1.record(); torch.cuda.synchronize()
    ts.append(e0.elapsed_time(e1) * 1e3)
print("after 6 conv launches each, us:", " ".join(f"{t:.0f}" for t in ts))
# a fresh output tensor per call, as the adaptor does it
ts = []
for _ in range(6):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); y = ops.groupnorm_cl(x, w, b, F=F, HW=H * W, groups=32); e1.record(); torch.cuda.synchronize()
    ts.append(e0.elapsed_time(e1) * 1e3); del y
print("fresh output each call, us:", " ".join(f"{t:.0f}" for t in ts))
