import torch, sys
sys.path.insert(0, ".")
from more4d_amd import ops
B, n, L, D = 1, 1, 2048, 128
C = n * D
q = torch.randn(B, 1280, C, device="cuda").bfloat16()
k = torch.randn(B, L, C, device="cuda").bfloat16()
vt = torch.randn(C, B * L, device="cuda").bfloat16()
out = torch.zeros(B, 1280, C, device="cuda", dtype=torch.bfloat16)
ops.attention(q, [ops.KV(k, vt, L * C, C, L, B * L, L)], B=B, Lq=1280, heads=n, head_dim=D, out=out)
torch.cuda.synchronize()
w = out.view(torch.int32).reshape(-1)[:256 * 16].reshape(256, 16).cpu()
for tid in (0, 1, 15, 16, 17, 63, 64, 128, 255):
    print(tid, [hex(x & 0xffffffff) for x in w[tid].tolist()])
