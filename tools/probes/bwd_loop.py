import sys, os, time
sys.path.insert(0, os.getcwd())
import torch
from more4d_amd import ops
B, Lq, n, D = 1, 21840, 40, 128
C = n * D
g = torch.Generator(device="cuda").manual_seed(0)
q = torch.randn(B * Lq, C, device="cuda", generator=g).bfloat16(); k = torch.randn_like(q); v = torch.randn_like(q); do = torch.randn_like(q)
vt = ops.transpose(v)
lse = torch.empty(B, n, Lq, device="cuda")
o = ops.attention(q, [ops.KV(k, vt, Lq * C, C, Lq, B * Lq, Lq)], B=B, Lq=Lq, heads=n, head_dim=D, q_bs=Lq * C, q_ls=C, lse=lse).view(B * Lq, C)
dq, dk, dv = torch.empty_like(q), torch.empty_like(k), torch.empty_like(v)
t0 = time.time()
while time.time() - t0 < 12:
    for _ in range(10):
        ops.attention_bwd(q, k, v, o, do, lse, B=B, Lq=Lq, Lk=Lq, Lk_rows=Lq, heads=n, head_dim=D, dq=dq, dk=dk, dv=dv)
    torch.cuda.synchronize()
print("done")
