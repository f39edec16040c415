import sys, os, torch
sys.path.insert(0, os.getcwd())
from more4d_amd import ops
def run(B, Lq, lens, n=40, D=128, iters=30):
    C = n * D
    q = torch.randn(B, Lq, C, device="cuda", dtype=torch.bfloat16)
    segs = []
    for Lk in lens:
        Lkp = (Lk + 7) // 8 * 8
        k = torch.randn(B, Lkp, C, device="cuda", dtype=torch.bfloat16)
        vt = torch.randn(C, B * Lkp, device="cuda", dtype=torch.bfloat16)
        segs.append(ops.KV(k, vt, Lkp * C, C, Lkp, B * Lkp, Lk))
    ns = 0b10 if len(lens) == 2 else 0
    out = ops.attention(q, segs, B=B, Lq=Lq, heads=n, head_dim=D, new_softmax=ns)
    for _ in range(3): ops.attention(q, segs, B=B, Lq=Lq, heads=n, head_dim=D, out=out, new_softmax=ns)
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); s.record()
    for _ in range(iters): ops.attention(q, segs, B=B, Lq=Lq, heads=n, head_dim=D, out=out, new_softmax=ns)
    e.record(); torch.cuda.synchronize()
    ms = s.elapsed_time(e) / iters
    print(dict(B=B, Lq=Lq, lens=lens, ms=round(ms, 4), tf=round(4 * B * Lq * sum(lens) * n * D / ms / 1e9, 1)), flush=True)
for B, Lq in ((2, 21840), (4, 21840), (2, 5460), (1, 21840)):
    for lens in ((512,), (1024,), (512, 257)):
        run(B, Lq, lens)
