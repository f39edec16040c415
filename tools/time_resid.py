import sys, os
sys.path.insert(0, os.getcwd())
import torch
from more4d_amd import ops
M = 43680
g = torch.Generator(device="cuda").manual_seed(1)
for name, N, K in (("o-proj", 5120, 5120), ("ffn_down", 5120, 13824)):
    a = torch.randn(M, K, device="cuda", generator=g).bfloat16(); w = (torch.randn(N, K, device="cuda", generator=g) * K ** -0.5).bfloat16()
    b = torch.randn(N, device="cuda", generator=g).bfloat16()
    resid = torch.zeros(M, N, device="cuda"); gate = torch.randn(2, N, device="cuda", generator=g)
    kw = dict(out=resid, epilogue=ops.EPI_RESID_GATE, gate=gate, gate_stride=N, rows_per_sample=M // 2)
    for _ in range(5): ops.gemm_bt(a, w, b, **kw)
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); s.record()
    n = 40
    for _ in range(n): ops.gemm_bt(a, w, b, **kw)
    e.record(); torch.cuda.synchronize()
    ms = s.elapsed_time(e) / n
    s2 = torch.cuda.Event(enable_timing=True); e2 = torch.cuda.Event(enable_timing=True)
    out = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
    for _ in range(5): ops.gemm_bt(a, w, b, out=out)
    torch.cuda.synchronize(); s2.record()
    for _ in range(n): ops.gemm_bt(a, w, b, out=out)
    e2.record(); torch.cuda.synchronize()
    ms2 = s2.elapsed_time(e2) / n
    print(f"{name}: gated residual {ms:.3f} ms = {2*M*N*K/ms/1e9:.0f} TF; bf16 store {ms2:.3f} ms = {2*M*N*K/ms2/1e9:.0f} TF")
