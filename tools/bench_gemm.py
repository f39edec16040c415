#!/usr/bin/env python
"""Micro-benchmark of m4d_gemm_bt at the DiT's shapes (run on the GPU box).  Random bf16 operands
(cdna guide rule 25: never zero-filled), 5 warm-up + 20 timed launches, HIP events on torch's stream.
Checks each result against torch.matmul (rocBLAS) first — tool only, not a product path."""
import sys, os, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from more4d_amd import ops

SHAPES = [("qkvo", 43680, 5120, 5120), ("ffn_up", 43680, 13824, 5120), ("ffn_down", 43680, 5120, 13824),
          ("v_t", 5120, 43680, 5120), ("sq4k", 4096, 4096, 4096), ("sq8k", 8192, 8192, 8192)]


def main():
    dev = "cuda"
    res = {}
    for name, M, N, K in SHAPES:
        a = torch.randn(M, K, device=dev, dtype=torch.bfloat16)
        w = torch.randn(N, K, device=dev, dtype=torch.bfloat16) * K ** -0.5
        b = torch.randn(N, device=dev, dtype=torch.bfloat16)
        out = ops.gemm_bt(a, w, b)
        ref = torch.nn.functional.linear(a[:2048], w, b)
        err = float((out[:2048].float() - ref.float()).abs().max() / ref.float().abs().max())
        for _ in range(5):
            ops.gemm_bt(a, w, b, out=out)
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize()
        s.record()
        n = 20
        for _ in range(n):
            ops.gemm_bt(a, w, b, out=out)
        e.record()
        torch.cuda.synchronize()
        ms = s.elapsed_time(e) / n
        tf = 2 * M * N * K / ms / 1e9
        res[name] = dict(M=M, N=N, K=K, ms=round(ms, 4), tflops=round(tf, 1), relerr=err)
        # packed-weight production kernel: weight side is W unless the weight is the M operand (v_t)
        if name == "v_t":
            pk = ops.pack_frag(a); args = (pk, w)
        else:
            pk = ops.pack_frag(w); args = (a, pk)
        kw = dict(bias_on_m=True) if name == "v_t" else {}
        bb = torch.randn(M if name == "v_t" else N, device=dev, dtype=torch.bfloat16)
        out2 = ops.gemm_bt(*args, bb, **kw)
        ref2 = ops.gemm_bt(a, w, bb, **kw)
        res[name]["packed_equal"] = bool(torch.equal(out2, ref2))
        for _ in range(5):
            ops.gemm_bt(*args, bb, out=out2, **kw)
        torch.cuda.synchronize(); s.record()
        for _ in range(n):
            ops.gemm_bt(*args, bb, out=out2, **kw)
        e.record(); torch.cuda.synchronize()
        ms2 = s.elapsed_time(e) / n
        res[name]["packed_ms"] = round(ms2, 4); res[name]["packed_tflops"] = round(2 * M * N * K / ms2 / 1e9, 1)
        print(name, res[name], flush=True)
    print(json.dumps(res))


if __name__ == "__main__":
    main()
