#!/usr/bin/env python
"""Micro-benchmark of m4d_attention at the DiT's shapes (GPU box): self-attention B=2, L=21840, 40 heads, D=128,
and the two cross-attention shapes.  Random bf16 data; HIP events on torch's stream."""
import sys, os, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from more4d_amd import ops

def run(name, B, Lq, Lk, n, D=128, iters=5):
    dev = "cuda"
    C = n * D
    q = torch.randn(B, Lq, C, device=dev, dtype=torch.bfloat16)
    Lkp = (Lk + 7) // 8 * 8
    k = torch.randn(B, Lkp, C, device=dev, dtype=torch.bfloat16)
    vt = torch.randn(C, B * Lkp, device=dev, dtype=torch.bfloat16)
    seg = [ops.KV(k, vt, Lkp * C, C, Lkp, B * Lkp, Lk)]
    out = ops.attention(q, seg, B=B, Lq=Lq, heads=n, head_dim=D)
    # spot check 256 rows of head 3 against fp32 math
    qq = q[0, :256, 3 * D:4 * D].float(); kk = k[0, :Lk, 3 * D:4 * D].float(); vv = vt[3 * D:4 * D, :Lk].float().t()
    ref = torch.softmax(qq @ kk.t() / D ** 0.5, -1) @ vv
    err = float((out[0, :256, 3 * D:4 * D].float() - ref).abs().max() / ref.abs().max())
    for _ in range(2):
        ops.attention(q, seg, B=B, Lq=Lq, heads=n, head_dim=D, out=out)
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); s.record()
    for _ in range(iters):
        ops.attention(q, seg, B=B, Lq=Lq, heads=n, head_dim=D, out=out)
    e.record(); torch.cuda.synchronize()
    ms = s.elapsed_time(e) / iters
    tf = 4 * B * Lq * Lk * n * D / ms / 1e9
    print(name, dict(B=B, Lq=Lq, Lk=Lk, ms=round(ms, 3), tflops=round(tf, 1), relerr=round(err, 5)), flush=True)

def run_i2v(B=2, Lq=21840, n=40, D=128, iters=50):
    """text (512) | image (257) softmaxes in one launch: the DiT's WanI2VCrossAttention call (separate K / V^T buffers per branch)"""
    dev = "cuda"
    C = n * D
    q = torch.randn(B, Lq, C, device=dev, dtype=torch.bfloat16)
    segs = []
    for Lk in (512, 257):
        Lkp = (Lk + 7) // 8 * 8
        k = torch.randn(B, Lkp, C, device=dev, dtype=torch.bfloat16)
        vt = torch.randn(C, B * Lkp, device=dev, dtype=torch.bfloat16)
        segs.append(ops.KV(k, vt, Lkp * C, C, Lkp, B * Lkp, Lk))
    out = ops.attention(q, segs, B=B, Lq=Lq, heads=n, head_dim=D, new_softmax=0b10)
    for _ in range(3):
        ops.attention(q, segs, B=B, Lq=Lq, heads=n, head_dim=D, out=out, new_softmax=0b10)
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); s.record()
    for _ in range(iters):
        ops.attention(q, segs, B=B, Lq=Lq, heads=n, head_dim=D, out=out, new_softmax=0b10)
    e.record(); torch.cuda.synchronize()
    ms = s.elapsed_time(e) / iters
    print("cross_i2v", dict(B=B, Lq=Lq, Lk=(512, 257), ms=round(ms, 3), tflops=round(4 * B * Lq * 769 * n * D / ms / 1e9, 1)), flush=True)


which = sys.argv[1:] or ["self", "cross_txt", "cross_img", "cross_i2v"]
if "self" in which:
    run("self", 2, 21840, 21840, 40)
if "cross_txt" in which:
    run("cross_txt", 2, 21840, 512, 40, iters=50)
if "cross_img" in which:
    run("cross_img", 2, 21840, 257, 40, iters=50)
if "cross_i2v" in which:
    run_i2v()
