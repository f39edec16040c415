#!/usr/bin/env python
"""Micro-benchmark of m4d_attention_bwd at the DiT's training shape (GPU box): self-attention B=1, L=21840, 40 heads,
D=128 (plus the text cross-attention shape).  Random bf16 data; spot-checks dq/dk/dv of one head against fp32 torch."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from more4d_amd import ops


def run(name, B, Lq, Lk, n, D=128, iters=3, check=True):
    dev, bf = "cuda", torch.bfloat16
    C = n * D
    Lkp = (Lk + 7) // 8 * 8
    g = torch.Generator(device=dev).manual_seed(0)
    q = torch.randn(B * Lq, C, device=dev, generator=g).to(bf)
    k = torch.randn(B * Lkp, C, device=dev, generator=g).to(bf)
    v = torch.randn(B * Lkp, C, device=dev, generator=g).to(bf)
    d_o = torch.randn(B * Lq, C, device=dev, generator=g).to(bf)
    vt = ops.transpose(v)
    lse = torch.empty(B, n, Lq, device=dev)
    o = ops.attention(q, [ops.KV(k, vt, Lkp * C, C, Lkp, B * Lkp, Lk)], B=B, Lq=Lq, heads=n, head_dim=D, q_bs=Lq * C, q_ls=C,
                      lse=lse).view(B * Lq, C)
    dq, dk, dv = torch.empty_like(q), torch.empty_like(k), torch.empty_like(v)
    kw = dict(B=B, Lq=Lq, Lk=Lk, Lk_rows=Lkp, heads=n, head_dim=D, dq=dq, dk=dk, dv=dv)
    ops.attention_bwd(q, k, v, o, d_o, lse, **kw)
    err = {}
    if check:
        h = 3
        sl = slice(h * D, (h + 1) * D)
        qq = q[:Lq, sl].float().requires_grad_(True)
        kk = k[:Lk, sl].float().requires_grad_(True)
        vv = v[:Lk, sl].float().requires_grad_(True)
        ref = torch.softmax(qq @ kk.t() / D ** 0.5, -1) @ vv
        gq, gk, gv = torch.autograd.grad(ref, (qq, kk, vv), d_o[:Lq, sl].float())
        for nm, a, r in (("dq", dq[:Lq, sl], gq), ("dk", dk[:Lk, sl], gk), ("dv", dv[:Lk, sl], gv)):
            err[nm] = round(float((a.float() - r).abs().max() / r.abs().max()), 4)
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); s.record()
    for _ in range(iters):
        ops.attention_bwd(q, k, v, o, d_o, lse, **kw)
    e.record(); torch.cuda.synchronize()
    ms = s.elapsed_time(e) / iters
    print(name, dict(B=B, Lq=Lq, Lk=Lk, ms=round(ms, 2), tflops_10=round(10 * B * Lq * Lk * n * D / ms / 1e9, 1), err=err), flush=True)


run("self", 1, 21840, 21840, 40)
run("cross_txt", 1, 21840, 512, 40, iters=10)
run("cross_img", 1, 21840, 257, 40, iters=10)
