#!/usr/bin/env python
"""A/B of the attention backward at the train step's self-attention shape (B = 1, L = 21 840, 40 heads, d = 128, bf16): the fused
dK/dV pass (M4D_ATTN_BWD_FUSED=2, default: attention_bwd_kvp.h; 1: the first fused kernel) against the separate dK and dV passes
(M4D_ATTN_BWD_FUSED=0), alternating subprocesses on ONE box (environment
switches are read once per process).  Prints ms per m4d_attention_bwd call (delta + transposes excluded: the ops.attention_bwd wrapper
is timed as a whole and the pure-kernel time separately) and the MFMA fraction on the 10 L^2 d convention.
    python tools/ab_attn_bwd.py [rounds]"""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def child():
    sys.path.insert(0, ROOT)
    import torch
    from more4d_amd import ops
    B, Lq, n, D = 1, 21840, 40, 128
    C = n * D
    g = torch.Generator(device="cuda").manual_seed(0)
    q = torch.randn(B * Lq, C, device="cuda", generator=g).bfloat16()
    k, v, do = (torch.randn(B * Lq, C, device="cuda", generator=g).bfloat16() for _ in range(3))
    do = do * 0.1
    vt = ops.transpose(v)
    lse = torch.empty(B, n, Lq, device="cuda")
    o = ops.attention(q, [ops.KV(k, vt, Lq * C, C, Lq, B * Lq, Lq)], B=B, Lq=Lq, heads=n, head_dim=D, q_bs=Lq * C, q_ls=C, lse=lse).view(B * Lq, C)
    dq, dk, dv = torch.empty_like(q), torch.empty_like(k), torch.empty_like(v)

    def call():
        ops.attention_bwd(q, k, v, o, do, lse, B=B, Lq=Lq, Lk=Lq, Lk_rows=Lq, heads=n, head_dim=D, dq=dq, dk=dk, dv=dv)
    for _ in range(8):
        call()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    N = 20
    for _ in range(N):
        call()
    e.record()
    torch.cuda.synchronize()
    ms = s.elapsed_time(e) / N
    digest = [float(t.float().abs().sum()) for t in (dq, dk, dv)]
    print(json.dumps({"ms": ms, "tf_10L2d": 10.0 * Lq * Lq * D * n * B / ms / 1e9, "digest": digest}))


if __name__ == "__main__":
    if os.environ.get("AB_CHILD"):
        child()
        sys.exit(0)
    rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 2
    for r in range(rounds):
        for name, val in (("separate dK, dV passes", "0"), ("first fused kernel (lock-step)", "1"), ("phased fused kernel (default)", "2")):
            env = dict(os.environ, AB_CHILD="1", M4D_ATTN_BWD_FUSED=val)
            out = subprocess.run([sys.executable, os.path.abspath(__file__)], env=env, capture_output=True, text=True)
            line = [ln for ln in out.stdout.splitlines() if ln.startswith("{")]
            d = json.loads(line[-1]) if line else {"error": out.stderr[-400:]}
            if "ms" in d:
                print(f"round {r}  {name:24s} {d['ms']:7.2f} ms   {d['tf_10L2d']:6.0f} TF = {d['tf_10L2d'] / 2500:.3f} of the bf16 MFMA peak   digest {d['digest']}")
            else:
                print(name, d)
