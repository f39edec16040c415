#!/usr/bin/env python
"""attn_bwd_dq64_kernel (M4D_ATTN_BWD64=1, the default) against an fp32 torch reference and against the two-waves-per-SIMD kernels
(M4D_ATTN_BWD64=0) on the same inputs, then timing at the training shape.
    python tools/check_bwd64.py [--time]        (M4D_LIB=<tag> selects a side build, e.g. the --plain stream)"""
import os
import subprocess
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
DEV = "cuda"
LN2 = 0.6931471805599453


def case(B, n, Lq, Lk, seed=0, spikes=False):
    """scale = ln 2 (the DiT folds the softmax scale into q): q is drawn at 1/sqrt(d) * log2(e) so that the scores have the usual spread"""
    from more4d_amd import ops
    D, dt = 128, torch.bfloat16
    C = n * D
    g = torch.Generator().manual_seed(seed)
    qq = torch.randn(B, Lq, n, D, generator=g) * (D ** -0.5 / LN2)
    kk, vv = (torch.randn(B, L, n, D, generator=g) for L in (Lk, Lk))
    dd = torch.randn(B, Lq, n, D, generator=g)
    if spikes:
        kk[0, Lk // 2 + 3, 0] = qq[0, 5, 0] * 60.0
        kk[0, Lk - 1, n - 1] = qq[0, Lq - 1, n - 1] * 50.0
    Lkp = (Lk + 7) // 8 * 8
    # q, k, v as column slices of one [rows, 3C] buffer when Lq == Lkp (the training layout: strided rows), else separate
    q = qq.reshape(B * Lq, C).to(dt).to(DEV)
    k = torch.full((B, Lkp, C), float("nan"), dtype=dt)     # rows beyond Lk must never be consumed
    v = torch.full((B, Lkp, C), float("nan"), dtype=dt)
    k[:, :Lk] = kk.reshape(B, Lk, C).to(dt)
    v[:, :Lk] = vv.reshape(B, Lk, C).to(dt)
    k, v = k.reshape(B * Lkp, C).to(DEV), v.reshape(B * Lkp, C).to(DEV)
    d_o = dd.reshape(B * Lq, C).to(dt).to(DEV)
    vt = torch.nan_to_num(v.float()).to(dt).t().contiguous()
    lse = torch.empty(B, n, Lq, device=DEV)
    o = ops.attention(q, [ops.KV(k, vt, Lkp * C, C, Lkp, B * Lkp, Lk)], B=B, Lq=Lq, heads=n, head_dim=D, q_bs=Lq * C, q_ls=C,
                      lse=lse, scale=LN2).view(B * Lq, C)
    dq, dk, dv = torch.full_like(q, float("nan")), torch.full_like(k, float("nan")), torch.full_like(v, float("nan"))
    ops.launch_counts(reset=True)
    ops.attention_bwd(q, k, v, o, d_o, lse, B=B, Lq=Lq, Lk=Lk, Lk_rows=Lkp, heads=n, head_dim=D, dq=dq, dk=dk,
                      dv=dv, scale=LN2)
    torch.cuda.synchronize()
    cnt = {a: b for a, b in ops.launch_counts().items() if b}
    # fp32 reference on the bf16-rounded operands
    qf = q.float().view(B, Lq, n, D).permute(0, 2, 1, 3).requires_grad_(True)
    kf = k.float().view(B, Lkp, n, D)[:, :Lk].permute(0, 2, 1, 3).requires_grad_(True)
    vf = torch.nan_to_num(v.float()).view(B, Lkp, n, D)[:, :Lk].permute(0, 2, 1, 3).requires_grad_(True)
    s = (qf @ kf.transpose(-1, -2)) * LN2
    ref = torch.softmax(s, -1) @ vf
    gq, gk, gv = torch.autograd.grad(ref, (qf, kf, vf), d_o.float().view(B, Lq, n, D).permute(0, 2, 1, 3))
    gq = gq.permute(0, 2, 1, 3).reshape(B * Lq, C)
    gk, gv = (x.permute(0, 2, 1, 3).reshape(B, Lk, C) for x in (gk, gv))
    bad = False
    msg = []
    for nm, got, want in (("dq", dq, gq), ("dk", dk.view(B, Lkp, C)[:, :Lk], gk), ("dv", dv.view(B, Lkp, C)[:, :Lk], gv)):
        fin = bool(torch.isfinite(got.float()).all())
        e = float((got.float() - want).abs().max() / want.abs().max())
        bad |= not fin or not e < 1.5e-2
        msg.append(f"{nm} err {e:.3e}{'' if fin else ' NONFINITE'}")
    pad = float(dk.view(B, Lkp, C)[:, Lk:].float().abs().sum() + dv.view(B, Lkp, C)[:, Lk:].float().abs().sum()) if Lkp > Lk else 0.0
    bad |= pad != 0.0
    digest = " ".join(f"{float(x.float().abs().double().nan_to_num().sum()):.9e}" for x in (dq, dk.view(B, Lkp, C)[:, :Lk], dv.view(B, Lkp, C)[:, :Lk]))
    print(f"B={B} n={n} Lq={Lq} Lk={Lk} spikes={spikes}: {', '.join(msg)} pad {pad} digest {digest}", cnt, "FAIL" if bad else "ok", flush=True)
    return bad, torch.cat([dq.float().flatten(), dk.view(B, Lkp, C)[:, :Lk].float().flatten(), dv.view(B, Lkp, C)[:, :Lk].float().flatten()])


MODES = ("0", "3")       # M4D_ATTN_BWD64: 0 = the two-waves-per-SIMD kernels, 1 = dq64 only, 3 = dq64 + kv64 (default)
CASES = ((1, 8, 1280, 2048), (1, 8, 1100, 2080), (2, 3, 700, 2300, 1, True), (2, 4, 2080, 2080), (1, 16, 4100, 4099, 2, True),
         (1, 2, 256, 21840))


def main():
    if len(sys.argv) > 1 and sys.argv[1] == "--child":
        mode, what = sys.argv[2], sys.argv[3]
        os.environ["M4D_ATTN_BWD64"] = mode
        from more4d_amd import ops
        if what == "time":
            B, n, L, D = 1, 40, 21840, 128
            C = n * D
            g = torch.Generator(device=DEV).manual_seed(0)
            qkv = (torch.randn(B * L, 3 * C, device=DEV, generator=g) * 1.0).to(torch.bfloat16)
            qkv[:, :C] *= D ** -0.5 / LN2
            q, k, v = qkv[:, :C], qkv[:, C:2 * C], qkv[:, 2 * C:]
            d_o = torch.randn(B * L, C, device=DEV, generator=g).to(torch.bfloat16)
            vt = ops.transpose(v.contiguous())
            lse = torch.empty(B, n, L, device=DEV)
            o = ops.attention(q.contiguous(), [ops.KV(k.contiguous(), vt, L * C, C, L, B * L, L)], B=B, Lq=L, heads=n, head_dim=D, q_bs=L * C,
                              q_ls=C, lse=lse, scale=LN2).view(B * L, C)
            dqkv = torch.empty_like(qkv)
            kw = dict(B=B, Lq=L, Lk=L, Lk_rows=L, heads=n, head_dim=D, dq=dqkv[:, :C], dk=dqkv[:, C:2 * C], dv=dqkv[:, 2 * C:], scale=LN2)
            for _ in range(2):
                ops.attention_bwd(q, k, v, o, d_o, lse, **kw)
            torch.cuda.synchronize()
            reps = 10
            t0 = time.perf_counter()
            for _ in range(reps):
                ops.attention_bwd(q, k, v, o, d_o, lse, **kw)
            torch.cuda.synchronize()
            ms = (time.perf_counter() - t0) / reps * 1e3
            tf = 10 * B * L * L * C / ms / 1e9
            print(f"mode {mode}: bwd {ms:.3f} ms  {tf:.0f} TF  frac {tf / 2500:.3f}  dqkv digest {float(dqkv.float().abs().double().sum()):.9e}",
                  flush=True)
            return
        bad = False
        outs = []
        for a in CASES:
            b_, dq = case(*a)
            bad |= b_
            outs.append(dq.cpu())
        torch.save(outs, f"/tmp/bwd64_mode{mode}.pt")
        print("RESULT mode", mode, "FAIL" if bad else "PASS", flush=True)
        sys.exit(1 if bad else 0)
    rc = 0
    for mode in MODES:
        rc |= subprocess.run([sys.executable, __file__, "--child", mode, "check"]).returncode
    try:
        a, b = torch.load("/tmp/bwd64_mode0.pt"), torch.load(f"/tmp/bwd64_mode{MODES[-1]}.pt")
        for i, (x, y) in enumerate(zip(a, b)):
            d = float((x.float() - y.float()).abs().max() / x.float().abs().max())
            print(f"case {i}: new vs old kernel max rel diff {d:.3e}", "FAIL" if not d < 1.2e-2 else "ok")
            rc |= int(not d < 1.2e-2)
    except Exception as ex:      # noqa: BLE001
        print("compare failed:", ex)
        rc |= 1
    if "--time" in sys.argv:
        for rnd in range(2):
            for mode in MODES + ("1",):
                subprocess.run([sys.executable, __file__, "--child", mode, "time"])
    sys.exit(rc)


if __name__ == "__main__":
    main()
