#!/usr/bin/env python
"""attn128q_kernel (M4D_ATTN_Q64=1) against an fp32 torch reference and against the production phased kernel, then timing.
    python tools/check_q64.py [--time]"""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
DEV = "cuda"


def run(mode, q, seg, **kw):
    os.environ["M4D_ATTN_Q64"] = mode
    from more4d_amd import ops
    return ops.attention(q, [seg], **kw)


def case(B, n, Lq, Lk, spikes=False, lse=False, seed=0):
    from more4d_amd import ops
    D, dt = 128, torch.bfloat16
    C = n * D
    g = torch.Generator().manual_seed(seed)
    qq, kk, vv = (torch.randn(B, L, n, D, generator=g) for L in (Lq, Lk, Lk))
    if spikes:
        kk[0, Lk // 2 + 3, 0] = qq[0, 5, 0] * 6.0
        kk[0, Lk - 1, n - 1] = qq[0, Lq - 1, n - 1] * 5.0
        kk[0, 70, 0] = qq[0, 300, 0] * 4.0
    Lkp = (Lk + 7) // 8 * 8
    kd = torch.zeros(B, Lkp, C, dtype=dt)
    kd[:, :Lk] = kk.reshape(B, Lk, C).to(dt)
    vt = torch.full((C, B * Lkp), float("nan"), dtype=dt)
    for b in range(B):
        vt[:, b * Lkp:b * Lkp + Lk] = vv[b].reshape(Lk, C).t().to(dt)
    kd, vt = kd.to(DEV), vt.to(DEV)
    seg = ops.KV(kd, vt, Lkp * C, C, Lkp, B * Lkp, Lk)
    qd = qq.reshape(B, Lq, C).to(DEV, dt).contiguous()
    kw = dict(B=B, Lq=Lq, heads=n, head_dim=D)
    qf, kf, vf = (x.to(dt).float().to(DEV).permute(0, 2, 1, 3) for x in (qq, kk, vv))
    s = (qf @ kf.transpose(-1, -2)) * D ** -0.5
    ref = (torch.softmax(s, -1) @ vf).permute(0, 2, 1, 3).reshape(B, Lq, C)
    ref_lse = torch.logsumexp(s, -1) * 1.4426950408889634
    mode = os.environ.get("M4D_ATTN_Q64", "0")
    l = torch.zeros(B, n, Lq, device=DEV) if lse else None
    ops.launch_counts(reset=True)
    o = ops.attention(qd, [seg], lse=l, **kw)
    torch.cuda.synchronize()
    cnt = ops.launch_counts()
    e = float((o.float() - ref).abs().max() / ref.abs().max())
    msg = f"mode {mode} B={B} n={n} Lq={Lq} Lk={Lk} spikes={spikes}: err {e:.3e}"
    if lse:
        msg += "  lse err %.3e" % float((l - ref_lse).abs().max())
    bad = not (e < 8e-3) or not bool(torch.isfinite(o.float()).all()) or (lse and not float((l - ref_lse).abs().max()) < 2e-2)
    print(msg, {k: v for k, v in cnt.items() if v}, "FAIL" if bad else "ok", flush=True)
    return bad


CASES = ((1, 8, 1280, 2048), (1, 8, 1280, 2080, True, True), (2, 3, 1100, 2300, True), (2, 4, 2080, 2080, False, True),
         (1, 16, 4100, 4099, True))


def main():
    # NOTE the switch is read once per process (M4D_ENV_ONCE): every mode runs in its own child
    if len(sys.argv) > 1 and sys.argv[1] == "--child":
        mode, what = sys.argv[2], sys.argv[3]
        os.environ["M4D_ATTN_Q64"] = mode
        from more4d_amd import ops
        D, dt = 128, torch.bfloat16
        if what == "time":
            B, n, L = 2, 40, 21840
            C = n * D
            g = torch.Generator(device=DEV).manual_seed(0)
            q = torch.randn(B, L, C, device=DEV, generator=g).to(dt)
            k = torch.randn(B, L, C, device=DEV, generator=g).to(dt)
            vt = torch.randn(C, B * L, device=DEV, generator=g).to(dt)
            seg = ops.KV(k, vt, L * C, C, L, B * L, L)
            out = torch.empty_like(q)
            for _ in range(3):
                ops.attention(q, [seg], B=B, Lq=L, heads=n, head_dim=D, out=out)
            torch.cuda.synchronize()
            reps = 20
            t0 = time.perf_counter()
            for _ in range(reps):
                ops.attention(q, [seg], B=B, Lq=L, heads=n, head_dim=D, out=out)
            torch.cuda.synchronize()
            ms = (time.perf_counter() - t0) / reps * 1e3
            tf = 4 * B * L * L * C / ms / 1e9
            print(f"mode {mode}: {ms:.3f} ms  {tf:.0f} TF  frac {tf / 2500:.3f}  digest {float(out.float().abs().sum()):.6e}", flush=True)
            return
        bad = False
        for a in CASES:
            bad |= case(*a)
        print("RESULT mode", mode, "FAIL" if bad else "PASS", flush=True)
        sys.exit(1 if bad else 0)
    import subprocess
    rc = 0
    for mode in ("0", "1"):
        rc |= subprocess.run([sys.executable, __file__, "--child", mode, "check"]).returncode
    if "--time" in sys.argv:
        for rnd in range(2):
            for mode in ("0", "1"):
                subprocess.run([sys.executable, __file__, "--child", mode, "time"])
    sys.exit(rc)


if __name__ == "__main__":
    main()
