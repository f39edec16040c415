#!/usr/bin/env python
"""Same-box A/B of ln_modulate's two forms (M4D_LN_ROWS is read once per process: one child each, alternating): bit comparison + sustained
bandwidth at the DiT's shapes (fp32 residual -> bf16, modulated / affine).  Usage: python tools/ab_ln.py [--reps 2]"""
import hashlib, json, os, subprocess, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def child():
    import torch
    from more4d_amd import ops
    B, L, C = 2, 21840, 5120
    g = torch.Generator(device="cuda").manual_seed(3)
    x = torch.randn(B * L, C, device="cuda", generator=g) * 3 + 0.5
    sc = torch.randn(B, 6, C, device="cuda", generator=g) * 0.3
    w = torch.randn(C, device="cuda", generator=g)
    bb = torch.randn(C, device="cuda", generator=g)
    res = {}
    for name, kw, od in (("modulate f32->bf16", dict(shift=sc[:, 0], scale=sc[:, 1], mod_stride=6 * C, rows_per_sample=L), torch.bfloat16),
                         ("affine f32->bf16", dict(ln_w=w, ln_b=bb), torch.bfloat16)):
        out = ops.ln_modulate(x, od, **kw)
        dig = hashlib.sha1(out.view(torch.int16).cpu().numpy().tobytes()).hexdigest()[:12]
        for _ in range(5):
            ops.ln_modulate(x, od, out=out, **kw)
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize()
        s.record()
        n = 60
        for _ in range(n):
            ops.ln_modulate(x, od, out=out, **kw)
        e.record()
        torch.cuda.synchronize()
        us = s.elapsed_time(e) / n * 1e3
        res[name] = dict(us=round(us, 1), tbs=round((x.numel() * 4 + out.numel() * 2) / us / 1e6, 2), digest=dig)
    print("RESULT " + json.dumps(res), flush=True)


if __name__ == "__main__":
    if "--child" in sys.argv:
        child()
    else:
        for rep in range(2):
            for v in ("0", "1"):
                r = subprocess.run([sys.executable, os.path.abspath(__file__), "--child"], env=dict(os.environ, M4D_LN_ROWS=v), capture_output=True, text=True, timeout=600)
                line = [l for l in r.stdout.splitlines() if l.startswith("RESULT ")]
                print(f"M4D_LN_ROWS={v}: " + (line[0][7:] if line else f"FAILED {r.stderr[-800:]}"), flush=True)
