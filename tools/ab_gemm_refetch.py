#!/usr/bin/env python
"""What the fabric re-fetch of the production GEMM costs: the persistent 256 x 256 kernel at the bench's three
shapes with its real tile order (5.3 x the algorithmic bytes through L2) against the tool build's ablation `M4D_GEMM_ABL=16` (every
workgroup reads the SAME panels: all L2 hits, ~zero fabric traffic, results wrong by design), alternating on one box, with the
shader clock and socket power sampled over each arm (bench.py: ClockMonitor).
    python tools/ab_gemm_refetch.py            (needs `python -m more4d_amd.build --ablations`; M4D_LIB=abl is set for the children)"""
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
SHAPES = ((43680, 5120, 5120), (43680, 13824, 5120), (43680, 5120, 13824))


def child(abl):
    import torch
    from bench import ClockMonitor
    from more4d_amd import ops
    for M, N, K in SHAPES:
        a = torch.randn(M, K, device="cuda", dtype=torch.bfloat16)
        w = torch.randn(N, K, device="cuda", dtype=torch.bfloat16) * K ** -0.5
        out = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
        for _ in range(10):
            ops.gemm_bt(a, w, None, out=out)
        torch.cuda.synchronize()
        mon = ClockMonitor(0).start()
        n = 0
        t0 = time.perf_counter()
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        while time.perf_counter() - t0 < 4.0:
            for _ in range(50):
                ops.gemm_bt(a, w, None, out=out)
            n += 50
            torch.cuda.synchronize()
        e.record()
        torch.cuda.synchronize()
        mon.stop()
        ms = s.elapsed_time(e) / n
        r = mon.region()
        ck, pw = r.get("clock_mhz", {}).get("mean", float("nan")), r.get("socket_power_w", {}).get("mean", float("nan"))
        print(f"abl {abl:2d}  {M}x{N}x{K}: {ms:.4f} ms  {2 * M * N * K / ms / 1e9:.0f} TF  clock {ck:.0f} MHz  power {pw:.0f} W", flush=True)


if __name__ == "__main__":
    if len(sys.argv) > 2 and sys.argv[1] == "--child":
        child(int(sys.argv[2]))
        sys.exit(0)
    for rep in range(2):
        for abl in (0, 16):
            subprocess.run([sys.executable, __file__, "--child", str(abl)], env={**os.environ, "M4D_LIB": "abl", "M4D_GEMM_ABL": str(abl)})
