#!/usr/bin/env python
"""Does filling the partial tile round pay at a rank's shard size?  M = 5 460 (cfg2 x sp4 / sp8 ranks) and M = 2 730:
440 tiles of 256 x 256 on 256 CUs = 1.72 -> 2 rounds.  Same box, alternating, clock and power sampled over each arm (bench.py: ClockMonitor):
    wide      the production kernel (gemm_bt256w, persistent form)                                   M4D_GEMM_VARIANT=5
    phased    the round-2/3 kernel (gemm_bt256p), one launch                                         M4D_GEMM_VARIANT=4
    phased+T  the same kernel with its split-K TAIL (full rounds, then the remaining tiles K-split over all CUs into an fp32
              workspace + a fix-up kernel: the hybrid the stream-K proposal amounts to)             M4D_GEMM_VARIANT=4 M4D_GEMM_TAIL=1
phased vs phased+T isolates the effect of the tail on ONE kernel; wide shows where the production kernel stands.
    python tools/ab_gemm_tail.py"""
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
SHAPES = ((5460, 5120, 5120), (5460, 13824, 5120), (5460, 5120, 13824), (2730, 5120, 5120), (43680, 5120, 5120))
ARMS = (("wide", {"M4D_GEMM_VARIANT": "5"}), ("phased", {"M4D_GEMM_VARIANT": "4"}), ("phased+T", {"M4D_GEMM_VARIANT": "4", "M4D_GEMM_TAIL": "1"}))


def child(name):
    import torch
    from bench import ClockMonitor
    from more4d_amd import ops
    for M, N, K in SHAPES:
        a = (torch.randn(M, K, device="cuda") * 0.5).bfloat16()
        w = (torch.randn(N, K, device="cuda") * K ** -0.5).bfloat16()
        b = torch.zeros(N, device="cuda", dtype=torch.bfloat16)
        out = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
        ops.launch_counts(reset=True)
        for _ in range(10):
            ops.gemm_bt(a, w, b, out=out)
        torch.cuda.synchronize()
        cls = [k for k, v in ops.launch_counts().items() if v]
        mon = ClockMonitor(0).start()
        n = 0
        t0 = time.perf_counter()
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        while time.perf_counter() - t0 < 3.0:
            for _ in range(100):
                ops.gemm_bt(a, w, b, out=out)
            n += 100
            torch.cuda.synchronize()
        e.record()
        torch.cuda.synchronize()
        mon.stop()
        ms = s.elapsed_time(e) / n
        r = mon.region()
        ck, pw = r.get("clock_mhz", {}).get("mean", float("nan")), r.get("socket_power_w", {}).get("mean", float("nan"))
        print(f"{name:9s} {M}x{N}x{K}: {ms * 1e3:8.1f} us  {2 * M * N * K / ms / 1e9:5.0f} TF  clock {ck:.0f} MHz  power {pw:.0f} W  {cls}"
              f"  digest {float(out.float().abs().sum()):.6e}", flush=True)


if __name__ == "__main__":
    if len(sys.argv) > 2 and sys.argv[1] == "--child":
        child(sys.argv[2])
        sys.exit(0)
    for rep in range(2):
        for name, env in ARMS:
            subprocess.run([sys.executable, __file__, "--child", name], env={**os.environ, **env})
