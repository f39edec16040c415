#!/usr/bin/env python
"""Per-rank compute time of the N-GPU denoise step on ONE GPU: the sequence-parallel group is replaced by a stand-in whose
"all-gather" replicates the local shard (same shapes, same kernels, no xGMI traffic).  step_time(1 GPU) / (N * this) is the
scaling the compute alone allows; the difference to the driver's measured N-GPU number is communication + skew.
    python tools/bench_shard.py --world 8 --mode cfg-sp|sp [--steps 2]"""
import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from bench import CFG_14B, ClockMonitor, build_model, standin_group  # noqa: E402
import more4d_amd.models.wan_transformer4d as _wt  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--world", type=int, default=8)
    ap.add_argument("--mode", choices=["sp", "cfg-sp"], default="cfg-sp")
    ap.add_argument("--steps", type=int, default=2)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--layers", type=int, default=40)
    args = ap.parse_args()
    dev = torch.device("cuda", 0)

    spw = args.world // 2 if args.mode == "cfg-sp" else args.world
    B = 1 if args.mode == "cfg-sp" else 2

    cfg = dict(CFG_14B)
    cfg["num_layers"] = args.layers
    model = build_model(cfg, dev, torch.bfloat16)
    if spw > 1:
        model.sp_world_size, model.sp_world_rank, model._sp = spw, 0, standin_group(spw)
    g = torch.Generator(device=dev).manual_seed(1234)
    F_, H_, W_ = 13, 60, 104
    x = torch.randn(B, 16, F_, H_, W_, generator=g, device=dev).bfloat16()
    y = torch.randn(B, 48, F_, H_, W_, generator=g, device=dev).bfloat16()
    full_ref = torch.randn(B, 16, H_, W_, generator=g, device=dev).bfloat16()
    ctx = [torch.randn(512, 4096, generator=g, device=dev) for _ in range(B)]
    clip = torch.randn(B, 257, 1280, generator=g, device=dev)
    Lv = F_ * (H_ // 2) * (W_ // 2)
    t = torch.full((B,), 500.0, device=dev)
    with torch.no_grad():
        cc = model.prepare_context(ctx, clip)
        for _ in range(args.warmup):
            model(x=x, t=t, context=cc, seq_len=Lv, y=y, full_ref=full_ref)
        torch.cuda.synchronize()
        mon = ClockMonitor(0).start()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            out = model(x=x, t=t, context=cc, seq_len=Lv, y=y, full_ref=full_ref)
        t_host = (time.perf_counter() - t0) / args.steps       # host time to ENQUEUE a step (must stay below the GPU time)
        torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / args.steps
    mon.stop()
    ck = mon.region()
    import hashlib
    digest = hashlib.sha1(out.float().cpu().numpy().tobytes()).hexdigest()[:12]
    print(json.dumps({"world": args.world, "mode": args.mode, "sp_world": spw, "batch_per_rank": B,
                      "effective_clock_mhz": ck.get("effective_clock_mhz"), "socket_power_w": ck.get("socket_power_w", {}).get("mean"),
                      "rank_ms_per_step": dt * 1e3, "host_enqueue_ms_per_step": t_host * 1e3, "finite": bool(torch.isfinite(out.float()).all()),
                      "sp_overlap": int(_wt._SP_OVERLAP), "digest": digest}))


if __name__ == "__main__":
    main()
