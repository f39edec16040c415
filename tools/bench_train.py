#!/usr/bin/env python
"""Train-step time of the 14B-shaped DiT (BASELINE.json configs[4] per-GPU work: batch 1, 49x480x832 latents,
fwd + per-block recompute + bwd + clip + AdamW, bf16).  Usage:
    python tools/bench_train.py [--layers 40] [--steps 2] [--warmup 1] [--profile-ops]
With WORLD_SIZE > 1 (torch.distributed.run) the model is wrapped in DDP over RCCL (data parallel, one sample/GPU).
Prints one JSON line: seconds per step, model FLOPs (4x forward) and the MFMA fraction."""
import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from bench import CFG_14B, MFMA_BF16_PEAK_TF, build_model, flops_per_forward  # noqa: E402


def run_train(model, cfg, dev, steps=2, warmup=1, fp32_state=False, act_budget=None, profile_ops=False, world=1, rank=0,
              local_rank=0, guidance=False, dp="sharded"):
    """Time `steps` training steps of `model` (already on `dev`, bf16): fwd + (recompute) + bwd + clip + AdamW.
    world > 1: data parallel, one sample per rank — dp="sharded" = more4d_amd.dist.data_parallel (bucketed reduce-scatter,
    sharded AdamW, parameter all-gather), dp="ddp" = torch DDP's bucketed all-reduce + the replicated optimizer (A/B)."""
    from more4d_amd import ops
    from more4d_amd.optim import AdamW, clip_grad_norm_
    model = model.train()
    for p_ in model.parameters():
        p_.requires_grad_(True)
    model.activation_budget_gb = act_budget
    net = model
    hp = dict(lr=2e-5, weight_decay=3e-2, eps=1e-10, state_dtype=torch.float32 if fp32_state else None)
    sdp = opt = None
    if world > 1 and dp == "sharded":
        from more4d_amd.dist.data_parallel import ShardedDataParallel
        sdp = ShardedDataParallel(model, **hp)
    else:
        if world > 1:
            from torch.nn.parallel import DistributedDataParallel as DDP
            net = DDP(model, device_ids=[local_rank], find_unused_parameters=True, gradient_as_bucket_view=True,
                      bucket_cap_mb=512)
        opt = AdamW(model.parameters(), **hp)

    g = torch.Generator(device=dev).manual_seed(1234 + rank)
    F_, H_, W_ = 13, 60, 104
    lat = torch.randn(1, 16, F_, H_, W_, generator=g, device=dev)
    noise = torch.randn(1, 16, F_, H_, W_, generator=g, device=dev)
    y = torch.randn(1, 48, F_, H_, W_, generator=g, device=dev).bfloat16()
    full_ref = torch.randn(1, 16, H_, W_, generator=g, device=dev).bfloat16()
    ctx = [torch.randn(512, 4096, generator=g, device=dev)]
    clip = torch.randn(1, 257, 1280, generator=g, device=dev)
    Lv = F_ * (H_ // 2) * (W_ // 2)
    L = Lv + (H_ // 2) * (W_ // 2)
    sigma = 0.7
    noisy = ((1 - sigma) * lat + sigma * noise).bfloat16()      # train_wan.py:1926
    target = noise - lat                                        # :1929
    t = torch.tensor([sigma * 1000.0], device=dev)
    extra = {}
    if guidance:    # the released recipe (--use_omnimae_guidance): precomputed OmniMAE patch features, ViT frozen
        extra["first_frame_features"] = (torch.randn(1, 196, 768, generator=g, device=dev), torch.randn(1, 768, generator=g, device=dev))

    timers = {}
    if profile_ops:
        for name in ("gemm_bt", "attention", "attention_bwd", "transpose", "ln_modulate", "ln_modulate_bwd", "rmsnorm_rope",
                     "rmsnorm_rope_bwd_", "colsum", "scale_cast", "resid_gate", "act_bwd_", "unary", "add", "adamw_", "sumsq", "guidance_bwd_",
                     "conv_cl", "bilinear_cl", "add_bcast"):
            orig = getattr(ops, name)

            def wrapped(*a, _o=orig, _n=name, **k):
                s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                s.record()
                r = _o(*a, **k)
                e.record()
                timers.setdefault(_n, []).append((s, e))
                return r
            setattr(ops, name, wrapped)
        import more4d_amd.autograd as ag
        import more4d_amd.models.wan_transformer4d as wt
        ag.ops = ops
        wt.ops = ops

    def step():
        pred = net(x=noisy, t=t, context=ctx, seq_len=Lv, clip_fea=clip, y=y, full_ref=full_ref, **extra)
        diff = pred.float() - target
        loss = (diff * diff * (diff.abs() <= 50).float()).mean()            # custom_mse_loss :1953-1963
        loss.backward()
        if sdp is not None:
            sdp.step(max_norm=0.05, total_norm=sdp.reduce_gradients())
            sdp.zero_grad()
        else:
            clip_grad_norm_(model.parameters(), 0.05, optimizer=opt)
            opt.step()
            opt.zero_grad(set_to_none=False)
        return loss

    for _ in range(warmup):
        step()
    timers.clear()
    torch.cuda.synchronize()
    if world > 1:
        torch.distributed.barrier()
    t0 = time.perf_counter()
    marks = [torch.cuda.Event(enable_timing=True) for _ in range(steps + 1)]
    marks[0].record()
    for i in range(steps):
        loss = step()
        marks[i + 1].record()
    torch.cuda.synchronize()
    if world > 1:
        torch.distributed.barrier()
    dt = (time.perf_counter() - t0) / steps
    gf, af = flops_per_forward(cfg, L, 1)
    # fwd + recompute + bwd: GEMMs 1 + 1 + 2, attention 1 + 1 + 2.5 (five matmuls per pair instead of two)
    model_flops = 4 * gf + 4.5 * af
    out = {"metric": "train-step seconds, 14B DiT fwd+recompute+bwd+AdamW, batch 1/GPU, 49x480x832 bf16",
           "value": dt, "unit": "s/step", "each_step_s": [marks[i].elapsed_time(marks[i + 1]) / 1e3 for i in range(steps)], "n_gpus": world, "layers": cfg["num_layers"], "loss": float(loss.detach()),
           "model_tflop": model_flops / 1e12, "mfma_frac": model_flops / dt / 1e12 / MFMA_BF16_PEAK_TF,
           "max_mem_gb": torch.cuda.max_memory_allocated() / 2 ** 30, "stored_blocks": [model.last_stored_blocks, model.last_full_blocks],
           "state_dtype": "float32" if fp32_state else "bfloat16",
           "data_parallel": "single GPU" if world == 1 else (
               "bucketed reduce-scatter + sharded AdamW + parameter all-gather (RCCL)" if sdp is not None else "torch DDP all-reduce")}
    if sdp is not None:
        out["optimizer_state_gb_per_rank"] = sdp.state_bytes() / 2 ** 30
        sdp.close()
    if timers:
        torch.cuda.synchronize()
        out["ops_ms_per_step"] = {k: round(sum(s.elapsed_time(e) for s, e in v) / steps, 2) for k, v in
                                  sorted(timers.items(), key=lambda kv: -sum(s.elapsed_time(e) for s, e in kv[1]))}
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--layers", type=int, default=40)
    ap.add_argument("--steps", type=int, default=2)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--fp32-state", action="store_true")
    ap.add_argument("--act-budget", type=float, default=None, help="GB of stored activations (default: automatic; 0 = recompute)")
    ap.add_argument("--profile-ops", action="store_true", help="HIP-event time per ops.* entry point (adds syncs)")
    ap.add_argument("--guidance", action="store_true", help="train with spatial guidance on (use_omnimae_guidance, random-init gates)")
    ap.add_argument("--dp", choices=["sharded", "ddp"], default="sharded")
    args = ap.parse_args()
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        dist.init_process_group("nccl", device_id=dev)

    cfg = dict(CFG_14B)
    cfg["num_layers"] = args.layers
    if args.guidance:
        cfg["use_omnimae_guidance"] = True
    model = build_model(cfg, dev, torch.bfloat16)
    out = run_train(model, cfg, dev, steps=args.steps, warmup=args.warmup, fp32_state=args.fp32_state,
                    act_budget=args.act_budget, profile_ops=args.profile_ops, world=world, rank=rank, local_rank=local_rank,
                    guidance=args.guidance, dp=args.dp)
    out["guidance"] = args.guidance
    if rank == 0:
        print(json.dumps(out))


if __name__ == "__main__":
    main()
