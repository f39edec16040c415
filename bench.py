#!/usr/bin/env python
"""bench.py — 4D-STraG denoise-steps/sec at 49x480x832, bf16, Wan2.1-14B-shaped DiT, on N MI355X.

One "step" = one pass of the hot path over one batch of synthetic input = the CFG pair of DiT forwards
(batch 2, L = 21 840 tokens incl. the reference-image row) + classifier-free guidance + Euler update, with
latents / control latents / embedded context resident in HBM before the timed region.  N > 1 shards the
token (= frame-major, T) axis across ranks with an RCCL all-gather of K / V^T per layer (strong scaling:
the N GPUs cooperate on the same sample).

Prints ONE JSON line (rank 0) with the driver's contract plus
  roofline     : the dominant kernel (bf16 MFMA GEMM), algorithmic FLOPs / HIP-event time over the timed region
  cpu_baseline : the CPU oracle (oracle/dit.py, fp32, torch threads) on a bounded sample, N=1 rank 0 only.
"""
import argparse
import json
import math
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

CFG_14B = dict(model_type="i2v", patch_size=(1, 2, 2), text_len=512, in_dim=64, dim=5120, ffn_dim=13824, freq_dim=256,
               text_dim=4096, out_dim=16, num_heads=40, num_layers=40, qk_norm=True, cross_attn_norm=True, eps=1e-6,
               add_ref_conv=True, in_dim_ref_conv=16, use_dino_guidance=False, use_omnimae_guidance=False)
MFMA_BF16_PEAK_TF = 2500.0   # dense, /opt/skills/guides/MI355X_MICROARCH.md


def build_model(cfg, device, dtype):
    """Random-init weights of the named architecture directly on the device (no checkpoint offline)."""
    from more4d_amd.models import WanTransformer4DModel
    with torch.device("meta"):
        m = WanTransformer4DModel(**cfg)
    m = m.to_empty(device=device)
    g = torch.Generator(device=device).manual_seed(0)
    with torch.no_grad():
        for name, p in m.named_parameters():
            if name.endswith("norm_q.weight") or name.endswith("norm_k.weight") or name.endswith("norm_k_img.weight") \
                    or name.endswith("norm3.weight") or ".proj.0.weight" in name or ".proj.4.weight" in name:
                p.fill_(1.0)
            elif name.endswith("bias"):
                p.zero_()
            elif name.endswith("modulation"):
                p.copy_(torch.randn(p.shape, generator=g, device=device) / p.shape[-1] ** 0.5)
            else:
                p.copy_(torch.randn(p.shape, generator=g, device=device) * 0.02)
    m.disable_riflex()   # rebuild the host-side freqs table (it was created on the meta device)
    return m.to(dtype).eval()


def flops_per_forward(cfg, L, B):
    d, ffn, nl = cfg["dim"], cfg["ffn_dim"], cfg["num_layers"]
    ctx = cfg["text_len"] + 257
    gemm = nl * (12 * L * d * d + 4 * L * d * ffn) + 2 * L * (256 + 64) * d
    attn = nl * (4 * L * L * d + 4 * L * ctx * d)
    return B * gemm, B * attn


class ClockMonitor:
    """Shader clock and socket power of ONE GPU, sampled by a thread while a region runs, so that two runs on different boxes (or
    two commits on the same box) can be normalised: the loop is power-limited, boxes differ by +-4 % with identical code.
    Sources, first that works: amdgpu sysfs of the device with torch's PCI address (pp_dpm_sclk: the starred level = current average
    gfx clock; hwmon power1_average / power1_input in microwatts), else `rocm-smi --showclocks --showpower --json` as a child.
    `region()` returns the mean / min / max of the samples taken between start() and stop() — `effective_clock_mhz` is the mean."""

    def __init__(self, index=0, period=0.25):
        import glob
        import threading
        self.period, self.samples, self._run, self._thr = period, [], False, None
        self.source, self._sclk, self._power = None, None, None
        try:
            pr = torch.cuda.get_device_properties(index)
            want = "%04x:%02x:%02x" % (getattr(pr, "pci_domain_id", 0), pr.pci_bus_id, pr.pci_device_id)
        except Exception:
            want = None
        cards = sorted(glob.glob("/sys/class/drm/card[0-9]*/device"))
        cands = [c for c in cards if os.path.exists(os.path.join(c, "pp_dpm_sclk"))]
        if want is not None:
            hit = [c for c in cands if os.path.basename(os.path.realpath(c)).lower().startswith(want)]
            cands = hit or (cands if len(cands) == 1 else [])
        if cands:
            d = cands[0]
            self._sclk = os.path.join(d, "pp_dpm_sclk")
            for f in ("power1_average", "power1_input"):
                g = glob.glob(os.path.join(d, "hwmon", "hwmon*", f))
                if g:
                    self._power = g[0]
                    break
            self.source = "sysfs " + os.path.basename(os.path.realpath(d))
        elif os.path.exists("/opt/rocm/bin/rocm-smi"):
            self.source = f"rocm-smi -d {index}"
        self._index = index
        self._threading = threading

    def _sample(self):
        if self._sclk:
            mhz = None
            with open(self._sclk) as fh:
                for line in fh:
                    if "*" in line:
                        mhz = float("".join(ch for ch in line.split(":")[1] if ch.isdigit() or ch == "."))
            w = None
            if self._power:
                with open(self._power) as fh:
                    w = float(fh.read()) / 1e6
            return mhz, w
        import subprocess
        r = subprocess.run(["/opt/rocm/bin/rocm-smi", "-d", str(self._index), "--showclocks", "--showpower", "--json"],
                           capture_output=True, text=True, timeout=10)
        card = next(iter(json.loads(r.stdout).values()))
        mhz = w = None
        for k, v in card.items():
            if k.startswith("sclk clock speed"):
                mhz = float("".join(ch for ch in str(v) if ch.isdigit() or ch == "."))
            if "Socket" in k and "Power" in k and "(W)" in k:
                w = float(v)
        return mhz, w

    def _loop(self):
        while self._run:
            try:
                self.samples.append((time.perf_counter(),) + tuple(self._sample()))
            except Exception as ex:      # a monitor must never sink the measurement
                self.error = repr(ex)[:200]
                return
            time.sleep(self.period)

    def start(self):
        if self.source is None:
            return self
        self.samples, self._run, self.error = [], True, None
        self._thr = self._threading.Thread(target=self._loop, daemon=True)
        self._thr.start()
        return self

    def stop(self):
        self._run = False
        if self._thr is not None:
            self._thr.join(timeout=15)

    def region(self, t0=None, t1=None):
        """statistics of the samples with t0 <= t <= t1 (perf_counter stamps; None = all)"""
        if self.source is None:
            return {"source": None, "note": "no amdgpu sysfs entry / rocm-smi for this device"}
        ss = [x for x in self.samples if (t0 is None or x[0] >= t0) and (t1 is None or x[0] <= t1)]
        out = {"source": self.source, "samples": len(ss), "period_s": self.period}
        if getattr(self, "error", None):
            out["error"] = self.error
        for i, key in ((1, "clock_mhz"), (2, "socket_power_w")):
            v = [x[i] for x in ss if x[i] is not None]
            if v:
                out[key] = {"mean": sum(v) / len(v), "min": min(v), "max": max(v)}
        if "clock_mhz" in out:
            out["effective_clock_mhz"] = out["clock_mhz"]["mean"]
        return out


class KernelTimer:
    """HIP-event timing of individual launches on the stream they are enqueued on (torch's current stream —
    the one ops.py hands to the C ABI)."""

    def __init__(self):
        self.rec = {}

    def wrap(self, ops_mod, name, flops_fn, class_fn=None):
        """class_fn(*args, **kwargs) -> suffix: launches are also summarised per class as `name:suffix`"""
        orig = getattr(ops_mod, name)
        timer = self

        def timed(*a, **k):
            s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            s.record()
            out = orig(*a, **k)
            e.record()
            rec = (s, e, flops_fn(*a, **k))
            timer.rec.setdefault(name, []).append(rec)
            if class_fn is not None:
                timer.rec.setdefault(f"{name}:{class_fn(*a, **k)}", []).append(rec)
            return out
        setattr(ops_mod, name, timed)
        return orig

    def summary(self):
        out = {}
        for name, lst in self.rec.items():
            ms = sum(s.elapsed_time(e) for s, e, _ in lst)
            fl = sum(f for _, _, f in lst)
            out[name] = dict(launches=len(lst), ms=ms, flops=fl, tflops=fl / ms / 1e9 if ms > 0 else 0.0)
        return out


def cpu_baseline(cfg, L, seconds_hint=30):
    """oracle/dit.py block_forward (fp32) at full 14B width and full L on the host cores: one block,
    extrapolated x num_layers x 2 (CFG) to a step.  Bounded sample: ~20-60 s of CPU work."""
    from oracle import dit as odit

    def block_shapes(dim, ffn):
        s = {"blocks.0.modulation": (1, 6, dim), "blocks.0.norm3.weight": (dim,), "blocks.0.norm3.bias": (dim,),
             "blocks.0.ffn.0.weight": (ffn, dim), "blocks.0.ffn.0.bias": (ffn,), "blocks.0.ffn.2.weight": (dim, ffn),
             "blocks.0.ffn.2.bias": (dim,), "blocks.0.cross_attn.norm_k_img.weight": (dim,)}
        for a in ("self_attn", "cross_attn"):
            for n in ("q", "k", "v", "o") + (("k_img", "v_img") if a == "cross_attn" else ()):
                s[f"blocks.0.{a}.{n}.weight"] = (dim, dim)
                s[f"blocks.0.{a}.{n}.bias"] = (dim,)
            s[f"blocks.0.{a}.norm_q.weight"] = (dim,)
            s[f"blocks.0.{a}.norm_k.weight"] = (dim,)
        return s
    torch.manual_seed(0)
    ocfg = odit.DiTConfig(dim=cfg["dim"], ffn_dim=cfg["ffn_dim"], num_heads=cfg["num_heads"], num_layers=1)
    sd = {k: torch.randn(v) * (0.02 if len(v) > 1 else 0.0) + (1.0 if "norm" in k and k.endswith("weight") else 0.0)
          for k, v in block_shapes(cfg["dim"], cfg["ffn_dim"]).items()}
    h, w = 30, 52
    f = L // (h * w)
    Lc = f * h * w
    x = torch.randn(1, Lc, cfg["dim"])
    e0 = torch.randn(1, 6, cfg["dim"]) * 0.1
    ctx = torch.randn(1, 257 + cfg["text_len"], cfg["dim"])
    t0 = time.time()
    with torch.no_grad():
        odit.block_forward(sd, 0, ocfg, x, e0, (f, h, w), ctx)
    dt = time.time() - t0
    step_s = dt * cfg["num_layers"] * 2
    return dict(value=1.0 / step_s, unit="denoise-steps/s", cores=torch.get_num_threads(), kind="port",
                sample=f"one 14B-width WanAttentionBlock (oracle/dit.py, fp32) at L={Lc}: {dt:.1f} s, "
                       f"x{cfg['num_layers']} layers x2 CFG => {step_s:.0f} s/step")


def secondary_figures(model, cfg, dev):
    """The other figures BASELINE.json's metric names, measured after (never inside) the timed region on the same GPU:
    VAE + adaptor round trip at 49x480x832 (configs[2]) and one 14B train step at batch 1 (configs[4] per-GPU work).  Each is
    best-effort: a failure is reported, it never sinks the headline measurement."""
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "tools"))
    out = {}
    try:
        import bench_vae
        vmon = ClockMonitor(dev.index or 0).start()
        try:
            r = bench_vae.run(49, 480, 832, iters=2, dev=dev, verbose=False)
        finally:
            vmon.stop()
        out["vae_roundtrip"] = {"ms": r["roundtrip_ms"], "parts_ms": r["ms"], "tflops": r["tflops"], "finite": r["finite"],
                                "clock": vmon.region(),
                                "workload": "enc-adaptor + encode + decode + dec-adaptor, 49x480x832x3 trajectories, bf16"}
        px = 480 * 832
        fl = px * (6.657e6 + 48 * 5.003e6) + px * (10.748e6 + 48 * 8.445e6)
        tf = fl / ((r["ms"]["encode"] + r["ms"]["decode"]) / 1e3) / 1e12
        out["roofline_vae"] = {"kernel": "conv_halo64_kernel + conv_halo_kernel (80 % of VAE kernel time; weights staged from the tiled copies of m4d_conv_pack_weights) inside vae.encode + vae.decode, whole-call FLOPs / wall time",
                               "bound": "mfma", "achieved": tf, "peak": MFMA_BF16_PEAK_TF, "unit": "TFLOP/s", "frac": tf / MFMA_BF16_PEAK_TF,
                               "algorithmic_tflop": fl / 1e12}
    except Exception as ex:
        out["vae_roundtrip"] = {"error": repr(ex)}
    torch.cuda.empty_cache()
    try:
        import bench_train
        from more4d_amd import ops
        import more4d_amd.autograd as ag
        kt = KernelTimer()

        def bwd_flops(q, k, v, o, d_o, lse, **kk):      # 10 L^2 d per head: S, dP, dV, dQ, dK (2 L^2 d each)
            return 10 * kk["B"] * kk["Lq"] * kk["Lk"] * kk["heads"] * kk["head_dim"]
        orig = kt.wrap(ops, "attention_bwd", bwd_flops)
        mon = ClockMonitor(dev.index or 0).start()
        try:
            r = bench_train.run_train(model, cfg, dev, steps=3, warmup=2)      # (two warm-up steps: a fresh box's first step pays for allocator growth)
        finally:
            ops.attention_bwd = orig
            mon.stop()
        ab = kt.summary().get("attention_bwd", {})
        out["roofline_attention_bwd"] = {
            "kernel": "attn_bwd_kv64_kernel (fused dK / dV, role-split) + attn_bwd_dq64_kernel (dQ), one wave per SIMD, for the self-attention; "
                      "attn_bwd_kvp_kernel + attn_bwd_dqp_kernel for the cross-attention calls; via m4d_attention_bwd (self + cross, 5 train steps "
                      "incl. warm-up)", "bound": "mfma",
            "achieved": ab.get("tflops", 0.0), "peak": MFMA_BF16_PEAK_TF, "unit": "TFLOP/s", "frac": ab.get("tflops", 0.0) / MFMA_BF16_PEAK_TF,
            "launches": ab.get("launches", 0), "flops_convention": "10 B Lq Lk heads head_dim per call"}
        del ag
        # next to 216 GB of live tensors a step occasionally pays for allocator growth (erratic beyond ~230 GB): the
        # figure is the MEDIAN of three timed steps, every step is listed, the mean is kept as s_per_step_mean
        each = sorted(r["each_step_s"])
        med = each[len(each) // 2]
        out["train_step"] = {"s_per_step": med, "s_per_step_mean": r["value"], "each_step_s": r["each_step_s"],
                             "mfma_frac": r["mfma_frac"] * r["value"] / med, "max_mem_gb": r["max_mem_gb"],
                             "stored_blocks": r["stored_blocks"], "clock": mon.region(),
                             "workload": "14B DiT fwd + bwd (+ recompute where activations are not stored) + clip + AdamW, batch 1, "
                                         "L=21840, bf16 params and optimizer state"}
    except Exception as ex:
        out["train_step"] = {"error": repr(ex)}
    return out


PROBE_SHAPES = ((5120, 5120, 6), (13824, 5120, 1), (5120, 13824, 1))      # (N, K, launches per DiT block) of the production GEMM kernel
GEMM_KERNELS = ("gemm_bt256w_kernel", "gemm_bt256p_kernel")               # 4-wave wide kernel (default) / phased kernel (M4D_GEMM_VARIANT=4)


def pmc_probe():
    """`bench.py --pmc-probe` (run by measure_traffic under rocprofv3): the dominant kernel at the bench's shapes — M = 43 680
    rows of the CFG pair, the per-block mix of PROBE_SHAPES — on N(0,1)-scaled operands, nothing else on the device."""
    from more4d_amd import ops
    dev = torch.device("cuda", 0)
    g = torch.Generator(device=dev).manual_seed(0)
    M = 2 * 21840
    for N, K, reps in PROBE_SHAPES:
        a = (torch.randn(M, K, generator=g, device=dev) * 0.5).bfloat16()
        w = (torch.randn(N, K, generator=g, device=dev) * K ** -0.5).bfloat16()
        b = torch.zeros(N, device=dev, dtype=torch.bfloat16)
        for _ in range(reps):
            ops.gemm_bt(a, w, b)
        torch.cuda.synchronize()
        del a, w
    return 0


def measure_traffic(timeout=240):
    """HBM-side traffic of the dominant kernel from the PMC counters, collected as MI355X_MICROARCH.md prescribes: separate
    rocprofv3 passes for FETCH_SIZE and WRITE_SIZE (they do not fit one pass), --kernel-trace only; FETCH_SIZE doubled (gfx950
    tallies 128-byte requests of wide coalesced reads at 64 B), WRITE_SIZE as reported; both count Infinity-Cache hits, i.e.
    they are L2-miss traffic.  Returns bytes per launch, averaged over the per-block launch mix, next to the algorithmic bytes."""
    import csv
    import glob
    import shutil
    import subprocess
    import tempfile
    exe = shutil.which("rocprofv3") or "/opt/rocm/bin/rocprofv3"
    if not os.path.exists(exe):
        return None, {"error": "rocprofv3 not found"}
    tmp = tempfile.mkdtemp(prefix="m4d_pmc_")
    env = dict(os.environ, TMPDIR=tmp)
    per = {}
    try:
        for counter in ("FETCH_SIZE", "WRITE_SIZE"):
            out = os.path.join(tmp, counter)
            cmd = [exe, "--kernel-trace", "--pmc", counter, "--output-format", "csv", "-d", out, "-o", "p", "--", sys.executable,
                   os.path.abspath(__file__), "--pmc-probe"]
            r = subprocess.run(cmd, cwd=tmp, env=env, capture_output=True, text=True, timeout=timeout)
            files = glob.glob(os.path.join(out, "**", "*counter_collection.csv"), recursive=True)
            if r.returncode != 0 or not files:
                return None, {"error": f"rocprofv3 --pmc {counter} failed (rc {r.returncode})", "stderr": r.stderr[-300:]}
            vals = {}
            for row in csv.DictReader(open(files[0])):
                if any(k in row["Kernel_Name"] for k in GEMM_KERNELS) and row["Counter_Name"] == counter:
                    vals[int(row["Dispatch_Id"])] = vals.get(int(row["Dispatch_Id"]), 0.0) + float(row["Counter_Value"])
            if not vals:
                return None, {"error": f"no {' / '.join(GEMM_KERNELS)} rows in the {counter} pass"}
            per[counter] = sum(vals.values()) / len(vals) * 1024.0          # the counters are reported in KiB
            per[counter + "_launches"] = len(vals)
            # the probe launches the shapes in PROBE_SHAPES order, `reps` launches each: bucket the dispatches by that order
            ordered = [vals[k] * 1024.0 for k in sorted(vals)]
            if len(ordered) == sum(r for _, _, r in PROBE_SHAPES):
                i = 0
                for N, K, reps in PROBE_SHAPES:
                    per.setdefault("per_shape", {}).setdefault(f"N{N}_K{K}", {})[counter] = sum(ordered[i:i + reps]) / reps
                    i += reps
    except Exception as ex:
        return None, {"error": repr(ex)}
    finally:
        shutil.rmtree(tmp, ignore_errors=True)
    M = 2 * 21840
    nl = sum(r for _, _, r in PROBE_SHAPES)
    algo = sum(r * 2 * (M * K + N * K + M * N) for N, K, r in PROBE_SHAPES) / nl      # bf16 A + W read once, out written once
    fetch, write = 2.0 * per["FETCH_SIZE"], per["WRITE_SIZE"]
    shapes = {}
    for N, K, reps in PROBE_SHAPES:
        d = per.get("per_shape", {}).get(f"N{N}_K{K}")
        if d and "FETCH_SIZE" in d and "WRITE_SIZE" in d:
            a_ = 2.0 * (M * K + N * K + M * N)
            shapes[f"M{M}_N{N}_K{K}"] = {"launches_per_block": reps, "fetch_x2_bytes": 2.0 * d["FETCH_SIZE"], "write_bytes": d["WRITE_SIZE"],
                                         "algorithmic_bytes": a_, "counter_over_algorithmic": (2.0 * d["FETCH_SIZE"] + d["WRITE_SIZE"]) / a_}
    return fetch + write, {
        "per_shape": shapes,
        "fetch_bytes_per_launch": fetch, "write_bytes_per_launch": write, "algorithmic_bytes_per_launch": algo,
        "counter_over_algorithmic": (fetch + write) / algo, "launches_profiled": per["FETCH_SIZE_launches"],
        "note": "rocprofv3 --kernel-trace --pmc, one pass per counter, inside bench.py after the timed region (bench.py --pmc-probe: the "
                "kernel alone at the bench shapes); FETCH_SIZE x2 (gfx950 correction of MI355X_MICROARCH.md), WRITE_SIZE raw; both are "
                "L2-miss (fabric-side) bytes incl. Infinity-Cache hits, so the ratio is re-fetch through L2, not DRAM traffic"}


def standin_group(spw):
    """A sequence-parallel group whose all-gather replicates the local shard: same shapes and kernels per rank, no xGMI traffic.
    step time with the real group minus step time with this one = the exposed (un-hidden) part of the collectives."""
    from more4d_amd.dist import SequenceParallelGroup

    class _StandIn(SequenceParallelGroup):
        def __init__(self):
            self.group, self.world_size, self.rank = None, spw, 0

        def all_gather(self, x, dim=1):
            return torch.cat([x] * self.world_size, dim=dim)

        def gather_start(self, x):
            x = x.contiguous()
            buf = torch.empty((self.world_size,) + tuple(x.shape), device=x.device, dtype=x.dtype)
            buf.copy_(x.unsqueeze(0).expand_as(buf))     # every peer slot = a copy of the local shard (real values: power)
            return buf, None, x

        def all_to_all(self, x):                          # Ulysses (M4D_SP_MODE=ulysses): chunk j would go to rank j — keep the local copy
            return x.contiguous().clone()
    return _StandIn()


def self_launch(n):
    """`python bench.py --gpus N` without a launcher: re-exec this script under torch.distributed.run, one rank per GPU
    (the reference's multi-GPU entry is `accelerate launch`, train_wan.sh:9; inference wan_transformer4d.py:1187-1198)."""
    import socket
    import subprocess
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.setdefault("OMP_NUM_THREADS", str(max(1, (os.cpu_count() or 8) // n)))
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    return subprocess.run(cmd, env=env).returncode


def launch_check(world, rank, local_rank, parallelism="auto"):
    """Bring the ranks up exactly as a measurement would (RCCL when GPUs are present, gloo otherwise), count them with an
    all-reduce, then build the process groups of BOTH layouts a measurement may use — the one --parallelism selects first, the
    other one the way the second pass of a measurement re-initialises it — and push one K-shaped all-gather, the head all-gather
    and (cfg-sp) one velocity exchange through each, checked against what every rank must receive; the stand-in group of the
    exposed-collective pass is checked for shape compatibility with the real one.  One JSON line on rank 0.  Runs in the
    GPU-less CPU suite at world 2 and 8."""
    import torch.distributed as dist
    gpu = torch.cuda.is_available() and torch.cuda.device_count() >= world
    layouts = {}
    if world > 1:
        if gpu:
            torch.cuda.set_device(local_rank)
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group("gloo")
        dev = torch.device("cuda", local_rank) if gpu else torch.device("cpu")
        ones = torch.ones(1, device=dev)
        dist.all_reduce(ones)
        n = int(ones.item())
        from more4d_amd.dist import cfg_exchange, get_cfg_parallel_rank, init_sequence_parallel
        first = parallelism == "cfg-sp" or (parallelism == "auto" and world % 2 == 0)
        for cfgp in ([first, not first] if world % 2 == 0 else [False]):
            name = "cfg-sp" if cfgp else "sp"
            try:
                sp = init_sequence_parallel(cfg_parallel=cfgp)
                br = get_cfg_parallel_rank()
                W = sp.world_size
                k = torch.full((2, 8), float(rank), device=dev)                      # a K shard: every peer's rows must arrive
                if W > 1:
                    buf, work, _ = sp.gather_start(k)
                    if work is not None:
                        work.wait()
                    base = rank - sp.rank                                            # first global rank of this group
                    ok = all(float(buf[r].mean()) == base + r for r in range(W))
                    ok = ok and tuple(sp.all_gather(k, dim=0).shape) == (2 * W, 8)
                    sb, _, _ = standin_group(W).gather_start(k)                      # the stand-in pass must hand over the same shapes
                    ok = ok and sb.shape == buf.shape
                else:
                    ok = True
                if br is not None:
                    v = cfg_exchange(torch.full((1, 4), float(br), device=dev))
                    ok = ok and v.shape == (2, 4) and float(v[0].mean()) == 0.0 and float(v[1].mean()) == 1.0
                layouts[name] = {"sp_world": W, "cfg_branch_of_rank0": br, "ok": bool(ok)}
            except Exception as ex:
                layouts[name] = {"ok": False, "error": repr(ex)[:300]}
        okt = torch.tensor([1.0 if all(v["ok"] for v in layouts.values()) else 0.0], device=dev)
        dist.all_reduce(okt, op=dist.ReduceOp.MIN)
        layouts["all_ranks_ok"] = bool(okt.item() == 1.0)
        dist.barrier()
        dist.destroy_process_group()
    else:
        n = 1
    if rank == 0:
        print(json.dumps({"launch_check": True, "n_gpus": world, "ranks": n, "backend": "nccl(RCCL)" if gpu else "gloo",
                          "layouts": layouts}))
    return 0 if (world == 1 or layouts.get("all_ranks_ok")) else 4


def train_mode(args, world, rank, local_rank, dev, rccl_ranks, overrides):
    """BASELINE configs[4]: one 14B DiT train step (fwd + recompute + bwd + clip + AdamW) per GPU at batch 1, data parallel over
    the N ranks (weak scaling): value = samples/s of the whole job."""
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import bench_train
    cfg = dict(CFG_14B)
    cfg["num_layers"] = args.layers
    if args.guidance:
        cfg["use_omnimae_guidance"] = True
    model = build_model(cfg, dev, torch.bfloat16)
    r = bench_train.run_train(model, cfg, dev, steps=args.steps, warmup=args.warmup, world=world, rank=rank, local_rank=local_rank,
                              guidance=args.guidance)
    dt = torch.tensor([r["value"]], device=dev, dtype=torch.float64)
    if world > 1:
        import torch.distributed as dist
        dist.all_reduce(dt, op=dist.ReduceOp.MAX)
    dt = float(dt)
    if rank == 0:
        print(json.dumps({
            "metric": "4D-STraG train-step samples/sec (DiT fwd+bwd+AdamW), 49x480x832 bf16", "value": world / dt, "unit": "samples/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": dt * 1e3, "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": "bf16", "data": "synthetic",
            "config": {"workload": "configs[4]: Wan2.1-14B-shaped DiT train step, batch 1 per GPU, latent 1x16x13x60x104, L=21840 tokens "
                                   "incl. the ref row, per-block recompute where activations are not stored, clip + AdamW (bf16 params "
                                   "and state); random-init weights", "layers": args.layers, "parallelism": r["data_parallel"],
                       "spatial_guidance": bool(args.guidance)},
            "valid": args.layers == 40 and not overrides and math.isfinite(r["loss"]), "env_overrides": overrides,
            "rccl_ranks": rccl_ranks, "model_tflop_per_sample": r["model_tflop"], "mfma_frac_whole_step": r["mfma_frac"],
            "max_mem_gb": r["max_mem_gb"], "stored_blocks": r["stored_blocks"], "loss": r["loss"],
            "optimizer_state_gb_per_rank": r.get("optimizer_state_gb_per_rank")}))
    if world > 1:
        import torch.distributed as dist
        dist.destroy_process_group()
    return 0


def m4d_overrides():
    """A/B switches present in the environment: a run with any of them is not the shipping configuration."""
    return sorted(k for k in os.environ if k.startswith("M4D_"))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--layers", type=int, default=40, help="debug only: != 40 marks the result invalid")
    ap.add_argument("--no-ref", action="store_true", help="drop the reference-image row (L=20280)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-kernel-timers", action="store_true")
    ap.add_argument("--no-standin", action="store_true", help="N>1: skip the communication-free re-run that measures the exposed collectives")
    ap.add_argument("--guidance", action="store_true",
                    help="spatial guidance on (use_omnimae_guidance=True, synthetic OmniMAE patch features): the released 4D-STraG "
                         "checkpoint's configuration; the headline number is quoted without it (SURVEY 8d config 2)")
    ap.add_argument("--no-secondary", action="store_true",
                    help="skip the secondary BASELINE.json figures (VAE round trip, train step) measured after the timed region")
    ap.add_argument("--mode", choices=["denoise", "train"], default="denoise",
                    help="denoise (default, BASELINE metric): the 4D-STraG denoise step, N>1 = T-sharded strong scaling; "
                         "train: BASELINE configs[4], one DiT train step per GPU, N>1 = data parallel with bucketed gradient "
                         "reduce-scatter + parameter all-gather over RCCL (weak scaling; value = samples/s)")
    ap.add_argument("--pmc-probe", action="store_true", help="internal: the dominant kernel alone, for the rocprofv3 counter passes")
    ap.add_argument("--no-traffic", action="store_true", help="skip the rocprofv3 PMC passes that fill roofline.traffic")
    ap.add_argument("--launch-check", action="store_true",
                    help="only bring up the N ranks (RCCL on GPUs, gloo without), all-reduce a one per rank, print the count")
    ap.add_argument("--no-other-layout", action="store_true",
                    help="N>1 even: skip the second pass that times the other layout (sp-N <-> cfg2 x sp(N/2)) for secondary.*_layout")
    ap.add_argument("--parallelism", choices=["auto", "sp", "cfg-sp"], default="auto",
                    help="N>1: 'sp' = all ranks shard the tokens of the CFG pair; 'cfg-sp' = the two CFG branches on the two "
                         "halves of the world, tokens sharded inside each half (auto: cfg-sp when N is even)")
    args = ap.parse_args()

    if args.pmc_probe:
        return pmc_probe()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        raise SystemExit(self_launch(args.gpus))       # plain `python bench.py --gpus N`: become the launcher of N ranks
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus != world:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with --nproc-per-node {args.gpus}")
    if args.launch_check:
        return launch_check(world, rank, local_rank, args.parallelism)
    overrides = m4d_overrides()
    from more4d_amd import _lib
    if _lib.ABLATION_BUILD:
        raise SystemExit("bench.py refuses the ablation build of the library (M4D_LIB=abl: kernels that skip work)")
    # M4D_BENCH_ONE_GPU=1 (tool; the line is then marked valid: false like every M4D_* override): all ranks share cuda:0 and talk gloo
    # (device tensors staged through the host) — the whole N-rank code path with the real kernels on a 1-GPU box, where RCCL refuses
    # two ranks on one device.  Timings of such a run mean nothing; it exists to catch a crash before an 8-GPU node does.
    one_gpu = os.environ.get("M4D_BENCH_ONE_GPU", "0") == "1"
    if one_gpu:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    rccl_ranks = 1
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        # bring-up = what --launch-check does; a failure here (no RCCL transport between the GPUs, a rank that never arrives, a
        # sub-group that cannot be created) is reported as ONE JSON diagnostic line instead of a bare non-zero exit code
        stage = "init_process_group(nccl)"
        try:
            if one_gpu:
                dist.init_process_group("gloo")
            else:
                dist.init_process_group("nccl", device_id=dev)
            stage = "all_reduce"
            ones = torch.ones(1, device=dev)
            dist.all_reduce(ones)                       # proves every rank is on the RCCL communicator
            rccl_ranks = int(ones.item())
            if args.mode == "denoise":
                stage = "new_group (sequence / CFG-parallel sub-groups)"
                probe = dist.new_group(list(range(world)))      # (left alive: an idle communicator costs nothing)
                dist.barrier(group=probe)
        except Exception as ex:
            print(json.dumps({"metric": "4D-STraG denoise-steps/sec, 49x480x832 bf16", "value": None, "n_gpus": world, "valid": False,
                              "error": {"stage": stage, "rank": rank, "local_rank": local_rank, "exception": repr(ex)[:600],
                                        "visible_gpus": torch.cuda.device_count(), "backend": "nccl (RCCL)",
                                        "ranks_seen": rccl_ranks,
                                        "env": {k: os.environ.get(k) for k in ("MASTER_ADDR", "MASTER_PORT", "WORLD_SIZE",
                                                                               "HSA_ENABLE_IPC_MODE_LEGACY", "NCCL_DEBUG")}}}),
                  flush=True)
            raise SystemExit(3)
        if rccl_ranks != world and rank == 0:
            print(json.dumps({"warning": f"RCCL all-reduce counted {rccl_ranks} ranks, WORLD_SIZE is {world}"}), file=sys.stderr, flush=True)
    if args.mode == "train":
        return train_mode(args, world, rank, local_rank, dev, rccl_ranks, overrides)

    from more4d_amd import ops
    from more4d_amd.pipeline import denoise_latents
    from more4d_amd.utils.fm_solvers import FlowDPMSolverMultistepScheduler, get_sampling_sigmas, retrieve_timesteps

    cfg = dict(CFG_14B)
    cfg["num_layers"] = args.layers
    if args.guidance:
        cfg["use_omnimae_guidance"] = True
    dtype = torch.bfloat16
    model = build_model(cfg, dev, dtype)
    branch = None

    def set_layout(cfgp):
        """(re)build the process groups of one layout: plain sp-N (every rank shards the tokens of the CFG pair: the layout
        north_star names) or cfg2 x sp(N/2); returns this rank's CFG branch (None = CFG batched on every rank)"""
        from more4d_amd.dist import get_cfg_parallel_rank, init_sequence_parallel
        try:
            init_sequence_parallel(cfg_parallel=cfgp)
        except Exception as ex:     # sub-group creation failed identically on every rank: plain T-sharding over WORLD
            if rank == 0:
                print(f"[bench] cfg-parallel groups unavailable ({ex!r}); falling back to sp{world}", file=sys.stderr)
            init_sequence_parallel(cfg_parallel=False)
        model.enable_multi_gpus_inference()
        return get_cfg_parallel_rank()

    cfgp = False
    if world > 1:
        cfgp = args.parallelism == "cfg-sp" or (args.parallelism == "auto" and world % 2 == 0)
        branch = set_layout(cfgp)

    # synthetic 49x480x832 trajectory latents: [1,16,13,60,104] (+48 control channels, ref row, context)
    g = torch.Generator(device=dev).manual_seed(1234)
    F_, H_, W_ = 13, 60, 104
    lat = torch.randn(1, 16, F_, H_, W_, generator=g, device=dev)
    y = torch.randn(1, 48, F_, H_, W_, generator=g, device=dev)
    full_ref = None if args.no_ref else torch.randn(1, 16, H_, W_, generator=g, device=dev)
    ctx = [torch.randn(512, 4096, generator=g, device=dev), torch.randn(77, 4096, generator=g, device=dev)]
    clip = torch.randn(1, 257, 1280, generator=g, device=dev)
    ffeat = (torch.randn(1, 196, 768, generator=g, device=dev), torch.randn(1, 768, generator=g, device=dev)) if args.guidance else None
    Lv = F_ * (H_ // 2) * (W_ // 2)
    L = Lv + (0 if args.no_ref else (H_ // 2) * (W_ // 2))

    sch = FlowDPMSolverMultistepScheduler(num_train_timesteps=1000, solver_order=1, shift=1.0)
    total_steps = args.warmup + args.steps
    ts, _ = retrieve_timesteps(sch, device=dev, sigmas=get_sampling_sigmas(50, 5.0))
    with torch.no_grad():
        # step-invariant, resident before timing (CFG-parallel ranks hold their own branch only)
        cc = model.prepare_context(ctx, torch.cat([clip, clip])) if branch is None else \
            model.prepare_context([ctx[branch]], clip)

        def run(i0, n, x):
            class _S:   # scheduler view starting at step i0
                @staticmethod
                def step_cfg_(lat_, v, gs, i, round_dtype=torch.float32):
                    return sch.step_cfg_(lat_, v, gs, i0 + i, round_dtype)
            return denoise_latents(model, _S, x, ts[i0:i0 + n], 6.0, cc, y=y, full_ref=full_ref, seq_len=Lv,
                                   first_frame_features=ffeat)

        x = run(0, args.warmup, lat) if args.warmup else lat
        kt = None
        if not args.no_kernel_timers:
            kt = KernelTimer()

            def gemm_flops(a, w, *aa, **kk):
                return 2 * (a.numel() // a.shape[-1]) * (w.numel() // w.shape[-1]) * a.shape[-1]

            def attn_flops(q, segs, *aa, **kk):
                klen = sum(max(0, s.len) for s in segs)
                return 4 * kk["B"] * kk["Lq"] * klen * kk["heads"] * kk["head_dim"]
            originals = {"gemm_bt": kt.wrap(ops, "gemm_bt", gemm_flops), "attention": kt.wrap(ops, "attention", attn_flops,
                                              lambda q, segs, **kk: "cross" if sum(sg.len for sg in segs) < 2048 else "self")}
        mon = ClockMonitor(local_rank).start()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        x = run(args.warmup, args.steps, x)
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        mon.stop()
        clock = mon.region(t0, t0 + dt)
        if kt is not None:
            for name, fn in originals.items():
                setattr(ops, name, fn)
    if world > 1:
        tmax = torch.tensor([dt], device=dev, dtype=torch.float64)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dt = float(tmax)
    ok = bool(torch.isfinite(x).all())
    exposed = None
    if world > 1 and not args.no_standin:
        # the same steps once more with the collectives replaced by local copies (per-rank compute only)
        import more4d_amd.dist as mdist
        real_sp, real_rank, real_exchange = model._sp, model.sp_world_rank, mdist.cfg_exchange
        if model.sp_world_size > 1:
            model._sp, model.sp_world_rank = standin_group(model.sp_world_size), 0
        mdist.cfg_exchange = lambda v: torch.cat([v, v])
        try:
            with torch.no_grad():
                dist.barrier()
                torch.cuda.synchronize()
                t1 = time.perf_counter()
                run(args.warmup, args.steps, x)
                torch.cuda.synchronize()
                dts = torch.tensor([time.perf_counter() - t1], device=dev, dtype=torch.float64)
            dist.all_reduce(dts, op=dist.ReduceOp.MAX)
            exposed = {"standin_ms_per_step": float(dts) / args.steps * 1e3,
                       "exposed_collective_ms_per_step": (dt - float(dts)) / args.steps * 1e3,
                       "note": "same ranks and shapes with every all-gather replaced by a local copy; max over ranks"}
        except Exception as ex:
            exposed = {"error": repr(ex)}
        finally:
            model._sp, model.sp_world_rank, mdist.cfg_exchange = real_sp, real_rank, real_exchange

    def layout_name(br):
        return (f"sp{world} (token/T-sharded, RCCL all-gather K,V^T)" if br is None else
                f"cfg2 x sp{world // 2} (CFG branches on the two halves; tokens T-sharded inside a half, RCCL "
                "all-gather K,V^T; one velocity exchange per step)")

    # N > 1, even: the OTHER layout in the same run, so that one SCALE pass answers both questions — the plain sp-N layout
    # north_star names and the cfg2 x sp(N/2) layout this build prefers.  Same inputs, same steps, its own
    # context cache; the headline `value` stays the layout chosen by --parallelism.
    other = None
    if world > 1 and world % 2 == 0 and not args.no_other_layout:
        def agreed(flag):
            """every rank learns whether ALL ranks got this far (a rank that threw must not leave its peers waiting in a barrier)"""
            t_ = torch.tensor([1.0 if flag else 0.0], device=dev)
            dist.all_reduce(t_, op=dist.ReduceOp.MIN)
            return bool(t_.item() > 0.5)
        err, x2, d2 = None, None, None
        try:
            try:
                br2 = set_layout(not cfgp)
                with torch.no_grad():
                    cc2 = model.prepare_context(ctx, torch.cat([clip, clip])) if br2 is None else model.prepare_context([ctx[br2]], clip)
            except Exception as ex:
                err = repr(ex)[:400]
            if agreed(err is None):

                def run2(i0, n, x_):
                    class _S:
                        @staticmethod
                        def step_cfg_(lat_, v, gs, i, round_dtype=torch.float32):
                            return sch.step_cfg_(lat_, v, gs, i0 + i, round_dtype)
                    return denoise_latents(model, _S, x_, ts[i0:i0 + n], 6.0, cc2, y=y, full_ref=full_ref, seq_len=Lv,
                                           first_frame_features=ffeat)
                try:
                    with torch.no_grad():
                        x2 = run2(0, max(1, args.warmup), lat)
                except Exception as ex:
                    err = repr(ex)[:400]
                if agreed(err is None):           # only now is it safe to meet in a barrier
                    dist.barrier()
                    torch.cuda.synchronize()
                    t2 = time.perf_counter()
                    try:
                        with torch.no_grad():
                            x2 = run2(max(1, args.warmup), args.steps, x2)
                    except Exception as ex:
                        err = repr(ex)[:400]
                    if agreed(err is None):
                        dist.barrier()
                        torch.cuda.synchronize()
                        d2 = torch.tensor([time.perf_counter() - t2], device=dev, dtype=torch.float64)
                        dist.all_reduce(d2, op=dist.ReduceOp.MAX)
            if d2 is not None:
                other = {"parallelism": layout_name(br2), "ms_per_step": float(d2) / args.steps * 1e3, "value": args.steps / float(d2),
                         "unit": "denoise-steps/s", "finite": bool(torch.isfinite(x2).all()), "steps": args.steps}
            else:
                other = {"error": err or "another rank failed in the other-layout pass"}
        finally:
            try:
                branch = set_layout(cfgp)          # the rest of main() (and anything the caller does next) sees the layout it chose
            except Exception as ex:
                other = {"error": f"restoring the layout failed: {ex!r}"[:400]}

    if rank == 0:
        gemm_fl, attn_fl = flops_per_forward(cfg, L, 2)
        step_flops = gemm_fl + attn_fl
        ms = dt / args.steps * 1e3
        out = {
            "metric": "4D-STraG denoise-steps/sec, 49x480x832 bf16", "value": args.steps / dt, "unit": "denoise-steps/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": ms, "higher_is_better": True,
            "scaling": "strong", "vs_baseline": None, "dtype": "bf16", "data": "synthetic",
            "config": {"workload": "configs[1] x CFG: Wan2.1-14B-shaped DiT (40 layers, d=5120, ffn=13824, 40 heads), "
                                   f"latent 1x16x13x60x104 (49x480x832 px), L={L} tokens"
                                   f"{'' if args.no_ref else ' incl. 1560 ref-row tokens'}, CFG batch 2, "
                                   "guidance + Euler fused; random-init weights",
                       "layers": args.layers, "tokens": L, "cfg_batch": 2, "spatial_guidance": bool(args.guidance),
                       "parallelism": "single GPU" if world == 1 else layout_name(branch)},
            "finite": ok, "valid": args.layers == 40 and ok and not overrides, "env_overrides": overrides,
            "rccl_ranks": rccl_ranks, "collectives": exposed,
            "step_tflop": step_flops / 1e12,
            "mfma_frac_whole_step": step_flops / (dt / args.steps) / 1e12 / (MFMA_BF16_PEAK_TF * world),
            # shader clock / socket power of rank 0's GPU over the timed region: what a run has to be normalised by before two boxes
            # are compared (same code: +-4 % box to box, all of it clock).  `mfma_frac_at_clock` = the whole-step MFMA fraction against
            # the peak AT THE CLOCK THE BOARD GRANTED (2.5 PF is quoted at 2.4 GHz): how busy the pipes were, independent of the box.
            "clock": clock,
        }
        if clock.get("effective_clock_mhz"):
            out["effective_clock_mhz"] = clock["effective_clock_mhz"]
            out["mfma_frac_at_clock"] = out["mfma_frac_whole_step"] * 2400.0 / clock["effective_clock_mhz"]
        if kt is not None:
            ks = kt.summary()
            gk = ks.get("gemm_bt", {})
            out["roofline"] = {
                "kernel": ("gemm_bt256p_kernel" if os.environ.get("M4D_GEMM_VARIANT") == "4" else "gemm_bt256w_kernel") +
                          " via m4d_gemm_bt (all DiT projections / FFN, bf16; flops-weighted over its launches)",
                "bound": "mfma", "achieved": gk.get("tflops", 0.0), "peak": MFMA_BF16_PEAK_TF, "unit": "TFLOP/s",
                "frac": gk.get("tflops", 0.0) / MFMA_BF16_PEAK_TF, "traffic": None, "traffic_unit": "bytes per launch",
                "launches": gk.get("launches", 0), "avg_launch_ms": gk.get("ms", 0.0) / max(1, gk.get("launches", 1)),
                "share_of_step_time": gk.get("ms", 0.0) / (dt * 1e3),
            }
            if world == 1 and not args.no_traffic:
                out["roofline"]["traffic"], out["roofline"]["traffic_detail"] = measure_traffic()
            ak = ks.get("attention", {})
            out["roofline_attention"] = {
                "kernel": "attn128q_kernel (self) + attn128x_kernel (cross) via m4d_attention", "bound": "mfma", "achieved": ak.get("tflops", 0.0),
                "peak": MFMA_BF16_PEAK_TF, "unit": "TFLOP/s", "frac": ak.get("tflops", 0.0) / MFMA_BF16_PEAK_TF,
                "launches": ak.get("launches", 0), "share_of_step_time": ak.get("ms", 0.0) / (dt * 1e3),
                "by_class": {c: {"achieved": ks.get(f"attention:{c}", {}).get("tflops", 0.0),
                                 "frac": ks.get(f"attention:{c}", {}).get("tflops", 0.0) / MFMA_BF16_PEAK_TF,
                                 "launches": ks.get(f"attention:{c}", {}).get("launches", 0),
                                 "share_of_step_time": ks.get(f"attention:{c}", {}).get("ms", 0.0) / (dt * 1e3)}
                             for c in ("self", "cross")},
            }
        if world == 1 and not args.no_secondary and args.layers == 40:
            out["secondary"] = secondary_figures(model, cfg, dev)
        if other is not None:
            out.setdefault("secondary", {})["sp_layout" if cfgp else "cfg_sp_layout"] = other
        if world == 1 and not args.no_cpu_baseline:
            try:
                out["cpu_baseline"] = cpu_baseline(cfg, L)
            except Exception as ex:  # the baseline must never sink the measurement
                out["cpu_baseline"] = {"error": repr(ex)}
        print(json.dumps(out))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
