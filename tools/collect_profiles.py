#!/usr/bin/env python
"""gpurun_out/final (tools/final_measure.sh) + gpurun_out/prof* -> profiles/<tag>_*: the tracked evidence for the round.
    python tools/collect_profiles.py r03"""
import csv
import glob
import json
import os
import shutil
import subprocess
import sys

tag = sys.argv[1] if len(sys.argv) > 1 else "r03"
src = "gpurun_out/final"
dst = "profiles"
os.makedirs(dst, exist_ok=True)


def cp(a, b):
    if os.path.exists(a):
        shutil.copy(a, os.path.join(dst, f"{tag}_{b}"))
        print("  ", b)


cp(f"{src}/bench.json", "bench.json")
cp(f"{src}/train_bench.json", "train_bench.json")
cp(f"{src}/pytest_gpu.log", "pytest_gpu.log")
cp(f"{src}/bench_shard.log", "bench_shard.log")
cp(f"{src}/gemm_timeline.log", "gemm_timeline.log")
cp(f"{src}/conv_timeline.log", "conv_timeline.log")
cp(f"{src}/ab_gemm.log", "ab_gemm.log")
cp(f"{src}/atomic_probe.log", "atomic_probe.log")
cp(f"{src}/race_screen.log", "race_screen.log")
cp(f"{src}/attn_clock.log", "attn_clock.log")
cp(f"{src}/attn_phases.log", "attn_phases.log")
cp(f"{src}/gemm_traffic_variants.log", "gemm_traffic_variants.log")
cp(f"{src}/mfma_rate_probe.log", "mfma_rate_probe.log")
cp(f"{src}/bw_probe.log", "bw_probe.log")
cp(f"{src}/ab_attn_bwd.log", "ab_attn_bwd_final.log")
cp(f"{src}/vae_train_bench.json", "vae_train_bench.json")
cp(f"{src}/bench_gn_planar.log", "bench_gn_planar.log")
cp(f"{src}/ab_q64_bench.log", "ab_q64_bench.log")
cp(f"{src}/q64_check_time.log", "q64_check_time.log")
cp(f"{src}/q64_stamps.log", "q64_stamps.log")
cp(f"{src}/launch_check_8.log", "launch_check_8.log")
cp(f"{src}/ab_bwd64_train.log", "ab_bwd64_train.log")
cp(f"{src}/bwd64_check_time.log", "bwd64_check_time.log")
cp(f"{src}/ab_gemm_refetch.log", "ab_gemm_refetch.log")
cp(f"{src}/bwd64_pmc_summary.md", "bwd64_pmc_summary.md")
cp(f"{src}/depth40_distance.log", "depth40_distance.log")
cp(f"{src}/conv64_check_time.log", "conv64_check_time.log")
cp(f"{src}/ab_gemm_tail.log", "ab_gemm_tail.log")
cp(f"{src}/ab_conv_tiled.log", "ab_conv_tiled.log")
cp(f"{src}/ab_sp_overlap.log", "ab_sp_overlap.log")
cp(f"{src}/ab_conv_narrow.log", "ab_conv_narrow.log")
for f in glob.glob(f"{src}/vae_train_ks/**/p_kernel_stats.csv", recursive=True):
    cp(f, "vae_train_kernel_stats.csv")
for f in glob.glob(f"{src}/train_ks/**/p_kernel_stats.csv", recursive=True):
    cp(f, "train_4layers_kernel_stats.csv")
subprocess.run([sys.executable, "tools/summarize_prof.py", tag], check=False)
subprocess.run([sys.executable, "tools/summarize_prof_vae.py", tag], check=False)

# per-shape fabric traffic of the production GEMM kernel, from the bench line's own PMC probe (bench.py:measure_traffic)
b = os.path.join(src, "bench.json")
if os.path.exists(b):
    d = json.load(open(b))
    td = d.get("roofline", {}).get("traffic_detail", {})
    lines = [f"# {tag}: fabric-side (L2-miss) traffic of the production GEMM kernel per shape", "",
             "Source: `bench.py`'s own probe (`measure_traffic`: `rocprofv3 --kernel-trace --pmc FETCH_SIZE` / `--pmc WRITE_SIZE`, one pass each,",
             "over `bench.py --pmc-probe` = the kernel alone at M = 43 680 and the per-block launch mix); FETCH_SIZE doubled (gfx950 correction,",
             "MI355X_MICROARCH.md), WRITE_SIZE raw; both include Infinity-Cache hits (re-fetch through L2, not DRAM traffic).", "",
             f"Kernel: `{d['roofline']['kernel'].split(' via')[0]}`; bench line: {d['value']:.4f} steps/s, GEMM {d['roofline']['achieved']:.0f} TF/s "
             f"(frac {d['roofline']['frac']:.3f}).", "",
             "| shape (M x N x K) | launches per block | FETCH x2 (GB) | WRITE (GB) | algorithmic (GB) | counters / algorithmic |", "|---|---|---|---|---|---|"]
    for k, v in td.get("per_shape", {}).items():
        lines.append(f"| {k.replace('_', ' x ').replace('M', '').replace('N', '').replace('K', '')} | {v['launches_per_block']} | "
                     f"{v['fetch_x2_bytes'] / 1e9:.2f} | {v['write_bytes'] / 1e9:.2f} | {v['algorithmic_bytes'] / 1e9:.2f} | {v['counter_over_algorithmic']:.2f} |")
    if "fetch_bytes_per_launch" in td:
        lines += ["", f"Launch-mix average (what `roofline.traffic` reports): fetch {td['fetch_bytes_per_launch'] / 1e9:.2f} GB + write "
                      f"{td['write_bytes_per_launch'] / 1e9:.2f} GB = {(td['fetch_bytes_per_launch'] + td['write_bytes_per_launch']) / 1e9:.2f} GB per launch vs "
                      f"{td['algorithmic_bytes_per_launch'] / 1e9:.2f} GB algorithmic = **{td['counter_over_algorithmic']:.2f}x**."]
    open(os.path.join(dst, f"{tag}_gemm_traffic.md"), "w").write("\n".join(lines) + "\n")
    print("   gemm_traffic.md")
