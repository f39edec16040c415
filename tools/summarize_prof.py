#!/usr/bin/env python
"""Summarise gpurun_out/prof (written by tools/prof.sh on the GPU box) into profiles/<tag>_*:
kernel stats CSV (verbatim), the bench JSON line, and a PMC table per kernel.
PMC notes (MI355X_MICROARCH.md): SQ_VALU_MFMA_BUSY_CYCLES counts cycles per SIMD (1024 SIMDs);
GRBM_GUI_ACTIVE is summed over the 8 XCDs; FETCH_SIZE (KiB) under-reports wide coalesced reads by 2x
(we print raw and x2) and counts Infinity-Cache hits, i.e. it is L2-miss traffic, not DRAM traffic."""
import collections
import csv
import json
import os
import shutil
import sys

tag = sys.argv[1] if len(sys.argv) > 1 else "r01"
src = sys.argv[2] if len(sys.argv) > 2 else "gpurun_out/prof"
dst = "profiles"
os.makedirs(dst, exist_ok=True)
shutil.copy(os.path.join(src, "trace", "bench_kernel_stats.csv"), os.path.join(dst, f"{tag}_kernel_stats.csv"))
for line in open(os.path.join(src, "trace_bench.log")):
    if line.startswith("{"):
        open(os.path.join(dst, f"{tag}_bench_under_rocprof.json"), "w").write(line)


def short(n):
    for k in ("gemm_bt256w_kernel", "gemm_bt256p_kernel", "gemm_bt256_kernel", "gemm_bt256pp_kernel", "gemm_packed_kernel", "gemm_bt_kernelIDF16b", "gemm_bt_kernelIfE",
              "attn128q_kernel", "attn128x_kernel", "attn128p_kernel", "attn128_kernel<8>", "attn128_kernel<4>", "attn128_kernelILi8", "attn128_kernelILi4", "attn_kernelIDF16bLi128",
              "ln_modulate_kernelIfDF16b", "rmsnorm_rope_kernelIDF16b", "conv_cl_kernel"):
        if k in n:
            return k
    return None


def agg(path):
    out = collections.defaultdict(lambda: collections.defaultdict(float))
    disp = collections.defaultdict(set)
    for x in csv.DictReader(open(path)):
        k = short(x["Kernel_Name"])
        if k:
            out[k][x["Counter_Name"]] += float(x["Counter_Value"])
            disp[k].add(x["Dispatch_Id"])
    return out, {k: len(v) for k, v in disp.items()}


m, nm = agg(os.path.join(src, "pmc_mfma", "bench_counter_collection.csv"))
f, nf = agg(os.path.join(src, "pmc_fetch", "bench_counter_collection.csv"))
w, nw = agg(os.path.join(src, "pmc_write", "bench_counter_collection.csv"))
lines = [f"# {tag}: PMC summary (bench.py --layers 2 --steps 1, one pass per counter group)", "",
         "| kernel | launches | MFMA busy / SIMD-cycles | FETCH_SIZE GiB/launch (raw, x2) | WRITE_SIZE GiB/launch |",
         "|---|---|---|---|---|"]
for k in m:
    gui = m[k]["GRBM_GUI_ACTIVE"] / 8.0
    util = m[k]["SQ_VALU_MFMA_BUSY_CYCLES"] / (gui * 1024) if gui else 0
    fe = f[k]["FETCH_SIZE"] / max(1, nf.get(k, 1)) / 2 ** 20
    wr = w[k]["WRITE_SIZE"] / max(1, nw.get(k, 1)) / 2 ** 20
    lines.append(f"| {k} | {nm[k]} | {util:.3f} | {fe:.3f}, {2 * fe:.3f} | {wr:.3f} |")
open(os.path.join(dst, f"{tag}_pmc_summary.md"), "w").write("\n".join(lines) + "\n")
print("\n".join(lines))
