#!/usr/bin/env python
"""gpurun_out/prof_vae (tools/prof_vae.sh) -> profiles/<tag>_vae_kernel_stats.csv + profiles/<tag>_vae_pmc_summary.md.
FETCH_SIZE is doubled (gfx950 tallies the 128-byte requests of wide coalesced reads at 64 B, MI355X_MICROARCH.md); both byte counters
are fabric-side (L2-miss) traffic including Infinity-Cache hits."""
import collections
import csv
import glob
import os
import shutil
import sys

tag = sys.argv[1] if len(sys.argv) > 1 else "r02"
src = sys.argv[2] if len(sys.argv) > 2 else "gpurun_out/prof_vae"
os.makedirs("profiles", exist_ok=True)
shutil.copy(glob.glob(os.path.join(src, "trace", "**", "*kernel_stats.csv"), recursive=True)[0], f"profiles/{tag}_vae_kernel_stats.csv")
for line in open(os.path.join(src, "trace.log")):
    if line.startswith("{"):
        open(f"profiles/{tag}_vae_bench_under_rocprof.json", "w").write(line)

import re

FIXED = ("conv_halo64_kernel", "conv_cl256_kernel", "conv_cl_kernel", "rmsnorm_silu_cl_kernel", "groupnorm_apply_kernel", "groupnorm_stats_kernel")
KEYS = []


def short(n):
    m_ = re.search(r"conv_halo_kernel<[^>]*>", n)           # every instantiation (KT, KH, TH, TW, NT, MT) on its own row
    k = m_.group(0) if m_ else next((k for k in FIXED if k in n), None)
    if k and k not in KEYS:
        KEYS.append(k)
    return k


def agg(d):
    out = collections.defaultdict(lambda: collections.defaultdict(float))
    disp = collections.defaultdict(set)
    f = glob.glob(os.path.join(src, d, "**", "*counter_collection.csv"), recursive=True)
    if not f:
        return out, {}
    for x in csv.DictReader(open(f[0])):
        k = short(x["Kernel_Name"])
        if k:
            out[k][x["Counter_Name"]] += float(x["Counter_Value"])
            disp[k].add(x["Dispatch_Id"])
    return out, {k: len(v) for k, v in disp.items()}


m, nm = agg("pmc_mfma")
f, nf = agg("pmc_fetch")
w, nw = agg("pmc_write")
lines = [f"# {tag}: VAE PMC summary (tools/bench_vae.py 17 480 832, one rocprofv3 pass per counter group)", "",
         "| kernel | launches | MFMA busy / SIMD-cycles | FETCH_SIZE MiB/launch (x2-corrected) | WRITE_SIZE MiB/launch |", "|---|---|---|---|---|"]
for k in sorted(KEYS, key=lambda k_: (not k_.startswith("conv_halo"), k_)):
    if k not in m:
        continue
    gui = m[k]["GRBM_GUI_ACTIVE"] / 8.0
    util = m[k]["SQ_VALU_MFMA_BUSY_CYCLES"] / (gui * 1024) if gui else 0
    fe = 2 * f[k]["FETCH_SIZE"] / max(1, nf.get(k, 1)) / 1024
    wr = w[k]["WRITE_SIZE"] / max(1, nw.get(k, 1)) / 1024
    lines.append(f"| {k} | {nm[k]} | {util:.3f} | {fe:.1f} | {wr:.1f} |")
open(f"profiles/{tag}_vae_pmc_summary.md", "w").write("\n".join(lines) + "\n")
print("\n".join(lines))
