#!/bin/bash
# PMC passes over a micro-benchmark (GPU box).  usage: tools/prof_kernel.sh <tag> <python script + args...>
# Each --pmc group is its own run; only --kernel-trace is combined with counters.
TAG=$1; shift
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/pmc_$TAG
rm -rf $OUT; mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
i=0
for G in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS GRBM_GUI_ACTIVE" \
         "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL SQ_INSTS_LDS SQ_INSTS_VALU SQ_INSTS_MFMA SQ_ACTIVE_INST_LDS" \
         "TCC_HIT_sum TCC_MISS_sum TCP_TCC_READ_REQ_sum" ; do
  i=$((i+1))
  rocprofv3 --kernel-trace --output-format csv --pmc $G -d $OUT/g$i -o p -- python $R/"$@" > $OUT/g$i.log 2>&1
done
python - <<PY
import csv, collections, glob
for f in sorted(glob.glob("$OUT/g*/p_counter_collection.csv")):
    agg = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.defaultdict(set)
    for x in csv.DictReader(open(f)):
        k = x["Kernel_Name"]
        if "gemm" in k or "attn" in k or "conv" in k:
            k = k[:70]
            agg[k][x["Counter_Name"]] += float(x["Counter_Value"]); n[k].add(x["Dispatch_Id"])
    for k, v in agg.items():
        print(k, "launches", len(n[k]))
        for c, val in sorted(v.items()):
            print("    %-28s %.4g  (per launch %.4g)" % (c, val, val / len(n[k])))
PY
