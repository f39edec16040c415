#!/bin/bash
# Per-kernel time table of a script (GPU box): tools/kstats.sh <tag> <python script + args...>
TAG=$1; shift
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/ks_$TAG
rm -rf $OUT; mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -o p -- python $R/"$@" > $OUT/run.log 2>&1
tail -5 $OUT/run.log
python - <<PY
import csv, glob
for f in glob.glob("$OUT/**/p_kernel_stats.csv", recursive=True):
    rows = list(csv.DictReader(open(f)))
    for r in rows[:14]:
        print("%-90s calls %5s avg %10.1f us  %5.1f%%" % (r["Name"][:90], r["Calls"], float(r["AverageNs"]) / 1e3, float(r["Percentage"])))
PY
find $OUT -name "*_kernel_trace.csv" -size +20M -delete
