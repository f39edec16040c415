#!/bin/bash
# Same-box A/B of two production builds inside bench.py (which refuses M4D_LIB): the shipping library and lib/libmore4d_hip_<tag>.so
# are swapped on disk between runs.  tools/ab_swap.sh <tag> [bench args]
TAG=$1; shift
ARGS="$@"
L=more4d_amd/lib
cp $L/libmore4d_hip.so /tmp/_new.so
run() { timeout 400 python bench.py --no-cpu-baseline --no-secondary --no-traffic --steps 4 --warmup 1 $ARGS 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); r=d['roofline']; a=d['roofline_attention']; print('$1', round(d['ms_per_step'],1), 'gemm', round(r['frac'],4), 'attn', round(a['frac'],4), a.get('by_class'))"; }
for rep in 1 2; do
  cp $L/libmore4d_hip_$TAG.so $L/libmore4d_hip.so; run $TAG
  cp /tmp/_new.so $L/libmore4d_hip.so; run new
done
