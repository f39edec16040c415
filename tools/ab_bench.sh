#!/bin/bash
# Sustained A/B of an environment switch inside bench.py (same box, alternating): tools/ab_bench.sh VAR v0 v1 [v2 ...]
VAR=$1; shift
for rep in 1 2; do
  for v in "$@"; do
    printf "%s=%s " "$VAR" "$v"
    env "$VAR=$v" timeout 300 python bench.py --no-cpu-baseline --no-secondary --steps 3 2>&1 | tail -1 | \
      python -c "import sys,json; d=json.loads(sys.stdin.read()); print('ms/step %.1f attn %.1f gemm %.1f finite %s' % (d['ms_per_step'], d['roofline_attention']['achieved'], d['roofline']['achieved'], d['finite']))"
  done
done
