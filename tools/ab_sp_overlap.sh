#!/bin/bash
# Same-box A/B of the T-sharded self-attention: remote-shard calls behind (M4D_SP_OVERLAP=0) or beside (=1) the local call.
# One child process per mode and world, alternating, with clocks; the output digests of the two modes must be equal.
cd "$(dirname "$0")/.."
for rep in 1 2; do
  for w in 8 4 2; do
    for m in 0 1; do
      M4D_SP_OVERLAP=$m timeout 600 python tools/bench_shard.py --world $w --mode cfg-sp --steps 3 --warmup 1 2>&1 | tail -1
    done
  done
done
