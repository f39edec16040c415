#!/bin/bash
# Same-box A/B of the VAE + adaptor round trip: conv weights staged from the plain order (M4D_CONV_TILED=0) or from the tiled copies
# (default), with and without conv_halo64 on channels-last inputs (M4D_CONV_HALO64=2).  One child per mode, alternating.
cd "$(dirname "$0")/.."
for rep in 1 2; do
  for m in "0 1" "1 1" "1 2"; do
    set -- $m
    echo "== M4D_CONV_TILED=$1 M4D_CONV_HALO64=$2"
    M4D_CONV_TILED=$1 M4D_CONV_HALO64=$2 timeout 600 python tools/bench_vae.py 2>&1 | grep -v amdgpu.ids | tail -4
  done
done
