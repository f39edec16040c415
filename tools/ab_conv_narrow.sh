#!/bin/bash
# Same-box A/B: conv_halo64_kernel on the 208-column maps (M4D_CONV_HALO64_NARROW=1, default) or the 24 x 16 kernel there (=0).
cd "$(dirname "$0")/.."
for rep in 1 2; do for m in 0 1; do echo "== M4D_CONV_HALO64_NARROW=$m"; M4D_CONV_HALO64_NARROW=$m timeout 600 python tools/bench_vae.py 2>&1 | grep -v amdgpu.ids | tail -1 | cut -c1-260; done; done
