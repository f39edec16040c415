#!/bin/bash
# Timing ablations of conv_halo_kernel (tool build: python -m more4d_amd.build --ablations; results are wrong by construction).
# bits: 1 no MFMA, 2 no fragment reads, 4 no weight DMA after the prologue, 8 no halo DMA after the first, 16 no barriers,
#       32 no epilogue
export M4D_LIB=abl
for a in ${@:-0 1 2 3 4 8 12 16 31}; do echo "CONV_ABL=$a"; M4D_CONV_ABL=$a python tools/bench_conv.py 2>&1 | grep -E "^dec 480x832 96->96 x4|^adaptor"; done
