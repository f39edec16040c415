#!/bin/bash
# Timing ablations of a GEMM structure (tool build, wrong results by design): tools/abl_gemm.sh VARIANT "M N K" abl...
# wide kernel (variant 5): 1 no DMA, 2 no fragment reads, 4 no barriers, 8 no MFMA
V=$1; SHAPE=$2; shift 2
for a in "$@"; do
  M4D_LIB=abl M4D_GEMM_VARIANT=$V M4D_GEMM_ABL=$a timeout 120 python tools/time_gemm.py $SHAPE 30 2>&1 | tail -1
done
