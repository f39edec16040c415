#!/bin/bash
# Same-box A/B of two builds of the library: tools/ab_lib.sh <tag> <python script + args...>   (lib/libmore4d_hip_<tag>.so vs the
# shipping library, alternating, two rounds).  Build the side library from another checkout and copy it to more4d_amd/lib/.
TAG=$1; shift
for rep in 1 2; do
  echo "== $TAG"; M4D_LIB=$TAG timeout 600 python "$@" 2>&1 | tail -1
  echo "== shipping"; timeout 600 python "$@" 2>&1 | tail -1
done
