#!/bin/bash
# Side build for same-box A/B: tools/side_lib.sh <tag> <source.hip> [-DX=Y ...]  compiles ONE source of csrc/ with extra flags (tool
# build, -DM4D_ABLATIONS) and links it with the other objects of the ablation build into more4d_amd/lib/libmore4d_hip_<tag>.so
# (select with M4D_LIB=<tag>; python -m more4d_amd.build --ablations must have run).
set -e
TAG=$1; SRC=$2; shift 2
R=$(cd "$(dirname "$0")/.." && pwd)
O=$R/more4d_amd/build
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -I $R/more4d_amd/csrc -I $R/include -Wno-unused-result -DM4D_ABLATIONS "$@" \
    -c $R/more4d_amd/csrc/$SRC -o $O/$SRC.$TAG.o
OBJS=$(for f in $R/more4d_amd/csrc/*.hip $R/more4d_amd/csrc/*.cpp; do b=$(basename $f); [ "$b" = "$SRC" ] && echo $O/$SRC.$TAG.o || echo $O/$b.abl.o; done)
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $R/more4d_amd/lib/libmore4d_hip_$TAG.so $OBJS
echo built libmore4d_hip_$TAG.so
