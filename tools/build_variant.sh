#!/bin/sh
# Tuning aid: build libfgnn_hip_<tag>.so with ONE source recompiled under extra -D flags (the rest from the normal objects).
#   sh tools/build_variant.sh <tag> <source.hip> "<-D flags>"      then   FGNN_HIP_LIB=.../libfgnn_hip_<tag>.so python tools/kbench.py ...
set -e
ROOT=$(cd "$(dirname "$0")/.." && pwd)
C=$ROOT/factor-graph-neural-network_amd/csrc
TAG=$1; SRC=$2; FLAGS=$3
mkdir -p $C/var_obj
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -I$ROOT/include -Wno-unused-function -Wno-pass-failed $FLAGS -c $C/$SRC -o $C/var_obj/$TAG.o
OBJS=$(ls $C/*.o | grep -v "/${SRC%.hip}.o")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $OBJS $C/var_obj/$TAG.o -o $ROOT/factor-graph-neural-network_amd/fgnn_amd/libfgnn_hip_$TAG.so
echo built libfgnn_hip_$TAG.so
