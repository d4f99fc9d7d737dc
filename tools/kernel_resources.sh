#!/bin/sh
# Tuning aid: per-kernel register / scratch / LDS use of one HIP source (compiled for gfx950 with the Makefile's flags).
#   sh tools/kernel_resources.sh mpconv_fwd_ws.hip [extra flags]
ROOT=$(cd "$(dirname "$0")/.." && pwd)
C=$ROOT/factor-graph-neural-network_amd/csrc
SRC=$1; shift
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -I$ROOT/include -Wno-unused-function -Wno-pass-failed "$@" \
    -Rpass-analysis=kernel-resource-usage -c $C/$SRC -o /tmp/kr_$$.o 2>&1 | grep "remark:" | \
    awk '/Function Name:/ {name=$5} /TotalSGPRs:/ {sg=$4} / VGPRs:/ {v=$4} /AGPRs:/ {a=$4} /ScratchSize/ {s=$5} /Occupancy/ {o=$5} /VGPRs Spill/ {sp=$5} /LDS Size/ {print name, "sgpr", sg, "vgpr", v, "agpr", a, "scratch", s, "spill", sp, "occ", o, "lds", $6}' | c++filt
rm -f /tmp/kr_$$.o
