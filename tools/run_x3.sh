#!/bin/sh
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r06x
mkdir -p $O
cd $R
FGNN_HIP_LIB=$R/factor-graph-neural-network_amd/fgnn_amd/libfgnn_hip_prof.so FGNN_PROF=1 FGNN_EXT_BWD_PIECES=${NP:-2} timeout 300 python tools/xbench.py 1024 2>&1 | grep "prof extq" | tail -16 | tee $O/prof_2.txt
