#!/bin/sh
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r06x
mkdir -p $O
cd $R
timeout 600 python tools/diag_syn_ops.py syn_hop 1024 2>&1 | grep -v "aggregator\|amdgpu.ids" | tee $O/diag_syn_ops.txt | head -60
