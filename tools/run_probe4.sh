#!/bin/sh
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r06p
mkdir -p $O
cd $R
AMD_LOG_LEVEL=4 AMD_LOG_LEVEL_FILE=/tmp/amdlog timeout 300 python tools/ubench/graph_fork_probe.py 4 > $O/probe_log.out 2>&1
for f in /tmp/amdlog*; do tail -c 1500000 $f > $O/amdlog1_tail.txt; done
wc -l $O/amdlog1_tail.txt
