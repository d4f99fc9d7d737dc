#!/bin/sh
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r06p
mkdir -p $O
cd $R
timeout 300 python tools/ubench/graph_fork_probe.py --time > $O/probe_time.txt 2>&1
cat $O/probe_time.txt
