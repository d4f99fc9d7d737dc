#!/bin/sh
# Run on the GPU box (round 5): per-shape efficiency of the streaming kernels around the operator.
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/${1:-r05f}
mkdir -p $O
cd $R
timeout 300 python tools/lbench.py > $O/lbench.log 2>&1
timeout 300 python tools/wbench.py --bf16 > $O/wbench.log 2>&1
timeout 300 python tools/wbench.py --bf16 --rows=196608 > $O/wbench_48.log 2>&1
timeout 300 python tools/tbench.py > $O/tbench.log 2>&1
timeout 300 python tools/sbench.py > $O/sbench.log 2>&1
timeout 300 python tools/ibench.py > $O/ibench.log 2>&1
tail -30 $O/lbench.log; tail -14 $O/wbench.log; tail -12 $O/tbench.log; tail -30 $O/sbench.log
