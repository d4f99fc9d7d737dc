#!/bin/sh
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r06p
mkdir -p $O
cd $R
run() { label=$1; shift; env "$@" FGNN_BENCH_HOST_TIMES=1 timeout 300 python bench.py --no-cpu-baseline --steps ${STEPS:-20} > $O/b_$label.json 2> $O/b_$label.err; echo "== $label: $(grep -A1 'host enqueue' $O/b_$label.err | cut -c1-600)"; }
run base FGNN_X=1
run mainfirst FGNN_MAIN_FIRST=1
run onestream FGNN_NO_SIDE_STREAM=1
run sync FGNN_BENCH_STEP_TIMES=1
STEPS=100 run base100 FGNN_X=1
run base_b FGNN_X=1
run mainfirst_b FGNN_MAIN_FIRST=1
