#!/bin/sh
# round 5: the factor states' gradient merge and the broadcast addend's node sum on the side stream
mkdir -p gpurun_out/r05ah
run() { env "$@" python bench.py --no-cpu-baseline 2>gpurun_out/r05ah/err.log > gpurun_out/r05ah/b.json; python -c "import json,sys; d=json.load(open('gpurun_out/r05ah/b.json')); print(' '.join(sys.argv[1:]) or 'default', d['ms_per_step'])" "$@"; }
run A=0
run FGNN_FAC_MERGE_SIDE=1
run FGNN_NODE_SUM_HOME=1
run FGNN_FAC_MERGE_SIDE=1 FGNN_NODE_SUM_HOME=1
run A=0
tail -3 gpurun_out/r05ah/err.log
