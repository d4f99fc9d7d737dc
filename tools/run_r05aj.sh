#!/bin/sh
mkdir -p gpurun_out/r05aj
run() { env "$@" python bench.py --no-cpu-baseline 2>gpurun_out/r05aj/err.log > gpurun_out/r05aj/b.json; python -c "import json,sys; d=json.load(open('gpurun_out/r05aj/b.json')); print(' '.join(sys.argv[1:]) or 'default', d['ms_per_step'])" "$@"; }
run A=default
run FGNN_DEFER_TO_END=1
run FGNN_DEFER_TO_END=2
run A=default
