#!/bin/sh
mkdir -p gpurun_out/r05ak
run() { env "$@" python bench.py --no-cpu-baseline 2>gpurun_out/r05ak/err.log > gpurun_out/r05ak/b.json; python -c "import json,sys; d=json.load(open('gpurun_out/r05ak/b.json')); print(' '.join(sys.argv[1:]) or 'default', d['ms_per_step'])" "$@"; }
run A=default
run FGNN_MERGED_WGRAD_SIDE=1
run FGNN_NO_MERGED_FAN_WGRADS=1
run A=default
run FGNN_MERGED_WGRAD_SIDE=1
tail -2 gpurun_out/r05ak/err.log
