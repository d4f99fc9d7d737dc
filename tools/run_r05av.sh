#!/bin/sh
mkdir -p gpurun_out/r05av
run() { env "$@" python bench.py --no-cpu-baseline --steps 40 2>gpurun_out/r05av/err.log > gpurun_out/r05av/b.json; python -c "import json,sys; d=json.load(open('gpurun_out/r05av/b.json')); print(' '.join(sys.argv[1:]) or 'default', round(d['ms_per_step'],3))" "$@"; }
for i in 1 2; do
run A=default
run FGNN_IID_FUSE_MAX_CIN=256
run FGNN_HEAD_WIDTHS=64,128,256
run FGNN_HEAD_WIDTHS=64
run FGNN_EARLY_JOIN=1
done
