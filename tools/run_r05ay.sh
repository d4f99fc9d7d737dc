#!/bin/sh
mkdir -p gpurun_out/r05ay
run() { env "$@" python bench.py --no-cpu-baseline --steps 40 2>gpurun_out/r05ay/err.log > gpurun_out/r05ay/b.json; python -c "import json,sys; d=json.load(open('gpurun_out/r05ay/b.json')); print(' '.join(sys.argv[1:]) or 'default', round(d['ms_per_step'],3))" "$@"; }
for i in 1 2; do
run A=all
run FGNN_FLUSH_MAX=1
run FGNN_FLUSH_MAX=2
run FGNN_FLUSH_MAX=3
run FGNN_FLUSH_MAX=4
done
tail -2 gpurun_out/r05ay/err.log
