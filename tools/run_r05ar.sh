#!/bin/sh
mkdir -p gpurun_out/r05ar
run() { env "$@" python bench.py --no-cpu-baseline --steps 40 2>gpurun_out/r05ar/err.log > gpurun_out/r05ar/b.json; python -c "import json,sys; d=json.load(open('gpurun_out/r05ar/b.json')); print(' '.join(sys.argv[1:]) or 'default', round(d['ms_per_step'],3))" "$@"; }
for i in 1 2 3 4; do
run A=default
run FGNN_BT_GRID=512 FGNN_BN_APPLY_GRID=1024
run FGNN_BT_GRID=512
run FGNN_BN_APPLY_GRID=1024
done
