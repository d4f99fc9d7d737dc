#!/bin/sh
mkdir -p gpurun_out/r05at
run() { env "$@" python bench.py --no-cpu-baseline --steps 40 2>gpurun_out/r05at/err.log > gpurun_out/r05at/b.json; python -c "import json,sys; d=json.load(open('gpurun_out/r05at/b.json')); print(' '.join(sys.argv[1:]) or 'default', round(d['ms_per_step'],3))" "$@"; }
for i in 1 2 3; do
run A=head768
run FGNN_BH_GRID_512=1
done
