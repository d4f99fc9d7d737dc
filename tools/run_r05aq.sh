#!/bin/sh
mkdir -p gpurun_out/r05aq
run() { env "$@" python bench.py --no-cpu-baseline --steps 40 2>gpurun_out/r05aq/err.log > gpurun_out/r05aq/b.json; python -c "import json,sys; d=json.load(open('gpurun_out/r05aq/b.json')); print(' '.join(sys.argv[1:]) or 'default', round(d['ms_per_step'],3))" "$@"; }
run A=default
run FGNN_BT_GRID=256
run FGNN_BT_GRID=384
run FGNN_BT_GRID=512
run FGNN_BT_GRID=640
run A=default
run FGNN_BT_GRID=512
run FGNN_BT_GRID=512 FGNN_BN_APPLY_GRID=2048
run FGNN_BT_GRID=512 FGNN_LF_GRID=768
run FGNN_BT_GRID=512 FGNN_BN_APPLY_GRID=1024
run A=default
