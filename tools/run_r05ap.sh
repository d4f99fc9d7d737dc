#!/bin/sh
# round 5: in-step sweep of the streaming kernels' workgroup counts
mkdir -p gpurun_out/r05ap
run() { env "$@" python bench.py --no-cpu-baseline 2>gpurun_out/r05ap/err.log > gpurun_out/r05ap/b.json; python -c "import json,sys; d=json.load(open('gpurun_out/r05ap/b.json')); print(' '.join(sys.argv[1:]) or 'default', round(d['ms_per_step'],3))" "$@"; }
run A=default
run FGNN_BN_APPLY_GRID=2048
run FGNN_BN_APPLY_GRID=8192
run FGNN_BN_GRID=256
run FGNN_BN_GRID=1024
run FGNN_BT_GRID=512
run FGNN_BT_GRID=2048
run FGNN_LF_GRID=512
run FGNN_LF_GRID=2048
run FGNN_WB_GRID=128
run A=default
