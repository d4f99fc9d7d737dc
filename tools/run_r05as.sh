#!/bin/sh
mkdir -p gpurun_out/r05as
run() { env "$@" python bench.py --no-cpu-baseline --steps 40 2>gpurun_out/r05as/err.log > gpurun_out/r05as/b.json; python -c "import json,sys; d=json.load(open('gpurun_out/r05as/b.json')); print(' '.join(sys.argv[1:]) or 'default', round(d['ms_per_step'],3))" "$@"; }
for i in 1 2 3; do
run A=default_512_768
run FGNN_BT_GRID_APPLY=512
run FGNN_BT_GRID_APPLY=1024
run FGNN_BT_GRID=768
done
python -m pytest tests/test_block_tail_gpu.py -x -q -m gpu 2>&1 | tail -2
