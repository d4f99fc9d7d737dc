#!/bin/sh
mkdir -p gpurun_out/r05au
python -m pytest tests/test_block_tail_gpu.py -x -q -m gpu 2>&1 | tail -2
python tools/sbench.py 2>&1 | grep -i "node" | head
run() { env "$@" python bench.py --no-cpu-baseline --steps 40 2>gpurun_out/r05au/err.log > gpurun_out/r05au/b.json; python -c "import json,sys; d=json.load(open('gpurun_out/r05au/b.json')); k=d['kernels'].get('node_sum_kernel',{}); print(' '.join(sys.argv[1:]) or 'default', round(d['ms_per_step'],3), k)" "$@"; }
run A=1
run A=2
run A=3
