#!/bin/sh
mkdir -p gpurun_out/r05ai
run() { env "$@" python bench.py --no-cpu-baseline 2>gpurun_out/r05ai/err.log > gpurun_out/r05ai/b.json; python -c "import json,sys; d=json.load(open('gpurun_out/r05ai/b.json')); print(' '.join(sys.argv[1:]) or 'default', d['ms_per_step'])" "$@"; }
run A=default
run FGNN_FAC_MERGE_SIDE=2
run FGNN_FAC_MERGE_SIDE=0 FGNN_NODE_SUM_HOME=0
python -m pytest tests/test_assemblies_gpu.py tests/test_parity_pins_gpu.py tests/test_soak_gpu.py tests/test_block_tail_gpu.py tests/test_dp_two_ranks_gpu.py -x -q -m gpu > gpurun_out/r05ai/tests.log 2>&1; tail -3 gpurun_out/r05ai/tests.log
