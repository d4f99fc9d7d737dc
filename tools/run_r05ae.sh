#!/bin/sh
# round 5: second form of the fan-in backward
mkdir -p gpurun_out/r05ae
python -m pytest tests/test_mpconv_gpu.py -x -q -m gpu -k "fan_in_backward_second or hyper_edge_backward" > gpurun_out/r05ae/t1.log 2>&1; tail -5 gpurun_out/r05ae/t1.log
python tools/kbench.py --dtype bf16 --only hyper --bwd --cold 8 > gpurun_out/r05ae/kbench_v2.log 2>&1; grep -v amdgpu gpurun_out/r05ae/kbench_v2.log | tail -8
FGNN_FANIN_BWD_V1=1 python tools/kbench.py --dtype bf16 --only hyper --bwd --cold 8 > gpurun_out/r05ae/kbench_v1.log 2>&1; grep -v amdgpu gpurun_out/r05ae/kbench_v1.log | tail -8
python bench.py --no-cpu-baseline > gpurun_out/r05ae/bench_v2.json 2>/dev/null; python -c "import json; d=json.load(open('gpurun_out/r05ae/bench_v2.json')); print('v2', d['ms_per_step'])"
FGNN_FANIN_BWD_V1=1 python bench.py --no-cpu-baseline > gpurun_out/r05ae/bench_v1.json 2>/dev/null; python -c "import json; d=json.load(open('gpurun_out/r05ae/bench_v1.json')); print('v1', d['ms_per_step'])"
