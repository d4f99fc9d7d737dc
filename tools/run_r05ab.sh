#!/bin/sh
# round 5: parked weight gradients on a third stream (FGNN_WGRAD_STREAM=1)
mkdir -p gpurun_out/r05ab
python -m pytest tests/test_assemblies_gpu.py -x -q -m gpu -k "merged_fan_out" > gpurun_out/r05ab/t2.log 2>&1; tail -3 gpurun_out/r05ab/t2.log
for v in 0 1; do
FGNN_WGRAD_STREAM=$v timeout 300 python bench.py --no-cpu-baseline > gpurun_out/r05ab/bench_ws$v.json 2> gpurun_out/r05ab/bench_ws$v.err; echo "rc $?"; python -c "import json; d=json.load(open('gpurun_out/r05ab/bench_ws$v.json')); print('wgrad stream $v', d['ms_per_step'])"
done
FGNN_WGRAD_STREAM=1 FGNN_NO_MERGED_FAN_WGRADS=1 timeout 300 python bench.py --no-cpu-baseline > gpurun_out/r05ab/bench_ws1_unmerged.json 2> gpurun_out/r05ab/bench_ws1_unmerged.err; echo "rc $?"; python -c "import json; d=json.load(open('gpurun_out/r05ab/bench_ws1_unmerged.json')); print('wgrad stream 1 unmerged', d['ms_per_step'])"
tail -5 gpurun_out/r05ab/bench_ws1.err
