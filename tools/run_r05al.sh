#!/bin/sh
# round 5: the degree-3 backward instance over 48 node rows (NL) instead of 64
mkdir -p gpurun_out/r05al
python -m pytest tests/test_mpconv_sg_gpu.py -x -q -m gpu -k "backward or bwd or reproducible" > gpurun_out/r05al/t1.log 2>&1; tail -3 gpurun_out/r05al/t1.log
for c in 1 8; do
python tools/kbench.py --dtype bf16 --regular --bwd --only parity --cold $c 2>&1 | grep -v amdgpu | grep "F->V" > gpurun_out/r05al/kbench_nl48_cold$c.log; cat gpurun_out/r05al/kbench_nl48_cold$c.log
FGNN_BWD_WS_NL64=1 python tools/kbench.py --dtype bf16 --regular --bwd --only parity --cold $c 2>&1 | grep -v amdgpu | grep "F->V" > gpurun_out/r05al/kbench_nl64_cold$c.log; cat gpurun_out/r05al/kbench_nl64_cold$c.log
done
python bench.py --no-cpu-baseline > gpurun_out/r05al/bench_nl48.json 2>/dev/null; python -c "import json; d=json.load(open('gpurun_out/r05al/bench_nl48.json')); print('nl48', d['ms_per_step'], d['roofline']['kernel'], d['roofline']['frac'])"
FGNN_BWD_WS_NL64=1 python bench.py --no-cpu-baseline > gpurun_out/r05al/bench_nl64.json 2>/dev/null; python -c "import json; d=json.load(open('gpurun_out/r05al/bench_nl64.json')); print('nl64', d['ms_per_step'], d['roofline']['kernel'], d['roofline']['frac'])"
