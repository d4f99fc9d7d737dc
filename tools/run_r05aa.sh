#!/bin/sh
# round 5, merged weight gradients: the new tests, then the bench line with and without the merge
mkdir -p gpurun_out/r05aa
python -m pytest tests/test_mpconv_gpu.py -x -q -m gpu -k "several_maps or weight_gradient" > gpurun_out/r05aa/t1.log 2>&1; tail -3 gpurun_out/r05aa/t1.log
python -m pytest tests/test_assemblies_gpu.py -x -q -m gpu -k "merged_fan_out" > gpurun_out/r05aa/t2.log 2>&1; tail -3 gpurun_out/r05aa/t2.log
python bench.py --no-cpu-baseline > gpurun_out/r05aa/bench_merged.json 2> gpurun_out/r05aa/bench_merged.err; python -c "import json; d=json.load(open('gpurun_out/r05aa/bench_merged.json')); print('merged', d['ms_per_step'])"
FGNN_NO_MERGED_FAN_WGRADS=1 python bench.py --no-cpu-baseline > gpurun_out/r05aa/bench_unmerged.json 2> gpurun_out/r05aa/bench_unmerged.err; python -c "import json; d=json.load(open('gpurun_out/r05aa/bench_unmerged.json')); print('unmerged', d['ms_per_step'])"
python bench.py --no-cpu-baseline > gpurun_out/r05aa/bench_merged2.json 2>/dev/null; python -c "import json; d=json.load(open('gpurun_out/r05aa/bench_merged2.json')); print('merged again', d['ms_per_step'])"
