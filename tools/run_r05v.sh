#!/bin/sh
# Run on the GPU box (round 5): merged fan-out gradients (ops.FanBox, linear_multi_b16_kernel).
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/${1:-r05v}
mkdir -p $O
cd $R
timeout 900 python -m pytest tests/test_mpconv_gpu.py -m gpu -q -k "linear_multi" > $O/pytest_lm.log 2>&1; tail -4 $O/pytest_lm.log
timeout 1500 python -m pytest tests/test_assemblies_gpu.py tests/test_parity_pins_gpu.py tests/test_block_tail_gpu.py -m gpu -q > $O/pytest_a.log 2>&1; tail -8 $O/pytest_a.log
for i in 1 2; do
python bench.py --no-cpu-baseline 2> $O/bench$i.err > $O/bench$i.json
python -c "import json,sys; d=json.loads(open('$O/bench$i.json').read().strip().splitlines()[-1]); print('bench', d['ms_per_step'], d['config']['peak_hbm_GB'])"
FGNN_NO_MERGED_FAN_GRADS=1 python bench.py --no-cpu-baseline 2> /dev/null > $O/bench_nomerge$i.json
python -c "import json,sys; d=json.loads(open('$O/bench_nomerge$i.json').read().strip().splitlines()[-1]); print('bench per-consumer sums', d['ms_per_step'], d['config']['peak_hbm_GB'])"
done
tail -3 $O/bench1.err
