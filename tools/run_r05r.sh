#!/bin/sh
# Run on the GPU box (round 5): fan-in forward with the whole sample's loads in flight.
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/${1:-r05r}
mkdir -p $O
cd $R
python tools/kbench.py --dtype bf16 --only "hyper  V->F" --cold 8 2> /dev/null | grep -v amdgpu.ids > $O/kbench_fanin.log; cut -c1-150 $O/kbench_fanin.log
python tools/kbench.py --dtype bf16 --only "hyper  V->F" --cold 8 --stats --argmax 2> /dev/null | grep -v amdgpu.ids > $O/kbench_fanin_stats.log; cut -c1-150 $O/kbench_fanin_stats.log
timeout 1200 python -m pytest tests/test_mpconv_gpu.py tests/test_mpconv_sg_gpu.py -m gpu -q -k "hyper or fanin or fan_in or single or degree" > $O/pytest_h.log 2>&1; tail -3 $O/pytest_h.log
timeout 1200 python -m pytest tests/test_assemblies_gpu.py tests/test_parity_pins_gpu.py -m gpu -q -x > $O/pytest_a.log 2>&1; tail -3 $O/pytest_a.log
for i in 1 2; do
python bench.py --no-cpu-baseline 2> /dev/null > $O/bench$i.json
python -c "import json,sys; d=json.loads(open('$O/bench$i.json').read().strip().splitlines()[-1]); print('bench', d['ms_per_step'])"
done
