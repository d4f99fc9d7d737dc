#!/bin/sh
# Run on the GPU box (round 5): contiguous slab fold of the weight-gradient kernels.
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/${1:-r05q}
mkdir -p $O
cd $R
python tools/wbench.py --bf16 2> /dev/null | grep -v amdgpu.ids > $O/wbench.log; cat $O/wbench.log
timeout 900 python -m pytest tests/test_mpconv_gpu.py -m gpu -q -k "wgrad or linear or pointwise or conv" > $O/pytest_w.log 2>&1; tail -3 $O/pytest_w.log
timeout 900 python -m pytest tests/test_assemblies_gpu.py -m gpu -q -x > $O/pytest_a.log 2>&1; tail -3 $O/pytest_a.log
for i in 1 2; do
python bench.py --no-cpu-baseline 2> /dev/null > $O/bench$i.json
python -c "import json,sys; d=json.loads(open('$O/bench$i.json').read().strip().splitlines()[-1]); print('bench', d['ms_per_step'])"
done
