#!/bin/sh
# Run on the GPU box (round 5): staggered block walk of the wide weight-gradient kernel; HBM bytes of the whole step.
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/${1:-r05p}
mkdir -p $O
cd $R
for s in 0 2 4 8 16; do
  FGNN_WB_STAGGER=$s python tools/wbench.py --bf16 2> /dev/null | grep -v amdgpu.ids > $O/wbench_stagger$s.log
  echo "stagger >= $s"; head -8 $O/wbench_stagger$s.log
done
for s in 0 4 8; do
  FGNN_WB_STAGGER=$s python bench.py --no-cpu-baseline 2> /dev/null > $O/bench_stagger$s.json
  python -c "import json,sys; d=json.loads(open('$O/bench_stagger$s.json').read().strip().splitlines()[-1]); print('stagger', $s, d['ms_per_step'])"
done
sh tools/profile_step_traffic.sh ${1:-r05p}/traffic
