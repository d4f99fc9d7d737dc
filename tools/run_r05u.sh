#!/bin/sh
# Run on the GPU box (round 5): recorded parameter-gradient folds (csrc/fold_batch.hip).
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/${1:-r05u}
mkdir -p $O
cd $R
timeout 900 python -m pytest tests/test_assemblies_gpu.py -m gpu -q -k "fold or parked or graph_replay or reproducible" > $O/pytest_a.log 2>&1; tail -5 $O/pytest_a.log
for i in 1 2; do
python bench.py --no-cpu-baseline 2> $O/bench$i.err > $O/bench$i.json
python -c "import json,sys; d=json.loads(open('$O/bench$i.json').read().strip().splitlines()[-1]); print('bench', d['ms_per_step'], d['config']['peak_hbm_GB'])"
FGNN_NO_DEFER_FOLDS=1 python bench.py --no-cpu-baseline 2> /dev/null > $O/bench_nodefer$i.json
python -c "import json,sys; d=json.loads(open('$O/bench_nodefer$i.json').read().strip().splitlines()[-1]); print('bench immediate folds', d['ms_per_step'], d['config']['peak_hbm_GB'])"
done
sh tools/profile_timeline.sh ${1:-r05u}/timeline > /dev/null 2>&1
grep -n "fold_batch\|wall_ms\|wgb_reduce\|bres_reduce" $O/timeline/timeline.txt | head
