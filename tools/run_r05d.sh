#!/bin/sh
# Run on the GPU box (round 5): full GPU suite + default bench + timeline with the launch sequence.
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/${1:-r05d}
mkdir -p $O
cd $R
timeout 1500 python -m pytest tests -m gpu -q --maxfail=25 > $O/pytest.log 2>&1
tail -5 $O/pytest.log
python bench.py --no-cpu-baseline > $O/bench_default.json 2> $O/bench_default.err
FGNN_F2F_SIDE=1 python bench.py --no-cpu-baseline > $O/bench_f2fside.json 2> /dev/null
python bench.py --no-cpu-baseline > $O/bench_default2.json 2> /dev/null
FGNN_F2F_SIDE=1 python bench.py --no-cpu-baseline > $O/bench_f2fside2.json 2> /dev/null
for f in default f2fside default2 f2fside2; do python - <<PY
import json
try:
    d = json.load(open('$O/bench_$f.json'))
    print('$f', round(d['ms_per_step'], 3), 'ms/step', round(d['value'] / 1e9, 3), 'G msg/s', 'frac', d['roofline']['frac'])
except Exception as e:
    print('$f', 'FAILED', e)
PY
done
sh tools/profile_timeline.sh ${1:-r05d}/timeline > /dev/null 2>&1
head -8 $O/timeline/timeline.txt
