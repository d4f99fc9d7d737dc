#!/bin/sh
# Run on the GPU box (round 5, first pass): the GPU test suite, then A/B bench lines of the two round-5 levers and a timeline.
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/${1:-r05a}
mkdir -p $O
cd $R
timeout 1500 python -m pytest tests -m gpu -q --maxfail=25 > $O/pytest.log 2>&1
tail -40 $O/pytest.log
python bench.py --no-cpu-baseline > $O/bench_default.json 2> $O/bench_default.err
FGNN_SEPARATE_FINALISERS=1 python bench.py --no-cpu-baseline > $O/bench_separate_finalisers.json 2> /dev/null
FGNN_NO_FANOUT_BROADCAST=1 python bench.py --no-cpu-baseline > $O/bench_no_broadcast.json 2> /dev/null
for f in default separate_finalisers no_broadcast; do python - <<PY
import json
try:
    d = json.load(open('$O/bench_$f.json'))
    print('$f', round(d['ms_per_step'], 3), 'ms/step', round(d['value'] / 1e9, 3), 'G msg/s', 'frac', d['roofline']['frac'])
except Exception as e:
    print('$f', 'FAILED', e)
PY
done
tail -5 $O/bench_default.err
sh tools/profile_timeline.sh ${1:-r05a}/timeline > /dev/null 2>&1
head -70 $O/timeline/timeline.txt
