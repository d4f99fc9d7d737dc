#!/bin/sh
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r06x
mkdir -p $O
cd $R
timeout 2400 python -m pytest tests -m gpu -q -x -k "ext or syn or factor_mpnn or fullsize or golden or sequential or parity_pins" 2>&1 | tail -12 | tee $O/tests_sel.txt
for w in syn_pw syn_hop; do
  timeout 600 python bench.py --workload $w --no-cpu-baseline > $O/bench_$w.json 2> $O/bench_$w.err
  python - "$O/bench_$w.json" $w <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print('== bench %s: %.3f ms/step, roofline %s' % (sys.argv[2], d['ms_per_step'], {k: v for k, v in d['roofline'].items() if k in ('kernel', 'frac', 'avg_launch_us', 'achieved')}))
PY
done
