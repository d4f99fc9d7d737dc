#!/bin/sh
# The ONE parameterised A/B script (run on the GPU box through gpurun; replaces the per-experiment tools/run_r05*.sh files):
#   gpurun --timeout 1800 -- sh tools/gpu_ab.sh OUT [-t "PYTEST ARGS"]... [-b "LABEL[:ENV=V ...][:BENCH ARGS]"]... [-c "SHELL COMMAND"]...
# Steps run in the order given.  -t: python -m pytest ARGS -m gpu -q (log: OUT/pytest_N.log, tail printed).  -b: one bench.py line with the
# environment assignments in front (log: OUT/bench_LABEL.json, ms/step and roofline fraction printed).  -c: any command, with $O = the
# output directory.  Everything lands under gpurun_out/OUT/ (merged back by gpurun).
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/$1
shift
mkdir -p $O
export O
cd $R
n=0
while [ $# -gt 1 ]; do
  kind=$1; arg=$2; shift 2
  case $kind in
    -t) n=$((n + 1))
        timeout 2400 python -m pytest $arg -m gpu -q > $O/pytest_$n.log 2>&1
        echo "== pytest $arg"; tail -6 $O/pytest_$n.log ;;
    -b) label=$(echo "$arg" | cut -d: -f1); envs=$(echo "$arg" | cut -s -d: -f2); bargs=$(echo "$arg" | cut -s -d: -f3)
        env $envs python bench.py --no-cpu-baseline $bargs > $O/bench_$label.json 2> $O/bench_$label.err
        python - "$O/bench_$label.json" "$label" <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    r = d.get('roofline') or {}
    print('== bench %-28s %8.3f ms/step  %8.3f G msg/s  frac %s (%s)' % (sys.argv[2], d['ms_per_step'], d['value'] / 1e9, r.get('frac'), r.get('kernel')))
except Exception as e:
    print('== bench %s FAILED: %s' % (sys.argv[2], e))
PY
        ;;
    -c) echo "== $arg"; sh -c "$arg" ;;
  esac
done
ls $O | head -50
