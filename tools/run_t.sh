#!/bin/sh
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r06x
mkdir -p $O
cd $R
timeout 2400 python -m pytest tests/test_concat_gpu.py tests -m gpu -q -x -k "concat or syn or factor_mpnn or fullsize or golden or sequential" 2>&1 | tail -6 | tee $O/tests_sel.txt
for w in syn_pw syn_hop; do
  timeout 600 python bench.py --workload $w --no-cpu-baseline > $O/bench_$w.json 2> $O/bench_$w.err
  echo "== bench $w: $(grep -o '"ms_per_step": [0-9.]*' $O/bench_$w.json | head -1)"
done
