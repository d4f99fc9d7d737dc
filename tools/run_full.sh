#!/bin/sh
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r06x
mkdir -p $O
cd $R
timeout 3000 python -m pytest tests -m gpu -q -x 2>&1 | tail -8 | tee $O/tests_full.txt
timeout 600 python bench.py --no-cpu-baseline > $O/bench_ldpc.json 2> $O/bench_ldpc.err
grep -o '"ms_per_step": [0-9.]*' $O/bench_ldpc.json | head -1
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
