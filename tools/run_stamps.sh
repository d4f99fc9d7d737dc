#!/bin/sh
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r06p
mkdir -p $O
cd $R
run() { label=$1; shift; env "$@" timeout 300 python bench.py --no-cpu-baseline --steps 20 > $O/s_$label.json 2> $O/s_$label.err; echo "== $label: $(grep -o '"ms_per_step": [0-9.]*' $O/s_$label.json | head -1)"; grep "^stamp" $O/s_$label.err > $O/stamps_$label.txt; wc -l $O/stamps_$label.txt; }
run base FGNN_STAMPS=1


