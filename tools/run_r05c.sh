#!/bin/sh
# Run on the GPU box (round 5): new tests of the table-driven backward, kbench of the operator, then scheduling knobs A/B.
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/${1:-r05c}
mkdir -p $O
cd $R
timeout 900 python -m pytest tests/test_mpconv_sg_gpu.py tests/test_block_tail_gpu.py -m gpu -q --maxfail=10 > $O/pytest.log 2>&1
tail -5 $O/pytest.log
for cold in 1 8; do
  python tools/kbench.py --dtype bf16 --regular --bwd --only parity --cold $cold > $O/kbench_bwd_cold$cold.log 2>&1
  FGNN_NO_BWD_TABLES=1 python tools/kbench.py --dtype bf16 --regular --bwd --only parity --cold $cold > $O/kbench_bwd_cold${cold}_notables.log 2>&1
done
grep -h "bwd" $O/kbench_bwd_cold1.log $O/kbench_bwd_cold1_notables.log | head -20
run() { tag=$1; shift; env "$@" python bench.py --no-cpu-baseline > $O/bench_$tag.json 2> $O/bench_$tag.err; python - <<PY
import json
try:
    d = json.load(open('$O/bench_$tag.json'))
    print('$tag', round(d['ms_per_step'], 3), 'ms/step', round(d['value'] / 1e9, 3), 'G msg/s', 'frac', d['roofline']['frac'], d['roofline']['kernel'])
except Exception as e:
    print('$tag', 'FAILED', e)
PY
}
run default FGNN_X=0
run notables FGNN_NO_BWD_TABLES=1
run f2fside FGNN_F2F_SIDE=1
run wgradside FGNN_WGRAD_SIDE=1
run both FGNN_F2F_SIDE=1 FGNN_WGRAD_SIDE=1
run default2 FGNN_X=0
