#!/bin/sh
# Run on the GPU box (round 5): the 128 -> 64 backward as two launches of the third-generation kernel.
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/${1:-r05j}
mkdir -p $O
cd $R
timeout 1500 python -m pytest tests/test_mpconv_sg_gpu.py tests/test_mpconv_gpu.py tests/test_parity_pins_gpu.py tests/test_fullsize_properties_gpu.py tests/test_assemblies_gpu.py -m gpu -q --maxfail=10 > $O/pytest.log 2>&1
tail -6 $O/pytest.log
python tools/kbench.py --dtype bf16 --regular --bwd --only "128->64" --cold 8 > $O/kbench_ksplit.log 2>&1
FGNN_NO_WS_KSPLIT=1 python tools/kbench.py --dtype bf16 --regular --bwd --only "128->64" --cold 8 > $O/kbench_noksplit.log 2>&1
grep -h "128->64" $O/kbench_ksplit.log $O/kbench_noksplit.log
run() { tag=$1; shift; env "$@" timeout 300 python bench.py --no-cpu-baseline $EXTRA > $O/bench_$tag.json 2> $O/bench_$tag.err; python - <<PY
import json
try:
    d = json.load(open('$O/bench_$tag.json'))
    print('%-12s %7.3f ms/step  %.3f G msg/s' % ('$tag', d['ms_per_step'], d['value'] / 1e9))
except Exception as e:
    print('$tag', 'FAILED', e)
PY
}
run ksplit FGNN_X=0
run noksplit FGNN_NO_WS_KSPLIT=1
run ksplit2 FGNN_X=0
EXTRA="--mode fwd" run fwd FGNN_X=0
