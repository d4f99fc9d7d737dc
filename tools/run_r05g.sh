#!/bin/sh
# Run on the GPU box (round 5): the MFMA edge MLP — parity tests, then the step with it and with the VALU kernels.
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/${1:-r05g}
mkdir -p $O
cd $R
timeout 900 python -m pytest tests/test_assemblies_gpu.py -m gpu -q --maxfail=10 -k "edge_mlp or fast_path or training_reduces or bitwise" > $O/pytest.log 2>&1
tail -15 $O/pytest.log
run() { tag=$1; shift; env "$@" timeout 300 python bench.py --no-cpu-baseline > $O/bench_$tag.json 2> $O/bench_$tag.err; python - <<PY
import json
try:
    d = json.load(open('$O/bench_$tag.json'))
    k = d['kernels']
    print('%-12s %7.3f ms/step  %.3f G msg/s' % ('$tag', d['ms_per_step'], d['value'] / 1e9), {n: (v['avg_us'], v['avg_us_in_step']) for n, v in k.items() if 'edge_mlp' in n})
except Exception as e:
    print('$tag', 'FAILED', e)
PY
}
run mfma FGNN_X=0
run valu FGNN_EDGE_MLP_VALU=1
run mfma2 FGNN_X=0
