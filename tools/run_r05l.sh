#!/bin/sh
# Run on the GPU box (round 5): the identity-list fan-in forward.
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/${1:-r05l}
mkdir -p $O
cd $R
timeout 1500 python -m pytest tests/test_mpconv_sg_gpu.py tests/test_mpconv_gpu.py tests/test_assemblies_gpu.py tests/test_parity_pins_gpu.py tests/test_block_tail_gpu.py -m gpu -q --maxfail=10 > $O/pytest.log 2>&1
tail -6 $O/pytest.log
python tools/kbench.py --dtype bf16 --only "hyper  V->F" --cold 8 --argmax > $O/kbench_fanin.log 2>&1
FGNN_NO_FANIN_ID=1 python tools/kbench.py --dtype bf16 --only "hyper  V->F" --cold 8 --argmax > $O/kbench_fanin_old.log 2>&1
grep -h "hyper" $O/kbench_fanin.log $O/kbench_fanin_old.log | cut -c1-175
for t in a b; do python bench.py --no-cpu-baseline > $O/bench_$t.json 2>/dev/null; FGNN_NO_FANIN_ID=1 python bench.py --no-cpu-baseline > $O/bench_old_$t.json 2>/dev/null; done
python - <<PY
import json
for f in ('a','old_a','b','old_b'):
    d=json.load(open('$O/bench_%s.json'%f)); print(f, round(d['ms_per_step'],3), d['roofline']['operator_fwd_frac'], d['roofline']['forward']['kernel'])
PY
