#!/bin/sh
# Run on the GPU box (round 5): the identity-list fan-in forward.
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/${1:-r05z}
mkdir -p $O
cd $R
timeout 900 python -m pytest tests/test_mpconv_sg_gpu.py -m gpu -q -k "identity_list" > $O/pytest_id.log 2>&1; tail -5 $O/pytest_id.log
python tools/kbench.py --dtype bf16 --only "hyper  V->F" --cold 8 --argmax 2> /dev/null | grep -v amdgpu.ids > $O/kbench_fanin.log; cut -c1-150 $O/kbench_fanin.log
FGNN_NO_FANIN_ID=1 python tools/kbench.py --dtype bf16 --only "hyper  V->F" --cold 8 --argmax 2> /dev/null | grep -v amdgpu.ids > $O/kbench_fanin_general.log; cut -c1-150 $O/kbench_fanin_general.log
python tools/kbench.py --dtype bf16 --only "hyper  V->F" --cold 8 2> /dev/null | grep -v amdgpu.ids > $O/kbench_fanin_eval.log; cut -c1-150 $O/kbench_fanin_eval.log
