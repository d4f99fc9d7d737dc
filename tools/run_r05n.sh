#!/bin/sh
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/${1:-r05n}
mkdir -p $O
cd $R
T=tests/test_assemblies_gpu.py::test_graph_replay_reproduces_eager_gradients_bitwise
for i in 1 2; do
echo default; timeout 300 python -m pytest $T -m gpu -q 2>&1 | tail -1
echo head_main; FGNN_HEAD_SIDE=0 timeout 300 python -m pytest $T -m gpu -q 2>&1 | tail -1
echo no_dot; FGNN_NO_INSTNORM_DOT=1 timeout 300 python -m pytest $T -m gpu -q 2>&1 | tail -1
echo both_off; FGNN_HEAD_SIDE=0 FGNN_NO_INSTNORM_DOT=1 timeout 300 python -m pytest $T -m gpu -q 2>&1 | tail -1
done
