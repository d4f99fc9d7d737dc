#!/bin/sh
# Run on the GPU box (round 5): RCCL C-API all-reduce inside the step graph (one-rank communicator), edge-MLP tests.
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/${1:-r05i}
mkdir -p $O
cd $R
timeout 1200 python -m pytest tests/test_dp_two_ranks_gpu.py -m gpu -q -x > $O/pytest_dp.log 2>&1
tail -8 $O/pytest_dp.log
timeout 600 python -m pytest tests/test_assemblies_gpu.py -m gpu -q -k "edge_mlp" > $O/pytest_em.log 2>&1
tail -4 $O/pytest_em.log
