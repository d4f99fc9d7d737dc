#!/bin/sh
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r06p
mkdir -p $O
cd $R
for v in base "DEBUG_CLR_GRAPH_PACKET_CAPTURE=0" "DEBUG_CLR_GRAPH_PACKET_CAPTURE=1" "DEBUG_HIP_GRAPH_BATCH_SIZE=1" "DEBUG_HIP_GRAPH_BATCH_SIZE=256" "DEBUG_HIP_FORCE_GRAPH_QUEUES=8" "GPU_MAX_HW_QUEUES=8" "AMD_DIRECT_DISPATCH=0"; do
  e=$v; [ "$v" = base ] && e="FGNN_X=1"
  env $e FGNN_BENCH_HOST_TIMES=1 timeout 300 python bench.py --no-cpu-baseline --steps 20 > $O/b.json 2> $O/b.err
  echo "== $v: $(grep 'host enqueue' $O/b.err)"
done
