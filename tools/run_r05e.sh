#!/bin/sh
# Run on the GPU box (round 5): runtime knobs that affect cross-queue dependencies inside the replayed hipGraph.
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/${1:-r05e}
mkdir -p $O
cd $R
run() { tag=$1; shift; env "$@" timeout 300 python bench.py --no-cpu-baseline > $O/bench_$tag.json 2> $O/bench_$tag.err; python - <<PY
import json
try:
    d = json.load(open('$O/bench_$tag.json'))
    print('%-28s %7.3f ms/step  %.3f G msg/s' % ('$tag', d['ms_per_step'], d['value'] / 1e9))
except Exception as e:
    print('$tag', 'FAILED', e)
PY
}
run base FGNN_X=0
run hsa_poll HSA_ENABLE_INTERRUPT=0
run cp_wait1 GPU_STREAMOPS_CP_WAIT=1
run cp_wait0 GPU_STREAMOPS_CP_WAIT=0
run graph_queues1 DEBUG_HIP_FORCE_GRAPH_QUEUES=1
run graph_queues2 DEBUG_HIP_FORCE_GRAPH_QUEUES=2
run graph_queues4 DEBUG_HIP_FORCE_GRAPH_QUEUES=4
run packet_capture0 DEBUG_CLR_GRAPH_PACKET_CAPTURE=0
run packet_capture1 DEBUG_CLR_GRAPH_PACKET_CAPTURE=1
run dyn_queues0 DEBUG_HIP_DYNAMIC_QUEUES=0
run one_stream FGNN_NO_SIDE_STREAM=1
run active_wait ROC_ACTIVE_WAIT_TIMEOUT=200
run base2 FGNN_X=0
