#!/bin/sh
# Run on the GPU box: kernel trace (start / end per launch) of a few training steps, attributed to wall time by tools/timeline.py.
#   gpurun --timeout 900 -- sh tools/profile_timeline.sh [outdir under gpurun_out] [extra bench.py flags]
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/${1:-timeline}
shift
mkdir -p $O
cd $R
rm -rf /tmp/tl
# FGNN_STEP_HEAD_START_MS: a spin kernel at the head of the captured step — the profiler-slowed host enqueues the whole step under it, so the
# trace shows the graph's own schedule (without it the second branch of every fork starts when the HOST reaches it: DESIGN 4.13)
FGNN_STEP_HEAD_START_MS=${FGNN_STEP_HEAD_START_MS:-60} timeout 600 rocprofv3 --kernel-trace --output-format csv -d /tmp/tl -o tl -- python bench.py --steps 4 --warmup 2 --mode train --no-cpu-baseline "$@" > $O/bench.log 2>&1
T=$(find /tmp/tl -name "*kernel_trace.csv" | head -1)
python tools/timeline.py $T --marker flat_adam_kernel --after fgnn_spin_kernel --json $O/timeline.json --seq $O/step_sequence.csv > $O/timeline.txt 2>&1     # (the capturable optimizer is two kernels: name the update itself)
tail -n 2000 $T > $O/trace_tail.csv
head -1 $T > $O/trace_head.csv
cat $O/timeline.txt | head -60
