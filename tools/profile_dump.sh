#!/bin/sh
# Run on the GPU box: kernel trace of a few training steps, then the per-queue kernel sequence between consecutive launches of one
# kernel (tools/timeline.py --dump).   gpurun -- sh tools/profile_dump.sh <outdir> <substring> <from> <count> [env assignments...]
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/$1; SUB=$2; FROM=$3; CNT=$4
shift 4
mkdir -p $O
cd $R
rm -rf /tmp/tld
env "$@" timeout 600 rocprofv3 --kernel-trace --output-format csv -d /tmp/tld -o tl -- python bench.py --steps 4 --warmup 2 --mode train --no-cpu-baseline > $O/bench.log 2>&1
T=$(find /tmp/tld -name "*kernel_trace.csv" | head -1)
python tools/timeline.py $T --dump "$SUB" --dump-from $FROM --dump-count $CNT > $O/dump.txt 2>&1
