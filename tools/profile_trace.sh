#!/bin/sh
# Run on the GPU box: per-dispatch kernel trace of `bench.py` (training), analysed by tools/timeline.py.
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out/trace
cd $R
rm -rf /tmp/tr
timeout 600 rocprofv3 --kernel-trace --output-format csv -d /tmp/tr -o tr -- python bench.py --steps 10 --warmup 3 --no-cpu-baseline $FGNN_BENCH_ARGS > /tmp/tr.log 2>&1
f=$(find /tmp/tr -name "*kernel_trace.csv" | head -1)
python tools/timeline.py $f 10 counts | tee $R/gpurun_out/trace/timeline.txt
