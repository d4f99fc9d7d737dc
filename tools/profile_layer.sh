#!/bin/sh
# Run on the GPU box (gpurun -- sh tools/profile_layer.sh): rocprofv3 kernel statistics of the bf16 LDPCModel inference forward
# (bench.py --mode fwd) with the one-kernel 64-wide layers (csrc/factor_layer_fwd.hip).  Output under gpurun_out/prof_layer/.
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/prof_layer
mkdir -p $OUT
cd $R
rm -rf /tmp/pl
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pl -o fwd -- python bench.py --mode fwd --steps 10 --warmup 3 --no-cpu-baseline > /tmp/pl.log 2>&1
grep "^{\"metric" /tmp/pl.log | tail -1 > $OUT/bench_fwd.json
find /tmp/pl -name "*kernel_stats*.csv" -exec cp {} $OUT/fwd_kernel_stats.csv \;
head -12 $OUT/fwd_kernel_stats.csv | cut -c1-150
cut -c1-200 $OUT/bench_fwd.json
