#!/bin/sh
# Run on the GPU box (e.g. `gpurun -- sh tools/profile_bench.sh`): rocprofv3 kernel statistics of `bench.py` in both modes,
# written under gpurun_out/${FGNN_PROF_OUT:-prof4}/ (merged back by gpurun); tools/refresh_profiles.py then copies them into profiles/r01/.
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out/${FGNN_PROF_OUT:-prof4}
cd $R
for mode in train fwd; do
  rm -rf /tmp/prof_$mode
  timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$mode -o $mode -- python bench.py --steps 10 --warmup 3 --mode $mode --no-cpu-baseline > /tmp/prof_$mode.log 2>&1
  grep "^{\"metric" /tmp/prof_$mode.log | tail -1 > $R/gpurun_out/${FGNN_PROF_OUT:-prof4}/bench_$mode.json
  find /tmp/prof_$mode -name "*kernel_stats*.csv" -exec cp {} $R/gpurun_out/${FGNN_PROF_OUT:-prof4}/ \;
done
# the training step on ONE stream: every kernel alone on the chip inside the replayed graph (the two-stream averages above carry
# the contention of whatever runs beside a launch on the other queue)
rm -rf /tmp/prof_train1
FGNN_NO_SIDE_STREAM=1 timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_train1 -o train1 -- python bench.py --steps 10 --warmup 3 --mode train --no-cpu-baseline > /tmp/prof_train1.log 2>&1
grep "^{\"metric" /tmp/prof_train1.log | tail -1 > $R/gpurun_out/${FGNN_PROF_OUT:-prof4}/bench_train1.json
find /tmp/prof_train1 -name "*kernel_stats*.csv" -exec cp {} $R/gpurun_out/${FGNN_PROF_OUT:-prof4}/ \;
