#!/bin/sh
# Run on the GPU box: everything profiles/r03/ is built from (tools/refresh_profiles_r03.py copies / summarises it).
#   gpurun --timeout 2400 -- sh tools/profile_r03.sh
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r03
mkdir -p $O
cd $R
# 1. kernel statistics of the benched step and of the inference forward (rocprofv3 --kernel-trace --stats)
FGNN_PROF_OUT=r03/prof sh tools/profile_bench.sh > /dev/null 2>&1
# 2. PMC passes (each counter set its own rocprofv3 run) of the second-generation parity kernels, PER INSTANCE
sh tools/profile_pmc_fwd.sh r03/pmc_fwd_v2f "parity V->F 64->64" "--regular --stats --argmax" > /dev/null 2>&1
sh tools/profile_pmc_fwd.sh r03/pmc_fwd_f2v "parity F->V 64->64" "--regular --stats --argmax" > /dev/null 2>&1
sh tools/profile_pmc_fwd.sh r03/pmc_bwd_v2f "parity V->F 64->64" "--regular --bwd" > /dev/null 2>&1
sh tools/profile_pmc_fwd.sh r03/pmc_bwd_f2v "parity F->V 64->64" "--regular --bwd" > /dev/null 2>&1
# 3. the fused block tail (csrc/block_tail.hip) at its widest shape
sh tools/profile_pmc_tail.sh r03/pmc_tail 393216,256 stats,apply,backward > /dev/null 2>&1
python tools/tbench.py > $O/tbench.log 2>&1
python tools/kbench.py --dtype bf16 --regular --stats --argmax > $O/kbench_fwd.log 2>&1
python tools/kbench.py --dtype bf16 --regular --bwd > $O/kbench_bwd.log 2>&1
# 4. bench lines: the default one (with the CPU baselines), f32 (what the reference scripts get through the shim),
#    per-sample tables (the reference's calling convention), inference
python bench.py > $O/bench_default.json 2> $O/bench_default.err
python bench.py --dtype f32 --no-cpu-baseline > $O/bench_f32.json 2> /dev/null
python bench.py --tables per_sample --no-cpu-baseline > $O/bench_per_sample.json 2> /dev/null
python bench.py --mode fwd --no-cpu-baseline > $O/bench_fwd.json 2> /dev/null
# 5. wall-time attribution of one replayed training step (kernel trace -> tools/timeline.py) and the wide parity shapes
sh tools/profile_timeline.sh r03/timeline > /dev/null 2>&1
python tools/kbench.py --dtype bf16 --regular --stats --argmax --only "64->128" > $O/kbench_fwd_wide.log 2>&1
FGNN_SG_NOSPLIT=1 python tools/kbench.py --dtype bf16 --regular --stats --argmax --only "64->128" >> $O/kbench_fwd_wide.log 2>&1
FGNN_SG_NOSPLIT=1 python tools/kbench.py --dtype bf16 --regular --bwd --only "64->128" > $O/kbench_bwd_wide_first_generation.log 2>&1
python tools/wbench.py --bf16 > $O/wbench.log 2>&1
ls -la $O
