#!/bin/sh
# Run on the GPU box: everything profiles/r04/ is built from (tools/refresh_profiles_r04.py copies / summarises it).
#   gpurun --timeout 2400 -- sh tools/profile_r04.sh
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r04
mkdir -p $O
cd $R
# 1. kernel statistics of the benched step and of the inference forward (rocprofv3 --kernel-trace --stats): the IN-GRAPH averages
FGNN_PROF_OUT=r04/prof sh tools/profile_bench.sh > /dev/null 2>&1
# 2. PMC passes (each counter set its own rocprofv3 run) of the third-generation parity kernels, PER INSTANCE
sh tools/profile_pmc_fwd.sh r04/pmc_fwd_v2f "parity V->F 64->64" "--regular --stats --argmax" > /dev/null 2>&1
sh tools/profile_pmc_fwd.sh r04/pmc_fwd_f2v "parity F->V 64->64" "--regular --stats --argmax" > /dev/null 2>&1
sh tools/profile_pmc_fwd.sh r04/pmc_bwd_v2f "parity V->F 64->64" "--regular --bwd" > /dev/null 2>&1
sh tools/profile_pmc_fwd.sh r04/pmc_bwd_f2v "parity F->V 64->64" "--regular --bwd" > /dev/null 2>&1
# 3. stand-alone device times: inputs resident in the infinity cache (one copy) and from HBM (8 rotating copies, as inside a step);
#    third generation and, with FGNN_NO_WS=1, the second generation it replaces
for cold in 1 8; do
  python tools/kbench.py --dtype bf16 --regular --stats --argmax --only parity --cold $cold > $O/kbench_fwd_cold$cold.log 2>&1
  python tools/kbench.py --dtype bf16 --regular --bwd --only parity --cold $cold > $O/kbench_bwd_cold$cold.log 2>&1
  FGNN_NO_WS=1 python tools/kbench.py --dtype bf16 --regular --stats --argmax --only parity --cold $cold > $O/kbench_fwd_cold${cold}_second_generation.log 2>&1
  FGNN_NO_WS=1 python tools/kbench.py --dtype bf16 --regular --bwd --only parity --cold $cold > $O/kbench_bwd_cold${cold}_second_generation.log 2>&1
done
# 4. bench lines: the default one (with the CPU baselines), inference, f32 (what the reference scripts get through the shim)
python bench.py > $O/bench_default.json 2> $O/bench_default.err
python bench.py --mode fwd --no-cpu-baseline > $O/bench_fwd.json 2> /dev/null
python bench.py --dtype f32 --no-cpu-baseline > $O/bench_f32.json 2> /dev/null
FGNN_NO_WS=1 python bench.py --no-cpu-baseline > $O/bench_second_generation_kernels.json 2> /dev/null
# 5. wall-time attribution of one replayed training step (kernel trace -> tools/timeline.py)
sh tools/profile_timeline.sh r04/timeline > /dev/null 2>&1
ls -la $O
