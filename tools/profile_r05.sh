#!/bin/sh
# Run on the GPU box: everything profiles/r05/ is built from (tools/refresh_profiles_r05.py copies / summarises it).
#   gpurun --timeout 3000 -- sh tools/profile_r05.sh
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r05
mkdir -p $O
cd $R
# 1. kernel statistics of the benched step and of the inference forward (rocprofv3 --kernel-trace --stats): the IN-GRAPH averages
FGNN_PROF_OUT=r05/prof sh tools/profile_bench.sh > /dev/null 2>&1
# 2. PMC passes (each counter set its own rocprofv3 run) of the third-generation parity kernels, PER INSTANCE
sh tools/profile_pmc_fwd.sh r05/pmc_fwd_v2f "parity V->F 64->64" "--regular --stats --argmax" > /dev/null 2>&1
sh tools/profile_pmc_fwd.sh r05/pmc_fwd_f2v "parity F->V 64->64" "--regular --stats --argmax" > /dev/null 2>&1
sh tools/profile_pmc_fwd.sh r05/pmc_bwd_v2f "parity V->F 64->64" "--regular --bwd" > /dev/null 2>&1
sh tools/profile_pmc_fwd.sh r05/pmc_bwd_f2v "parity F->V 64->64" "--regular --bwd" > /dev/null 2>&1
# 3. stand-alone device times of the operator: inputs resident in the infinity cache (one copy) / from HBM (8 rotating copies)
for cold in 1 8; do
  python tools/kbench.py --dtype bf16 --regular --stats --argmax --only parity --cold $cold > $O/kbench_fwd_cold$cold.log 2>&1
  python tools/kbench.py --dtype bf16 --regular --bwd --only parity --cold $cold > $O/kbench_bwd_cold$cold.log 2>&1
done
python tools/kbench.py --dtype bf16 --only hyper --cold 8 > $O/kbench_hyper_fwd_cold8.log 2>&1
python tools/kbench.py --dtype bf16 --only hyper --bwd --cold 8 > $O/kbench_hyper_bwd_cold8.log 2>&1
# 4. per-shape efficiency of the streaming kernels around the operator (library GEMMs beside the hand-written maps)
python tools/lbench.py > $O/lbench.log 2>&1
python tools/wbench.py --bf16 > $O/wbench.log 2>&1
python tools/tbench.py > $O/tbench.log 2>&1
python tools/sbench.py > $O/sbench.log 2>&1
python tools/mbench.py > $O/mbench.log 2>&1
python tools/wmbench.py > $O/wmbench.log 2>&1
# 5. bench lines: the default one (with the CPU baselines), inference, f32, and the A/B switches of this round's levers
python bench.py > $O/bench_default.json 2> $O/bench_default.err
python bench.py --mode fwd --no-cpu-baseline > $O/bench_fwd.json 2> /dev/null
python bench.py --dtype f32 --no-cpu-baseline > $O/bench_f32.json 2> /dev/null
FGNN_NO_FANOUT_BROADCAST=1 python bench.py --no-cpu-baseline > $O/bench_no_fanout_broadcast.json 2> /dev/null
FGNN_INKERNEL_FINALISERS=1 python bench.py --no-cpu-baseline > $O/bench_inkernel_finalisers.json 2> /dev/null
FGNN_EDGE_MLP_VALU=1 python bench.py --no-cpu-baseline > $O/bench_edge_mlp_valu.json 2> /dev/null
FGNN_F2F_SIDE=0 python bench.py --no-cpu-baseline > $O/bench_f2f_main.json 2> /dev/null
FGNN_NO_SIDE_STREAM=1 python bench.py --no-cpu-baseline > $O/bench_one_stream.json 2> /dev/null
FGNN_BWD_TABLES=1 python bench.py --no-cpu-baseline > $O/bench_bwd_tables.json 2> /dev/null
FGNN_NO_MERGED_FAN_GRADS=1 python bench.py --no-cpu-baseline > $O/bench_no_merged_fan_grads.json 2> /dev/null
FGNN_NO_MERGED_FAN_WGRADS=1 python bench.py --no-cpu-baseline > $O/bench_no_merged_fan_wgrads.json 2> /dev/null
FGNN_WGRAD_STREAM=1 python bench.py --no-cpu-baseline > $O/bench_wgrad_third_stream.json 2> /dev/null
FGNN_NO_DEFER_FOLDS=1 python bench.py --no-cpu-baseline > $O/bench_immediate_folds.json 2> /dev/null
FGNN_NO_INSTNORM_DOT=1 python bench.py --no-cpu-baseline > $O/bench_no_instnorm_dot.json 2> /dev/null
FGNN_NO_FAST_REGRESSOR=1 python bench.py --no-cpu-baseline > $O/bench_torch_regressor.json 2> /dev/null
# 6. the all-host-cores CPU sample (BASELINE.md asks for os.cpu_count(); the line's default is 16 threads: see bench.py)
timeout 900 python bench.py --cpu-baseline-only --cpu-batch 256 --cpu-threads $(nproc) > $O/cpu_baseline_all_cores.json 2> $O/cpu_baseline_all_cores.err
# 7. wall-time attribution of one replayed training step (kernel trace -> tools/timeline.py)
sh tools/profile_timeline.sh r05/timeline > /dev/null 2>&1
python tools/critical_path.py $O/timeline/step_sequence.csv > $O/timeline/critical_path.txt 2>&1
# 8. HBM bytes of every kernel of the step (two --pmc passes over bench.py; per-step launch counts from the sequence of 7.)
cp $O/timeline/step_sequence.csv /tmp/seq_r05.csv
FGNN_STEP_SEQ=/tmp/seq_r05.csv sh tools/profile_step_traffic.sh r05/traffic > /dev/null 2>&1
ls -la $O
