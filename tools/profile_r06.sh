#!/bin/sh
# Run on the GPU box: everything profiles/r06/ is built from (python tools/refresh_profiles_r05.py r06 copies / summarises it).
#   gpurun --timeout 3000 -- sh tools/profile_r06.sh
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r06
mkdir -p $O
cd $R
# 1. kernel statistics of the benched step (two streams AND one stream) and of the inference forward: the IN-GRAPH averages
#    (bench.py's roofline.frac is taken from the two-stream table)
FGNN_PROF_OUT=r06/prof sh tools/profile_bench.sh > /dev/null 2>&1
# 2. PMC passes (each counter set its own rocprofv3 run) of the parity kernels, PER INSTANCE
sh tools/profile_pmc_fwd.sh r06/pmc_fwd_v2f "parity V->F 64->64" "--regular --stats --argmax" > /dev/null 2>&1
sh tools/profile_pmc_fwd.sh r06/pmc_fwd_f2v "parity F->V 64->64" "--regular --stats --argmax" > /dev/null 2>&1
sh tools/profile_pmc_fwd.sh r06/pmc_bwd_v2f "parity V->F 64->64" "--regular --bwd" > /dev/null 2>&1
sh tools/profile_pmc_fwd.sh r06/pmc_bwd_f2v "parity F->V 64->64" "--regular --bwd" > /dev/null 2>&1
# 2b. HBM bytes of the LDS-staged wide weight-gradient kernel (round 6) over tools/wbench.py's shapes
mkdir -p $O/pmc_wgrad
for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf /tmp/pw_$c
  timeout 300 rocprofv3 --kernel-trace --pmc $c --output-format csv -d /tmp/pw_$c -o $c -- python tools/wbench.py --bf16 > /tmp/pw_$c.log 2>&1
  find /tmp/pw_$c -name "*counter_collection.csv" -exec cp {} $O/pmc_wgrad/$c.csv \;
done
# 3. stand-alone device times of the operator: inputs resident in the infinity cache (one copy) / from HBM (8 rotating copies)
for cold in 1 8; do
  python tools/kbench.py --dtype bf16 --regular --stats --argmax --only parity --cold $cold > $O/kbench_fwd_cold$cold.log 2>&1
  python tools/kbench.py --dtype bf16 --regular --bwd --only parity --cold $cold > $O/kbench_bwd_cold$cold.log 2>&1
done
# 4. the streaming kernels around the operator; the weight gradients with and without the LDS-staged kernel
python tools/wbench.py --bf16 > $O/wbench.log 2>&1
FGNN_WG_NOLDS=1 python tools/wbench.py --bf16 > $O/wbench_register_direct.log 2>&1
python tools/wmbench.py > $O/wmbench.log 2>&1
FGNN_WG_NOLDS=1 python tools/wmbench.py > $O/wmbench_register_direct.log 2>&1
python tools/mbench.py > $O/mbench.log 2>&1
python tools/sbench.py > $O/sbench.log 2>&1
# 5. bench lines: the default one (with the CPU baselines), inference, f32, the synthetic-PGM configurations, this round's A/B
python bench.py > $O/bench_default.json 2> $O/bench_default.err
python bench.py --mode fwd --no-cpu-baseline > $O/bench_fwd.json 2> /dev/null
python bench.py --dtype f32 --no-cpu-baseline > $O/bench_f32.json 2> /dev/null
python bench.py --workload syn_pw > $O/bench_syn_pw.json 2> /dev/null
python bench.py --workload syn_hop > $O/bench_syn_hop.json 2> /dev/null
python bench.py --workload syn_hop --hop-order 8 --no-cpu-baseline > $O/bench_syn_hop8.json 2> /dev/null
FGNN_WG_NOLDS=1 python bench.py --no-cpu-baseline > $O/bench_wgrad_register_direct.json 2> /dev/null
FGNN_NO_SIDE_STREAM=1 python bench.py --no-cpu-baseline > $O/bench_one_stream.json 2> /dev/null
# 6. an unchanged training loop (the reference's model composition and loop lines): as written / fast path eager / fast path replayed
python tools/fastpath_step.py 4096 20 > $O/fastpath_step.log 2>&1
# 7. wall-time attribution of one replayed training step, its critical path, and the HBM bytes of every kernel of the step
sh tools/profile_timeline.sh r06/timeline > /dev/null 2>&1
python tools/critical_path.py $O/timeline/step_sequence.csv > $O/timeline/critical_path.txt 2>&1
cp $O/timeline/step_sequence.csv /tmp/seq_r06.csv
FGNN_STEP_SEQ=/tmp/seq_r06.csv sh tools/profile_step_traffic.sh r06/traffic > /dev/null 2>&1
# 8. kernel statistics of the synthetic-PGM training steps (configs 2 / 5)
sh tools/profile_syn.sh > /dev/null 2>&1
mkdir -p $O/syn && cp $R/gpurun_out/prof_syn/* $O/syn/ 2>/dev/null
# 9. the synthetic-PGM operator's backward (configs 2 / 5): exact-f32 kernel / three / two bf16 pieces — call times, gradients against the
#    exact kernel's —, its PMC passes, and the training steps with the exact kernel (the round's A/B)
mkdir -p $O/ext
FGNN_EXT_BWD_PIECES=0 python tools/xbench.py --save=/tmp/ext_exact.pt 2>&1 | grep -v amdgpu.ids > $O/ext/xbench_exact_f32.txt
FGNN_EXT_BWD_PIECES=3 python tools/xbench.py --compare=/tmp/ext_exact.pt 2>&1 | grep -v amdgpu.ids > $O/ext/xbench_three_pieces.txt
FGNN_EXT_BWD_PIECES=2 python tools/xbench.py --compare=/tmp/ext_exact.pt 2>&1 | grep -v amdgpu.ids > $O/ext/xbench_two_pieces.txt
sh tools/profile_pmc_ext.sh r06/ext/pmc > /dev/null 2>&1
FGNN_EXT_BWD_PIECES=0 python bench.py --workload syn_hop --no-cpu-baseline > $O/bench_syn_hop_exact_f32_backward.json 2> /dev/null
FGNN_EXT_BWD_PIECES=0 python bench.py --workload syn_pw --no-cpu-baseline > $O/bench_syn_pw_exact_f32_backward.json 2> /dev/null
FGNN_EXT_BWD_PIECES=3 python bench.py --workload syn_hop --no-cpu-baseline > $O/bench_syn_hop_three_pieces.json 2> /dev/null
# 10. what a replayed two-branch hipGraph really does, with NO profiler attached: dependent-kernel latency in one- / two-branch graphs, the
#     fork probe, and device stamps inside the benched step (when each layer's two chains start and end) with the host's lead
mkdir -p $O/graph
python tools/ubench/graph_chain_latency.py 2>&1 | grep -v amdgpu.ids > $O/graph/chain_latency.txt
python tools/ubench/graph_fork_probe.py --time 2>&1 | grep -v amdgpu.ids > $O/graph/fork_probe_time.txt
FGNN_STAMPS=1 FGNN_BENCH_HOST_TIMES=1 python bench.py --no-cpu-baseline --steps 20 > $O/graph/bench_stamps.json 2> $O/graph/bench_stamps.err
grep -E "^stamp|host enqueue|step: host" $O/graph/bench_stamps.err > $O/graph/step_stamps.txt
FGNN_BENCH_HOST_TIMES=1 FGNN_NO_SIDE_STREAM=1 python bench.py --no-cpu-baseline --steps 20 2>&1 > /dev/null | grep -E "host enqueue" > $O/graph/host_times_one_stream.txt
# 11. inference: the one-row hyper-factor block on the wave-per-sample kernel (A/B of mpconv_block_rows1_kernel)
FGNN_NO_BLOCK_ROWS1=1 python bench.py --mode fwd --no-cpu-baseline > $O/bench_fwd_wave_per_sample_fanout.json 2> /dev/null
ls -la $O
