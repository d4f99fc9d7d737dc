#!/bin/sh
# Run on the GPU box: SQ issue counters (three 8-counter passes) + FETCH / WRITE passes over the INFERENCE forward (bench.py --mode fwd), for
# the one-kernel-per-block / per-layer kernels (factor_layer_fwd_kernel, mpconv_block_fwd_kernel, mpconv_block_fanout / fanin_kernel).
#   sh tools/profile_pmc_infer.sh <outdir under gpurun_out>        (each counter set is its OWN rocprofv3 run with --kernel-trace only)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/${1:-pmc_infer}
mkdir -p $OUT
cd $R
run() { # name, counters
    rm -rf /tmp/pi_$1
    timeout 400 rocprofv3 --kernel-trace --pmc $2 --output-format csv -d /tmp/pi_$1 -o $1 -- \
        python bench.py --mode fwd --steps 3 --warmup 1 --no-cpu-baseline > /tmp/pi_$1.log 2>&1
    find /tmp/pi_$1 -name "*counter_collection.csv" -exec cp {} $OUT/$1.csv \;
}
run issue_a "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_LDS"
run issue_b "SQ_INSTS_SALU SQ_INSTS_VMEM SQ_INSTS_MFMA SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS"
run issue_c "SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_MISC SQ_INSTS_SMEM SQ_VALU_MFMA_BUSY_CYCLES SQ_INST_CYCLES_VMEM SQ_WAVES_EQ_64 SQ_INSTS_VALU_MFMA_MOPS_BF16"
run fetch "FETCH_SIZE"
run write "WRITE_SIZE"
# phase timeline of the layer kernel (a -DFGNN_ENABLE_PROF build, loaded through FGNN_HIP_LIB)
make -C factor-graph-neural-network_amd/csrc prof > /tmp/pi_prof_build.log 2>&1
FGNN_HIP_LIB=$R/factor-graph-neural-network_amd/fgnn_amd/libfgnn_hip_prof.so FGNN_PROF=1 python bench.py --mode fwd --steps 1 --warmup 1 --no-graph --no-cpu-baseline > /dev/null 2> $OUT/layer_phase_timeline.txt
ls -la $OUT
