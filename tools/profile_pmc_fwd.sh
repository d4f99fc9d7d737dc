#!/bin/sh
# Run on the GPU box: SQ issue counters (two 8-counter passes) + FETCH/WRITE passes of ONE kbench shape.
#   sh tools/profile_pmc_fwd.sh <outdir> "<shape substring>" [extra kbench flags, e.g. --bwd]
# Each counter set is its OWN rocprofv3 run with --kernel-trace only.
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/$1
SHAPE="$2"
EXTRA="$3"
mkdir -p $OUT
cd $R
run() { # name, counters
    rm -rf /tmp/pm_$1
    timeout 300 rocprofv3 --kernel-trace --pmc $2 --output-format csv -d /tmp/pm_$1 -o $1 -- \
        python tools/kbench.py --layout cl --dtype bf16 --only "$SHAPE" --iters 3 $EXTRA > /tmp/pm_$1.log 2>&1
    find /tmp/pm_$1 -name "*counter_collection.csv" -exec cp {} $OUT/$1.csv \;
}
run issue_a "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_LDS"
run issue_b "SQ_INSTS_SALU SQ_INSTS_VMEM SQ_INSTS_MFMA SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS"
run issue_c "SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_MISC SQ_INSTS_SMEM SQ_VALU_MFMA_BUSY_CYCLES SQ_INST_CYCLES_VMEM SQ_WAVES_EQ_64 SQ_INSTS_VALU_MFMA_MOPS_BF16"
run fetch "FETCH_SIZE"
run write "WRITE_SIZE"
