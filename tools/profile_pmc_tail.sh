#!/bin/sh
# Run on the GPU box: SQ issue counters (three 8-counter passes, each its OWN rocprofv3 run with --kernel-trace only) of the
# fused block-tail kernels at one shape.   sh tools/profile_pmc_tail.sh <outdir> <R,Cout> <modes e.g. stats>
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/$1
mkdir -p $OUT
cd $R
run() { # name, counters
    rm -rf /tmp/pm_$1
    TB_SHAPE=$SHAPE TB_MODES=$MODES timeout 300 rocprofv3 --kernel-trace --pmc $2 --output-format csv -d /tmp/pm_$1 -o $1 -- \
        python tools/tbench.py 3 > /tmp/pm_$1.log 2>&1
    find /tmp/pm_$1 -name "*counter_collection.csv" -exec cp {} $OUT/$1.csv \;
}
SHAPE=$2
MODES=$3
run issue_a "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_LDS"
run issue_b "SQ_INSTS_SALU SQ_INSTS_VMEM SQ_INSTS_MFMA SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS"
run issue_c "SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_MISC SQ_INSTS_SMEM SQ_VALU_MFMA_BUSY_CYCLES SQ_INST_CYCLES_VMEM SQ_WAVES_EQ_64 SQ_INSTS_VALU_MFMA_MOPS_BF16"
run fetch "FETCH_SIZE"
run write "WRITE_SIZE"
