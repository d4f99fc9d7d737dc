#!/bin/sh
# Run on the GPU box (e.g. `gpurun -- sh tools/profile_pmc.sh`): the PMC passes behind profiles/r01/pmc_*.{csv,json}.
# Each counter set is its OWN rocprofv3 run with --kernel-trace only (never combined with sys / hip / hsa tracing), over
# tools/kbench.py at the LDPC parity 64->64 shapes; FETCH_SIZE is corrected x2 for gfx950 when the JSON is written
# (MI355X_MICROARCH.md).  Outputs land under gpurun_out/pmc/.
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out/pmc
cd $R
run() { # name, counters, shape, extra kbench flags
    rm -rf /tmp/pm_$1
    timeout 300 rocprofv3 --kernel-trace --pmc $2 --output-format csv -d /tmp/pm_$1 -o $1 -- \
        python tools/kbench.py --layout cl --dtype bf16 --only "$3" --iters 3 $4 > /tmp/pm_$1.log 2>&1
    find /tmp/pm_$1 -name "*counter_collection.csv" -exec cp {} $R/gpurun_out/pmc/$1.csv \;
}
for dir in "V->F" "F->V"; do
    tag=$(echo $dir | tr -d '>-')
    run fwd_fetch_$tag "FETCH_SIZE" "parity $dir 64->64" ""
    run fwd_write_$tag "WRITE_SIZE" "parity $dir 64->64" ""
    run bwd_fetch_$tag "FETCH_SIZE" "parity $dir 64->64" "--bwd"
    run bwd_write_$tag "WRITE_SIZE" "parity $dir 64->64" "--bwd"
done
run fwd_issue_a "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_LDS" "parity V->F 64->64" ""
run fwd_issue_b "SQ_INSTS_SALU SQ_INSTS_VMEM SQ_INSTS_MFMA SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS" "parity V->F 64->64" ""
run bwd_issue_a "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_LDS" "parity V->F 64->64" "--bwd"
run bwd_issue_b "SQ_INSTS_SALU SQ_INSTS_VMEM SQ_INSTS_MFMA SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS" "parity V->F 64->64" "--bwd"
