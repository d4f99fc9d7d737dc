#!/bin/sh
# Run on the GPU box: HBM bytes of EVERY kernel of the benched training step (two rocprofv3 --kernel-trace --pmc passes, FETCH_SIZE and
# WRITE_SIZE, each its own run) -> tools/step_traffic.py sums them per kernel and per step: how close the whole step is to the HBM
# roofline, not just its dominant kernel.     sh tools/profile_step_traffic.sh <outdir under gpurun_out>
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/${1:-step_traffic}
mkdir -p $OUT
cd $R
for c in FETCH_SIZE WRITE_SIZE; do
    rm -rf /tmp/st_$c
    timeout 900 rocprofv3 --kernel-trace --pmc $c --output-format csv -d /tmp/st_$c -o $c -- \
        python bench.py --steps 2 --warmup 1 --no-cpu-baseline > $OUT/$c.log 2>&1
    find /tmp/st_$c -name "*counter_collection.csv" -exec cp {} $OUT/$c.csv \;
done
python tools/step_traffic.py $OUT/FETCH_SIZE.csv $OUT/WRITE_SIZE.csv ${FGNN_STEP_SEQ:-profiles/r05/train_step_sequence.csv} > $OUT/step_traffic.txt
head -40 $OUT/step_traffic.txt
