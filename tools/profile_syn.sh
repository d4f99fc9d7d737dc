#!/bin/sh
# Run on the GPU box (gpurun -- sh tools/profile_syn.sh): rocprofv3 kernel statistics of `bench.py --workload syn_pw|syn_hop`
# (training step) and FETCH_SIZE / WRITE_SIZE passes (each its own run) of the two synthetic-PGM operator kernels at the
# order-9 shape, batch 1024.  Outputs under gpurun_out/prof_syn/; tools/refresh_profiles_syn.py copies them into profiles/r02/.
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/prof_syn
mkdir -p $OUT
cd $R
for w in syn_pw syn_hop; do
  rm -rf /tmp/ps_$w
  timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/ps_$w -o $w -- python bench.py --workload $w --steps 10 --warmup 3 --no-cpu-baseline > /tmp/ps_$w.log 2>&1
  grep "^{\"metric" /tmp/ps_$w.log | tail -1 > $OUT/bench_$w.json
  find /tmp/ps_$w -name "*kernel_stats*.csv" -exec cp {} $OUT/${w}_kernel_stats.csv \;
done
pmc() { # name, counter, extra kbench flags
  rm -rf /tmp/pm_$1
  timeout 300 rocprofv3 --kernel-trace --pmc $2 --output-format csv -d /tmp/pm_$1 -o $1 -- \
      python tools/kbench.py --syn --dtype f32 --batch 1024 --shared-et --only "syn hop 64->64" --iters 3 $3 > /tmp/pm_$1.log 2>&1
  find /tmp/pm_$1 -name "*counter_collection.csv" -exec cp {} $OUT/$1.csv \;
}
pmc pmc_fwd_ext_fetch FETCH_SIZE ""
pmc pmc_fwd_ext_write WRITE_SIZE ""
pmc pmc_bwd_ext_fetch FETCH_SIZE "--bwd"
pmc pmc_bwd_ext_write WRITE_SIZE "--bwd"
