#!/bin/sh
# gpurun -- sh tools/run_probe.sh : the graph fork probe under rocprofv3, then today's default bench line
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r06p
mkdir -p $O
cd $R
rm -rf /tmp/pr
timeout 300 rocprofv3 --kernel-trace --output-format csv -d /tmp/pr -o pr -- python tools/ubench/graph_fork_probe.py > $O/probe.log 2>&1
T=$(find /tmp/pr -name "*kernel_trace.csv" | head -1)
python tools/ubench/graph_fork_probe.py --parse $T > $O/probe.txt 2>&1
cat $O/probe.txt
timeout 600 python bench.py --no-cpu-baseline > $O/bench_base.json 2> $O/bench_base.err
tail -c 600 $O/bench_base.json
