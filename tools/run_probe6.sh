#!/bin/sh
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r06p
mkdir -p $O /tmp/dotd
cd /tmp/dotd
DEBUG_HIP_GRAPH_DOT_PRINT=1 timeout 300 python $R/bench.py --no-cpu-baseline --steps 3 > $O/b_dot.json 2> $O/b_dot.err
ls -la /tmp/dotd | head; ls -la $R/*.dot 2>/dev/null | head
cp /tmp/dotd/graph_* $O/
grep -il dot $O/b_dot.err | head -2; grep -i "dot" $O/b_dot.err | head -5
