#!/bin/sh
# Run on the GPU box: the same as tools/profile_timeline.sh for the inference forward (the step window is marked by the two
# edge-type MLP launches that open every forward).   gpurun -- sh tools/profile_timeline_fwd.sh [outdir]
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/${1:-timeline_fwd}
mkdir -p $O
cd $R
rm -rf /tmp/tlf
timeout 600 rocprofv3 --kernel-trace --output-format csv -d /tmp/tlf -o tl -- python bench.py --steps 6 --warmup 2 --mode fwd --no-cpu-baseline > $O/bench.log 2>&1
T=$(find /tmp/tlf -name "*kernel_trace.csv" | head -1)
python tools/timeline.py $T --marker edge_mlp_fwd --marker-stride 2 --marker-skip ${2:-4} --json $O/timeline.json > $O/timeline.txt 2>&1
cat $O/timeline.txt | head -70
