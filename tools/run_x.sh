#!/bin/sh
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r06x
mkdir -p $O
cd $R
FGNN_EXT_BWD_PIECES=0 timeout 300 python tools/xbench.py --save=/tmp/exact.pt > $O/xbench_0.txt 2>&1
for np in ${NPS:-2}; do
  echo "== FGNN_EXT_BWD_PIECES=$np"
  FGNN_EXT_BWD_PIECES=$np timeout 300 python tools/xbench.py --compare=/tmp/exact.pt 2>&1 | grep -v "amdgpu.ids" | cut -c1-20,82-125,200-400 | tee $O/xbench_$np.txt
done
echo "== prof NP=2"
FGNN_HIP_LIB=$R/factor-graph-neural-network_amd/fgnn_amd/libfgnn_hip_prof.so FGNN_PROF=1 FGNN_EXT_BWD_PIECES=2 timeout 300 python tools/xbench.py 1024 2>&1 | grep "prof extq" | tail -8 | tee $O/prof_2.txt
