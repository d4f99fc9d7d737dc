#!/bin/sh
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd $R
for v in "" "FGNN_FH_DBG=1" "FGNN_FH_DBG=2" "FGNN_FH_DBG=3" "FGNN_FH_GRID=256" "FGNN_FH_GRID=512" "FGNN_FH_GRID=768"; do
  echo "== $v"; env $v python tools/kbench.py --dtype bf16 --only "hyper  V->F 64->64" --cold 8 2> /dev/null | grep -v amdgpu.ids | cut -c1-110
done
