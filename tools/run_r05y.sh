#!/bin/sh
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd $R
sh tools/profile_timeline.sh ${1:-r05y}/timeline > /dev/null 2>&1
cp gpurun_out/${1:-r05y}/timeline/step_sequence.csv /tmp/seq.csv
sed -i 's#profiles/r05/train_step_sequence.csv#/tmp/seq.csv#' tools/profile_step_traffic.sh
sh tools/profile_step_traffic.sh ${1:-r05y}/traffic > /dev/null 2>&1
head -60 gpurun_out/${1:-r05y}/traffic/step_traffic.txt
