#!/bin/sh
# Run on the GPU box: the whole GPU suite + one bench line.
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/${1:-r05o}
mkdir -p $O
cd $R
timeout 2400 python -m pytest tests -m gpu -q > $O/pytest.log 2>&1
tail -12 $O/pytest.log
python bench.py --no-cpu-baseline > $O/bench.json 2> /dev/null
cat $O/bench.json | cut -c1-300
