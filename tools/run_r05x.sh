#!/bin/sh
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/${1:-r05x}
mkdir -p $O
cd $R
timeout 1500 python -m pytest tests/test_assemblies_gpu.py -m gpu -q -k "loss or merged or fold or graph_replay or reproducible or fast_path" > $O/pytest_a.log 2>&1; tail -4 $O/pytest_a.log
for i in 1 2 3; do
python bench.py --no-cpu-baseline 2> $O/bench$i.err > $O/bench$i.json
python -c "import json,sys; d=json.loads(open('$O/bench$i.json').read().strip().splitlines()[-1]); print('bench', d['ms_per_step'], d['config']['peak_hbm_GB'])"
done
python bench.py --dtype f32 --no-cpu-baseline 2> /dev/null > $O/bench_f32.json
python -c "import json,sys; d=json.loads(open('$O/bench_f32.json').read().strip().splitlines()[-1]); print('bench f32', d['ms_per_step'])"
python bench.py --mode fwd --no-cpu-baseline 2> /dev/null > $O/bench_fwd.json
python -c "import json,sys; d=json.loads(open('$O/bench_fwd.json').read().strip().splitlines()[-1]); print('bench fwd', d['ms_per_step'])"
