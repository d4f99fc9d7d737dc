#!/bin/sh
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/${1:-r05x}
mkdir -p $O
cd $R
timeout 1500 python -m pytest tests/test_assemblies_gpu.py tests/test_parity_pins_gpu.py tests/test_fullsize_properties_gpu.py -m gpu -q > $O/pytest_a.log 2>&1; tail -4 $O/pytest_a.log
for i in 1 2 3; do
python bench.py --no-cpu-baseline 2> $O/bench$i.err > $O/bench$i.json
python -c "import json,sys; d=json.loads(open('$O/bench$i.json').read().strip().splitlines()[-1]); print('bench', d['ms_per_step'], d['config']['peak_hbm_GB'])"
done
FGNN_NO_FANIN_ID=1 python bench.py --no-cpu-baseline 2> /dev/null > $O/bench_noid.json
python -c "import json,sys; d=json.loads(open('$O/bench_noid.json').read().strip().splitlines()[-1]); print('bench general fan-in', d['ms_per_step'])"
python bench.py --mode fwd --no-cpu-baseline 2> /dev/null > $O/bench_fwd.json
python -c "import json,sys; d=json.loads(open('$O/bench_fwd.json').read().strip().splitlines()[-1]); print('bench fwd', d['ms_per_step'])"
