#!/bin/sh
# Run on the GPU box (round 5): the step's turn-around — classifier closing pair as one kernel, regressor head through the map
# kernels, the loss as one launch each way.
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/${1:-r05m}
mkdir -p $O
cd $R
timeout 900 python -m pytest tests/test_mpconv_gpu.py -m gpu -q -k "instnorm or instance_norm" > $O/pytest_in.log 2>&1
tail -3 $O/pytest_in.log
timeout 1500 python -m pytest tests/test_assemblies_gpu.py tests/test_parity_pins_gpu.py -m gpu -q > $O/pytest_model.log 2>&1
tail -8 $O/pytest_model.log
for i in 1 2; do
python bench.py --no-cpu-baseline > $O/bench_a$i.json 2> /dev/null
FGNN_NO_FAST_REGRESSOR=1 python bench.py --no-cpu-baseline > $O/bench_torch_regressor$i.json 2> /dev/null
FGNN_NO_INSTNORM_DOT=1 python bench.py --no-cpu-baseline > $O/bench_no_dot$i.json 2> /dev/null
done
python bench.py --mode fwd --no-cpu-baseline > $O/bench_fwd.json 2> /dev/null
python bench.py --dtype f32 --no-cpu-baseline > $O/bench_f32.json 2> /dev/null
python - <<'PY'
import json,glob,os
for f in sorted(glob.glob(os.environ.get('GRAFT_REPO_ROOT','.')+'/gpurun_out/r05m/bench_*.json')):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1]); print(os.path.basename(f), d['ms_per_step'], d['roofline']['frac'])
    except Exception as e: print(f, 'ERR', e)
PY
