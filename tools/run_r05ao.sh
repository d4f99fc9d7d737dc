#!/bin/sh
mkdir -p gpurun_out/r05ao
python -c "import torch; print('priority range', torch.cuda.Stream.priority_range() if hasattr(torch.cuda.Stream,'priority_range') else None); s=torch.cuda.Stream(priority=-1); print(s.priority)"
run() { env "$@" python bench.py --no-cpu-baseline 2>gpurun_out/r05ao/err.log > gpurun_out/r05ao/b.json; python -c "import json,sys; d=json.load(open('gpurun_out/r05ao/b.json')); print(' '.join(sys.argv[1:]) or 'default', d['ms_per_step'])" "$@"; }
run A=default
run FGNN_HIGH_PRIO=1
run FGNN_HIGH_PRIO=1 FGNN_WGRAD_STREAM=1
run FGNN_WGRAD_STREAM=1
tail -2 gpurun_out/r05ao/err.log
