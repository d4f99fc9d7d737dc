#!/bin/sh
mkdir -p gpurun_out/r05ax
L=$PWD/factor-graph-neural-network_amd/fgnn_amd/libfgnn_hip_d3.so
echo base; python tools/tbench.py 2>&1 | grep -v amdgpu | cut -c1-60
echo depth3; FGNN_HIP_LIB=$L python tools/tbench.py 2>&1 | grep -v amdgpu | cut -c1-60
run() { env "$@" python bench.py --no-cpu-baseline --steps 40 2>gpurun_out/r05ax/err.log > gpurun_out/r05ax/b.json; python -c "import json,sys; d=json.load(open('gpurun_out/r05ax/b.json')); print(' '.join(sys.argv[1:])[:30] or 'default', round(d['ms_per_step'],3))" "$@"; }
for i in 1 2 3; do run A=base; run FGNN_HIP_LIB=$L; done
