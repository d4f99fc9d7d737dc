#!/bin/sh
# round 5: SOLO staging roles in the degree-6 backward instance
mkdir -p gpurun_out/r05am
python -m pytest tests/test_mpconv_sg_gpu.py -x -q -m gpu -k "backward or bwd or reproducible" > gpurun_out/r05am/t1.log 2>&1; tail -3 gpurun_out/r05am/t1.log
for c in 1 8; do
python tools/kbench.py --dtype bf16 --regular --bwd --only parity --cold $c 2>&1 | grep -v amdgpu | grep "V->F" > gpurun_out/r05am/kbench_solo_cold$c.log; cat gpurun_out/r05am/kbench_solo_cold$c.log
FGNN_BWD_WS_NO_SOLO=1 python tools/kbench.py --dtype bf16 --regular --bwd --only parity --cold $c 2>&1 | grep -v amdgpu | grep "V->F" > gpurun_out/r05am/kbench_nosolo_cold$c.log; cat gpurun_out/r05am/kbench_nosolo_cold$c.log
done
