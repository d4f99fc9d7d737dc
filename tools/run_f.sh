#!/bin/sh
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
timeout 1800 python -m pytest tests/test_block_tail_gpu.py tests/test_assemblies_gpu.py tests/test_parity_pins_gpu.py -m gpu -q -x 2>&1 | tail -4
