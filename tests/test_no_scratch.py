"""No kernel on the benched LDPC paths may spill: a spill reload is a memory operation whose wait also waits for every load issued before
it — in kernels that keep the next tile / sample in flight that was an HBM round trip per tile.  The check is the
compiler's own resource report (`-Rpass-analysis=kernel-resource-usage`: ScratchSize) for the sources of those kernels, cross-compiled for
gfx950 — no GPU needed."""
import os
import re
import shutil
import subprocess
from concurrent.futures import ThreadPoolExecutor

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, 'factor-graph-neural-network_amd', 'csrc')
HIPCC = shutil.which('hipcc') or '/opt/rocm/bin/hipcc'

# source -> the kernels (mangled-name substrings) that must have no scratch; other instances in the same source are reported, not asserted
MUST = {
    'factor_layer_fwd.hip': ['factor_layer_fwd_kernel'],
    'mpconv_block_fwd.hip': ['mpconv_block_fwd_kernel', 'mpconv_block_rows1_kernel', 'mpconv_block_fanin_kernel', 'mpconv_block_fanout_kernel'],
    'linear_fwd_b16.hip': ['linear_fwd_b16_kernel', 'linear_instnorm_fwd_kernel', 'linear_multi_b16_kernel'],
    'mpconv_bwd_ws.hip': ['mpconv_bwd_ws_kernel'],
    'block_tail.hip': ['block_tail_kernel', 'block_head_bwd_kernel'],
    'mpconv_fwd_ws.hip': ['mpconv_fwd_ws_kernelILi64E'],          # (the 128-input instances spill at their 128-register budget: consumers only, profiles/r06/README.md)
}


def _scratch(src):
    cmd = [HIPCC, '--offload-arch=gfx950', '-O3', '-std=c++17',      # (the Makefile's flags)
           '-I' + os.path.join(ROOT, 'include'), '-I' + CSRC, '-Wno-unused-function',
           '-Wno-pass-failed', '-fPIC'] + (['-fno-slp-vectorize'] if src == 'block_tail.hip' else []) + ['-c', '--cuda-device-only', os.path.join(CSRC, src), '-o', os.devnull,
           '-Rpass-analysis=kernel-resource-usage']
    err = subprocess.run(cmd, capture_output=True, text=True, timeout=1500).stderr
    out, name = {}, None
    for line in err.splitlines():
        m = re.search(r'Function Name: (\S+)', line)
        if m:
            name = m.group(1)
        m = re.search(r'ScratchSize \[bytes/lane\]: (\d+)', line)
        if m and name:
            out[name] = int(m.group(1))
    return out


@pytest.mark.skipif(not os.path.exists(HIPCC), reason='no hipcc')
def test_benched_kernels_have_no_scratch():
    with ThreadPoolExecutor(max_workers=len(MUST)) as ex:
        reports = dict(zip(MUST, ex.map(_scratch, MUST)))
    bad = []
    for src, subs in MUST.items():
        rep = reports[src]
        assert rep, 'no resource report for %s' % src
        for sub in subs:
            hits = {k: v for k, v in rep.items() if sub in k}
            assert hits, (src, sub)
            bad += ['%s: %s spills %d bytes per lane' % (src, k, v) for k, v in hits.items() if v]
    assert not bad, '\n'.join(bad)
