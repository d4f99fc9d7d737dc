import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (os.path.join(ROOT, 'factor-graph-neural-network_amd'), os.path.join(ROOT, 'oracle'),
          os.path.join(ROOT, 'tests'), ROOT):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line('markers', 'gpu: needs a real MI355X (run with -m gpu on the GPU box)')


@pytest.fixture(scope='session')
def dev():
    import torch
    if not torch.cuda.is_available():
        pytest.skip('no ROCm device')
    return torch.device('cuda:0')


@pytest.fixture(params=['finaliser_launch', 'in_kernel_fold'])
def finaliser_mode(request, dev):
    """Both ways a statistics-producing launch can be finalised (include/fgnn_hip.h: fgnn_set_inkernel_finalisers): the small
    finaliser launch behind it (the default) and the producer's last workgroup (csrc/fgnn_gridfold.h)."""
    from fgnn_amd import _hip
    L = _hip.lib()
    was = L.fgnn_set_inkernel_finalisers(1 if request.param == 'in_kernel_fold' else 0)
    yield request.param
    L.fgnn_set_inkernel_finalisers(was)
