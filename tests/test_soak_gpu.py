"""GPU soak: the benched training step repeated on fixed inputs and parameters gives the bit-identical flat gradient every time,
launched eagerly on the two streams and replayed as a hipGraph, and the two agree bit for bit.  Every kernel of the step is
atomic-free with fixed summation orders, so a synchronisation error anywhere in the step (round 4 found one in the forward
operator's wave hand-off that no parity tolerance would ever see) shows here as a rare difference.  tools/soak_step.py is the
long form (batch 4096, 60 + 60 iterations)."""
import contextlib
import io
import os
import sys

import pytest
import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'factor-graph-neural-network_amd'))

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize('B', [1100, 300])
def test_training_step_is_bit_reproducible_eager_and_replayed(B, dev):
    import fgnn_amd
    from fgnn_amd.datapath import LdpcDataPath
    from fgnn_amd.dp import FlatGradBucket
    from fgnn_amd.graph import StepGraph
    torch.manual_seed(1)
    with contextlib.redirect_stdout(io.StringIO()):
        model = fgnn_amd.LDPCModel(2, 6, 4, aggregator='max').to(dev).train()
    data = LdpcDataPath(dev).sample(B, seed=3, dtype=torch.bfloat16)
    bucket = FlatGradBucket(model.parameters(), flatten_params=True)
    label = data[6][:, :48].float().contiguous()

    def compute():
        bucket.zero()
        with torch.autocast('cuda', dtype=torch.bfloat16):
            logits, snr = model(*data[:6])
        (torch.nn.functional.binary_cross_entropy_with_logits(logits.float().reshape(-1), label.reshape(-1))
         + snr.float().square().mean()).backward()

    firsts = {}
    for mode in ('eager', 'graph'):
        step = compute if mode == 'eager' else StepGraph(compute)
        for it in range(25):
            step()
            torch.cuda.synchronize()
            g = bucket.flat.clone()
            if it == 0:
                firsts[mode] = g
                assert bool(torch.isfinite(g).all()) and float(g.abs().max()) > 0
            else:
                assert torch.equal(g, firsts[mode]), '%s: iteration %d differs from iteration 0 in %d elements' % (
                    mode, it, int((g != firsts[mode]).sum()))
    assert torch.equal(firsts['eager'], firsts['graph'])
