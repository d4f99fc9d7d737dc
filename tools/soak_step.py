#!/usr/bin/env python3
"""Soak: the benched training step (forward + backward, no optimizer) repeated on FIXED inputs and parameters — eagerly on the
two streams and as hipGraph replays — must give the bit-identical flat gradient every time (every kernel of the step is
atomic-free with fixed summation orders; a synchronisation error anywhere shows here as a rare difference).
    python tools/soak_step.py [batch] [iterations]"""
import contextlib, io, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'factor-graph-neural-network_amd'))
import torch
import fgnn_amd
from fgnn_amd.datapath import LdpcDataPath
from fgnn_amd.dp import FlatGradBucket
from fgnn_amd.graph import StepGraph
dev = torch.device('cuda:0')
B = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 60
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
    (torch.nn.functional.binary_cross_entropy_with_logits(logits.float().reshape(-1), label.reshape(-1)) + snr.float().square().mean()).backward()


bad = 0
for mode in ('eager', 'graph'):
    step = compute if mode == 'eager' else StepGraph(compute)
    first = None
    for it in range(iters):
        step()
        torch.cuda.synchronize()
        g = bucket.flat.clone()
        if first is None:
            first = g
            assert bool(torch.isfinite(g).all()) and float(g.abs().max()) > 0
        elif not torch.equal(g, first):
            bad += 1
            d = (g != first)
            print('%s iteration %d: %d of %d gradient elements differ (max |diff| %.3e)' % (mode, it, int(d.sum()), g.numel(), float((g - first).abs().max())), flush=True)
    print('%s: %d iterations at batch %d, gradient norm %.6e' % (mode, iters, B, float(first.norm())), flush=True)
print('SOAK', 'FAILED' if bad else 'OK', bad)
sys.exit(1 if bad else 0)
