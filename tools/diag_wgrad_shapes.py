#!/usr/bin/env python3
"""Which (rows, Cin, Cout) the node-wise weight-gradient kernel is launched with in one LDPCModel training step (B = 4096)."""
import collections, contextlib, io, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'factor-graph-neural-network_amd'))
import torch
import fgnn_amd
from fgnn_amd import _hip
from fgnn_amd.datapath import LdpcDataPath
from fgnn_amd.dp import FlatGradBucket
dev = torch.device('cuda:0')
B = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
with contextlib.redirect_stdout(io.StringIO()):
    model = fgnn_amd.LDPCModel(2, 6, 4, aggregator='max').to(dev).train()
data = LdpcDataPath(dev).sample(B, seed=1, dtype=torch.bfloat16)
bucket = FlatGradBucket(model.parameters(), flatten_params=True)
L = _hip.lib()
seen = collections.Counter()
for name in ('fgnn_linear_wgrad',):
    orig = getattr(L, name)
    def wrap(*a, _o=orig):
        seen[(int(a[2]), int(a[3]), int(a[4]))] += 1
        return _o(*a)
    setattr(L, name, wrap)
for it in range(2):
    seen.clear()
    bucket.zero()
    with torch.autocast('cuda', dtype=torch.bfloat16):
        logits, snr = model(*data[:6])
    (logits.float().square().mean() + snr.float().square().mean()).backward()
torch.cuda.synchronize()
for (R, ci, co), n in sorted(seen.items(), key=lambda kv: -kv[1]):
    print('rows %7d (%3d per codeword)  %3d -> %3d   x %d' % (R, R // B, ci, co, n))
print('total', sum(seen.values()))
