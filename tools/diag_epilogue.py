#!/usr/bin/env python3
"""Repeat one RAW forward launch of the parity operator on fixed inputs and report run-to-run differences (a race shows as such):
which samples / destinations / channels of y, argmax and the statistics partials differ.   python tools/diag_epilogue.py B reps"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for d in ('factor-graph-neural-network_amd', 'tests', 'oracle'):
    sys.path.insert(0, os.path.join(ROOT, d))
import torch
import test_mpconv_sg_gpu as T
from fgnn_amd import ops, _hip
dev = torch.device('cuda:0')
B = int(sys.argv[1]) if len(sys.argv) > 1 else 1100
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 20
shapes = [(64, 64, 96, 48, 6), (64, 64, 48, 96, 3), (128, 64, 48, 96, 3), (64, 64, 37, 95, 3), (128, 64, 96, 48, 6), (64, 128, 96, 48, 6),
          (64, 128, 48, 96, 3), (64, 64, 40, 20, 3)]
if len(sys.argv) > 3:
    shapes = [shapes[int(sys.argv[3])]]
for shape in shapes:
    nin, nou, N, M, k = shape
    x, idx, et, W, bias, g = T._problem(shape, B, dev, seed=B)
    xd, idxd, etd = T._dev_views(x, idx, et, dev)
    Wd, bd = W.to(dev), bias.to(dev)
    for stats in (True, False):
        first, nbad = None, 0
        for r in range(reps):
            y, am = ops.mpconv_forward_raw(xd, idxd, etd, Wd, bd, nou, 4, 0, _hip.AGG_MAX, want_argmax=True, bn=T.H.bn_spec_for(nou, dev) if stats else None)
            torch.cuda.synchronize()
            cur = (y.clone(), am.clone())
            if first is None:
                first = cur
                continue
            dy, da = (cur[0] != first[0]), (cur[1] != first[1])
            if bool(dy.any()) or bool(da.any()):
                nbad += 1
                bs = dy.flatten(1).any(1).nonzero().flatten().tolist()
                msg = '   run %d: y differs in %d elements, argmax in %d; samples %s (per-workgroup turn %s, workgroup %s)' % (
                    r, int(dy.sum()), int(da.sum()), bs[:10], [b // 256 for b in bs[:10]], [b % 256 for b in bs[:10]])
                if bs:
                    where = dy[bs[0]].nonzero()        # [channel, destination, 0]
                    msg += '; sample %d: channels %s, destinations %s' % (bs[0], sorted(set(where[:, 0].tolist()))[:12], sorted(set(where[:, 1].tolist()))[:12])
                print(msg, flush=True)
        print(shape, 'B', B, 'stats epilogue' if stats else 'plain         ', '%d of %d runs differ from the first' % (nbad, reps - 1), flush=True)
