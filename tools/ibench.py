#!/usr/bin/env python3
"""iid_mapping_in (1x1 map -> InstanceNorm -> ReLU) at B = 4096: the fused kernel (with / without storing z) against the staged pair."""
import contextlib, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'factor-graph-neural-network_amd'))
import torch
from fgnn_amd.mpnn import blocks, iid_mapping_in
dev = torch.device('cuda:0')
B = 4096
for N in (96, 48):
    for cin, cout in [(64, 64), (64, 128), (128, 256), (256, 256), (256, 128), (128, 64)]:
        m = iid_mapping_in(cin, cout).to(dev)
        xs = [torch.randn(B, N, 1, cin, device=dev).bfloat16().permute(0, 3, 1, 2) for _ in range(4)]     # rotate: inputs from HBM
        res = []
        for fused, grad in ((True, False), (True, True), (False, True)):
            blocks.FUSE_IID_IN = fused
            def run(i):
                with torch.autocast('cuda', dtype=torch.bfloat16), (contextlib.nullcontext() if grad else torch.no_grad()):
                    return m(xs[i % 4].requires_grad_(grad))
            for i in range(3): run(i)
            torch.cuda.synchronize()
            s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            s.record()
            for i in range(20): run(i)
            e.record(); torch.cuda.synchronize()
            res.append(s.elapsed_time(e) / 20 * 1e3)
        blocks.FUSE_IID_IN = True
        mb = B * N * (cin + cout) * 2 / 1e6
        print('N %2d  %3d -> %3d   fused %6.1f us (%4.0f GB/s)   fused + z %6.1f us   staged %6.1f us' % (N, cin, cout, res[0], mb / res[0] * 1e3, res[1], res[2]), flush=True)
