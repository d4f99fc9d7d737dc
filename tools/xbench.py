#!/usr/bin/env python3
"""The synthetic-PGM operator (f32, 16 edge types, ORIG_WITH_DIFF, 64 -> 64; BASELINE configs 2 / 5) stand-alone: forward and
backward call times at the shapes of train_syn_pw_factor.py / train_syn_hop_factor.py (60 nodes, degree 2 / 9).
    python tools/xbench.py [batch ...]          FGNN_EXT_BWD_PIECES=0|2|3 selects the backward kernel (read once per process)"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'factor-graph-neural-network_amd'))
import torch
from fgnn_amd import _hip, ops

dev = torch.device('cuda:0')
args = [a for a in sys.argv[1:] if not a.startswith('--')]
save = next((a.split('=', 1)[1] for a in sys.argv[1:] if a.startswith('--save=')), None)          # write the gradients (torch.save)
compare = next((a.split('=', 1)[1] for a in sys.argv[1:] if a.startswith('--compare=')), None)    # ... and compare with a saved set
batches = [int(a) for a in args] or [256, 1024]
saved = {}
ref = torch.load(compare) if compare else None


def timed(fn, iters=10):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    best = 1e9
    for _ in range(3):
        a.record()
        for _ in range(iters):
            fn()
        b.record()
        torch.cuda.synchronize()
        best = min(best, a.elapsed_time(b) / iters * 1e3)
    return best


for B in batches:
    for N, k in ((60, 2), (60, 9)):
        g = torch.Generator().manual_seed(N + k)
        x = torch.randn(B, N, 1, 64, generator=g).to(dev).permute(0, 3, 1, 2).requires_grad_(True)
        idx = torch.randint(0, N, (1, N, k), generator=g).to(dev).expand(B, -1, -1)
        et = torch.randn(1, 16, N, k, generator=g).to(dev).requires_grad_(True)
        W = (torch.randn(128, 1024, generator=g) * 0.1).to(dev).requires_grad_(True)
        bias = torch.randn(64, generator=g).to(dev).requires_grad_(True)
        gy = torch.randn(B, N, 1, 64, generator=g).to(dev).permute(0, 3, 1, 2)
        z = ops.mpconv(x, idx, et.expand(B, -1, -1, -1), W, bias, 64, 16, 2, _hip.AGG_MAX)
        kf = _hip.lib().fgnn_last_kernel().decode()
        tf = timed(lambda: ops.mpconv(x, idx, et.expand(B, -1, -1, -1), W, bias, 64, 16, 2, _hip.AGG_MAX))
        tb = timed(lambda: torch.autograd.grad(z, [x, et, W, bias], gy, retain_graph=True))
        kb = _hip.lib().fgnn_last_kernel().decode()
        grads = torch.autograd.grad(z, [x, et, W, bias], gy, retain_graph=True)
        key = '%d_%d_%d' % (B, N, k)
        if save:
            saved[key] = [t.detach().cpu() for t in grads]
        acc = ''
        if ref is not None and key in ref:
            acc = '   vs saved: ' + ' '.join('%s %.1e' % (n, float((a.cpu().double() - b.double()).abs().max() / b.double().abs().max()))
                                             for n, a, b in zip(('gx', 'getype', 'gW', 'gbias'), grads, ref[key]))
        flops_b = B * 3 * 2 * 64 * 64 * 2048          # P recomputed + gx + gW (the three GEMMs of the backward)
        print('B %4d N %d k %d: forward %7.1f us (%s)   backward call %7.1f us = %5.1f TFLOP/s f32-equivalent (%s)'
              % (B, N, k, tf, kf, tb, flops_b / tb / 1e6, kb) + acc, flush=True)
if save:
    torch.save(saved, save)
