#!/usr/bin/env python3
"""Micro-benchmark of the MERGED weight-gradient launch (fgnn_linear_wgrad_multi: several maps over one state) against the same
maps one launch each (fgnn_linear_wgrad), cold operands (rotating copies), folds included."""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'factor-graph-neural-network_amd'))
import torch
from fgnn_amd import _hip, ops
dev = torch.device('cuda:0')
L = _hip.lib()
P = _hip._ptr
NCOPY = 6
for R in (4096 * 96, 4096 * 48):
    for cin, couts in [(64, (64, 64, 64)), (64, (64, 128, 64)), (128, (64, 256, 64)), (256, (64, 64)), (256, (64, 128, 64)), (128, (64, 64, 64)),
                       (64, (64, 64)), (128, (64, 256)), (256, (64, 128))]:
        n = len(couts)
        xs = [torch.randn(R, cin, device=dev).to(torch.bfloat16) for _ in range(NCOPY)]
        gys = [[torch.randn(R, c, device=dev).to(torch.bfloat16) for c in couts] for _ in range(NCOPY)]
        gws = [torch.zeros(c, cin, device=dev) for c in couts]
        gbs = [torch.zeros(c, device=dev) for c in couts]
        cc = (ctypes.c_int32 * n)(*couts)
        nb = int(L.fgnn_linear_wgrad_multi_workspace_bytes(R, cin, n, cc))
        wsm = torch.empty(nb // 4, device=dev)
        ws1 = [torch.empty(int(L.fgnn_linear_wgrad_workspace_bytes(R, cin, c)) // 4, device=dev) for c in couts]
        it = [0]
        def multi():
            k = it[0] % NCOPY; it[0] += 1
            _hip.check(L.fgnn_linear_wgrad_multi(P(xs[k]), R, cin, n, (ctypes.c_void_p * n)(*[P(t) for t in gys[k]]), cc,
                                                 (ctypes.c_void_p * n)(*[P(t) for t in gws]), (ctypes.c_void_p * n)(*[P(t) for t in gbs]),
                                                 P(wsm), nb, _hip.stream_ptr()))
        def single():
            k = it[0] % NCOPY; it[0] += 1
            for gy, gw, gb, c, w in zip(gys[k], gws, gbs, couts, ws1):
                _hip.check(L.fgnn_linear_wgrad(P(xs[k]), P(gy), R, cin, c, _hip.BF16, P(gw), P(gb), P(w), w.numel() * 4, _hip.stream_ptr()))
        res = []
        for fn in (multi, single):
            for _ in range(3): fn()
            torch.cuda.synchronize()
            s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            s.record()
            for _ in range(12): fn()
            e.record(); torch.cuda.synchronize()
            res.append(s.elapsed_time(e) / 12 * 1e3)
        mbm = 2 * R * (cin + sum(couts)) / 1e6
        mbs = 2 * R * (n * cin + sum(couts)) / 1e6
        print('R=%6d %3d -> %-15s merged %7.1f us (%5.2f TB/s of %4.0f MB) | one by one %7.1f us (%5.2f TB/s of %4.0f MB)'
              % (R, cin, couts, res[0], mbm / res[0], mbm, res[1], mbs / res[1], mbs), flush=True)
        del xs, gys
