#!/usr/bin/env python3
"""Micro-benchmark of the node-wise-map weight-gradient kernel against rocBLAS (torch matmul)."""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'factor-graph-neural-network_amd'))
import torch
from fgnn_amd import _hip, ops
dev = torch.device('cuda:0')
L = _hip.lib()
R = 4096 * 96
for a in sys.argv[1:]:
    if a.startswith('--rows='):
        R = int(a.split('=')[1])          # e.g. --rows=61440: the synthetic-PGM maps (1024 graphs x 60 nodes)
for dt in (torch.bfloat16,) if '--bf16' in sys.argv else (torch.float32,) if '--f32' in sys.argv else (torch.bfloat16, torch.float32):
    for cin, cout in [(64, 64), (64, 128), (128, 64), (64, 256), (256, 64), (128, 256), (256, 128), (256, 256), (96, 64), (7, 64), (2, 64), (64, 4)]:
        x = torch.randn(R, cin, device=dev).to(dt); gy = torch.randn(R, cout, device=dev).to(dt)
        gw = torch.zeros(cout, cin, device=dev); gb = torch.zeros(cout, device=dev)
        ws = ops._workspace(dev, int(L.fgnn_linear_wgrad_workspace_bytes(R, cin, cout)))
        def mine():
            _hip.check(L.fgnn_linear_wgrad(_hip._ptr(x), _hip._ptr(gy), R, cin, cout, _hip.dtype_code(x), _hip._ptr(gw), _hip._ptr(gb), _hip._ptr(ws), ws.numel() * 4, _hip.stream_ptr()))
        def blas():
            return gy.t() @ x
        res = []
        for fn in (mine, blas):
            for _ in range(3): fn()
            torch.cuda.synchronize()
            s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            s.record()
            for _ in range(10): fn()
            e.record(); torch.cuda.synchronize()
            res.append(s.elapsed_time(e) / 10 * 1e3)
        mb = (x.numel() + gy.numel()) * x.element_size() / 1e6
        print('%s %4d x %4d  mine %8.1f us (%6.1f GB/s)   rocBLAS %8.1f us' % (str(dt)[6:], cin, cout, res[0], mb / res[0] * 1e3, res[1]))
