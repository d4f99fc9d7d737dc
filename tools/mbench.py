#!/usr/bin/env python3
"""csrc/linear_fwd_b16.hip::linear_multi_b16_kernel (the merged gradient of a layer state, ops.FanBox) on cold tensors at the LDPC
step's shapes: us per launch and algorithmic TB/s (sources + addends read once, output written once)."""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'factor-graph-neural-network_amd'))
sys.path.insert(0, os.path.join(ROOT, 'tools'))
import torch
from fgnn_amd import _hip
from lbench import graph_time          # noqa: E402  (runs lbench's table on import: keep the output short with MBENCH_ONLY)
dev = torch.device('cuda:0')
L = _hip.lib()
P = _hip._ptr
SHAPES = [  # (nodes, K, C, addends)
    (96, (64, 64, 64), 64, 1), (96, (64, 128, 64), 64, 1), (96, (64, 256, 64), 128, 1), (96, (64, 256, 64), 256, 2), (96, (64, 128, 64), 256, 2),
    (96, (64, 64, 64), 128, 1), (48, (64, 0, 64), 64, 1), (48, (64, 256, 0), 256, 2), (48, (64, 256, 0), 128, 1), (48, (64, 128, 0), 256, 1)]
for N, K, C, nadd in SHAPES:
    R = 4096 * N
    per = R * (sum(K) + C * (1 + nadd)) * 2
    copies = max(2, int(2.4e9 // per))
    sets = []
    for _ in range(copies):
        xs = [torch.randn(R, k, device=dev).bfloat16() if k else None for k in K]
        adds = [torch.randn(R, C, device=dev).bfloat16() for _ in range(nadd)]
        sets.append((xs, adds, torch.empty(R, C, device=dev, dtype=torch.bfloat16)))
    Ws = [torch.randn(k, C, device=dev) * 0.1 if k else None for k in K]
    ks = (ctypes.c_int32 * 3)(*K)
    arr = lambda ts: (ctypes.c_void_p * 3)(*([P(t) for t in ts] + [None] * (3 - len(ts))))
    st = {'i': 0}
    def run():
        st['i'] = (st['i'] + 1) % copies
        xs, adds, y = sets[st['i']]
        _hip.check(L.fgnn_linear_multi_forward(arr(xs), ks, arr(Ws), arr(adds), P(y), R, C, _hip.stream_ptr()))
    t = graph_time(run, 24)
    print('R=%d K=%s -> %3d (+%d addends): %6.1f us (%.2f TB/s, %.0f TFLOP/s) %s' % (R, K, C, nadd, t, per / 1e6 / t, 2.0 * R * sum(K) * C / t / 1e6,
                                                                                 L.fgnn_last_kernel().decode()))
