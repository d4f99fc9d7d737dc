#!/usr/bin/env python3
"""Node-wise map forward: csrc/linear_fwd_b16.hip against torch (hipBLASLt) on cold tensors, R = 4096 x 96 rows (and x 48)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'factor-graph-neural-network_amd'))
import torch
from fgnn_amd import _hip
dev = torch.device('cuda:0')
L = _hip.lib()
sys.path.insert(0, os.path.join(ROOT, 'tools'))


def graph_time(run, iters=48):
    for _ in range(3): run()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph(); side = torch.cuda.Stream(); side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        run()
        with torch.cuda.graph(g, stream=side):
            for _ in range(iters): run()
    torch.cuda.current_stream().wait_stream(side)
    g.replay(); torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record(); g.replay(); e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / iters * 1e3


for N in ((96, 48) if __name__ == '__main__' else ()):
    for cin, cout in [(64, 64), (64, 128), (128, 64), (128, 128), (64, 256), (256, 64), (128, 256), (256, 128), (256, 256)]:
        R = 4096 * N
        K = max(2, int(2.4e9 // (R * (cin + cout) * 2)))
        xs = [torch.randn(R, cin, device=dev).bfloat16() for _ in range(K)]
        ys = [torch.empty(R, cout, device=dev, dtype=torch.bfloat16) for _ in range(K)]
        W = torch.randn(cout, cin, device=dev) * 0.1
        Wb = W.bfloat16()
        bias = torch.randn(cout, device=dev)
        bb = bias.bfloat16()
        st = {'i': 0}
        def nxt():
            st['i'] = (st['i'] + 1) % K
            return st['i']
        def f_hip():
            i = nxt()
            _hip.check(L.fgnn_linear_forward(_hip._ptr(xs[i]), _hip._ptr(W), _hip._ptr(bias), _hip._ptr(ys[i]), R, cin, cout, None, None, None, 0, _hip.stream_ptr()))
        def f_t():
            i = nxt()
            torch.addmm(bb, xs[i], Wb.t(), out=ys[i])
        t1, t2 = graph_time(f_hip), graph_time(f_t)
        mb = R * (cin + cout) * 2 / 1e6
        print('R=%d %3d->%3d: linear_fwd_b16 %6.1f us (%.2f TB/s) | torch addmm %6.1f us (%.2f TB/s)' % (R, cin, cout, t1, mb / t1, t2, mb / t2))
