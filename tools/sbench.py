#!/usr/bin/env python3
"""Device time of the streaming kernels around the message operator (BatchNorm apply / backward, n-way sum) on LDPC-sized
activations, replayed from a hipGraph (no launch gaps), against torch's copy of the same bytes.
  python tools/sbench.py            env FGNN_BN_APPLY_GRID / FGNN_BN_GRID select the grids"""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'factor-graph-neural-network_amd'))
import torch
from fgnn_amd import _hip, ops
dev = torch.device('cuda:0')
L = _hip.lib()


def graph_time(run, iters=20):
    for _ in range(3):
        run()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        run()
        with torch.cuda.graph(g, stream=side):
            for _ in range(iters):
                run()
    torch.cuda.current_stream().wait_stream(side)
    g.replay()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    g.replay()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / iters * 1e3


for dt in (torch.bfloat16,):
    for C, N in [(64, 96), (64, 48), (128, 96), (128, 48)]:
        R = 4096 * N
        x = torch.randn(R, C, device=dev).to(dt)
        gy = torch.randn(R, C, device=dev).to(dt)
        y = torch.empty_like(x)
        scale, shift = torch.rand(C, device=dev) + 0.5, torch.randn(C, device=dev)
        mean, invstd = torch.randn(C, device=dev) * 0.1, torch.rand(C, device=dev) + 0.5
        gamma, beta = torch.rand(C, device=dev) + 0.5, torch.randn(C, device=dev)
        dsum = torch.zeros(2, C, device=dev)
        gw, gb = torch.zeros(C, device=dev), torch.zeros(C, device=dev)
        ws = torch.empty(int(L.fgnn_bn_workspace_bytes(R, C)) // 4, device=dev)
        mb = x.numel() * x.element_size() / 1e6
        code = _hip.dtype_code(x)
        t_copy = graph_time(lambda: y.copy_(x))
        t_app = graph_time(lambda: _hip.check(L.fgnn_bn_apply(_hip._ptr(x), _hip._ptr(y), R, C, code, _hip._ptr(scale), _hip._ptr(shift),
                                                              0.01, None, None, None, None, _hip.stream_ptr())))
        t_app1 = graph_time(lambda: _hip.check(L.fgnn_bn_apply(_hip._ptr(x), _hip._ptr(y), R, C, code, _hip._ptr(scale), _hip._ptr(shift),
                                                               0.01, _hip._ptr(gy), None, None, None, _hip.stream_ptr())))
        t_bwd = graph_time(lambda: _hip.check(L.fgnn_bn_backward(_hip._ptr(x), _hip._ptr(gy), _hip._ptr(y), R, C, code, _hip._ptr(mean),
                                                                 _hip._ptr(invstd), _hip._ptr(gamma), _hip._ptr(beta), 0.01,
                                                                 _hip._ptr(gw), _hip._ptr(gb), _hip._ptr(ws), ws.numel() * 4,
                                                                 _hip._ptr(ops._fold_scratch(dev)), _hip.stream_ptr())))
        print('bf16 R=%d C=%3d (%.0f MB): copy %5.1f us (%.2f TB/s) | bn_apply %5.1f us (%.2f TB/s) | +1 addend %5.1f us (%.2f TB/s) | '
              'bn_backward (reduce+final+apply) %5.1f us (%.2f TB/s of 5T)'
              % (R, C, mb, t_copy, 2 * mb / t_copy, t_app, 2 * mb / t_app, t_app1, 3 * mb / t_app1, t_bwd, 5 * mb / t_bwd))

# the same kernels on tensors that are NOT cache-resident: 24 tensor pairs (2.4 GB at 50 MB each) cycled inside the graph,
# far beyond the 256 MB infinity cache — the regime of a training step
print('cold (24 rotating tensor pairs):')
for C, N in [(64, 96), (128, 96)]:
    R = 4096 * N
    K = 24
    xs = [torch.randn(R, C, device=dev).bfloat16() for _ in range(K)]
    ys = [torch.empty_like(xs[0]) for _ in range(K)]
    scale, shift = torch.rand(C, device=dev) + 0.5, torch.randn(C, device=dev)
    mean, invstd = torch.randn(C, device=dev) * 0.1, torch.rand(C, device=dev) + 0.5
    gamma, beta = torch.rand(C, device=dev) + 0.5, torch.randn(C, device=dev)
    gw, gb = torch.zeros(C, device=dev), torch.zeros(C, device=dev)
    ws = torch.empty(int(L.fgnn_bn_workspace_bytes(R, C)) // 4, device=dev)
    mb = xs[0].numel() * 2 / 1e6
    code = _hip.dtype_code(xs[0])
    state = {'i': 0}

    def nxt():
        state['i'] = (state['i'] + 1) % K
        return state['i']

    def f_copy():
        i = nxt(); ys[i].copy_(xs[i])

    def f_app():
        i = nxt()
        _hip.check(L.fgnn_bn_apply(_hip._ptr(xs[i]), _hip._ptr(ys[i]), R, C, code, _hip._ptr(scale), _hip._ptr(shift), 0.01, None, None,
                                   None, None, _hip.stream_ptr()))

    def f_bwd():
        i = nxt()
        _hip.check(L.fgnn_bn_backward(_hip._ptr(xs[i]), _hip._ptr(xs[(i + 7) % K]), _hip._ptr(ys[i]), R, C, code, _hip._ptr(mean), _hip._ptr(invstd),
                                      _hip._ptr(gamma), _hip._ptr(beta), 0.01, _hip._ptr(gw), _hip._ptr(gb), _hip._ptr(ws), ws.numel() * 4,
                                      _hip._ptr(ops._fold_scratch(dev)), _hip.stream_ptr()))

    t_copy, t_app, t_bwd = graph_time(f_copy, 48), graph_time(f_app, 48), graph_time(f_bwd, 48)
    print('bf16 R=%d C=%3d (%.0f MB): copy %5.1f us (%.2f TB/s) | bn_apply %5.1f us (%.2f TB/s) | bn_backward %5.1f us (%.2f TB/s of 5T)'
          % (R, C, mb, t_copy, 2 * mb / t_copy, t_app, 2 * mb / t_app, t_bwd, 5 * mb / t_bwd))

# node-wise maps (1x1 convolutions) and their weight gradients, cold: the other two streaming families of the step
print('node-wise maps, cold (24 rotating tensor pairs):')
for Cin, Cout, N in [(64, 64, 96), (64, 64, 48), (128, 64, 96), (64, 128, 96), (256, 256, 48), (256, 128, 48)]:
    R = 4096 * N
    K = 24
    xs = [torch.randn(R, Cin, device=dev).bfloat16() for _ in range(K)]
    ys = [torch.empty(R, Cout, device=dev, dtype=torch.bfloat16) for _ in range(K)]
    W = torch.randn(Cout, Cin, device=dev) * 0.1
    bias = torch.randn(Cout, device=dev)
    npart = int(L.fgnn_linear_forward_partials(R, Cin, Cout))
    parts = torch.empty(max(npart, 1) * 2 * Cout, device=dev)
    stats4 = torch.empty(4, Cout, device=dev)         # the BatchNorm behind the map, finalised by the map's last workgroup
    fin = _hip.bn_final(stats4, torch.ones(Cout, device=dev), torch.zeros(Cout, device=dev), torch.zeros(Cout, device=dev),
                        torch.ones(Cout, device=dev), torch.zeros((), device=dev, dtype=torch.int64), 0.1, 1e-5, R)
    gW, gb = torch.zeros(Cout, Cin, device=dev), torch.zeros(Cout, device=dev)
    wsb = int(L.fgnn_linear_wgrad_workspace_bytes(R, Cin, Cout))
    ws = torch.empty(max(wsb, 4) // 4, device=dev)
    state = {'i': 0}

    def nxt():
        state['i'] = (state['i'] + 1) % K
        return state['i']

    def f_fwd(stats):
        i = nxt()
        _hip.check(L.fgnn_linear_forward(_hip._ptr(xs[i]), _hip._ptr(W), _hip._ptr(bias), _hip._ptr(ys[i]), R, Cin, Cout,
                                         _hip._ptr(parts) if stats else None, fin if stats else None,
                                         _hip._ptr(ops._fold_scratch(dev)) if stats else None, 0, _hip.stream_ptr()))

    def f_wg():
        i = nxt()
        _hip.check(L.fgnn_linear_wgrad(_hip._ptr(xs[i]), _hip._ptr(ys[(i + 5) % K]), R, Cin, Cout, 1, _hip._ptr(gW), _hip._ptr(gb),
                                       _hip._ptr(ws), wsb, _hip.stream_ptr()))

    mb_in, mb_out = R * Cin * 2 / 1e6, R * Cout * 2 / 1e6
    t0, t1, t2 = graph_time(lambda: f_fwd(False), 48), graph_time(lambda: f_fwd(True), 48), graph_time(f_wg, 48)
    print('bf16 R=%d %3d->%3d: forward %5.1f us (%.2f TB/s) | with statistics epilogue %5.1f us (%.2f TB/s) | weight gradient %5.1f us '
          '(%.2f TB/s)' % (R, Cin, Cout, t0, (mb_in + mb_out) / t0, t1, (mb_in + mb_out) / t1, t2, (mb_in + mb_out) / t2))

# InstanceNorm (+ReLU) over the node axis, cold
print('InstanceNorm, cold:')
for C, N in [(64, 96), (128, 96), (256, 96), (64, 48), (256, 48)]:
    Bn = 4096
    K = max(3, int(2.4e9 // (Bn * N * C * 2 * 2)))
    xs = [torch.randn(Bn, N, C, device=dev).bfloat16() for _ in range(K)]
    ys = [torch.empty_like(xs[0]) for _ in range(K)]
    gs = [torch.randn(Bn, N, C, device=dev).bfloat16() for _ in range(K)]
    state = {'i': 0}
    def nxt():
        state['i'] = (state['i'] + 1) % K
        return state['i']
    def f_fwd():
        i = nxt()
        _hip.check(L.fgnn_instnorm_forward(_hip._ptr(xs[i]), _hip._ptr(ys[i]), Bn, N, C, 1, 1, _hip.stream_ptr()))
    def f_bwd():
        i = nxt()
        _hip.check(L.fgnn_instnorm_backward(_hip._ptr(xs[i]), _hip._ptr(gs[i]), _hip._ptr(ys[i]), Bn, N, C, 1, 1, _hip.stream_ptr()))
    mb = Bn * N * C * 2 / 1e6
    t0, t1 = graph_time(f_fwd, 48), graph_time(f_bwd, 48)
    print('bf16 [4096, %d, %3d] (%.0f MB): forward %6.1f us (%.2f TB/s of 2T) | backward %6.1f us (%.2f TB/s of 3T)' % (N, C, mb, t0, 2 * mb / t0, t1, 3 * mb / t1))
