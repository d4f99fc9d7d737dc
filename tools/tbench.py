"""Stand-alone device time of the fused block-tail kernels (csrc/block_tail.hip), per mode and shape, rotating over NBUF sets of
tensors so that the infinity cache does not flatter the numbers.   python tools/tbench.py [reps]"""
import os
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'factor-graph-neural-network_amd'))
import torch
from fgnn_amd import _hip

dev = torch.device('cuda:0')
L = _hip.lib()
P = _hip._ptr
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 20
NBUF = 4
SHAPES = [(196608, 64), (393216, 64), (393216, 128), (196608, 256), (393216, 256), (4096, 256)]
if os.environ.get('TB_SHAPE'):          # e.g. TB_SHAPE=393216,256 (profiling one shape)
    SHAPES = [tuple(int(v) for v in os.environ['TB_SHAPE'].split(','))]
MODES = os.environ.get('TB_MODES', 'stats,apply,backward').split(',')


def timeit(fn):
    for i in range(3):
        fn(i % NBUF)
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for i in range(reps):
        fn(i % NBUF)
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / reps * 1e3


for R, Cout in SHAPES:
    g = torch.Generator().manual_seed(1)
    e = [torch.randn(R, 64, generator=g).bfloat16().to(dev) for _ in range(NBUF)]
    gout = [torch.randn(R, Cout, device=dev).bfloat16() for _ in range(NBUF)]
    adds = [[torch.randn(R, Cout, device=dev).bfloat16() for _ in range(3)] for _ in range(NBUF)]
    out = [torch.empty(R, Cout, device=dev, dtype=torch.bfloat16) for _ in range(NBUF)]
    a2 = [torch.empty(R, 64, device=dev, dtype=torch.bfloat16) for _ in range(NBUF)]
    s2, t2 = torch.rand(64, device=dev) + 0.5, torch.randn(64, device=dev) * 0.3
    W2, b2 = torch.randn(Cout, 64, device=dev) * 0.2, torch.randn(Cout, device=dev)
    st3 = torch.rand(4, Cout, device=dev) + 0.5
    gam = torch.rand(Cout, device=dev) + 0.5
    ws = torch.zeros(int(L.fgnn_bn_workspace_bytes(R, Cout)) // 4, device=dev)
    np2 = int(L.fgnn_block_tail_backward_partials(R, Cout))
    part2 = torch.zeros(128, device=dev)              # BatchNorm2's backward sums [2][64], finalised by the grad kernel
    wsb = torch.zeros(2048 * Cout + 2 * Cout + 1024 * 128, device=dev)
    gw, gb = torch.zeros(Cout, device=dev), torch.zeros(Cout, device=dev)
    st = _hip.stream_ptr()
    from fgnn_amd import ops
    fold = ops._fold_scratch(dev)
    stf = torch.empty(4, Cout, device=dev)
    fin = _hip.bn_final(stf, gam, gb, torch.zeros(Cout, device=dev), torch.ones(Cout, device=dev), None, 0.1, 1e-5, R)
    t_stats = 0.0 if 'stats' not in MODES else timeit(lambda i: _hip.check(L.fgnn_block_tail_stats(P(e[i]), P(s2), P(t2), 0.0, P(W2), P(b2), R, Cout, P(ws), fin, P(fold), st)))
    res = ['stats %6.1f us (%4.2f TB/s)' % (t_stats, 2 * R * 64 / max(t_stats, 1e-9) / 1e6)]
    for nadd in ((0, 3) if 'apply' in MODES else ()):
        ap = [P(a) for a in adds[0][:nadd]] + [None] * (3 - nadd)
        t = timeit(lambda i: _hip.check(L.fgnn_block_tail_apply(P(e[i]), P(s2), P(t2), 0.0, P(W2), P(b2), P(st3[2]), P(st3[3]), 0.01,
                                                                *([P(a) for a in adds[i][:nadd]] + [None] * (3 - nadd)), None, P(out[i]), P(a2[i]), R, Cout, st)))
        res.append('apply+%d %6.1f us (%4.2f TB/s)' % (nadd, t, 2 * R * (64 + (1 + nadd) * Cout) / t / 1e6))
    t = 1e-9 if 'backward' not in MODES else timeit(lambda i: _hip.check(L.fgnn_block_tail_backward(P(e[i]), P(s2), P(t2), 0.0, P(W2), P(b2), P(st3[0]), P(st3[1]), P(gam), P(st3[2]),
                                                               P(st3[3]), 0.01, P(gout[i]), P(out[i]), P(a2[i]), P(gw), P(gb), P(s2), P(t2), None, None, P(part2), R,
                                                               Cout, P(wsb), wsb.numel() * 4, P(fold), st)))
    res.append('backward %6.1f us (%4.2f TB/s)' % (t, 2 * R * (3 * 64 + 3 * Cout) / t / 1e6))
    print('R %6d Cout %3d: ' % (R, Cout) + ' | '.join(res), flush=True)
