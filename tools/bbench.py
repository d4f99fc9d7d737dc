#!/usr/bin/env python3
"""One-kernel inference blocks (csrc/mpconv_block_fwd.hip) at the LDPC model's shapes, cold inputs, hipGraph replay:
us per launch without / with an addend, against the bytes the block has to move.   python tools/bbench.py [batch]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'factor-graph-neural-network_amd'))
import torch
import fgnn_amd
from fgnn_amd.mpnn import mp_conv_residual, mp_conv_type
from fgnn_amd.tables import LdpcGraph

B = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
dev = torch.device('cuda:0')
g = LdpcGraph()
f2v = torch.from_numpy(g.var_to_factors).to(dev)[None].expand(B, -1, -1)     # [B, 96, 3]
v2f = torch.from_numpy(g.factor_to_vars).to(dev)[None].expand(B, -1, -1)     # [B, 48, 6]
K = 6


def graph_time(run, iters=24):
    for _ in range(3):
        run()
    torch.cuda.synchronize()
    gr = torch.cuda.CUDAGraph()
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        run()
        with torch.cuda.graph(gr, stream=s):
            for _ in range(iters):
                run()
    torch.cuda.current_stream().wait_stream(s)
    gr.replay()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    gr.replay()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / iters * 1e3


print('batch %d' % B)
for name, nin, nout, idx, N, M, k, net in [
        ('V->F  64 ->  64', 64, 64, v2f, 96, 48, 6, 4), ('F->V  64 ->  64', 64, 64, f2v, 48, 96, 3, 4),
        ('V->F 128 -> 256', 128, 256, v2f, 96, 48, 6, 4), ('F->V 128 -> 256', 128, 256, f2v, 48, 96, 3, 4),
        ('V->F 256 -> 256', 256, 256, v2f, 96, 48, 6, 4), ('F->V 256 -> 256', 256, 256, f2v, 48, 96, 3, 4),
        ('V->F 256 -> 128', 256, 128, v2f, 96, 48, 6, 4), ('F->V 256 -> 128', 256, 128, f2v, 48, 96, 3, 4),
        ('fan-in  256 -> 256', 256, 256, None, 96, 1, 96, 1), ('fan-out 256 -> 256', 256, 256, None, 1, 96, 1, 1)]:
    torch.manual_seed(0)
    blk = mp_conv_residual(nin, 64, net, extension=mp_conv_type.NO_EXTENSION, with_residual=False, aggregator='max', nout=nout).to(dev).eval()
    xs = [torch.randn(B, N, 1, nin, device=dev).bfloat16().permute(0, 3, 1, 2) for _ in range(K)]
    ads = [torch.randn(B, M, 1, nout, device=dev).bfloat16().permute(0, 3, 1, 2) for _ in range(K)]
    if net == 4:
        et = torch.randn(B, M, k, 4, device=dev).bfloat16().permute(0, 3, 1, 2)
        ix = idx
    elif M == 1:
        et = torch.ones(1, 1, 1, 96, device=dev).bfloat16().expand(B, -1, -1, -1)
        ix = torch.arange(96, device=dev).reshape(1, 1, 96).expand(B, -1, -1)
    else:
        et = torch.ones(1, 1, 96, 1, device=dev).bfloat16().expand(B, -1, -1, -1)
        ix = torch.zeros(1, 96, 1, dtype=torch.int64, device=dev).expand(B, -1, -1)
    turn = [0]

    def run(with_addend):
        turn[0] = (turn[0] + 1) % K
        with torch.no_grad():
            y = blk._fused_eval(xs[turn[0]], ix, et, ads[turn[0]] if with_addend else None)
        assert y is not None
    t0, t1 = graph_time(lambda: run(False)), graph_time(lambda: run(True))
    mb = B * (N * nin + M * nout) * 2 / 1e6 + (B * M * k * 4 * 2 / 1e6 if net == 4 else 0)
    mba = mb + B * M * nout * 2 / 1e6
    print('%-20s plain %6.1f us (%5.1f MB, %4.2f TB/s) | + addend %6.1f us (%5.1f MB, %4.2f TB/s)' % (name, t0, mb, mb / t0, t1, mba, mba / t1))
