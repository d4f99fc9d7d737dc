#!/usr/bin/env python3
"""Per-shape micro-benchmark of the fused message operator (forward and backward) on one GPU.
Prints, for each LDPC operator shape at batch B: us per launch, algorithmic GB/s (SURVEY §8d bytes)
and f32 TFLOP/s of the projection.  Used to tune kernels; bench.py remains the contract benchmark."""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'factor-graph-neural-network_amd'))
import torch  # noqa: E402
from fgnn_amd import _hip, ops  # noqa: E402

SHAPES = [  # (name, nin, nou, net, N, M, k[, extension, aggregator])
    ('parity V->F 64->64', 64, 64, 4, 96, 48, 6),
    ('parity F->V 64->64', 64, 64, 4, 48, 96, 3),
    ('parity V->F 64->128', 64, 128, 4, 96, 48, 6),
    ('parity F->V 64->128', 64, 128, 4, 48, 96, 3),
    ('parity V->F 128->64', 128, 64, 4, 96, 48, 6),
    ('parity F->V 128->64', 128, 64, 4, 48, 96, 3),
    ('hyper  V->F 64->64', 64, 64, 1, 96, 1, 96),
    ('hyper  F->V 64->64', 64, 64, 1, 1, 96, 1),
    ('hyper  V->F 64->128', 64, 128, 1, 96, 1, 96),
    ('hyper  F->V 64->128', 64, 128, 1, 1, 96, 1),
    ('hyper  V->F 128->64', 128, 64, 1, 96, 1, 96),
    ('hyper  F->V 128->64', 128, 64, 1, 1, 96, 1),
    # BASELINE configs 2 / 5: factor_mpnn on 30-node synthetic PGMs (30 variables + 30 factors per graph, 16 edge types,
    # ORIG_WITH_DIFF extension; pairwise factors k = 2, degree-9 high-order factors k = 9); run with --dtype f32 --batch 256 / 1024
    ('syn pw  64->64 max', 64, 64, 16, 60, 60, 2, 2, 0),
    ('syn pw  64->128 lse', 64, 128, 16, 60, 60, 2, 2, 1),
    ('syn hop 64->64 max', 64, 64, 16, 60, 60, 9, 2, 0),
    ('syn hop 128->128 max', 128, 128, 16, 60, 60, 9, 2, 0),
]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--batch', type=int, default=4096)
    ap.add_argument('--iters', type=int, default=20)
    ap.add_argument('--dtype', default='f32')
    ap.add_argument('--layout', default='cl', choices=['cl', 'nchw'])
    ap.add_argument('--bwd', action='store_true')
    ap.add_argument('--regular', action='store_true', help='regular random graphs (constant in-degree) for the parity shapes')
    ap.add_argument('--only', default='')
    ap.add_argument('--random-hyper', action='store_true', help='hyper V->F shapes with a random neighbour table instead of the identity list')
    ap.add_argument('--syn', action='store_true', help='the synthetic-PGM shapes instead of the LDPC ones')
    ap.add_argument('--shared-et', action='store_true', help='edge types shared by the batch ([1, net, M, k] expanded), as in the synthetic-PGM scripts')
    ap.add_argument('--argmax', action='store_true', help='forward that also stores the argmax (what training needs)')
    ap.add_argument('--stats', action='store_true', help='forward with the BatchNorm statistics epilogue (training form)')
    ap.add_argument('--no-etgrad', action='store_true', help='backward without the edge-weight gradient (getype = NULL)')
    ap.add_argument('--etgrad', action='store_true', help='hyper shapes: also ask for the edge-weight gradient')
    ap.add_argument('--cold', type=int, default=1, help='rotate over this many copies of the activations (> 256 MB in total: every launch '
                    'reads from HBM, as inside a training step, instead of from the infinity cache)')
    a = ap.parse_args()
    dev = torch.device('cuda:0')
    dt = torch.float32 if a.dtype == 'f32' else torch.bfloat16
    for shape in SHAPES:
        name, nin, nou, net, N, M, k = shape[:7]
        ext, agg = (shape[7], shape[8]) if len(shape) > 7 else (0, _hip.AGG_MAX)
        if a.only and a.only not in name:
            continue
        if not a.only and (name.startswith('syn') != a.syn):
            continue
        g = torch.Generator(device='cpu').manual_seed(0)
        B = a.batch
        x = torch.randn(B, nin, N, 1, generator=g).to(dev, dt)
        if a.layout == 'cl':
            x = x.permute(0, 2, 3, 1).contiguous().permute(0, 3, 1, 2)
        if M == 1 and k == N and not a.random_hyper:
            idx = torch.arange(N).reshape(1, 1, N).to(dev).expand(B, -1, -1)       # the LDPC hyper-factor's table (train_ldpc.py:40-46): every variable, in order
        elif a.regular and (M * k) % N == 0:
            # a random REGULAR bipartite graph (every source node appears M k / N times), like the 96.3.963 code
            slots = torch.arange(N).repeat_interleave(M * k // N)[torch.randperm(M * k, generator=g)]
            idx = slots.reshape(1, M, k).to(dev).expand(B, -1, -1)
        else:
            idx = torch.randint(0, N, (1, M, k), generator=g).to(dev).expand(B, -1, -1)
        if net == 1:
            et = torch.ones(1, 1, M, k, device=dev, dtype=dt).expand(B, -1, -1, -1)
        elif a.shared_et:
            et = torch.randn(1, net, M, k, generator=g).to(dev, dt).expand(B, -1, -1, -1)
        else:
            et = torch.randn(B, M, k, net, generator=g).to(dev, dt).permute(0, 3, 1, 2)
        W = (torch.randn(nin * (1 if ext == 0 else 2), nou * net, generator=g) * 0.1).to(dev)
        bias = torch.randn(nou, generator=g).to(dev)
        nbytes = ops.algorithmic_bytes(x, idx, et, nou, net, ext, agg)
        flops = 2.0 * B * N * nin * (1 if ext == 0 else 2) * nou * net + 2.0 * B * M * k * nou * net * (1 if ext == 0 else 2)

        K = max(1, a.cold)
        xs = [x] + [x.clone(memory_format=torch.preserve_format) for _ in range(K - 1)]
        ets = [et] + [(et.clone(memory_format=torch.preserve_format) if et.stride(0) != 0 else et) for _ in range(K - 1)]
        turn = [0]

        spec = None
        if a.stats:     # the BatchNorm behind the operator, finalised by the operator's own launch
            spec = (torch.ones(nou, device=dev), torch.zeros(nou, device=dev), torch.zeros(nou, device=dev), torch.ones(nou, device=dev),
                    torch.zeros((), device=dev, dtype=torch.int64), 0.1, 1e-5)

        def fwd():
            turn[0] = (turn[0] + 1) % K
            return ops.mpconv_forward_raw(xs[turn[0]], idx, ets[turn[0]], W, bias, nou, net, ext, agg,
                                          want_argmax=a.bwd or a.stats or a.argmax, bn=spec)

        if not a.bwd:
            run = fwd
        else:
            import ctypes
            L = _hip.lib()
            y, amax = ops.mpconv_forward_raw(x, idx, et, W, bias, nou, net, ext, agg, want_argmax=True)
            gz = torch.randn_like(y)
            gx = torch.empty_like(x)
            get = None if ((net == 1 and not a.etgrad) or a.no_etgrad) else torch.empty((B, net, M, k), device=dev, dtype=dt)
            gw = torch.zeros_like(W)
            gb = torch.zeros(nou, device=dev)
            dsc = _hip.make_desc(x, idx, et, nou, net, ext, agg, False, gz)
            dsc.reserved = ops.max_in_degree(idx, N)
            if get is not None and et.stride(0) == 0 and L.fgnn_mpconv_backward_reduces_getype(ctypes.byref(dsc)):
                dsc.reserved |= _hip.DESC_GETYPE_REDUCED              # shared edge weights: batch-summed gradient
                get = torch.empty((1, net, M, k), device=dev, dtype=torch.float32)
            nbytes = (x.element_size() * (x.numel() + gz.numel()) + et.element_size() * net * M * k *
                      (1 if et.stride(0) == 0 else B) + 8 * M * k + B * nou * M + x.element_size() * (gx.numel() + (get.numel() if get is not None else 0))
                      + 8 * W.numel())
            flops *= 3.0
            wsb = ops._workspace(dev, int(L.fgnn_mpconv_backward_workspace_bytes(ctypes.byref(dsc))))
            tables = ops.backward_tables(idx, dsc)          # per-graph tables (ops.BACKWARD_TABLES off — the default: every launch builds its own)
            cp = lambda t: None if t is None else t.clone(memory_format=torch.preserve_format)
            sets = [(x, et, gz, amax, gx, get)] + [(xs[i], ets[i], cp(gz), cp(amax), cp(gx), cp(get)) for i in range(1, K)]

            def run():
                turn[0] = (turn[0] + 1) % K
                x_, et_, gz_, am_, gx_, get_ = sets[turn[0]]
                _hip.check(L.fgnn_mpconv_backward_with_tables(
                    ctypes.byref(dsc), _hip._ptr(x_), _hip._ptr(idx), _hip._ptr(et_), _hip._ptr(W),
                    _hip._ptr(gz_), None, _hip._ptr(am_), _hip._ptr(gx_), _hip._ptr(get_), _hip._ptr(gw),
                    _hip._ptr(gb), _hip._ptr(wsb), wsb.numel() * 4, _hip._ptr(tables), _hip.stream_ptr()))
        for _ in range(3):
            run()
        torch.cuda.synchronize()
        # host-side cost of one call (Python + ctypes), GPU idle work excluded
        import time
        t0 = time.perf_counter()
        for _ in range(a.iters):
            run()
        host_us = (time.perf_counter() - t0) / a.iters * 1e6
        torch.cuda.synchronize()
        # device time: replay a captured graph of `iters` back-to-back launches (no host gaps)
        graph = torch.cuda.CUDAGraph()
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            run()
            with torch.cuda.graph(graph, stream=side):
                for _ in range(a.iters):
                    run()
        torch.cuda.current_stream().wait_stream(side)
        graph.replay()
        torch.cuda.synchronize()
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        graph.replay()
        e.record()
        torch.cuda.synchronize()
        us = s.elapsed_time(e) / a.iters * 1e3
        print('%-22s %s %-4s %s  %8.1f us (host %6.1f us/call)  %7.1f GB/s (%.1f%% of 8 TB/s)  %6.1f TFLOP/s  '
              '[%d B, %.2f MB/call] %s'
              % (name, a.dtype, a.layout, 'bwd' if a.bwd else 'fwd', us, host_us, nbytes / us / 1e3,
                 nbytes / us / 1e3 / 80.0, flops / us / 1e6, B, nbytes / 1e6,
                 _hip.lib().fgnn_last_kernel().decode()))


if __name__ == '__main__':
    main()
