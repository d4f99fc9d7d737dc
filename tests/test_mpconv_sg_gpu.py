"""GPU parity of the second-generation parity-check kernels (csrc/mpconv_fwd_sg.hip, csrc/mpconv_bwd_sg.hip): the
calls of the LDPC model with a batch-shared graph, bf16 channel-fastest storage, 4 edge types, max aggregation.
Forward against the f32 ORACLE on the same bf16-rounded inputs (2^-6 of the output range: P and the output are rounded
to bf16), the argmax under the near-tie rule, the statistics epilogue against a direct reduction of the stored output;
backward against torch autograd through the forward's own routing."""
import pytest
import torch

import fgnn_oracle as O
import helpers as H

pytestmark = pytest.mark.gpu

# (nin, nou, N, M, k): the six parity-check call shapes of LDPCModel, then ragged / degenerate ones
SHAPES = [(64, 64, 96, 48, 6), (64, 64, 48, 96, 3), (64, 128, 96, 48, 6), (64, 128, 48, 96, 3), (128, 64, 96, 48, 6),
          (128, 64, 48, 96, 3), (64, 64, 91, 47, 6), (64, 64, 37, 95, 3), (64, 64, 96, 1, 6), (64, 64, 40, 20, 3),
          (64, 128, 33, 7, 6)]
IDS = ['x'.join(map(str, s)) for s in SHAPES]


def _problem(shape, B, dev, seed=0):
    nin, nou, N, M, k = shape
    g = torch.Generator().manual_seed(seed + N + 7 * M + nou)
    x = torch.randn(B, N, 1, nin, generator=g).bfloat16()                   # channel-fastest in memory
    idx = torch.randint(0, N, (1, M, k), generator=g)
    et = torch.randn(B, M, k, 4, generator=g).bfloat16()                      # edge-type fastest in memory
    if k > 2:                                                                  # same neighbour, same edge weights: exact ties
        idx[0, ::5, k - 1] = idx[0, ::5, 0]
        et[:, ::5, k - 1, :] = et[:, ::5, 0, :]
    W = (torch.randn(nin, nou * 4, generator=g) * 0.1).bfloat16().float()
    bias = torch.randn(nou, generator=g)
    return x, idx, et, W, bias, g


def _dev_views(x, idx, et, dev):
    B = x.shape[0]
    return (x.to(dev).permute(0, 3, 1, 2), idx.to(dev).expand(B, -1, -1), et.to(dev).permute(0, 3, 1, 2))


@pytest.mark.parametrize('shape', SHAPES, ids=IDS)
@pytest.mark.parametrize('mode', ['train', 'affine_relu', 'relu_only', 'affine_argmax'])
def test_sg_forward_vs_oracle(shape, mode, dev):
    from fgnn_amd import _hip, ops
    nin, nou, N, M, k = shape
    B = 37                                                                     # not a multiple of anything
    x, idx, et, W, bias, g = _problem(shape, B, dev)
    scale, shift = torch.rand(nou, generator=g) + 0.5, torch.randn(nou, generator=g)
    pre = O.mp_conv({'filters': W, 'bias': bias}, '', x.permute(0, 3, 1, 2).float(), idx.expand(B, -1, -1).contiguous(),
                    et.permute(0, 3, 1, 2).float(), nou=nou, net=4, extension=0, aggregator='max', relu=False)
    kw = dict(want_argmax=mode in ('train', 'affine_argmax'))
    ref = pre
    if mode in ('affine_relu', 'affine_argmax'):
        ref = ref * scale[None, :, None, None] + shift[None, :, None, None]
        kw.update(post_scale=scale.to(dev), post_shift=shift.to(dev))
    if mode in ('affine_relu', 'relu_only'):
        ref = torch.relu(ref)
        kw.update(relu=True)
    xd, idxd, etd = _dev_views(x, idx, et, dev)
    y, am = ops.mpconv_forward_raw(xd, idxd, etd, W.to(dev), bias.to(dev), nou, 4, 0, _hip.AGG_MAX, **kw)
    # 64 -> 128 runs as two launches of the 64 -> 64 kernel over the halves of the output channels
    kern = _hip.lib().fgnn_last_kernel().decode()
    # the third-generation kernel (mpconv_fwd_ws.hip) takes the 64-channel-input calls it can tile, mpconv_fwd_sg.hip the rest
    assert ('mpconv_fwd_ws' in kern or 'mpconv_fwd_sg' in kern) and kern.endswith(' x2') == (nou == 128), kern
    if (M * k) % 2 == 0:
        assert 'mpconv_fwd_ws' in kern, kern
    assert y.dtype == torch.bfloat16 and y.shape == ref.shape and y.stride(1) == 1
    err = float((y.float().cpu() - ref).abs().max() / ref.abs().max())
    assert err <= 2.0 ** -6, err
    if am is not None:
        # near-tie rule: the named neighbour's f32 message is the maximum up to the bf16 rounding of P
        e = O.mp_conv({'filters': W}, '', x.permute(0, 3, 1, 2).float(), idx.expand(B, -1, -1).contiguous(),
                      et.permute(0, 3, 1, 2).float(), nou=nou, net=4, extension=0, aggregator=None, relu=False)
        a = am.cpu().long()
        assert int(a.max()) < k
        gap = float((e.max(dim=3, keepdim=True)[0] - e.gather(3, a)).max())
        assert gap <= 2.0 ** -6 * float(e.abs().max()), gap
        # exact ties (duplicated neighbour in the last slot) resolve to the first occurrence
        dup = a[:, :, ::5, :]
        if k > 2:
            assert int((dup == k - 1).sum()) == 0


@pytest.mark.parametrize('shape', SHAPES[:6], ids=IDS[:6])
def test_sg_forward_matches_first_generation_kernel(shape, dev, monkeypatch):
    """Same inputs through mpconv_fwd_b16.hip (per-sample copy of the table defeats the shared-graph dispatch): both
    round P to bf16, so the outputs agree to one bf16 ulp of the output range and the argmax wherever messages differ."""
    from fgnn_amd import _hip, ops
    nin, nou, N, M, k = shape
    B = 64
    x, idx, et, W, bias, g = _problem(shape, B, dev, seed=3)
    xd, idxd, etd = _dev_views(x, idx, et, dev)
    y1, a1 = ops.mpconv_forward_raw(xd, idxd, etd, W.to(dev), bias.to(dev), nou, 4, 0, _hip.AGG_MAX, want_argmax=True)
    kern = _hip.lib().fgnn_last_kernel().decode()
    assert 'mpconv_fwd_ws' in kern or 'mpconv_fwd_sg' in kern, kern
    monkeypatch.setattr(ops, 'DEDUPE_GRAPHS', False)
    y2, a2 = ops.mpconv_forward_raw(xd, idxd.contiguous(), etd, W.to(dev), bias.to(dev), nou, 4, 0, _hip.AGG_MAX,
                                    want_argmax=True)
    assert 'mpconv_fwd_b16' in _hip.lib().fgnn_last_kernel().decode()
    assert float((y1.float() - y2.float()).abs().max()) <= 2.0 ** -7 * float(y2.float().abs().max())
    assert float((a1 != a2).float().mean()) <= 2e-3                       # near-ties only


@pytest.mark.parametrize('shape', SHAPES[:2] + SHAPES[4:8], ids=IDS[:2] + IDS[4:8])
@pytest.mark.parametrize('B', [5, 300, 1100])
def test_sg_statistics_epilogue(shape, B, dev):
    """Training-mode mp_conv_v2 on a shared graph: the forward's epilogue leaves (sum, sum of squares) of the STORED
    output for the BatchNorm behind it; result, running statistics and gradients equal BatchNorm's own reduction pass."""
    from fgnn_amd import _hip, ops
    from fgnn_amd.mpnn import mp_conv_type, mp_conv_v2
    nin, nou, N, M, k = shape
    x, idx, et, W, bias, g = _problem(shape, B, dev, seed=B)
    xd, idxd, etd = _dev_views(x, idx, et, dev)
    gy = torch.randn(B, M, 1, nou, generator=g).bfloat16().to(dev).permute(0, 3, 1, 2)

    def run(epilogue):
        torch.manual_seed(7)
        m = mp_conv_v2(nin, nou, 4, extension=mp_conv_type.NO_EXTENSION, aggregtor='max').to(dev).train()
        xx = xd.detach().clone().requires_grad_(True)
        ops.STATS_EPILOGUE = epilogue
        ops.TIMER = ops.KernelTimer()
        try:
            y = m(xx, idxd, etd)
            names = set(ops.TIMER.summary())
        finally:
            ops.STATS_EPILOGUE = True
            ops.TIMER = None
        y.backward(gy)
        return (y.detach().float(), m.bn.running_mean.clone(), m.bn.running_var.clone(), xx.grad.float(),
                m.filters.grad.clone(), names)

    a, b = run(True), run(False)
    assert any('mpconv_fwd_sg' in n or 'mpconv_fwd_ws' in n for n in a[5]), a[5]
    assert not any(n.startswith('bn_stats') for n in a[5]), a[5]           # the epilogue is what ran
    assert any(n.startswith('bn_stats') for n in b[5]), b[5]
    assert H.rel_err(a[0], b[0]) <= 2.0 ** -7 and H.rel_err(a[3], b[3]) <= 2.0 ** -6
    assert H.rel_err(a[1], b[1]) <= 1e-5 and H.rel_err(a[2], b[2]) <= 1e-4 and H.rel_err(a[4], b[4]) <= 2e-3


@pytest.mark.parametrize('shape', SHAPES[:2], ids=IDS[:2])
def test_ws_forward_is_bit_reproducible_run_to_run(shape, dev, finaliser_mode):
    """The wave-specialised forward synchronises its producer and consumer waves with LDS counters, not barriers: a protocol error
    shows as rare run-to-run differences, not as a parity failure (round 4: a single cumulative "image consumed" counter let fast
    consumer waves' signals for the next sample stand in for a slow wave's, the image was overwritten under it: ~1 launch in 3 at
    B = 1100 had a few wrong values in the last destinations of one sample).  120 launches on the same inputs, all bit-identical,
    and identical to the barrier-synchronised second-generation kernel is covered by the parity tests above."""
    from fgnn_amd import _hip, ops
    nin, nou, N, M, k = shape
    B = 1100
    x, idx, et, W, bias, g = _problem(shape, B, dev, seed=B)
    xd, idxd, etd = _dev_views(x, idx, et, dev)
    Wd, bd = W.to(dev), bias.to(dev)
    from fgnn_amd.mpnn import pointwise
    for stats in (True, False):
        first = None
        spec = H.bn_spec_for(nou, dev) if stats else None
        for r in range(60):
            y, am = ops.mpconv_forward_raw(xd, idxd, etd, Wd, bd, nou, 4, 0, _hip.AGG_MAX, want_argmax=True, bn=spec)
            st = pointwise.take_pending_stats(y.permute(0, 2, 3, 1).reshape(B * M, nou)) if stats else None
            assert (st is not None) == stats
            if first is None:
                first = (y.clone(), am.clone(), None if st is None else st.clone())
                if stats:
                    # the 256 partial rows folded and the BatchNorm finalised by the same call (a finaliser launch, or the launch's
                    # LAST workgroup: csrc/fgnn_gridfold.h): against an f64 restatement on the stored (bf16) output
                    rows = y.permute(0, 2, 3, 1).reshape(B * M, nou).double()
                    mean, var = rows.mean(0), rows.var(0, unbiased=False)
                    assert H.rel_err(st[0], mean) <= 1e-5 and H.rel_err(st[1], torch.rsqrt(var + 1e-5)) <= 1e-5
                    assert H.rel_err(st[2], spec[0].double() * torch.rsqrt(var + 1e-5)) <= 1e-5
            else:
                assert torch.equal(y, first[0]) and torch.equal(am, first[1]), 'launch %d differs from launch 0' % r
                if stats:            # ... whichever workgroup arrived last: bit-identical statistics
                    assert torch.equal(st, first[2]), 'statistics of launch %d differ from launch 0' % r
        if stats:
            assert int(spec[4]) == 60 and int(ops._fold_scratch(dev)[:65].abs().sum()) == 0


def _regular_table(N, M, k, g):
    """A random bipartite graph in which every source node appears M k / N times (like the 96.3.963 code: 3 / 6)."""
    assert (M * k) % N == 0
    slots = torch.arange(N).repeat_interleave(M * k // N)[torch.randperm(M * k, generator=g)]
    return slots.reshape(1, M, k)


BWD_SHAPES = [(96, 48, 6), (48, 96, 3), (64, 32, 6), (32, 64, 3), (96, 16, 6), (16, 32, 3)]      # (N, M, k), 64 -> 64 channels


@pytest.mark.parametrize('shape', BWD_SHAPES, ids=lambda s: 'x'.join(map(str, s)))
@pytest.mark.parametrize('B', [1, 37, 600])
def test_sg_backward_vs_routed_reference(shape, B, dev):
    """Shared regular graph, 64 -> 64 channels: csrc/mpconv_bwd_sg.hip against torch autograd through the forward's own
    routing (same bf16-rounded x / etype / gz, f32 arithmetic): what remains is the bf16 rounding of P, dP, the routed
    products and the bf16 outputs -> 2^-6 of each gradient's range.  Ragged batches (B = 37: chunks of different length per
    workgroup; B = 600: several samples per workgroup) and the real LDPC degrees (3 in / 6 out, 6 in / 3 out)."""
    from fgnn_amd import _hip, ops
    N, M, k = shape
    nin = nou = 64
    net = 4
    g = torch.Generator().manual_seed(100 + N + B)
    x = torch.randn(B, N, 1, nin, generator=g).bfloat16()
    idx = _regular_table(N, M, k, g)
    et = torch.randn(B, M, k, net, generator=g).bfloat16()
    W = torch.randn(nin, nou * net, generator=g) * 0.1
    bias = torch.randn(nou, generator=g)
    gz = torch.randn(B, M, 1, nou, generator=g).bfloat16()
    xd = x.to(dev).permute(0, 3, 1, 2).requires_grad_(True)
    etd = et.to(dev).permute(0, 3, 1, 2).requires_grad_(True)
    idxd = idx.to(dev).expand(B, -1, -1)
    Wd, bd = W.to(dev).requires_grad_(True), bias.to(dev).requires_grad_(True)
    z = ops.mpconv(xd, idxd, etd, Wd, bd, nou, net, 0, _hip.AGG_MAX)
    _, am = ops.mpconv_forward_raw(xd.detach(), idxd, etd.detach(), W.to(dev), bias.to(dev), nou, net, 0, _hip.AGG_MAX,
                                   want_argmax=True)
    z.backward(gz.to(dev).permute(0, 3, 1, 2))
    kern = _hip.lib().fgnn_last_kernel().decode()          # third generation (mpconv_bwd_ws.hip) where it tiles, else the second
    assert 'mpconv_bwd_ws' in kern or 'mpconv_bwd_sg' in kern, kern
    xr = x[:, :, 0, :].float().clone().requires_grad_(True)                                # [B,N,nin]
    er = et.float().clone().requires_grad_(True)                                           # [B,M,k,net]
    Wr, br = W.clone().requires_grad_(True), bias.clone().requires_grad_(True)
    P = torch.einsum('bnc,cq->bnq', xr, Wr).reshape(B, N, nou, net)
    E = (P[:, idx[0]] * er[:, :, :, None, :]).sum(-1)                                      # [B,M,k,nou]
    sel = am.cpu().long()[..., 0].permute(0, 2, 1)[:, :, None, :]                          # [B,M,1,nou]
    zr = E.gather(2, sel)[:, :, 0, :] + br[None, None, :]                                  # [B,M,nou]
    zr.backward(gz[:, :, 0, :].float())
    tol = 2.0 ** -6
    assert H.rel_err(xd.grad.float().permute(0, 2, 3, 1)[:, :, 0, :], xr.grad) <= tol
    assert H.rel_err(etd.grad.float().permute(0, 2, 3, 1), er.grad) <= tol
    assert H.rel_err(Wd.grad, Wr.grad) <= tol
    assert H.rel_err(bd.grad, br.grad) <= tol


# (nin, nou, N, M, k): the second-generation backward (64 -> 64) and the wide shapes of LDPCModel's layers 2 / 6
ORACLE_BWD_SHAPES = [(64, 64, 96, 48, 6), (64, 64, 48, 96, 3), (64, 128, 96, 48, 6), (64, 128, 48, 96, 3), (128, 64, 96, 48, 6),
                     (128, 64, 48, 96, 3)]


@pytest.mark.parametrize('shape', ORACLE_BWD_SHAPES[:2], ids=lambda s: 'x'.join(map(str, s)))
def test_bf16_backward_vs_oracle_autograd_at_the_benched_batch(shape, dev):
    """The same check at BASELINE config 3's batch, 4096 codewords, for the two 64 -> 64 parity shapes: there a workgroup of the
    third-generation kernels walks 16 samples (LDS-DMA two ahead, images handed between phases), which the small batches above
    never exercise against VALUES — only the bit-reproducibility tests ran that regime."""
    _backward_vs_oracle_autograd(shape, 4096, dev)


@pytest.mark.parametrize('shape', ORACLE_BWD_SHAPES, ids=lambda s: 'x'.join(map(str, s)))
@pytest.mark.parametrize('B', [37, 600])
def test_bf16_backward_vs_oracle_autograd(shape, B, dev):
    _backward_vs_oracle_autograd(shape, B, dev)


def _backward_vs_oracle_autograd(shape, B, dev):
    """The bf16 backward kernels of the parity calls against the ORACLE's autograd directly — `O.mp_conv` in f32 on the same
    bf16-rounded inputs, routing by the oracle's OWN maxima (not by the kernel's argmax).  The upstream gradient is zero on
    the outputs whose best two messages lie closer than the bf16 rounding of P can resolve (a coin flip for any finite
    precision: there the two sides may legitimately route differently); everywhere else the routes must agree, and every
    gradient must match to 2^-6 of its range.  Reference: mp_nn.py:115-175."""
    from fgnn_amd import _hip, ops
    nin, nou, N, M, k = shape
    net = 4
    g = torch.Generator().manual_seed(300 + N + nou + B)
    x = torch.randn(B, N, 1, nin, generator=g).bfloat16()
    idx = _regular_table(N, M, k, g)
    et = torch.randn(B, M, k, net, generator=g).bfloat16()
    W = (torch.randn(nin, nou * net, generator=g) * 0.1).bfloat16().float()     # what the bf16 matrix cores multiply by
    bias = torch.randn(nou, generator=g)
    gz = torch.randn(B, M, 1, nou, generator=g).bfloat16()
    # messages in f32 (reference order), gap between the best two per output
    P = torch.einsum('bnc,cq->bnq', x[:, :, 0, :].float(), W).reshape(B, N, nou, net)
    E = (P[:, idx[0]] * et.float()[:, :, :, None, :]).sum(-1)                              # [B,M,k,nou]
    top2 = E.topk(2, dim=2)[0]
    # what bf16 cannot resolve: P is rounded to bf16 (2^-9 relative per term) before the 4-term edge-type contraction
    mag = (P[:, idx[0]].abs() * et.float().abs()[:, :, :, None, :]).sum(-1).amax(dim=2)    # [B,M,nou]
    firm = (top2[:, :, 0] - top2[:, :, 1]) > 2.0 ** -7 * mag
    assert float(firm.float().mean()) > 0.9, float(firm.float().mean())
    gz = (gz[:, :, 0, :].float() * firm).bfloat16()[:, :, None, :]
    xo = x.permute(0, 3, 1, 2).float().contiguous().requires_grad_(True)                   # [B,nin,N,1]
    eo = et.permute(0, 3, 1, 2).float().contiguous().requires_grad_(True)                  # [B,net,M,k]
    sd = {'filters': W.clone().requires_grad_(True), 'bias': bias.clone().requires_grad_(True)}
    zo = O.mp_conv(sd, '', xo, idx.expand(B, -1, -1).contiguous(), eo, nou=nou, net=net, extension=0, aggregator='max',
                   relu=False)
    zo.backward(gz.permute(0, 3, 1, 2).float())
    xd = x.to(dev).permute(0, 3, 1, 2).requires_grad_(True)
    etd = et.to(dev).permute(0, 3, 1, 2).requires_grad_(True)
    Wd, bd = W.to(dev).requires_grad_(True), bias.to(dev).requires_grad_(True)
    z = ops.mpconv(xd, idx.to(dev).expand(B, -1, -1), etd, Wd, bd, nou, net, 0, _hip.AGG_MAX)
    z.backward(gz.to(dev).permute(0, 3, 1, 2))
    kern = _hip.lib().fgnn_last_kernel().decode()
    if nin == 64:       # 64 -> 64 in one launch of the second-generation kernel, 64 -> 128 as two over the output halves
        assert ('mpconv_bwd_ws' in kern or 'mpconv_bwd_sg' in kern) and kern.endswith(' x2') == (nou == 128), kern
        if nou == 64:
            assert 'mpconv_bwd_ws' in kern, kern
    else:               # 128 -> 64 (round 5): the third-generation kernel, two launches over the halves of the input channels
        assert 'mpconv_bwd_ws' in kern and kern.endswith(' k2'), kern
    assert H.rel_err(z.float(), zo) <= 2.0 ** -6
    tol = 2.0 ** -6
    assert H.rel_err(xd.grad.float(), xo.grad) <= tol, kern
    assert H.rel_err(etd.grad.float(), eo.grad) <= tol, kern
    assert H.rel_err(Wd.grad, sd['filters'].grad) <= tol, kern
    assert H.rel_err(bd.grad, sd['bias'].grad) <= tol, kern


def test_sg_backward_is_bitwise_reproducible_and_matches_first_generation(dev, monkeypatch):
    """No atomics, fixed summation orders: two runs give identical bits; and the first-generation kernel (taken when the
    in-degree is not passed) agrees within bf16 rounding."""
    from fgnn_amd import _hip, ops
    N, M, k, B = 96, 48, 6, 300
    g = torch.Generator().manual_seed(5)
    x = torch.randn(B, N, 1, 64, generator=g).bfloat16().to(dev).permute(0, 3, 1, 2)
    idx = _regular_table(N, M, k, g).to(dev).expand(B, -1, -1)
    et = torch.randn(B, M, k, 4, generator=g).bfloat16().to(dev).permute(0, 3, 1, 2)
    W, bias = (torch.randn(64, 256, generator=g) * 0.1).to(dev), torch.randn(64, generator=g).to(dev)
    gz = torch.randn(B, M, 1, 64, generator=g).bfloat16().to(dev).permute(0, 3, 1, 2)

    def run():
        xd, ed = x.detach().requires_grad_(True), et.detach().requires_grad_(True)
        Wd, bd = W.detach().requires_grad_(True), bias.detach().requires_grad_(True)
        ops.mpconv(xd, idx, ed, Wd, bd, 64, 4, 0, _hip.AGG_MAX).backward(gz)
        return xd.grad, ed.grad, Wd.grad, bd.grad, _hip.lib().fgnn_last_kernel().decode()

    a, b = run(), run()
    assert 'mpconv_bwd_ws' in a[4] or 'mpconv_bwd_sg' in a[4]
    assert all(torch.equal(u, v) for u, v in zip(a[:4], b[:4]))
    monkeypatch.setattr(ops, 'max_in_degree', lambda *_: 0)
    c = run()
    assert 'mpconv_bwd_b16' in c[4]
    for u, v in zip(a[:4], c[:4]):
        assert H.rel_err(u.float(), v.float()) <= 2.0 ** -6


@pytest.mark.parametrize('shape', [(96, 48, 6), (48, 96, 3)], ids=['V->F', 'F->V'])
def test_ws_backward_is_bit_reproducible_over_many_launches(shape, dev):
    """The third-generation backward moves its inputs by LDS-DMA two samples ahead and hands images between phases and wave roles
    with four barriers per sample; a missing wait shows as rare run-to-run differences (as the forward's did), not as a parity
    failure.  40 launches at a batch that gives the workgroups uneven sample counts, all four gradients bit-identical."""
    from fgnn_amd import _hip, ops
    N, M, k = shape
    B = 1100
    g = torch.Generator().manual_seed(9)
    x = torch.randn(B, N, 1, 64, generator=g).bfloat16().to(dev).permute(0, 3, 1, 2)
    idx = _regular_table(N, M, k, g).to(dev).expand(B, -1, -1)
    et = torch.randn(B, M, k, 4, generator=g).bfloat16().to(dev).permute(0, 3, 1, 2)
    W, bias = (torch.randn(64, 256, generator=g) * 0.1).to(dev), torch.randn(64, generator=g).to(dev)
    gz = torch.randn(B, M, 1, 64, generator=g).bfloat16().to(dev).permute(0, 3, 1, 2)
    first = None
    for r in range(40):
        xd, ed = x.detach().requires_grad_(True), et.detach().requires_grad_(True)
        Wd, bd = W.detach().requires_grad_(True), bias.detach().requires_grad_(True)
        ops.mpconv(xd, idx, ed, Wd, bd, 64, 4, 0, _hip.AGG_MAX).backward(gz)
        cur = (xd.grad.clone(), ed.grad.clone(), Wd.grad.clone(), bd.grad.clone())
        if first is None:
            assert 'mpconv_bwd_ws' in _hip.lib().fgnn_last_kernel().decode()
            first = cur
        else:
            assert all(torch.equal(u, v) for u, v in zip(cur, first)), 'launch %d differs from launch 0' % r


@pytest.mark.parametrize('shape', SHAPES[:6] + [SHAPES[9]], ids=IDS[:6] + [IDS[9]])
@pytest.mark.parametrize('nadd', [1, 3])
def test_inference_forward_with_addends_in_the_epilogue(shape, nadd, dev):
    """fgnn_mpconv_forward_addends: y = ReLU(scale * (operator + bias) + shift) + addends (the layer's running sum / residual / skip
    terms) from ONE launch of the third-generation kernel in its inference mode — against the plain launch plus an f32 sum of the
    same bf16 tensors, rounded once.  Shapes the kernel family does not take return 0 and leave y without the addends."""
    import ctypes
    from fgnn_amd import _hip, ops
    nin, nou, N, M, k = shape
    B = 300
    x, idx, et, W, bias, g = _problem(shape, B, dev, seed=3)
    xd, idxd, etd = _dev_views(x, idx, et, dev)
    Wd, bd = W.to(dev), bias.to(dev)
    sc, sh = (torch.rand(nou, generator=g) + 0.5).to(dev), (torch.randn(nou, generator=g) * 0.2).to(dev)
    y0, _ = ops.mpconv_forward_raw(xd, idxd, etd, Wd, bd, nou, 4, 0, _hip.AGG_MAX, post_scale=sc, post_shift=sh, relu=True)
    adds = [torch.randn(B, M, 1, nou, generator=g).bfloat16().to(dev).permute(0, 3, 1, 2) for _ in range(nadd)]
    assert all(a.stride() == y0.stride() for a in adds)
    y1, _ = ops.mpconv_forward_raw(xd, idxd, etd, Wd, bd, nou, 4, 0, _hip.AGG_MAX, post_scale=sc, post_shift=sh, relu=True, addends=adds)
    kern = _hip.lib().fgnn_last_kernel().decode()
    ref = y0.float()
    for a in adds:
        ref = ref + a.float()
    # the kernel adds to the UNROUNDED activation (one rounding), the reference to the rounded one: one bf16 ulp of the sum
    assert float((y1.float() - ref).abs().max()) <= 2.0 ** -7 * max(1.0, float(ref.abs().max())), kern
    if 'mpconv_fwd_ws' in kern:
        d = _hip.make_desc(xd, ops.shared_graph_view(idxd), etd, nou, 4, 0, _hip.AGG_MAX, True, y1)
        y2 = torch.empty_like(y1)
        P = _hip._ptr
        ap = [P(a) for a in adds] + [None] * (3 - nadd)
        rc = _hip.lib().fgnn_mpconv_forward_addends(ctypes.byref(d), P(xd), P(ops.shared_graph_view(idxd)), P(etd), P(Wd), P(bd), P(sc), P(sh),
                                                    ap[0], ap[1], ap[2], P(y2), _hip.stream_ptr())
        assert rc == 1 and torch.equal(y2, y1)


@pytest.mark.parametrize('shape', [(64, 96, 48, 6), (64, 48, 96, 3), (128, 96, 48, 6), (64, 64, 32, 6), (64, 16, 32, 3)],
                         ids=lambda s: 'x'.join(map(str, s)))
def test_ws_backward_with_prebuilt_tables_is_bit_identical(shape, dev, monkeypatch):
    """fgnn_mpconv_backward_with_tables: the transposed incidence built ONCE per graph (one small launch, remembered on the table's
    owner) and copied by the backward's workgroups == the tables every workgroup of every launch used to build for itself: all four
    gradients bit-identical, for the 64 -> 64 shapes, the two-launch 64 -> 128 form and graphs smaller than the LDS layout."""
    from fgnn_amd import _hip, ops
    nou, N, M, k = shape
    B = 300
    g = torch.Generator().manual_seed(N + M)
    x = torch.randn(B, N, 1, 64, generator=g).bfloat16().to(dev).permute(0, 3, 1, 2)
    idx = _regular_table(N, M, k, g).to(dev).expand(B, -1, -1)
    et = torch.randn(B, M, k, 4, generator=g).bfloat16().to(dev).permute(0, 3, 1, 2)
    W, bias = (torch.randn(64, nou * 4, generator=g) * 0.1).to(dev), torch.randn(nou, generator=g).to(dev)
    gz = torch.randn(B, M, 1, nou, generator=g).bfloat16().to(dev).permute(0, 3, 1, 2)

    def run(tables):
        monkeypatch.setattr(ops, 'BACKWARD_TABLES', tables)
        xd, ed = x.detach().requires_grad_(True), et.detach().requires_grad_(True)
        Wd, bd = W.detach().requires_grad_(True), bias.detach().requires_grad_(True)
        ops.mpconv(xd, idx, ed, Wd, bd, nou, 4, 0, _hip.AGG_MAX).backward(gz)
        return xd.grad, ed.grad, Wd.grad, bd.grad, _hip.lib().fgnn_last_kernel().decode()

    a = run(True)
    assert 'mpconv_bwd_ws' in a[4], a[4]
    owner = idx._base if idx._base is not None else idx
    assert getattr(owner, '_fgnn_bwd_tables', (None, None))[1] is not None
    b, c = run(False), run(True)
    for u, v, w in zip(a[:4], b[:4], c[:4]):
        assert torch.equal(u, v) and torch.equal(u, w)


@pytest.mark.parametrize('width', [(64, 64), (64, 128), (128, 64)], ids=lambda w: '%dto%d' % w)
@pytest.mark.parametrize('N', [96, 50, 16, 128, 1 + 16])
@pytest.mark.parametrize('want_argmax', [True, False], ids=['train', 'eval'])
def test_identity_list_fanin_forward_vs_general_kernel_and_oracle(width, N, want_argmax, dev, monkeypatch):
    """The hyper-factor's V -> F call (ONE destination listening to all N variables in order, train_ldpc.py:40-46): when the
    neighbour table is the identity list the forward reduces over the nodes in the matrix-core accumulators
    (mpconv_fwd_fanin_id_kernel, FGNN_DESC_IDENTITY_LIST) instead of walking an LDS image of P neighbour by neighbour.  Against the
    general kernel (which rounds the projected values to bf16 before the edge weight: agreement to bf16 rounding, the same winner
    wherever the general kernel's margin is clear), against the f32 oracle, exact ties -> the first node, bit-reproducible."""
    from fgnn_amd import _hip, ops
    nin, nou = width
    B = 37
    g = torch.Generator().manual_seed(N + nin)
    x = torch.randn(B, N, 1, nin, generator=g).bfloat16()
    x[3, min(5, N - 1)] = x[3, 0]                                      # an exact tie between two nodes of one sample: first occurrence wins
    idx = torch.arange(N).reshape(1, 1, N)
    et = (torch.rand(B, 1, 1, N, generator=g) + 0.5).bfloat16()
    et[3, 0, 0, min(5, N - 1)] = et[3, 0, 0, 0]
    W = torch.randn(nin, nou, generator=g) * 0.1
    bias = torch.randn(nou, generator=g)
    xd = x.to(dev).permute(0, 3, 1, 2)
    idxd = idx.to(dev).expand(B, -1, -1)
    etd = et.to(dev)
    Wd, bd = W.to(dev), bias.to(dev)
    y1, a1 = ops.mpconv_forward_raw(xd, idxd, etd, Wd, bd, nou, 1, 0, _hip.AGG_MAX, want_argmax=want_argmax)
    assert _hip.lib().fgnn_last_kernel().decode().startswith('mpconv_fwd_fanin_id_kernel')
    y1b, a1b = ops.mpconv_forward_raw(xd, idxd, etd, Wd, bd, nou, 1, 0, _hip.AGG_MAX, want_argmax=want_argmax)
    assert torch.equal(y1, y1b) and (a1 is None or torch.equal(a1, a1b))
    monkeypatch.setattr(ops, 'is_identity_list', lambda t: False)
    y0, a0 = ops.mpconv_forward_raw(xd, idxd, etd, Wd, bd, nou, 1, 0, _hip.AGG_MAX, want_argmax=True)
    assert _hip.lib().fgnn_last_kernel().decode().startswith('mpconv_fwd_fanin_kernel')
    assert H.rel_err(y1.float(), y0.float()) <= 2.0 ** -6
    if want_argmax:
        assert float((a1 != a0).float().mean()) <= 0.05                # (near-ties may resolve differently: f32 vs bf16-rounded projections)
        if N > 5:
            assert int((a1[3] == 5).sum()) == 0                        # the exact tie went to node 0, never to its copy
    ref = O.mp_conv({'filters': W.bfloat16().float(), 'bias': bias}, '', x.permute(0, 3, 1, 2).float().contiguous(), idx.expand(B, -1, -1).contiguous(),
                    et.float(), nou=nou, net=1, extension=0, aggregator='max', relu=False)
    assert H.rel_err(y1.float().cpu(), ref) <= 2.0 ** -7
