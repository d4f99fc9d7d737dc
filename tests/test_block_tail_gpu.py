"""GPU: the training-mode fused tail of `mp_conv_residual` (csrc/block_tail.hip) — BatchNorm2 + ReLU ->
conv2 -> BatchNorm3 + LeakyReLU (+ addends) with conv2's wide output recomputed instead of stored.
  * every kernel mode through the C ABI against a plain f32 torch restatement of the same op chain (bf16 rounding points as
    documented: a2 and the matrix-core operand of W2 are bf16, everything else f32);
  * the whole block, forward + backward, against the staged path (FUSE_TRAIN_TAIL off) and against the f32 oracle."""
import ctypes

import pytest
import torch

import fgnn_oracle as O
import helpers as H

pytestmark = pytest.mark.gpu


def _ref_chain(e, s2, t2, slope2, W2, b2):
    a2 = torch.nn.functional.leaky_relu(e.float() * s2 + t2, slope2).bfloat16()
    z3 = a2.float() @ W2.bfloat16().float().t() + b2
    return a2, z3


@pytest.mark.parametrize('Cout', [64, 128, 256])
@pytest.mark.parametrize('R', [1, 37, 4096 + 5, 48 * 300])
def test_block_tail_kernels_vs_torch(R, Cout, dev, finaliser_mode):
    from fgnn_amd import _hip
    L = _hip.lib()
    P = _hip._ptr
    g = torch.Generator().manual_seed(R + Cout)
    e = torch.randn(R, 64, generator=g).bfloat16().to(dev)
    s2, t2 = (torch.rand(64, generator=g) + 0.5).to(dev), (torch.randn(64, generator=g) * 0.3).to(dev)
    W2, b2 = (torch.randn(Cout, 64, generator=g) * 0.2).to(dev), torch.randn(Cout, generator=g).to(dev)
    slope2, slope3 = 0.0, 0.01
    a2_ref, z3 = _ref_chain(e, s2, t2, slope2, W2, b2)
    npart = int(L.fgnn_block_tail_partials(R, Cout))
    assert 1 <= npart <= 1024
    ws = torch.zeros(int(L.fgnn_bn_workspace_bytes(R, Cout)) // 4, device=dev)
    a2 = torch.empty_like(e)
    st = _hip.stream_ptr()
    # ---- mode 0: statistics partials, and BatchNorm3 finalised by the same CALL (a finaliser launch or the kernel's last workgroup) ----
    from fgnn_amd import ops
    fold = ops._fold_scratch(dev)
    gamma3, beta3 = (torch.rand(Cout, generator=g) + 0.5).to(dev), torch.randn(Cout, generator=g).to(dev)
    rm, rv = torch.zeros(Cout, device=dev), torch.ones(Cout, device=dev)
    nbt = torch.zeros((), device=dev, dtype=torch.int64)
    st3 = torch.full((4, Cout), float('nan'), device=dev)
    fin = _hip.bn_final(st3, gamma3, beta3, rm, rv, nbt, 0.1, 1e-5, R, 3 * R)       # (population 3 R: as if every row stood for three)
    _hip.check(L.fgnn_block_tail_stats(P(e), P(s2), P(t2), slope2, P(W2), P(b2), R, Cout, P(ws), fin, P(fold), st))
    part = ws[:npart * 2 * Cout].reshape(npart, 2, Cout).double().sum(0)      # sums of z3 - b2
    zz = z3.double()
    z0 = zz - b2.double()
    assert H.rel_err(part[0], z0.sum(0)) <= 1e-4 and H.rel_err(part[1], (z0 * z0).sum(0)) <= 1e-4
    assert int(nbt) == 1 and int(fold[:65].abs().sum()) == 0               # the ticket counters are back at zero
    n3 = 3.0 * R
    assert H.rel_err(rm, 0.1 * zz.mean(0)) <= 1e-4
    assert H.rel_err(rv - 0.9, 0.1 * zz.var(0, unbiased=False) * n3 / (n3 - 1.0)) <= 2e-3
    mean, var = zz.mean(0), zz.var(0, unbiased=False)
    assert H.rel_err(st3[0], mean) <= 1e-4
    if R > 1:               # (a single row: variance 0 on one side, rounding noise on the other — invstd ~ eps^-1/2 either way)
        assert H.rel_err(st3[1], 1.0 / torch.sqrt(var + 1e-5)) <= 1e-3
    # ---- mode 1: apply with 0 .. 3 addends ----
    adds = [torch.randn(R, Cout, generator=g).bfloat16().to(dev) for _ in range(3)]
    for nadd in range(4):
        out = torch.empty(R, Cout, device=dev, dtype=torch.bfloat16)
        ap = [P(a) for a in adds[:nadd]] + [None] * (3 - nadd)
        a2.zero_()
        _hip.check(L.fgnn_block_tail_apply(P(e), P(s2), P(t2), slope2, P(W2), P(b2), P(st3[2]), P(st3[3]), slope3, ap[0], ap[1],
                                           ap[2], None, P(out), P(a2) if nadd != 1 else None, R, Cout, st))
        if nadd != 1:
            assert H.rel_err(a2.float(), a2_ref.float()) <= 2.0 ** -8        # (the kernel's affine is one fused multiply-add)
        ref = torch.nn.functional.leaky_relu(z3 * st3[2] + st3[3], slope3)
        for a in adds[:nadd]:
            ref = ref + a.float()
        assert H.rel_err(out.float(), ref) <= 2.0 ** -7, nadd
    # ---- modes 2 + 3: BatchNorm3 backward sums, gz3 and ga2 ----
    gout = torch.randn(R, Cout, generator=g).bfloat16().to(dev)
    gz3, ga2 = torch.empty(R, Cout, device=dev, dtype=torch.bfloat16), torch.empty(R, 64, device=dev, dtype=torch.bfloat16)
    gw3, gb3 = torch.zeros(Cout, device=dev), torch.zeros(Cout, device=dev)
    # BatchNorm2's statistics as the forward would have left them (any values do for the arithmetic under test)
    mean2, invstd2 = (torch.randn(64, generator=g) * 0.1).to(dev), (torch.rand(64, generator=g) + 0.5).to(dev)
    gw2, gb2 = torch.zeros(64, device=dev), torch.zeros(64, device=dev)
    dsum2 = torch.full((2, 64), float('nan'), device=dev)
    wsb = torch.zeros((2048 * Cout + 2 * Cout + 1024 * 128), device=dev)
    _hip.check(L.fgnn_block_tail_backward(P(e), P(s2), P(t2), slope2, P(W2), P(b2), P(st3[0]), P(st3[1]), P(gamma3), P(st3[2]),
                                          P(st3[3]), slope3, P(gout), P(gz3), P(ga2), P(gw3), P(gb3), P(mean2), P(invstd2), P(gw2),
                                          P(gb2), P(dsum2), R, Cout, P(wsb), wsb.numel() * 4, P(fold), st))
    assert int(fold[:65].abs().sum()) == 0
    zl = z3.detach().clone().requires_grad_(True)
    gam, bet = gamma3.clone().requires_grad_(True), beta3.clone().requires_grad_(True)
    zh = (zl - zl.mean(0)) * torch.rsqrt(zl.var(0, unbiased=False) + 1e-5)
    torch.nn.functional.leaky_relu(zh * gam + bet, slope3).backward(gout.float())
    scale = max(1.0, float(zl.grad.abs().max()))
    if R > 1:
        assert float((gz3.float() - zl.grad).abs().max()) <= 2.0 ** -6 * scale
        assert H.rel_err(gw3, gam.grad) <= 2e-3 and H.rel_err(gb3, bet.grad) <= 2e-3
    ga_ref = gz3.float() @ W2.bfloat16().float()
    assert H.rel_err(ga2.float(), ga_ref) <= 2.0 ** -7
    # BatchNorm2's backward sums ride along, finalised by the grad kernel: dbeta = sum g2', dgamma = invstd2 (sum g2' e - mean2 sum g2'),
    # g2' = ga2 act2'(pre2) (from the f32 accumulators)
    pos = (e.float() * s2 + t2) > 0
    g2 = torch.where(pos, ga_ref, ga_ref * slope2).double()
    dbeta = g2.sum(0)
    dgamma = invstd2.double() * ((g2 * e.double()).sum(0) - mean2.double() * dbeta)
    tol = 2e-3 * max(1.0, float(dgamma.abs().max()), float((g2 * e.double()).sum(0).abs().max()))
    assert H.rel_err(dsum2[0], dbeta) <= 2e-3 and float((dsum2[1].double() - dgamma).abs().max()) <= tol
    assert torch.equal(gb2, dsum2[0]) and torch.equal(gw2, dsum2[1])


@pytest.mark.parametrize('Cout', [64, 256])
def test_block_tail_apply_broadcasts_a_per_sample_addend(Cout, dev):
    """fgnn_block_tail_apply / fgnn_bn_apply with `addend_period`: an addend of ONE row per `period` output rows (the hyper-factor's
    message to the 96 variables of a codeword, carried as [B][C]) == the same addend materialised."""
    from fgnn_amd import _hip
    L = _hip.lib()
    P = _hip._ptr
    Bn, M = 37, 96
    R = Bn * M
    g = torch.Generator().manual_seed(Cout)
    e = torch.randn(R, 64, generator=g).bfloat16().to(dev)
    s2, t2 = (torch.rand(64, generator=g) + 0.5).to(dev), (torch.randn(64, generator=g) * 0.3).to(dev)
    W2, b2 = (torch.randn(Cout, 64, generator=g) * 0.2).to(dev), torch.randn(Cout, generator=g).to(dev)
    s3, t3 = (torch.rand(Cout, generator=g) + 0.5).to(dev), torch.randn(Cout, generator=g).to(dev)
    full = torch.randn(R, Cout, generator=g).bfloat16().to(dev)
    small = torch.randn(Bn, Cout, generator=g).bfloat16().to(dev)
    big = small[:, None, :].expand(Bn, M, Cout).reshape(R, Cout).contiguous()
    st = _hip.stream_ptr()
    outs = []
    for adds, periods in (((full, big, None), None), ((full, small, None), (ctypes.c_int32 * 3)(1, M, 1)),
                          ((small, full, None), (ctypes.c_int32 * 3)(M, 1, 1))):
        out = torch.empty(R, Cout, device=dev, dtype=torch.bfloat16)
        _hip.check(L.fgnn_block_tail_apply(P(e), P(s2), P(t2), 0.0, P(W2), P(b2), P(s3), P(t3), 0.01, P(adds[0]), P(adds[1]), None,
                                           periods, P(out), None, R, Cout, st))
        outs.append(out)
    assert torch.equal(outs[0], outs[1])
    assert H.rel_err(outs[2].float(), outs[0].float()) <= 2.0 ** -7          # (the other order of two bf16-exact f32 additions)
    x = torch.randn(R, Cout, generator=g).bfloat16().to(dev)
    ya, yb = torch.empty_like(x), torch.empty_like(x)
    _hip.check(L.fgnn_bn_apply(P(x), P(ya), R, Cout, _hip.BF16, P(s3), P(t3), 0.0, P(big), P(full), None, None, st))
    _hip.check(L.fgnn_bn_apply(P(x), P(yb), R, Cout, _hip.BF16, P(s3), P(t3), 0.0, P(small), P(full), None, (ctypes.c_int32 * 3)(M, 1, 1), st))
    assert torch.equal(ya, yb)
    # and the backward of the broadcast: the node sum
    gs = torch.empty(Bn, Cout, device=dev, dtype=torch.bfloat16)
    _hip.check(L.fgnn_node_sum(P(full), P(gs), Bn, M, Cout, _hip.BF16, st))
    assert H.rel_err(gs.float(), full.float().reshape(Bn, M, Cout).sum(1)) <= 2.0 ** -8


CASES = [  # nin, nout, N, M, k, net
    (64, 64, 96, 48, 6, 4), (64, 64, 48, 96, 3, 4), (128, 256, 96, 48, 6, 4), (256, 256, 48, 96, 3, 4), (256, 128, 96, 48, 6, 4),
    (64, 64, 96, 1, 96, 1), (256, 256, 1, 96, 1, 1)]


@pytest.mark.parametrize('case', CASES, ids=lambda c: 'x'.join(map(str, c)))
@pytest.mark.parametrize('nadd', [0, 2, 4])
def test_fused_training_tail_matches_staged_path_and_oracle(case, nadd, dev):
    """`mp_conv_residual` in training mode, bf16, fused tail on / off.  Forward: same output (2^-6) and running statistics, and
    within 2^-5 of the f32 oracle's block.  Backward: both paths share the operator kernels and their routing; what differs is
    where bf16 rounding enters the BatchNorm backward sums, which batch-statistics BatchNorm amplifies — so every gradient is
    held against the ORACLE's autograd: the fused path's distance may not exceed 1.5 x the staged path's
    (+ 2^-5: max-norm of quantities that are themselves 8 - 30 % off the f32 result in either path); gradients that are pure cancellation noise (biases in front of a batch-statistics BatchNorm) are skipped."""
    from fgnn_amd import ops
    from fgnn_amd.mpnn import blocks, mp_conv_residual, mp_conv_type
    nin, nout, N, M, k, net = case
    B = 40
    g = torch.Generator().manual_seed(nin + nout + M + nadd)
    torch.manual_seed(5)
    m = mp_conv_residual(nin, 64, net, extension=mp_conv_type.NO_EXTENSION, with_residual=False, aggregator='max',
                         nout=None if nout == nin else nout).to(dev).train()
    with torch.no_grad():
        m.mp_conv.filters.mul_(10.0)                         # the constructor's U(-.01, .01) would leave every message tiny
    x = torch.randn(B, N, 1, nin, generator=g).bfloat16().to(dev).permute(0, 3, 1, 2)
    if M == 1:
        idx = torch.arange(N).reshape(1, 1, N)
    else:
        idx = torch.randint(0, N, (1, M, k), generator=g)
    idx = idx.to(dev).expand(B, -1, -1)
    et = torch.randn(B, M, k, net, generator=g).bfloat16().to(dev).permute(0, 3, 1, 2)
    adds = [torch.randn(B, M, 1, nout, generator=g).bfloat16().to(dev).permute(0, 3, 1, 2) for _ in range(nadd)]
    gy = torch.randn(B, M, 1, nout, generator=g).bfloat16().to(dev).permute(0, 3, 1, 2)
    sd0 = {k_: v.clone() for k_, v in m.state_dict().items()}

    def run(fused):
        m.load_state_dict(sd0)
        for q in m.parameters():
            q.grad = None
        ops._WS.clear()                                      # (a cold workspace: the tail must size it before the operator runs)
        blocks.FUSE_TRAIN_TAIL = fused
        try:
            xd, ed = x.detach().requires_grad_(True), et.detach().requires_grad_(True)
            ad = [a.detach().requires_grad_(True) for a in adds]
            with torch.autocast('cuda', dtype=torch.bfloat16):
                y = m(xd, idx, ed, addend=list(ad) if ad else None)
            y.backward(gy)
        finally:
            blocks.FUSE_TRAIN_TAIL = True
        out = {'y': y.detach().float().cpu(), 'gx': xd.grad.float().cpu(), 'get': ed.grad.float().cpu()}
        out.update({n: q.grad.detach().float().cpu() for n, q in m.named_parameters()})
        stats = {n: v.detach().float().clone() for n, v in m.state_dict().items() if 'running' in n}
        return out, [a.grad.float() for a in ad], stats

    rec = []
    ops.TIMER = type('T', (), {'run': staticmethod(lambda sym, nb, nf, launch, use_note=True, **kw: (rec.append(sym), launch()))})()
    try:
        f, fa, fs = run(True)
    finally:
        ops.TIMER = None
    assert any('block_tail_apply' in r for r in rec) and any('block_tail_backward' in r for r in rec), rec
    s, sa, ss = run(False)
    assert H.rel_err(f['y'], s['y']) <= 2.0 ** -6
    for a, b in zip(fa, sa):
        assert torch.equal(a, b)                             # an addend's gradient is the upstream gradient itself
    for n in fs:
        assert H.rel_err(fs[n], ss[n]) <= 1e-3, n
    # the f32 oracle's block, forward and autograd, on the same (bf16-rounded) inputs
    sd = {k_: v.detach().cpu().float().clone() for k_, v in sd0.items()}
    names = [n for n, _ in m.named_parameters()]
    for n in names:
        sd[n].requires_grad_(True)
    xo, eo = x.float().cpu().contiguous().requires_grad_(True), et.float().cpu().contiguous().requires_grad_(True)
    ref = O.residual_block(sd, '', xo, idx.cpu().contiguous(), eo, net=net, extension=0, aggregator='max', with_residual=False,
                           training=True)
    ref.backward(gy.float().cpu())
    with torch.no_grad():
        for a in adds:
            ref = ref + a.float().cpu()
    assert H.rel_err(f['y'], ref) <= 2.0 ** -5
    r = {'gx': xo.grad, 'get': eo.grad}
    r.update({n: sd[n].grad for n in names})
    gmax = max(float(v.abs().max()) for v in r.values() if v is not None)
    checked = 0
    for n, rv in r.items():
        if rv is None or float(rv.abs().max()) < 1e-4 * gmax:        # no gradient / cancellation noise
            continue
        ef, es = H.rel_err(f[n], rv), H.rel_err(s[n], rv)
        assert ef <= 1.5 * es + 2.0 ** -5, (n, ef, es)
        checked += 1
    assert checked >= 8


@pytest.mark.parametrize('Cin', [64, 128, 256])
@pytest.mark.parametrize('R', [2, 37, 4096 + 5, 96 * 300])
def test_block_head_backward_kernel_vs_torch(R, Cin, dev, finaliser_mode):
    """fgnn_block_head_backward through the C ABI: BatchNorm1 + LeakyReLU backward (batch statistics) and conv1's input gradient
    against autograd through the same chain in f32 torch (the kernel rounds gz1 to bf16 before the product, as the staged path
    does): gz1 and gx to 2^-7 of their range, the BatchNorm parameter gradients to 1e-3."""
    from fgnn_amd import _hip, ops
    L, P = _hip.lib(), _hip._ptr
    g = torch.Generator().manual_seed(R + Cin)
    z1 = (torch.randn(R, 64, generator=g) * 1.5 + 0.3).bfloat16()
    ga1 = torch.randn(R, 64, generator=g).bfloat16()
    gamma, beta = torch.rand(64, generator=g) + 0.5, torch.randn(64, generator=g) * 0.3
    W1 = torch.randn(64, Cin, generator=g) * 0.1
    slope, eps = 0.01, 1e-5
    zf = z1.float().requires_grad_(True)
    mean, var = zf.mean(0), zf.var(0, unbiased=False)
    invstd = torch.rsqrt(var + eps)
    gam, bet = gamma.clone().requires_grad_(True), beta.clone().requires_grad_(True)
    a1 = torch.nn.functional.leaky_relu((zf - mean) * invstd * gam + bet, slope)
    a1.backward(ga1.float())
    gz_ref = zf.grad
    gx_ref = gz_ref.bfloat16().float() @ W1.bfloat16().float()
    d = lambda t: t.to(dev).contiguous()
    gz, gx = torch.empty(R, 64, device=dev, dtype=torch.bfloat16), torch.empty(R, Cin, device=dev, dtype=torch.bfloat16)
    gw, gb = torch.zeros(64, device=dev), torch.zeros(64, device=dev)
    ws = ops._workspace(dev, int(L.fgnn_bn_workspace_bytes(R, 64)))
    z1d, gad, md, isd, gd, bd, Wd = d(z1), d(ga1), d(mean.detach()), d(invstd.detach()), d(gamma), d(beta), d(W1)
    fold = ops._fold_scratch(dev)
    _hip.check(L.fgnn_block_head_backward(P(z1d), P(gad), P(md), P(isd), P(gd), P(bd), slope, P(Wd), P(gz), P(gx), P(gw), P(gb), R, Cin,
                                          P(ws), ws.numel() * 4, P(fold), _hip.stream_ptr()))
    assert _hip.lib().fgnn_last_kernel().decode() == 'block_head_bwd_kernel<%d>' % (Cin // 64)
    assert H.rel_err(gz.float().cpu(), gz_ref) <= 2.0 ** -7
    assert H.rel_err(gx.float().cpu(), gx_ref) <= 2.0 ** -7
    assert H.rel_err(gw.cpu(), gam.grad) <= 1e-3 and H.rel_err(gb.cpu(), bet.grad) <= 1e-3
    with pytest.raises(_hip.FgnnHipError):
        _hip.check(L.fgnn_block_head_backward(P(z1d), P(gad), P(md), P(isd), P(gd), P(bd), slope, P(Wd), P(gz), P(gx), P(gw), P(gb), R, 96,
                                              P(ws), ws.numel() * 4, P(fold), _hip.stream_ptr()))


@pytest.mark.parametrize('case', CASES[:5], ids=lambda c: 'x'.join(map(str, c)))
def test_fused_training_head_matches_staged_head(case, dev, monkeypatch):
    """`mp_conv_residual` in training mode with the head's backward fused (blocks._BlockHead) and staged: the forward runs the
    same kernels (bit-identical output and running statistics); in the backward only the rounding of BatchNorm1's input gradient
    and the summation order of conv1's input-gradient product differ."""
    from fgnn_amd import ops
    from fgnn_amd.mpnn import blocks, mp_conv_residual, mp_conv_type
    nin, nout, N, M, k, net = case
    B = 40
    g = torch.Generator().manual_seed(nin + nout + M)
    torch.manual_seed(6)
    m = mp_conv_residual(nin, 64, net, extension=mp_conv_type.NO_EXTENSION, with_residual=False, aggregator='max',
                         nout=None if nout == nin else nout).to(dev).train()
    with torch.no_grad():
        m.mp_conv.filters.mul_(10.0)
    x = torch.randn(B, N, 1, nin, generator=g).bfloat16().to(dev).permute(0, 3, 1, 2)
    idx = torch.randint(0, N, (1, M, k), generator=g).to(dev).expand(B, -1, -1)
    et = torch.randn(B, M, k, net, generator=g).bfloat16().to(dev).permute(0, 3, 1, 2)
    gy = torch.randn(B, M, 1, nout, generator=g).bfloat16().to(dev).permute(0, 3, 1, 2)
    sd0 = {k_: v.clone() for k_, v in m.state_dict().items()}

    def run(fused):
        m.load_state_dict(sd0)
        for q in m.parameters():
            q.grad = None
        monkeypatch.setattr(blocks, "FUSE_TRAIN_HEAD", fused)
        monkeypatch.setattr(blocks, "_HEAD_WIDTHS", (64, 128, 256))
        rec = []
        ops.TIMER = type('T', (), {'run': staticmethod(lambda sym, nb, nf, launch, use_note=True, **kw: (rec.append(sym), launch()))})()
        try:
            xd, ed = x.detach().requires_grad_(True), et.detach().requires_grad_(True)
            with torch.autocast('cuda', dtype=torch.bfloat16):
                y = m(xd, idx, ed)
            y.backward(gy)
        finally:
            ops.TIMER = None
        out = {'y': y.detach().float(), 'gx': xd.grad.float(), 'get': ed.grad.float()}
        out.update({n: q.grad.detach().float().clone() for n, q in m.named_parameters()})
        out.update({n: v.detach().float().clone() for n, v in m.state_dict().items() if 'running' in n})
        return out, rec

    f, frec = run(True)
    s, srec = run(False)
    assert any('block_head_backward' in r for r in frec) and not any('block_head_backward' in r for r in srec), (frec, srec)
    assert torch.equal(f['y'], s['y'])
    for n in f:
        if 'running' in n or n == 'get' or not n.startswith('conv1') and n not in ('gx',):
            assert torch.equal(f[n], s[n]), n                 # everything behind the head sees the same tensors
    for n in ('gx', 'conv1.0.weight', 'conv1.1.weight', 'conv1.1.bias'):
        assert H.rel_err(f[n], s[n]) <= 2.0 ** -7, (n, H.rel_err(f[n], s[n]))


@pytest.mark.parametrize('width', [(64, 64), (128, 256), (256, 256), (256, 128), 'plain64to128', 'plain128to64'], ids=str)
def test_single_source_fanout_runs_on_one_row_per_sample(width, dev, monkeypatch, finaliser_mode):
    """The LDPC hyper-factor -> variables call (/root/reference/train_ldpc.py:40-46,82-88: ONE source node, `hnn_idx_f2v` == 0,
    `hetype_f2v` == 1): every one of the 96 destinations receives the same message, so the block (or the plain operator of the
    64 <-> 128 layers, factor_mpnn_sp.py:88-91) is computed on one row per sample and handed on as a broadcast
    (ops.single_source_fanout / ops.broadcast_nodes).  Against the SAME module run on the 96 materialised rows (switch off):
    output, BatchNorm running statistics — the unbiased variance counts B * 96 rows either way —, the gradient of the input and of
    every parameter; and against the f32 oracle's block."""
    from fgnn_amd import ops
    from fgnn_amd.mpnn import mp_conv_residual, mp_conv_type, mp_conv_v2
    B, M = 48, 96
    plain = isinstance(width, str)
    nin, nout = ((64, 128) if width == 'plain64to128' else (128, 64)) if plain else width
    g = torch.Generator().manual_seed(nin + 3 * nout)
    torch.manual_seed(7)
    if plain:
        m = mp_conv_v2(nin, nout, 1, extension=mp_conv_type.NO_EXTENSION, aggregtor='max').to(dev).train()
        filt = m.filters
    else:
        m = mp_conv_residual(nin, 64, 1, extension=mp_conv_type.NO_EXTENSION, with_residual=False, aggregator='max',
                             nout=None if nout == nin else nout).to(dev).train()
        filt = m.mp_conv.filters
    with torch.no_grad():
        filt.mul_(10.0)
    x = torch.randn(B, 1, 1, nin, generator=g).bfloat16().to(dev).permute(0, 3, 1, 2)
    idx = torch.zeros(1, M, 1, dtype=torch.int64, device=dev).expand(B, -1, -1)
    w = torch.full((1, 1, 1, 1), 1.0, device=dev, dtype=torch.bfloat16)
    et = w.expand(B, 1, M, 1)                                   # one weight for all destinations: a stride-0 node axis
    gy = torch.randn(B, M, 1, nout, generator=g).bfloat16().to(dev).permute(0, 3, 1, 2)
    sd0 = {k_: v.clone() for k_, v in m.state_dict().items()}

    def run(broadcast):
        monkeypatch.setattr(ops, 'FANOUT_BROADCAST', broadcast)
        m.load_state_dict(sd0)
        for q in m.parameters():
            q.grad = None
        xd = x.detach().requires_grad_(True)
        with torch.autocast('cuda', dtype=torch.bfloat16):
            y = m(xd, idx, et)
        assert (getattr(y, '_fgnn_bcast_src', None) is not None) == broadcast and tuple(y.shape) == (B, nout, M, 1)
        y.backward(gy)
        out = {'y': y.detach().float().cpu(), 'gx': xd.grad.float().cpu()}
        out.update({n: q.grad.detach().float().cpu() for n, q in m.named_parameters() if q.grad is not None})
        stats = {n: v.detach().double().cpu().clone() for n, v in m.state_dict().items() if 'running' in n or 'tracked' in n}
        return out, stats

    b, bs = run(True)
    f, fs = run(False)
    assert H.rel_err(b['y'], f['y']) <= 2.0 ** -6
    assert float((b['y'] - b['y'][:, :, :1, :]).abs().max()) == 0.0
    for n in fs:        # (the conv1 BatchNorm sees B rows either way; the two behind the operator B x 96 identical ones)
        assert H.rel_err(bs[n], fs[n]) <= 1e-3, n
    # the f32 oracle on the same (bf16-rounded) inputs, the 96 rows materialised as the reference does
    sd = {k_: v.detach().cpu().float().clone() for k_, v in sd0.items()}
    names = [n for n, _ in m.named_parameters()]
    for n in names:
        sd[n].requires_grad_(True)
    xo = x.float().cpu().contiguous().requires_grad_(True)
    eo = et.float().cpu().contiguous()
    if plain:
        ref = O.mp_conv(sd, '', xo, idx.cpu().contiguous(), eo, nou=nout, net=1, extension=0, aggregator='max', relu=True, training=True)
    else:
        ref = O.residual_block(sd, '', xo, idx.cpu().contiguous(), eo, net=1, extension=0, aggregator='max', with_residual=False,
                               training=True)
    ref.backward(gy.float().cpu())
    assert H.rel_err(b['y'], ref) <= 2.0 ** -5
    for n, v in bs.items():
        if 'running' in n:
            assert H.rel_err(v, sd[n].double()) <= 2e-2, n     # (the oracle updated its running statistics in place)
    r = {'gx': xo.grad}
    r.update({n: sd[n].grad for n in names})
    gmax = max(float(v.abs().max()) for v in r.values() if v is not None)
    checked = 0
    for n, ref_g in r.items():
        if ref_g is None or n not in b or float(ref_g.abs().max()) < 1e-3 * gmax:
            continue                 # pure cancellation noise (a bias in front of a batch-statistics BatchNorm)
        # in the Frobenius norm: the two paths round the pre-activation differently (one bf16 rounding of P more on the one-row path), so
        # an element within rounding of the (Leaky)ReLU kink takes the other slope in one of them — one such flip moves a whole column
        # of a weight gradient by ~1 / sqrt(B) of its norm (9 % of the max-norm seen for the plain 128 -> 64 filters, 0.2 % materialised)
        fro = lambda u, v: float((u.double() - v.double()).norm() / v.double().norm().clamp_min(1e-30))
        db, df = fro(b[n], ref_g), fro(f[n], ref_g)
        assert db <= 1.5 * df + 2.0 ** -4, (n, db, df)
        checked += 1
    assert checked >= 3


@pytest.mark.gpu
@pytest.mark.parametrize('dtype', [torch.bfloat16, torch.float32], ids=['bf16', 'f32'])
@pytest.mark.parametrize('B,M,C', [(4096, 96, 64), (515, 96, 128), (300, 48, 256), (77, 5, 64), (64, 96, 32), (33, 96, 192), (1, 1, 64)])
def test_node_sum_vs_torch(B, M, C, dtype, dev):
    """fgnn_node_sum (csrc/sum_n.hip: the gradient of a per-sample row broadcast over the sample's nodes, several threads per 16-byte
    channel chunk with a fixed lane-permute tree): against a float64 sum, bit-identical on a second run; channel counts on both
    sides of the power-of-two rule, node counts below and above the split threshold, a batch that is not a multiple of the block."""
    from fgnn_amd import _hip
    L = _hip.lib()
    g = torch.Generator().manual_seed(B + M + C)
    x = torch.randn(B, M, C, generator=g).to(dev).to(dtype)
    outs = []
    for _ in range(2):
        out = torch.full((B, C), 7.0, device=dev, dtype=dtype)
        _hip.check(L.fgnn_node_sum(_hip._ptr(x), _hip._ptr(out), B, M, C, _hip.dtype_code(x), _hip.stream_ptr()))
        torch.cuda.synchronize()
        outs.append(out.clone())
    ref = x.double().sum(1)
    tol = 2.0 ** -8 if dtype == torch.bfloat16 else 1e-6
    assert float((outs[0].double() - ref).abs().max()) <= tol * max(1.0, float(ref.abs().max()))
    assert torch.equal(outs[0], outs[1])


def _parity_block(dev, seed=3):
    from fgnn_amd.mpnn import mp_conv_residual, mp_conv_type
    torch.manual_seed(seed)
    m = mp_conv_residual(64, 64, 4, extension=mp_conv_type.NO_EXTENSION, with_residual=False, aggregator='max').to(dev).train()
    with torch.no_grad():
        m.mp_conv.filters.mul_(10.0)
    return m


def test_dropped_producer_statistics_do_not_finalise_a_batchnorm_twice(dev):
    """Round-5 advisory: a statistics-producing launch given ``bn=`` finalises that BatchNorm itself (momentum, counter).  When the
    consumer cannot use the pending statistics — here the operator's output is NOT channel-fastest (a plain-contiguous input), so
    the fused tail copies the rows and drops them — the second statistics pass must leave the running buffers and
    num_batches_tracked alone: ONE momentum update and ONE count per forward, the same as for the channel-fastest input."""
    B, N, M, k = 40, 96, 48, 6
    g = torch.Generator().manual_seed(11)
    xr = torch.randn(B, N, 1, 64, generator=g).bfloat16().to(dev)
    idx = torch.randint(0, N, (1, M, k), generator=g).to(dev).expand(B, -1, -1)
    et = torch.randn(B, M, k, 4, generator=g).bfloat16().to(dev).permute(0, 3, 1, 2)
    outs = []
    for plain in (False, True):
        m = _parity_block(dev)
        x = xr.permute(0, 3, 1, 2)
        if plain:
            x = x.contiguous()                   # [B, C, N, 1] channel-slowest: every row view downstream needs a copy
            assert not x.permute(0, 2, 3, 1).is_contiguous()
        with torch.autocast('cuda', dtype=torch.bfloat16):
            y = m(x, idx, et)
        sd = m.state_dict()
        for n, v in sd.items():
            if n.endswith('num_batches_tracked'):
                assert int(v) == 1, (plain, n, int(v))
        outs.append((y.float(), {n: v.float().clone() for n, v in sd.items() if 'running' in n}))
    (ya, sa), (yb, sb) = outs
    assert H.rel_err(ya, yb) <= 2.0 ** -6
    for n in sa:
        assert H.rel_err(sa[n], sb[n]) <= 2e-3, n


def test_training_block_takes_a_bare_tensor_addend(dev):
    """Round-5 advisory: ``mp_conv_residual.forward`` documents ``addend`` as a tensor, a list or a callable; a bare multi-element
    tensor in TRAINING mode used to be truth-tested (``not addend`` -> 'Boolean value of Tensor ... is ambiguous')."""
    B, N, M, k = 24, 96, 48, 6
    g = torch.Generator().manual_seed(12)
    x = torch.randn(B, N, 1, 64, generator=g).bfloat16().to(dev).permute(0, 3, 1, 2)
    idx = torch.randint(0, N, (1, M, k), generator=g).to(dev).expand(B, -1, -1)
    et = torch.randn(B, M, k, 4, generator=g).bfloat16().to(dev).permute(0, 3, 1, 2)
    a = torch.randn(B, M, 1, 64, generator=g).bfloat16().to(dev).permute(0, 3, 1, 2)
    m = _parity_block(dev)
    sd0 = {n: v.clone() for n, v in m.state_dict().items()}
    with torch.autocast('cuda', dtype=torch.bfloat16):
        y_t = m(x, idx, et, addend=a)
        m.load_state_dict(sd0)
        y_l = m(x, idx, et, addend=[a])
        m.load_state_dict(sd0)
        y_0 = m(x, idx, et)
    assert torch.equal(y_t, y_l)
    assert H.rel_err(y_t.float(), y_0.float() + a.float()) <= 2.0 ** -6


@pytest.mark.parametrize('C', [5, 12])
def test_node_sum_of_a_width_outside_the_kernel(C, dev):
    """Round-5 advisory: the broadcast fan-out is taken for any channel count; its backward (pointwise.node_sum) must then exist for
    any channel count (fgnn_node_sum itself wants 16-byte chunks)."""
    from fgnn_amd.mpnn import pointwise
    g = torch.randn(6 * 7, C, device=dev).bfloat16()
    out = pointwise.node_sum(g, 7)
    assert out.shape == (6, C)
    assert H.rel_err(out.float(), g.float().view(6, 7, C).sum(1)) <= 2.0 ** -7


@pytest.mark.parametrize('nin,nout', [(256, 256), (128, 256), (256, 128), (64, 64)])
def test_inference_fanout_block_runs_on_one_row_and_rides_as_a_row_addend(nin, nout, dev, monkeypatch):
    """Round 6, inference: the hyper-factor -> variables `mp_conv_residual` (one source node, identical single edges,
    /root/reference/train_ldpc.py:40-46,82-88) is formed on ONE row per codeword (`fgnn_mpconv_block_forward_fanout` with M = 1) and
    handed on as a broadcast; the parity F->V block's kernel takes it as a per-sample ROW addend (`fgnn_mpconv_block_forward_rows`).
    Both equal the materialised forms bit for bit: the 96 rows the reference writes are identical rows."""
    from fgnn_amd import _hip, ops
    from fgnn_amd.mpnn import mp_conv_residual, mp_conv_type
    B, M = 48, 96
    g = torch.Generator().manual_seed(nin + nout)
    torch.manual_seed(7)
    hyper = mp_conv_residual(nin, 64, 1, extension=mp_conv_type.NO_EXTENSION, with_residual=False, aggregator='max',
                             nout=None if nout == nin else nout).to(dev).eval()
    parity = mp_conv_residual(nin, 64, 4, extension=mp_conv_type.NO_EXTENSION, with_residual=False, aggregator='max',
                              nout=None if nout == nin else nout).to(dev).eval()
    with torch.no_grad():
        for m in (hyper, parity):
            m.mp_conv.filters.mul_(10.0)
            for bn in (m.conv1[1], m.mp_conv.bn, m.conv2[1]):
                bn.running_mean.normal_(0, 0.2, generator=None)
                bn.running_var.uniform_(0.5, 1.5)
    x1 = torch.randn(B, 1, 1, nin, generator=g).bfloat16().to(dev).permute(0, 3, 1, 2)
    idx1 = torch.zeros(B, M, 1, dtype=torch.int64, device=dev)
    et1 = torch.ones(B, 1, M, 1, device=dev, dtype=torch.bfloat16)
    xf = torch.randn(B, 48, 1, nin, generator=g).bfloat16().to(dev).permute(0, 3, 1, 2)
    idx = torch.randint(0, 48, (1, M, 3), generator=g).to(dev).expand(B, -1, -1)
    et = torch.randn(B, M, 3, 4, generator=g).bfloat16().to(dev).permute(0, 3, 1, 2)
    other = torch.randn(B, M, 1, nout, generator=g).bfloat16().to(dev).permute(0, 3, 1, 2)
    with torch.no_grad():
        h = hyper(x1, idx1, et1)
        # (the one-row form has a kernel of its own since round 6: 16 samples per wave on the matrix cores, csrc/mpconv_block_fwd.hip
        # mpconv_block_rows1_kernel — the same products and rounding points in another summation order)
        import os
        one_row = 'mpconv_block_fanout_kernel' if os.environ.get('FGNN_NO_BLOCK_ROWS1') else 'mpconv_block_rows1_kernel'      # (the A/B switch)
        assert one_row in _hip.lib().fgnn_last_kernel().decode(), _hip.lib().fgnn_last_kernel()
        assert getattr(h, '_fgnn_bcast_src', None) is not None and h.shape == (B, nout, M, 1) and h.stride(2) == 0
        monkeypatch.setattr(ops, 'FANOUT_BROADCAST', False)
        h_full = hyper(x1, idx1, et1)
        assert 'mpconv_block_fanout_kernel' in _hip.lib().fgnn_last_kernel().decode()
        monkeypatch.setattr(ops, 'FANOUT_BROADCAST', True)
        assert getattr(h_full, '_fgnn_bcast_src', None) is None and h_full.stride(2) != 0
        assert H.rel_err(h.contiguous().float(), h_full.contiguous().float()) <= 2.0 ** -7       # a bf16 rounding flips here and there
        differ = float((h.contiguous() != h_full.contiguous()).float().mean())
        assert differ <= 0.02, differ
        h_rows = h_full[:, :, :1, :].expand(-1, -1, M, -1)                                        # the materialised rows ARE identical rows
        assert torch.equal(h_full.contiguous(), h_rows.contiguous())
        bsrc = h._fgnn_bcast_src
        y_row = parity(xf, idx, et, addend=[other, h])
        assert 'mpconv_block_fwd_kernel' in _hip.lib().fgnn_last_kernel().decode()
        y_full = parity(xf, idx, et, addend=[other, h.contiguous(memory_format=torch.channels_last)])      # the same values, materialised
        assert bsrc is not None and torch.equal(y_row, y_full)
        y_first = parity(xf, idx, et, addend=[h, other])         # the row addend in the first slot
        assert H.rel_err(y_first.float(), y_full.float()) <= 2.0 ** -7      # (other summation order of the two addends)
        y_ref = parity(xf, idx, et, addend=[other, h_full])      # against the wave-per-sample kernel's rows
        assert H.rel_err(y_row.float(), y_ref.float()) <= 2.0 ** -7


@pytest.mark.parametrize('case', [(64, 64, 96, 48, 6, 4), (128, 256, 96, 48, 6, 4), (256, 128, 48, 96, 3, 4), (64, 64, 96, 1, 96, 1)],
                         ids=lambda c: 'x'.join(map(str, c)))
@pytest.mark.parametrize('B', [40, 171])
def test_conv2_weight_gradient_from_the_reduce_pass_moments(case, B, dev, monkeypatch):
    """Round 6: conv2's weight gradient as a closed form of three moments the tail's reduce pass accumulates (csrc/block_tail.hip:
    gW2 = diag(s3) M1 + diag(A) (W2b Gram) + K v^T) against the kernel form (gz3 and a2 stored, fgnn_linear_wgrad over them) on the same
    block: every other gradient is bit-identical (the moments do not touch them), conv2.weight agrees to the kernel form's own
    rounding (it contracts a bf16-rounded gz3), conv2.bias receives nothing (its gradient is identically zero in front of a
    batch-statistics BatchNorm; the kernel form adds rounding noise), and a second run is bit-identical."""
    from fgnn_amd.mpnn import blocks, mp_conv_residual, mp_conv_type
    nin, nout, N, M, k, net = case
    g = torch.Generator().manual_seed(nin + nout + M + B)
    torch.manual_seed(5)
    m = mp_conv_residual(nin, 64, net, extension=mp_conv_type.NO_EXTENSION, with_residual=False, aggregator='max',
                         nout=None if nout == nin else nout).to(dev).train()
    with torch.no_grad():
        m.mp_conv.filters.mul_(10.0)
    x = torch.randn(B, N, 1, nin, generator=g).bfloat16().to(dev).permute(0, 3, 1, 2)
    idx = (torch.arange(N).reshape(1, 1, N) if M == 1 else torch.randint(0, N, (1, M, k), generator=g)).to(dev).expand(B, -1, -1)
    et = torch.randn(B, M, k, net, generator=g).bfloat16().to(dev).permute(0, 3, 1, 2)
    gy = torch.randn(B, M, 1, nout, generator=g).bfloat16().to(dev).permute(0, 3, 1, 2)
    sd0 = {k_: v.clone() for k_, v in m.state_dict().items()}

    def run(moments):
        monkeypatch.setattr(blocks, 'TAIL_WGRAD_MOMENTS', moments)
        m.load_state_dict(sd0)
        for q in m.parameters():
            q.grad = None
        xd, ed = x.detach().requires_grad_(True), et.detach().requires_grad_(True)
        with torch.autocast('cuda', dtype=torch.bfloat16):
            y = m(xd, idx, ed)
        y.backward(gy)
        out = {'y': y.detach().clone(), 'gx': xd.grad.clone(), 'get': ed.grad.clone()}
        out.update({n: q.grad.detach().clone() for n, q in m.named_parameters()})
        return out
    a, b, a2 = run(True), run(False), run(True)
    for n in a:
        assert torch.equal(a[n], a2[n]), n                     # bit-reproducible
        if n not in ('conv2.0.weight', 'conv2.0.bias'):
            assert torch.equal(a[n], b[n]), n                  # untouched by the moments
    gw_m, gw_k = a['conv2.0.weight'].float(), b['conv2.0.weight'].float()
    scale = float(gw_k.abs().max())
    assert scale > 0 and float((gw_m - gw_k).abs().max()) <= 2e-2 * scale, (float((gw_m - gw_k).abs().max()), scale)
    assert float(a['conv2.0.bias'].abs().max()) == 0.0
    assert float(b['conv2.0.bias'].abs().max()) <= 1e-2 * scale + 1e-3        # (the kernel form's bias gradient: rounding noise around zero)
