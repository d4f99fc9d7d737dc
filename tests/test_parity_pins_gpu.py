"""GPU parity pins that tie the BENCHED configuration and every fused kernel to the f32 oracle / the real
reference's golden vectors ("what's weak" 1-6):

  * the bf16 whole-model path bench.py times (LDPCModel under autocast, hipGraph-less here) against the f32 oracle on
    a pick of the 4096-codeword data-path batch — eval AND train-mode forward — with SURVEY §8d's config-3 criterion:
    logits within 2e-2 of the logit range and hard decisions agreeing on >= 99.9 % of the bits;
  * the 12 `mp_conv_residual` vectors of tests/golden/block.npz (written by the REAL reference) on the HIP path,
    forward and backward;
  * whole-model gradients against the oracle's autograd, per parameter (cosine similarity);
  * callable / None aggregators against the oracle.
"""
import pytest
import torch

import fgnn_oracle as O
import helpers as H

pytestmark = pytest.mark.gpu


def _trained_like_ldpc(dev, seed=3, warm_batches=3):
    """LDPCModel with its constructor's own initialisation (the reference's U(-.01,.01) filters etc.) and BatchNorm
    running statistics populated by a few training-mode forward passes on data-path batches."""
    import fgnn_amd
    from fgnn_amd.datapath import LdpcDataPath
    torch.manual_seed(seed)
    m = fgnn_amd.LDPCModel(2, 6, 4, aggregator='max').to(dev).train()
    dp = LdpcDataPath(dev)
    with torch.no_grad():
        for i in range(warm_batches):
            m(*dp.sample(256, seed=50 + i)[:6])
    return m, dp


def _decision_agreement(a, b, margin):
    """Fraction of hard decisions (logit > 0) on which a and b agree, ignoring bits the reference itself decides with
    |logit| < margin (a coin flip for ANY finite-precision implementation)."""
    a, b = a.float().cpu(), b.float().cpu()
    firm = b.abs() >= margin
    return float(((a > 0) == (b > 0))[firm].float().mean()), float(firm.float().mean())


@pytest.mark.parametrize('mode', ['eval', 'eval_per_block_kernels', 'train', 'train_benched_batch'])
def test_benched_bf16_model_vs_f32_oracle(mode, dev):
    """The configuration bench.py measures — bf16 activations / messages / matrix cores, f32 parameters — against the
    f32 ORACLE (reference op order) on the same inputs and parameters.
    eval: the full 4096-codeword data-path batch runs on the GPU (one-kernel 64-wide layers, one-kernel inference blocks
    and all; `eval_per_block_kernels`: the same with the one-kernel layers switched off); 96 codewords picked from it are
    decoded by the oracle.  train: BatchNorm uses batch statistics, so the same 128 picked codewords
    form the batch on both sides (and the oracle's updated running statistics are compared as well); `train_benched_batch`: the
    same with ALL 4096 codewords as the batch on both sides — the batch bench.py times (the oracle's forward takes ~10 s of host
    time there).
    Criterion: max |logit error| <= 2e-2 of the logit range; hard decisions agree on >= 99.9 % of
    the bits the oracle decides firmly (|logit| >= 2e-2 of the range)."""
    from fgnn_amd.mpnn import assemblies
    per_block = mode == 'eval_per_block_kernels'
    if per_block:
        mode = 'eval'
    full = mode == 'train_benched_batch'
    if full:
        mode = 'train'
    m, dp = _trained_like_ldpc(dev)
    B = 4096
    data = dp.sample(B, seed=12, dtype=torch.bfloat16)[:6]
    data32 = dp.sample(B, seed=12, dtype=torch.float32)[:6]
    # the bf16 feature tensors are the f32 ones rounded once (same draws)
    assert H.rel_err(data[0].float(), data32[0]) <= 2.0 ** -8
    npick = 96 if mode == 'eval' else (B if full else 128)
    pick = torch.randperm(B, generator=torch.Generator().manual_seed(5))[:npick].to(dev)
    amp = torch.autocast('cuda', dtype=torch.bfloat16)
    sd = {k: v.detach().cpu().clone() for k, v in m.state_dict().items()}
    if mode == 'eval':
        m.eval()
        assemblies.FUSE_EVAL_LAYERS = not per_block
        try:
            with torch.no_grad(), amp:
                logits, snr = m(*data)
        finally:
            assemblies.FUSE_EVAL_LAYERS = True
        logits, snr = logits[pick], snr[pick]
    else:
        m.train()
        with torch.no_grad(), amp:
            logits, snr = m(*[t[pick] for t in data])
    # the oracle sees what the GPU saw: the bf16-rounded inputs, in f32 arithmetic
    o_in = [t[pick].cpu().contiguous() for t in data]
    o_in = [t.float() if t.is_floating_point() else t for t in o_in]
    with torch.no_grad():
        lo, so = O.ldpc_model(sd, *o_in, training=(mode == 'train'))
    rng = float(lo.abs().max())
    err = float((logits.float().cpu() - lo).abs().max()) / rng
    agree, firm = _decision_agreement(logits, lo, 2e-2 * rng)
    serr = float((snr.float().cpu() - so).abs().max()) / max(1.0, float(so.abs().max()))
    print('bf16 LDPCModel %s vs f32 oracle: logit err %.3e of range %.3g, decisions agree %.4f (firm bits %.3f), '
          'snr head err %.2e' % (mode, err, rng, agree, firm, serr))
    assert rng > 0.05                                     # a live output, not a vanished one
    assert err <= 2e-2, err
    assert agree >= 0.999, agree
    assert serr <= 2e-2, serr
    if mode == 'train':
        # the train-mode forward also moved the BatchNorm running statistics: same update on both sides
        worst = 0.0
        for k, v in m.state_dict().items():
            if k.endswith('running_var') or k.endswith('running_mean'):
                worst = max(worst, float((v.float().cpu() - sd[k]).abs().max()) / max(1.0, float(sd[k].abs().max())))
        print('bf16 train-mode running statistics vs oracle: worst %.2e' % worst)
        assert worst <= 2e-2, worst


def _block_cases():
    z = H.load('block.npz')
    return [tuple(int(v) for v in row) for row in z['meta']]


@pytest.mark.parametrize('case', _block_cases(), ids=lambda c: 'b%02d_ext%d_res%d_nout%d_%s' % (
    c[0], c[1], c[2], c[3], 'train' if c[4] else 'eval'))
def test_residual_block_matches_reference_golden(case, dev):
    """`mp_conv_residual` (mp_nn_residual.py:39-56) on the HIP path against the 12 vectors the REAL reference wrote:
    both `with_residual` settings, the `nout` override, NO_EXTENSION and DIFF, train and eval; output and the
    gradients w.r.t. x, etype, the operator's filters and conv1's weight."""
    from fgnn_amd.mpnn import mp_conv_residual, mp_conv_type
    cid, ext, with_res, nout, train, nin, nmed, net = case
    z = H.load('block.npz')
    pre = 'b%02d.' % cid
    g = lambda n: torch.from_numpy(z[pre + n])
    sd = {k[len(pre) + 3:]: torch.from_numpy(z[k]).clone() for k in z.files if k.startswith(pre + 'sd.')}
    m = mp_conv_residual(nin, nmed, net, extension=mp_conv_type(ext), with_residual=bool(with_res), aggregator='max',
                         nout=None if nout < 0 else nout)
    m.load_state_dict(sd)
    m = m.to(dev).train(bool(train))
    x = g('x').to(dev).requires_grad_(True)
    et = g('etype').to(dev).requires_grad_(True)
    y = m(x, g('idx').to(dev), et)
    # train mode: BatchNorm statistics over 24 values; the vectors' own conditioning (f32 oracle vs f64 oracle)
    tol = 1e-4
    if train:
        d = lambda v: v.double() if torch.is_tensor(v) and v.is_floating_point() else v
        with torch.no_grad():
            y32 = O.residual_block({k: v.clone() for k, v in sd.items()}, '', g('x'), g('idx'), g('etype'), net=net,
                                   extension=ext, aggregator='max', with_residual=bool(with_res), training=True)
            y64 = O.residual_block({k: d(v).clone() for k, v in sd.items()}, '', d(g('x')), g('idx'), d(g('etype')), net=net,
                                   extension=ext, aggregator='max', with_residual=bool(with_res), training=True)
        tol = max(1e-4, 8.0 * H.rel_err(y32, y64))
    assert H.rel_err(y, g('y')) <= tol
    y.backward(g('gy').to(dev))
    gtol = 10 * tol if train else 1e-4
    assert H.rel_err(x.grad, g('gx')) <= gtol
    assert H.rel_err(et.grad, g('getype')) <= gtol
    assert H.rel_err(m.mp_conv.filters.grad, g('gfilters')) <= gtol
    assert H.rel_err(m.conv1[0].weight.grad, g('gconv1')) <= gtol


@pytest.mark.parametrize('bn_mode', ['eval_stats', 'batch_stats'])
def test_whole_model_gradients_vs_oracle_autograd(bn_mode, dev):
    """LDPCModel, f32, B = 16, closed-form parameters: every parameter's gradient from the hand-written backward
    kernels against the ORACLE's autograd (reference op order on the CPU) — cosine similarity per parameter tensor and
    the relative error of the whole flattened gradient.  `eval_stats`: BatchNorm normalises with its running statistics
    (a fixed affine: the chain of message operators, node-wise maps, InstanceNorms and edge MLPs is compared with
    nothing amplifying rounding); `batch_stats`: training mode proper."""
    import fgnn_amd
    z = H.load('ldpc_model.npz')
    m = fgnn_amd.LDPCModel(2, 6, 4, aggregator='max')
    m.load_state_dict(H.fill_state_dict(m.state_dict()))
    m = m.to(dev)
    m.train(bn_mode == 'batch_stats')
    inputs = [torch.from_numpy(z['tin%d' % i]) for i in range(6)]
    tgt = (torch.arange(16 * 48).reshape(16, 48) % 3 == 0).float()

    def loss_of(logits, snr):
        return torch.nn.functional.binary_cross_entropy_with_logits(logits.reshape(-1), tgt.to(logits.device).reshape(-1)) \
            + 0.1 * torch.nn.functional.mse_loss(snr.reshape(-1), torch.ones(16, device=snr.device))

    sd = {k: v.detach().cpu().clone() for k, v in m.state_dict().items()}
    names = [n for n, p in m.named_parameters() if p.requires_grad]
    for n in names:
        sd[n].requires_grad_(True)
    loss_of(*O.ldpc_model(sd, *inputs, training=(bn_mode == 'batch_stats'))).backward()
    loss_of(*m(*[t.to(dev) for t in inputs])).backward()
    got = dict(m.named_parameters())
    worst_cos, worst_name, flat_g, flat_o = 1.0, '', [], []
    gmax = max(float(sd[n].grad.abs().max()) for n in names if sd[n].grad is not None)
    for n in names:
        go = sd[n].grad
        gg = got[n].grad
        if go is None:
            assert gg is None or float(gg.abs().max()) == 0.0, n
            continue
        assert gg is not None, n
        a, b = gg.detach().double().cpu().reshape(-1), go.double().reshape(-1)
        flat_g.append(a), flat_o.append(b)
        if float(b.abs().max()) < 1e-6 * gmax:            # numerically-zero gradient (dead branch): direction undefined
            assert float(a.abs().max()) < 1e-4 * gmax, n
            continue
        cos = float(torch.dot(a, b) / (a.norm() * b.norm()))
        if cos < worst_cos:
            worst_cos, worst_name = cos, n
    a, b = torch.cat(flat_g), torch.cat(flat_o)
    rel = float((a - b).norm() / b.norm())
    print('LDPCModel gradients vs oracle autograd (%s): worst per-parameter cosine %.6f (%s), whole-gradient relative '
          'error %.2e' % (bn_mode, worst_cos, worst_name, rel))
    assert worst_cos >= 0.999, (worst_cos, worst_name)
    assert rel <= 1e-2, rel


@pytest.mark.parametrize('bn_mode', ['eval_stats', 'batch_stats'])
def test_benched_bf16_whole_model_gradients_vs_oracle_autograd(bn_mode, dev):
    """The bf16 twin of the test above — what bench.py TIMES: LDPCModel under bf16 autocast (bf16 activations, messages and
    matrix cores, the second-generation backward kernels, bf16 weight-gradient kernels), forward + backward on 128 data-path
    codewords, against the f32 ORACLE's autograd on the same bf16-rounded inputs and the same parameters.

    What bf16 itself costs on this model is MEASURED, not assumed: the same oracle run under torch's CPU bf16 autocast (bf16
    matmuls / convolutions, f32 elsewhere — an independent bf16 implementation of the reference's op order) is 19 % (running
    statistics) / 42 % (batch statistics) away from its own f32 gradient — eight max-routed layers re-route near-ties and
    BatchNorm rescales the rounding (measured on the GPU box, tools/diag_bf16_grad.py; VERDICT r02's 5e-2 / cosine 0.99 is not
    attainable by any bf16 arithmetic here).  The HIP path must be no further from the f32 oracle than that noise floor:
    whole-gradient relative error <= 1.15 x the CPU-bf16 oracle's + 1e-2, cosine >= its cosine - 1e-2, and the same per
    parameter group for every group that carries >= 1 % of the gradient norm.  (The kernels themselves are pinned tightly —
    2^-6 against the oracle's autograd with near-ties masked — in test_mpconv_sg_gpu.py::test_bf16_backward_vs_oracle_autograd;
    the f32 path at 1e-2 / 0.999 above.)"""
    import re
    m, dp = _trained_like_ldpc(dev)
    B = 128
    data = dp.sample(B, seed=31, dtype=torch.bfloat16)
    inputs = data[:6]
    label = data[6][:, :48].float().contiguous()
    train = bn_mode == 'batch_stats'
    m.train(train)

    def loss_of(logits, snr, label):
        return torch.nn.functional.binary_cross_entropy_with_logits(logits.float().reshape(-1), label.reshape(-1)) \
            + 0.1 * torch.nn.functional.mse_loss(snr.float().reshape(-1), torch.ones(B, device=snr.device))

    sd0 = {k: v.detach().cpu().clone() for k, v in m.state_dict().items()}
    names = [n for n, p in m.named_parameters() if p.requires_grad]
    o_in = [t.cpu().contiguous() for t in inputs]
    o_in = [t.float() if t.is_floating_point() else t for t in o_in]

    def oracle(autocast):
        sd = {k: v.clone() for k, v in sd0.items()}
        for n in names:
            sd[n].requires_grad_(True)
        with torch.autocast('cpu', dtype=torch.bfloat16, enabled=autocast):
            out = O.ldpc_model(sd, *o_in, training=train)
        loss_of(*out, label.cpu()).backward()
        return {n: sd[n].grad.double() for n in names if sd[n].grad is not None}

    ref, floor = oracle(False), oracle(True)
    with torch.autocast('cuda', dtype=torch.bfloat16):
        logits, snr = m(*inputs)
    loss_of(logits, snr, label).backward()
    got = {n: p.grad.detach().double().cpu() for n, p in m.named_parameters() if p.grad is not None}
    live = [n for n in names if n in ref]
    assert all(n in got for n in live)

    def dist(g, keys):
        a = torch.cat([g[n].reshape(-1) for n in keys])
        b = torch.cat([ref[n].reshape(-1) for n in keys])
        return float((a - b).norm() / b.norm()), float(torch.dot(a, b) / (a.norm() * b.norm()))

    rel_hip, cos_hip = dist(got, live)
    rel_floor, cos_floor = dist(floor, live)
    print('bf16 LDPCModel gradients vs f32 oracle autograd (%s, 128 codewords): HIP rel err %.3e cosine %.5f; the oracle under '
          'CPU bf16 autocast: %.3e / %.5f' % (bn_mode, rel_hip, cos_hip, rel_floor, cos_floor))
    assert rel_hip <= 1.15 * rel_floor + 1e-2, (rel_hip, rel_floor)
    assert cos_hip >= cos_floor - 1e-2, (cos_hip, cos_floor)
    total = float(torch.cat([ref[n].reshape(-1) for n in live]).norm())
    groups = {}
    for n in live:
        mm = re.match(r'main\.(\w+?_\d(?:_\d)?)\.', n)
        groups.setdefault(mm.group(1) if mm else n.rsplit('.', 1)[0], []).append(n)
    checked = 0
    for key, keys in sorted(groups.items()):
        if float(torch.cat([ref[n].reshape(-1) for n in keys]).norm()) < 1e-2 * total:
            continue
        rh, ch = dist(got, keys)
        rf, cf = dist(floor, keys)
        assert rh <= 1.5 * rf + 3e-2 and ch >= cf - 1e-1, (key, rh, rf, ch, cf)     # (per group the two bf16 realisations scatter: measured 0.77 vs 0.82 on v2f_3_1)
        checked += 1
    assert checked >= 8


@pytest.mark.parametrize('bn_mode', ['eval_stats', 'batch_stats'])
def test_benched_bf16_whole_model_gradients_routing_forced(bn_mode, dev, monkeypatch):
    """The un-forced comparison above is loose (19 % / 41 %); the suspicion was RE-ROUTING: eight max-routed layers deep, a
    near-tie decided differently by bf16 sends the gradient down another edge, and the f32 oracle then differentiates a different
    piecewise-linear function.  This test removes routing from the comparison: every operator's argmax is taken from the HIP
    forward (ops.ROUTE_TAP) and IMPOSED on the oracle (`aggregator` = gather along the recorded index, mp_nn.py:71-75,160-175),
    so both sides differentiate the SAME routes.

    Measured (r04, printed below): with the routes forced the HIP gradient is 18.9 % / ~40 % from the f32 oracle — exactly where
    it was un-forced (18.8 % / 40.5 %).  Re-routing is NOT what the distance is made of: it is the bf16 rounding of activations
    and upstream gradients, amplified by this model's conditioning (the f32 HIP path sits at 1e-2 / cosine 0.999 of the same
    oracle, test above; every bf16 kernel at 2^-6 of the oracle's autograd, test_mpconv_sg_gpu.py).  So the forced-route
    comparison cannot be a 3e-2 pin either.  What it CAN pin: an independent bf16 implementation of the same chain along the same
    routes — the oracle under torch's CPU bf16 autocast, routes forced — must not be closer to the f32 result than the HIP path
    by more than 15 %, in total and per parameter group; i.e. the HIP chain adds no error of its own beyond what bf16 costs."""
    import re
    from fgnn_amd import ops
    m, dp = _trained_like_ldpc(dev)
    B = 128
    data = dp.sample(B, seed=31, dtype=torch.bfloat16)
    inputs = data[:6]
    label = data[6][:, :48].float().contiguous()
    train = bn_mode == 'batch_stats'
    m.train(train)

    def loss_of(logits, snr, label):
        return torch.nn.functional.binary_cross_entropy_with_logits(logits.float().reshape(-1), label.reshape(-1)) \
            + 0.1 * torch.nn.functional.mse_loss(snr.float().reshape(-1), torch.ones(B, device=snr.device))

    sd0 = {k: v.detach().cpu().clone() for k, v in m.state_dict().items()}
    names = [n for n, p in m.named_parameters() if p.requires_grad]
    by_ptr = {p.data_ptr(): n for n, p in m.named_parameters()}
    routes = {}

    def tap(filters, amax):
        n = by_ptr[filters.data_ptr()]
        assert n.endswith('filters') and n not in routes, n
        routes[n[:-len('filters')]] = amax.detach().cpu().long()

    monkeypatch.setattr(ops, 'ROUTE_TAP', tap)
    with torch.autocast('cuda', dtype=torch.bfloat16):
        logits, snr = m(*inputs)
    monkeypatch.setattr(ops, 'ROUTE_TAP', None)
    loss_of(logits, snr, label).backward()
    got = {n: p.grad.detach().double().cpu() for n, p in m.named_parameters() if p.grad is not None}
    assert len(routes) == 32, sorted(routes)                               # every VF / FV operator of the 8 layers

    orig = O.mp_conv
    used = set()

    def forced(sd, prefix, x, nn_idx, etype, *, aggregator, **kw):
        if prefix in routes:
            used.add(prefix)
            idx = routes[prefix]
            return orig(sd, prefix, x, nn_idx, etype, aggregator=lambda e: e.gather(3, idx), **kw)
        return orig(sd, prefix, x, nn_idx, etype, aggregator=aggregator, **kw)

    monkeypatch.setattr(O, 'mp_conv', forced)
    o_in = [t.cpu().contiguous() for t in inputs]
    o_in = [t.float() if t.is_floating_point() else t for t in o_in]

    def oracle(autocast):
        sd = {k: v.clone() for k, v in sd0.items()}
        for n in names:
            sd[n].requires_grad_(True)
        used.clear()
        with torch.autocast('cpu', dtype=torch.bfloat16, enabled=autocast):
            out = O.ldpc_model(sd, *o_in, training=train)
        loss_of(*out, label.cpu()).backward()
        assert used == set(routes)
        return {n: sd[n].grad.double() for n in names if sd[n].grad is not None}

    ref, floor = oracle(False), oracle(True)
    live = [n for n in names if n in ref]

    def dist(g, keys):
        a = torch.cat([g[n].reshape(-1) for n in keys])
        b = torch.cat([ref[n].reshape(-1) for n in keys])
        return float((a - b).norm() / b.norm()), float(torch.dot(a, b) / (a.norm() * b.norm()))

    rel, cos = dist(got, live)
    rel_f, cos_f = dist(floor, live)
    print('bf16 LDPCModel gradients vs the f32 oracle along the SAME routes (%s, 128 codewords): HIP rel err %.3e cosine %.6f; '
          'the oracle under CPU bf16 autocast, same routes: %.3e / %.6f' % (bn_mode, rel, cos, rel_f, cos_f))
    assert rel <= 1.15 * rel_f + 1e-2, (rel, rel_f)
    assert cos >= cos_f - 1e-2, (cos, cos_f)
    total = float(torch.cat([ref[n].reshape(-1) for n in live]).norm())
    groups = {}
    for n in live:
        mm = re.match(r'main\.(\w+?_\d(?:_\d)?)\.', n)
        groups.setdefault(mm.group(1) if mm else n.rsplit('.', 1)[0], []).append(n)
    checked = 0
    for key, keys in sorted(groups.items()):
        if float(torch.cat([ref[n].reshape(-1) for n in keys]).norm()) < 1e-2 * total:
            continue
        rh, ch = dist(got, keys)
        rf, cf = dist(floor, keys)
        assert rh <= 1.5 * rf + 3e-2 and ch >= cf - 1e-1, (key, rh, rf, ch, cf)     # (per group the two bf16 realisations scatter: measured 0.77 vs 0.82 on v2f_3_1)
        checked += 1
    assert checked >= 8


@pytest.mark.parametrize('ext', [0, 1, 2])
@pytest.mark.parametrize('agg', ['none', 'sum', 'topk'])
def test_callable_and_none_aggregators_vs_oracle(ext, agg, dev):
    """mp_nn.py:89-90,162-163: a user callable (or None = no aggregation) instead of the string aggregators.  The
    per-edge messages come from the HIP operator (`mp_conv_v2.edge_messages`), forward and backward vs the oracle."""
    from fgnn_amd.mpnn import mp_conv_type, mp_conv_v2
    fn = {'none': None, 'sum': lambda e: e.sum(dim=3, keepdim=True),
          'topk': lambda e: e.topk(2, dim=3)[0].mean(dim=3, keepdim=True)}[agg]
    B, nin, nou, net, N, k = 3, 6, 5, 3, 9, 4
    M = N if ext else 7
    g = torch.Generator().manual_seed(17 + ext)
    m = mp_conv_v2(nin, nou, net, bn=False, extension=mp_conv_type(ext), aggregtor=fn)
    with torch.no_grad():
        m.filters.copy_(torch.randn(m.filters.shape, generator=g) * 0.3)
    sd = {k_: v.detach().clone().requires_grad_(True) for k_, v in m.state_dict().items()}
    m = m.to(dev)
    x = torch.randn(B, nin, N, 1, generator=g)
    idx = torch.randint(0, N, (B, M, k), generator=g)
    et = torch.randn(B, net, M, k, generator=g)
    xo, eo = x.clone().requires_grad_(True), et.clone().requires_grad_(True)
    ref = O.mp_conv(sd, '', xo, idx, eo, nou=nou, net=net, extension=ext, aggregator=fn, relu=True)
    xd, ed = x.to(dev).requires_grad_(True), et.to(dev).requires_grad_(True)
    y = m(xd, idx.to(dev), ed)
    assert y.shape == ref.shape == (B, nou, M, k if fn is None else 1)
    assert H.rel_err(y, ref) <= 2e-5
    gy = torch.randn(ref.shape, generator=g)
    ref.backward(gy)
    y.backward(gy.to(dev))
    assert H.rel_err(xd.grad, xo.grad) <= 1e-4 and H.rel_err(ed.grad, eo.grad) <= 1e-4
    assert H.rel_err(m.filters.grad, sd['filters'].grad) <= 1e-4 and H.rel_err(m.bias.grad, sd['bias'].grad) <= 1e-4


def test_index_range_check_is_available(dev):
    """The kernels clamp neighbour ids (never fault); with the debug check on, a bad table raises like torch.gather."""
    from fgnn_amd import _hip, ops
    x = torch.randn(2, 4, 5, 1, device=dev)
    idx = torch.tensor([[[0, 7]], [[1, 2]]], device=dev)
    et = torch.ones(2, 1, 1, 2, device=dev)
    W = torch.randn(4, 3, device=dev)
    y, _ = ops.mpconv_forward_raw(x, idx, et, W, None, 3, 1, 0, _hip.AGG_MAX)        # clamped to node 4
    assert torch.isfinite(y).all()
    ops.CHECK_INDICES = True
    try:
        with pytest.raises(IndexError):
            ops.mpconv_forward_raw(x, idx, et, W, None, 3, 1, 0, _hip.AGG_MAX)
        ops.mpconv_forward_raw(x, idx.clamp(max=4), et, W, None, 3, 1, 0, _hip.AGG_MAX)
    finally:
        ops.CHECK_INDICES = False


def test_repeated_tables_take_the_shared_graph_path(dev):
    """The reference passes B copies of one neighbour table (`.repeat(B,1,1)` / DataLoader collation).  A content check
    (once per table) turns them into the batch-shared form the fast kernels want; tables that really differ per sample
    are left alone; results are bitwise those of an `expand`-ed table."""
    from fgnn_amd import _hip, ops
    B, nin, nou, net, N, M, k = 24, 64, 64, 4, 96, 48, 6
    g = torch.Generator().manual_seed(2)
    x = torch.randn(B, N, 1, nin, generator=g).bfloat16().to(dev).permute(0, 3, 1, 2)
    et = torch.randn(B, M, k, net, generator=g).bfloat16().to(dev).permute(0, 3, 1, 2)
    W = (torch.randn(nin, nou * net, generator=g) * 0.1).to(dev)
    one = torch.randint(0, N, (1, M, k), generator=g).to(dev)
    rep = one.repeat(B, 1, 1)
    view = ops.shared_graph_view(rep)
    assert view.stride(0) == 0 and view.shape == rep.shape and torch.equal(view, rep)
    assert ops.shared_graph_view(rep).data_ptr() == view.data_ptr()                 # remembered, not re-checked
    differ = rep.clone()
    differ[B // 2, 3, 1] = (differ[B // 2, 3, 1] + 1) % N
    assert ops.shared_graph_view(differ) is differ
    rep[5, 0, 0] = (rep[5, 0, 0] + 1) % N                                           # in-place edit: version bump, re-checked
    assert ops.shared_graph_view(rep) is rep
    rep = one.repeat(B, 1, 1)
    outs = []
    for idx in (rep, one.expand(B, -1, -1)):
        xd, ed = x.detach().requires_grad_(True), et.detach().requires_grad_(True)
        Wd = W.detach().requires_grad_(True)
        z = ops.mpconv(xd, idx, ed, Wd, None, nou, net, 0, _hip.AGG_MAX)
        z.backward(torch.ones_like(z))
        outs.append((z.detach(), xd.grad, ed.grad, Wd.grad))
    for a, b in zip(*outs):
        assert torch.equal(a, b)


def test_fused_flat_adam_matches_torch_adam(dev):
    """dp.FlatAdam's single kernel (csrc/flat_adam.hip) against torch.optim.Adam on the same parameters and gradients,
    weight decay and the folded gradient scale included, over several steps (bias corrections change every step)."""
    from fgnn_amd.dp import FlatAdam, FlatGradBucket
    torch.manual_seed(0)
    make = lambda: torch.nn.Sequential(torch.nn.Linear(37, 53), torch.nn.Tanh(), torch.nn.Linear(53, 3)).to(dev)
    a = make()
    b = make()
    b.load_state_dict(a.state_dict())
    ref = torch.optim.Adam(a.parameters(), lr=3e-3, weight_decay=1e-2)
    bucket = FlatGradBucket(b.parameters(), flatten_params=True)
    opt = FlatAdam(bucket, lr=3e-3, weight_decay=1e-2)
    x, y = torch.randn(64, 37, device=dev), torch.randn(64, 3, device=dev)
    for _ in range(7):
        ref.zero_grad()
        torch.nn.functional.mse_loss(a(x), y).backward()
        ref.step()
        bucket.zero()
        (2.0 * torch.nn.functional.mse_loss(b(x), y)).backward()          # twice the gradient ...
        opt.step(grad_scale=0.5)                                          # ... halved inside the kernel
    for pa, pb in zip(a.parameters(), b.parameters()):
        assert torch.allclose(pa, pb, rtol=2e-5, atol=2e-7), float((pa - pb).abs().max())
