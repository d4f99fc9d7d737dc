"""GPU: the reference's model bodies (LDPCModel/FactorNN, factor_mpnn, mp_sequential config 1)
built from fgnn_amd's drop-in classes with closed-form parameters, against the golden outputs of
the REAL reference (eval: <= 1e-4, the north-star tolerance; train: conditioning-limited, see
tests/test_oracle_golden.py) and against the CPU oracle."""
import numpy as np
import pytest
import torch

import fgnn_oracle as O
import helpers as H

pytestmark = pytest.mark.gpu


def _tol(z, mode):
    """eval: the north-star 1e-4, flat — every fixture's own f32-vs-f64 distance (`eval_cond`) is <= 2e-6.
    train: batch-statistics BatchNorm amplifies f32 rounding by up to 1e4 on these small fixtures (a 1e-7 input
    perturbation moves the REFERENCE's own output by ~1e-3); the fixture stores how far the reference's f32 result is
    from an f64 run of the same maths (`train_cond`) and the HIP path must stay within 8x of that."""
    if mode == 'eval':
        return 1e-4
    return max(1e-4, 8.0 * float(z['train_cond']))


def _block_vs_oracle(blk, x, idx, et, add, y):
    """A fused one-kernel inference block against the f32 ORACLE (`O.residual_block`, reference op order) on the same
    bf16-rounded inputs and the module's f32 parameters: what differs is the bf16 rounding of the three weight
    matrices and of the intermediates the kernel keeps in LDS -> 2^-5 of the output range."""
    sd = {k: v.detach().float().cpu() for k, v in blk.state_dict().items()}
    with torch.no_grad():
        ref = O.residual_block(sd, '', x.float().cpu().contiguous(), idx.cpu().contiguous(), et.float().cpu().contiguous(),
                               net=blk.mp_conv.nedge_types, extension=0, aggregator='max', with_residual=False,
                               training=False)
        if add is not None:
            ref = ref + add.float().cpu()
    err = float((y.float().cpu() - ref).abs().max() / ref.abs().max())
    assert err <= 2.0 ** -5, err
    return err


def _ldpc(dev):
    import fgnn_amd
    m = fgnn_amd.LDPCModel(2, 6, 4, aggregator='max')
    m.load_state_dict(H.fill_state_dict(m.state_dict()))
    return m.to(dev)


def test_ldpc_model_eval_matches_reference(dev):
    z = H.load('ldpc_model.npz')
    m = _ldpc(dev).eval()
    inputs = [torch.from_numpy(z['in%d' % i]).to(dev) for i in range(6)]
    with torch.no_grad():
        logits, snr = m(*inputs)
    assert H.rel_err(logits, torch.from_numpy(z['eval_logits'])) <= 1e-4
    assert H.rel_err(snr, torch.from_numpy(z['eval_snr'])) <= 1e-4


def test_ldpc_model_train_step_matches_reference(dev):
    z = H.load('ldpc_model.npz')
    m = _ldpc(dev).train()
    inputs = [torch.from_numpy(z['tin%d' % i]).to(dev) for i in range(6)]
    logits, snr = m(*inputs)
    assert H.rel_err(logits, torch.from_numpy(z['train_logits'])) <= _tol(z, 'train')
    tgt = (torch.arange(16 * 48, device=dev).reshape(16, 48) % 3 == 0).float()
    loss = torch.nn.functional.binary_cross_entropy_with_logits(logits.view(-1), tgt.view(-1)) \
        + 0.1 * torch.nn.functional.mse_loss(snr.view(-1), torch.ones(16, device=dev))
    assert abs(loss.item() - float(z['train_loss'])) <= 1e-3 * max(1.0, abs(float(z['train_loss'])))
    loss.backward()
    names = [n for n, _ in sorted(m.named_parameters())]
    assert names == list(z['param_names'])
    # Whole-model gradients are not comparable across implementations at these sizes (max-argmax
    # flips + batch-stat BatchNorm: the reference's own gradient norms move by 25% between two CPU
    # processes), so gradients are pinned per operator (test_mpconv_gpu.py) and checked here for
    # self-consistency: the directional derivative along the computed gradient must match a
    # central finite difference of the loss.
    params = [p for p in m.parameters() if p.grad is not None]
    gnorm = torch.sqrt(sum((p.grad.double() ** 2).sum() for p in params)).item()
    direction = [p.grad / gnorm for p in params]

    def loss_at(eps):
        with torch.no_grad():
            for p, dvec in zip(params, direction):
                p.add_(dvec, alpha=eps)
            lg, sn = m(*inputs)
            val = torch.nn.functional.binary_cross_entropy_with_logits(lg.view(-1), tgt.view(-1)) \
                + 0.1 * torch.nn.functional.mse_loss(sn.view(-1), torch.ones(16, device=dev))
            for p, dvec in zip(params, direction):
                p.add_(dvec, alpha=-eps)
        return val.double().item()

    eps = 1e-4
    fd = (loss_at(eps) - loss_at(-eps)) / (2 * eps)
    assert abs(fd - gnorm) <= 0.1 * gnorm, (fd, gnorm)


def test_ldpc_model_eval_vs_oracle_bigger_batch(dev):
    from fgnn_amd.ldpc import synthetic_batch
    m = _ldpc(dev).eval()
    batch = synthetic_batch(32, dev, seed=3)
    with torch.no_grad():
        logits, snr = m(*batch[:6])
        sd = {k: v.cpu() for k, v in m.state_dict().items()}
        lo, so = O.ldpc_model(sd, *[t.cpu().contiguous() for t in batch[:6]], training=False)
    assert H.rel_err(logits, lo) <= 1e-4
    assert H.rel_err(snr, so) <= 1e-4


@pytest.mark.parametrize('tag', H.SYN_TAGS)
def test_factor_mpnn_matches_reference(tag, dev):
    """factor_mpnn on the synthetic-PGM tables (pairwise + chain factors; pairwise + degree-9 / degree-8 budget
    factors: BASELINE configs 2 and 5) against the REAL reference's outputs: eval at a flat 1e-4."""
    import fgnn_amd
    z = H.load('factor_mpnn_%s.npz' % tag)
    hop_dim, pw_idx, pw_ef, hi_idx, hi_ef = H.syn_setup(tag)
    model = fgnn_amd.factor_mpnn(2, [4, hop_dim], O.SYN_DIMS, [16, 16])
    em_pw, em_hi = H.syn_edge_models(hi_ef)
    model, em_pw, em_hi = model.to(dev), em_pw.to(dev), em_hi.to(dev)
    B = z['nfeature'].shape[0]
    t = lambda a: torch.from_numpy(a).to(dev)
    for mode in ('eval', 'train'):
        model.load_state_dict(H.syn_fill(model.state_dict()))
        model.train(mode == 'train')
        with torch.no_grad():
            et_pw = em_pw(t(pw_ef)[None]).expand(B, -1, -1, -1)      # the scripts .repeat(); same values
            et_hi = em_hi(t(hi_ef)[None]).expand(B, -1, -1, -1)
            gs = [[t(pw_idx)[None].expand(B, -1, -1), et_pw], [t(hi_idx)[None].expand(B, -1, -1), et_hi]]
            pred, ff = model(t(z['nfeature']), [t(z['pws']), t(z['hi_feat'])], gs)
        tol = _tol(z, mode)
        e_pred, e_ff = H.rel_err(pred, torch.from_numpy(z[mode + '_pred'])), H.rel_err(ff[1], torch.from_numpy(z[mode + '_ff1']))
        print('factor_mpnn_%s %s: pred err %.2e, factor-feature err %.2e (tolerance %.1e)' % (tag, mode, e_pred, e_ff, tol))
        assert e_pred <= tol, mode
        assert e_ff <= (tol if mode == 'eval' else 4 * tol), mode


def test_sequential_config1_matches_reference(dev):
    from fgnn_amd import tables
    from fgnn_amd.mpnn import mp_conv_residual, mp_conv_type, mp_conv_v2, mp_sequential
    z = H.load('sequential_cfg1.npz')
    C = torch.nn.Conv2d
    bnrelu = lambda c: (torch.nn.BatchNorm2d(c), torch.nn.ReLU(inplace=True))
    model = mp_sequential(
        mp_conv_v2(2, 64, 16, extension=mp_conv_type.ORIG_WITH_NEIGHBOR),
        mp_conv_residual(64, 64, 16), C(64, 128, 1), *bnrelu(128),
        mp_conv_residual(128, 64, 16), C(128, 256, 1), *bnrelu(256),
        mp_conv_residual(256, 64, 16), C(256, 128, 1), *bnrelu(128),
        mp_conv_residual(128, 64, 16), C(128, 64, 1), *bnrelu(64),
        mp_conv_residual(64, 64, 16), C(64, 2, 1))
    emodel = torch.nn.Sequential(C(1, 64, 1), torch.nn.ReLU(), C(64, 16, 1))
    emodel.load_state_dict(H.fill_state_dict(emodel.state_dict()))
    model, emodel = model.to(dev), emodel.to(dev)
    idx, ef = tables.knn_table(30, 8)
    x = torch.from_numpy(z['x']).to(dev)
    B = x.shape[0]
    for mode in ('eval', 'train'):
        model.load_state_dict(H.fill_state_dict(model.state_dict()))
        model.train(mode == 'train')
        with torch.no_grad():
            et = emodel(torch.from_numpy(ef).to(dev)[None]).repeat(B, 1, 1, 1)
            y = model(x, torch.from_numpy(idx).to(dev)[None].repeat(B, 1, 1), et)
        assert H.rel_err(y, torch.from_numpy(z[mode + '_y'])) <= _tol(z, mode), mode


@pytest.mark.parametrize('dtype', ['f32', 'bf16'])
def test_ldpc_training_reduces_loss(dtype, dev):
    """A dozen Adam steps of the full LDPCModel on one fixed synthetic batch must drive the loss down in
    both precisions (f32, and bf16 activations + bf16 matrix cores under autocast) — an end-to-end check
    that every hand-written backward (message operator, node-wise maps, norms) descends."""
    import fgnn_amd
    from fgnn_amd.dp import FlatGradBucket
    from fgnn_amd.ldpc import synthetic_batch
    torch.manual_seed(0)
    m = fgnn_amd.LDPCModel(2, 6, 4, aggregator='max').to(dev).train()
    dt = torch.float32 if dtype == 'f32' else torch.bfloat16
    data = synthetic_batch(64, dev, seed=5, dtype=dt)
    bucket = FlatGradBucket(m.parameters())
    opt = torch.optim.Adam(bucket.params, lr=2e-3)
    amp = torch.autocast(device_type='cuda', dtype=torch.bfloat16, enabled=(dtype == 'bf16'))
    losses = []
    for _ in range(12):
        bucket.zero()
        with amp:
            logits, snr = m(*data[:6])
        loss = torch.nn.functional.binary_cross_entropy_with_logits(logits.float().view(-1), data[6].view(-1))
        loss.backward()
        opt.step()
        losses.append(loss.item())
    assert all(l == l for l in losses), losses          # no NaN
    assert losses[-1] < 0.8 * losses[0], losses


@pytest.mark.parametrize('dtype', ['f32', 'bf16'])
def test_gradients_accumulated_in_place_equal_returned_gradients(dtype, dev):
    """With a pre-allocated ``param.grad`` (dp.FlatGradBucket) the weight / bias gradient kernels add
    straight into it and autograd gets None (ops.grad_sink); the result must be what ordinary returned
    gradients give — and a second backward must accumulate, exactly like AccumulateGrad."""
    import fgnn_amd
    from fgnn_amd import ops
    from fgnn_amd.dp import FlatGradBucket
    from fgnn_amd.ldpc import synthetic_batch
    torch.manual_seed(1)
    m = fgnn_amd.LDPCModel(2, 6, 4, aggregator='max').to(dev).train()
    dt = torch.float32 if dtype == 'f32' else torch.bfloat16
    data = synthetic_batch(16, dev, seed=9, dtype=dt)
    amp = torch.autocast(device_type='cuda', dtype=torch.bfloat16, enabled=(dtype == 'bf16'))
    state = {k: v.clone() for k, v in m.state_dict().items()}

    def backward():
        m.load_state_dict(state)                        # BatchNorm running stats back to the start
        with amp:
            logits, snr = m(*data[:6])
        loss = torch.nn.functional.binary_cross_entropy_with_logits(logits.float().view(-1), data[6].view(-1))
        (loss + snr.mean()).backward()

    ops.ACCUMULATE_INTO_GRAD = False
    try:
        backward()
    finally:
        ops.ACCUMULATE_INTO_GRAD = True
    want = torch.cat([(p.grad if p.grad is not None else torch.zeros_like(p)).reshape(-1)
                      for p in m.parameters() if p.requires_grad]).clone()
    for p in m.parameters():
        p.grad = None
    bucket = FlatGradBucket(m.parameters())
    backward()
    assert float(bucket.flat.abs().max()) > 0
    tol = 1e-6 if dtype == 'f32' else 1e-5             # same kernels, same data: only 0 + s vs s
    got = lambda: torch.cat([p.grad.reshape(-1) for p in bucket.params])       # (the flat buffer itself pads to 16-byte boundaries)
    assert float((got() - want).abs().max()) <= tol * float(want.abs().max())
    backward()                                          # accumulates
    assert float((got() - 2 * want).abs().max()) <= 2 * tol * float(want.abs().max())


@pytest.mark.parametrize('shape', [(96, 48, 6), (48, 96, 3), (91, 47, 6), (37, 95, 3)], ids=lambda s: 'x'.join(map(str, s)))
@pytest.mark.parametrize('width', [(64, 64), (128, 256), (256, 256), (256, 128), (64, 128), (128, 64)], ids=lambda w: '%dto%d' % w)
@pytest.mark.parametrize('with_addend', [False, True], ids=['plain', 'addend'])
def test_fused_inference_block_matches_staged_path(shape, width, with_addend, dev):
    """Eval-mode bf16 mp_conv_residual(64, 64) as ONE kernel (csrc/mpconv_block_fwd.hip) against the same block
    run stage by stage (streaming GEMM, affine+activation, operator kernel, ...), non-trivial BatchNorm running
    statistics; both round every intermediate to bf16, the fused one skips the HBM round trips."""
    from fgnn_amd import _hip
    from fgnn_amd.mpnn import blocks, mp_conv_residual, mp_conv_type
    N, M, k = shape
    nin, nout = width
    B = 9
    g = torch.Generator().manual_seed(N + 3 * M + nin)
    blk = mp_conv_residual(nin, 64, 4, extension=mp_conv_type.NO_EXTENSION, with_residual=False, aggregator='max',
                           nout=nout)
    with torch.no_grad():
        for bn in (blk.conv1[1], blk.mp_conv.bn, blk.conv2[1]):
            C = bn.num_features
            bn.running_mean.copy_(torch.randn(C, generator=g) * 0.1)
            bn.running_var.copy_(torch.rand(C, generator=g) + 0.5)
            bn.weight.copy_(torch.rand(C, generator=g) + 0.5)
            bn.bias.copy_(torch.randn(C, generator=g) * 0.2)
        blk.mp_conv.filters.copy_(torch.randn(64, 256, generator=g) * 0.1)
    blk = blk.to(dev).eval()
    x = torch.randn(B, N, 1, nin, generator=g).bfloat16().to(dev).permute(0, 3, 1, 2)
    idx = torch.randint(0, N, (B, M, k), generator=g).to(dev)
    et = torch.randn(B, M, k, 4, generator=g).bfloat16().to(dev).permute(0, 3, 1, 2)
    add = torch.randn(B, M, 1, nout, generator=g).bfloat16().to(dev).permute(0, 3, 1, 2) if with_addend else None
    with torch.no_grad():
        y = blk(x, idx, et, addend=add)
        assert 'mpconv_block_fwd' in _hip.lib().fgnn_last_kernel().decode()
        blocks.FUSE_EVAL_BLOCKS = False
        try:
            ref = blk(x, idx, et, addend=add)
        finally:
            blocks.FUSE_EVAL_BLOCKS = True
    assert y.shape == ref.shape and y.dtype == torch.bfloat16
    err = float((y.float() - ref.float()).abs().max() / ref.float().abs().max())
    assert err <= 2.0 ** -5, err
    _block_vs_oracle(blk, x, idx, et, add, y)           # ... and against the f32 oracle, not only against ourselves


def test_ldpc_model_bf16_inference_fused_blocks_vs_staged(dev):
    """Whole LDPCModel, eval mode, bf16 autocast: the run that takes the one-kernel residual blocks against the
    same model with them disabled (every stage a separate kernel).  Both are bf16 pipelines that round at
    different points, so they are compared in the mean over a batch, not bit for bit."""
    import fgnn_amd
    from fgnn_amd.ldpc import synthetic_batch
    from fgnn_amd.mpnn import blocks
    torch.manual_seed(3)
    m = fgnn_amd.LDPCModel(2, 6, 4, aggregator='max').to(dev)
    data = synthetic_batch(64, dev, seed=11, dtype=torch.bfloat16)
    m.train()
    with torch.no_grad(), torch.autocast(device_type='cuda', dtype=torch.bfloat16):
        for _ in range(3):
            m(*data[:6])                                # populate the BatchNorm running statistics
    m.eval()
    with torch.no_grad(), torch.autocast(device_type='cuda', dtype=torch.bfloat16):
        a, sa = m(*data[:6])
        blocks.FUSE_EVAL_BLOCKS = False
        try:
            b, sb = m(*data[:6])
        finally:
            blocks.FUSE_EVAL_BLOCKS = True
    a, b = a.float(), b.float()
    assert torch.isfinite(a).all()
    assert float((a - b).abs().mean()) <= 0.03 * float(b.abs().mean()) + 1e-3
    assert float((sa.float() - sb.float()).abs().mean()) <= 0.03 * float(sb.float().abs().mean()) + 1e-3


@pytest.mark.parametrize('width', [(64, 64), (128, 256), (256, 256), (256, 128)], ids=lambda w: '%dto%d' % w)
@pytest.mark.parametrize('M', [96, 37])
@pytest.mark.parametrize('with_addend', [False, True], ids=['plain', 'addend'])
def test_fused_inference_fanout_block_matches_staged_path(width, M, with_addend, dev):
    """The one-kernel inference block around the hyper-factor fan-out call (one source, M destinations, k = 1, one
    edge type) against the staged path."""
    from fgnn_amd import _hip
    from fgnn_amd.mpnn import blocks, mp_conv_residual, mp_conv_type
    nin, nout = width
    B = 21
    g = torch.Generator().manual_seed(M + nin)
    blk = mp_conv_residual(nin, 64, 1, extension=mp_conv_type.NO_EXTENSION, with_residual=False, aggregator='max',
                           nout=nout)
    with torch.no_grad():
        for bn in (blk.conv1[1], blk.mp_conv.bn, blk.conv2[1]):
            C = bn.num_features
            bn.running_mean.copy_(torch.randn(C, generator=g) * 0.1)
            bn.running_var.copy_(torch.rand(C, generator=g) + 0.5)
            bn.weight.copy_(torch.rand(C, generator=g) + 0.5)
            bn.bias.copy_(torch.randn(C, generator=g) * 0.2)
        blk.mp_conv.filters.copy_(torch.randn(64, 64, generator=g) * 0.2)
    blk = blk.to(dev).eval()
    x = torch.randn(B, 1, 1, nin, generator=g).bfloat16().to(dev).permute(0, 3, 1, 2)
    idx = torch.zeros(B, M, 1, dtype=torch.int64, device=dev)
    et = (torch.rand(B, M, 1, 1, generator=g) + 0.5).bfloat16().to(dev).permute(0, 3, 1, 2)
    add = torch.randn(B, M, 1, nout, generator=g).bfloat16().to(dev).permute(0, 3, 1, 2) if with_addend else None
    with torch.no_grad():
        y = blk(x, idx, et, addend=add)
        assert 'mpconv_block_fanout' in _hip.lib().fgnn_last_kernel().decode()
        blocks.FUSE_EVAL_BLOCKS = False
        try:
            ref = blk(x, idx, et, addend=add)
        finally:
            blocks.FUSE_EVAL_BLOCKS = True
    err = float((y.float() - ref.float()).abs().max() / ref.float().abs().max())
    assert err <= 2.0 ** -5, err
    _block_vs_oracle(blk, x, idx, et, add, y)


@pytest.mark.parametrize('shape', [(5, 7, 96, 3, 4), (3, 7, 48, 6, 4), (2, 5, 7, 2, 3), (4, 8, 1, 96, 1), (70, 7, 96, 3, 4)])
@pytest.mark.parametrize('layout', ['nchw', 'channels_last'])
@pytest.mark.parametrize('glayout', ['edge_major', 'type_major'])
def test_fused_edge_mlp_matches_fp32_reference(shape, layout, glayout, dev):
    """emodel_f2v / emodel_v2f (train_ldpc.py:32-38) as one forward kernel + one recomputing backward: output within
    bf16 rounding of the fp32 MLP on the same bf16 inputs, parameter gradients within 1e-4 (f32 accumulation) of
    autograd's on the fp32 MLP — for either memory layout of the inputs and of the incoming gradient."""
    import fgnn_amd
    from fgnn_amd import _hip
    from fgnn_amd.edge_mlp import EdgeMLP
    B, cin, M, k, net = shape
    g = torch.Generator().manual_seed(B * 131 + cin)
    mlp = EdgeMLP(cin, 64, net).to(dev)
    x = torch.randn(B, cin, M, k, generator=g).to(dev).bfloat16()
    if layout == 'channels_last':
        x = x.permute(0, 2, 3, 1).contiguous().permute(0, 3, 1, 2)
    gy = torch.randn(B, net, M, k, generator=g).to(dev).bfloat16()
    with torch.no_grad():       # rows with a hidden pre-activation within rounding of the ReLU kink carry no gradient: there the kernel (bf16 hi + lo
        # operands, f32 accumulation in the matrix cores' order) and torch's f32 GEMM may legitimately take different sides — one such
        # row among 20 160 x 64 moved one weight-gradient column by 0.4 % (gpurun_out/r05g)
        pre = x.float().permute(0, 2, 3, 1).reshape(-1, cin) @ mlp[0].weight.view(64, cin).t() + mlp[0].bias
        firm = (pre.abs().min(dim=1)[0] > 1e-4).view(B, 1, M, k)
        gy = gy * firm
    if glayout == 'type_major':
        gy = gy.permute(0, 2, 3, 1).contiguous().permute(0, 3, 1, 2)
    y = mlp(x)
    # round 5: on the matrix cores (hi + lo bf16 operands) for the layout the reference's collated tensors have — plane-major sample
    # blocks of whole 16-row tiles; the VALU kernels keep every other layout
    mfma = layout == 'nchw' and (M * k) % 16 == 0
    assert _hip.lib().fgnn_last_kernel().decode() == ('edge_mlp_fwd_mfma_kernel' if mfma else 'edge_mlp_fwd_kernel')
    assert y.shape == (B, net, M, k) and y.permute(0, 2, 3, 1).is_contiguous()
    y.backward(gy)
    assert _hip.lib().fgnn_last_kernel().decode() == ('edge_mlp_bwd_mfma_kernel' if (mfma and (glayout == 'edge_major' or net == 1))
                                                      else 'edge_mlp_bwd_kernel')
    got = [p.grad.clone() for p in mlp.parameters()]
    # fp32 reference of the same three ops
    w1, b1, w2, b2 = [p.detach().clone().requires_grad_(True) for p in mlp.parameters()]
    xr = x.float().permute(0, 2, 3, 1).reshape(-1, cin)
    yr = torch.relu(xr @ w1.view(64, cin).t() + b1) @ w2.view(net, 64).t() + b2
    yr4 = yr.view(B, M, k, net).permute(0, 3, 1, 2)
    assert (y.float() - yr4).abs().max().item() <= 2.0 ** -8 * max(1.0, yr4.abs().max().item())
    yr4.backward(gy.float())
    for a, r in zip(got, (w1, b1, w2, b2)):
        assert H.rel_err(a.view(-1), r.grad.view(-1)) <= 1e-4
    # a second backward accumulates into the existing .grad in place
    mlp(x).backward(gy)
    for a, p in zip(got, mlp.parameters()):
        assert H.rel_err(p.grad.view(-1), 2 * a.view(-1)) <= 1e-5


def test_edge_mlp_keeps_reference_state_dict_and_staged_fallback(dev):
    from fgnn_amd.edge_mlp import EdgeMLP
    mlp = EdgeMLP(7, 64, 4).to(dev)
    assert list(mlp.state_dict().keys()) == ['0.weight', '0.bias', '2.weight', '2.bias']
    x = torch.randn(3, 7, 96, 3, device=dev)
    y32 = mlp(x)                                        # f32 inputs: the three children as written
    y16 = mlp(x.bfloat16())
    assert H.rel_err(y16.float(), y32) <= 2e-2


@pytest.mark.parametrize('dtype', [torch.float32, torch.bfloat16], ids=['f32', 'bf16'])
@pytest.mark.parametrize('nadd', [1, 2, 3, 4])
@pytest.mark.parametrize('training', [True, False], ids=['train', 'eval'])
def test_batchnorm_act_with_several_addends(dtype, nadd, training, dev):
    """act(BN(x)) + a1 + a2 + a3 in the apply kernel (FactorNN's `messages + old state + skip link`,
    factor_mpnn_sp.py:139-170) == torch's batch_norm + activation + explicit adds; each addend's gradient is the
    output gradient."""
    from fgnn_amd.mpnn.pointwise import BatchNormAct2d
    g = torch.Generator().manual_seed(11 * nadd)
    B, C, N = 6, 64, 96
    bn = BatchNormAct2d(C, slope=0.01).to(dev).train(training)
    ref = torch.nn.BatchNorm2d(C).to(dev).train(training)
    with torch.no_grad():
        bn.weight.copy_(torch.rand(C, generator=g) + 0.5); bn.bias.copy_(torch.randn(C, generator=g) * 0.1)
        bn.running_mean.copy_(torch.randn(C, generator=g) * 0.1); bn.running_var.copy_(torch.rand(C, generator=g) + 0.5)
    ref.load_state_dict(bn.state_dict())
    cl = lambda t: t.to(dev).to(dtype).permute(0, 2, 3, 1).contiguous().permute(0, 3, 1, 2)
    x = cl(torch.randn(B, C, N, 1, generator=g)).requires_grad_(training)
    adds = [cl(torch.randn(B, C, N, 1, generator=g)).requires_grad_(training) for _ in range(nadd)]
    if training:
        y = bn(x, addend=adds)
    else:
        with torch.no_grad():
            y = bn(x, addend=adds)
    xr = x.detach().float().requires_grad_(training)
    ar = [a.detach().float().requires_grad_(training) for a in adds]
    yr = torch.nn.functional.leaky_relu(ref(xr), 0.01)
    for a in ar:
        yr = yr + a
    tol = 1e-5 if dtype == torch.float32 else 2.0 ** -7
    assert H.rel_err(y.float(), yr) <= tol
    if training:
        gy = cl(torch.randn(B, C, N, 1, generator=g))
        y.backward(gy)
        yr.backward(gy.float())
        assert H.rel_err(x.grad.float(), xr.grad) <= (1e-4 if dtype == torch.float32 else 2.0 ** -6)
        for a in adds:
            assert torch.equal(a.grad, gy)


@pytest.mark.parametrize('side_stream', [True, False], ids=['two_streams', 'one_stream'])
def test_training_steps_are_bitwise_reproducible(side_stream, dev):
    """No atomics anywhere on the path and every cross-stream hand-over ordered: two runs of the same three bf16
    training steps (same seed, same data) end in bit-identical parameters — also with the hyper-factor branch of
    each layer on its side stream — and both stream layouts agree to rounding."""
    import fgnn_amd
    from fgnn_amd import ops
    from fgnn_amd.dp import FlatAdam, FlatGradBucket
    from fgnn_amd.ldpc import synthetic_batch

    def run(two):
        old = ops.SIDE_STREAM
        ops.SIDE_STREAM = two
        try:
            torch.manual_seed(1234)
            m = fgnn_amd.LDPCModel(2, 6, 4).to(dev).train()
            bucket = FlatGradBucket(m.parameters(), flatten_params=True)
            opt = FlatAdam(bucket, lr=1e-3, weight_decay=1e-8)
            data = synthetic_batch(512, dev, seed=21, dtype=torch.bfloat16)
            for _ in range(3):
                bucket.zero()
                with torch.autocast('cuda', dtype=torch.bfloat16):
                    logits, snr = m(*data[:6])
                loss = torch.nn.functional.binary_cross_entropy_with_logits(logits.float(), data[6]) + 0.1 * snr.float().pow(2).mean()
                loss.backward()
                opt.step()
            torch.cuda.synchronize()
            return bucket.flat_param.detach().clone(), float(loss)
        finally:
            ops.SIDE_STREAM = old

    a, la = run(side_stream)
    b, lb = run(side_stream)
    assert torch.equal(a, b) and la == lb
    c, lc = run(not side_stream)
    assert abs(la - lc) <= 2e-2 * max(1.0, abs(lc))


@pytest.mark.parametrize('width', [(64, 64), (128, 256), (256, 256), (256, 128), (64, 128)])
@pytest.mark.parametrize('N', [96, 50, 16])
@pytest.mark.parametrize('with_addend', [False, True], ids=['plain', 'addend'])
@pytest.mark.parametrize('shared', [True, False], ids=['shared_tables', 'per_sample_tables'])
def test_fused_inference_fanin_block_matches_staged_path(width, N, with_addend, shared, dev):
    """The one-kernel inference block around the hyper-factor fan-in call (one destination listening to all N nodes in
    order, one edge type with per-neighbour weights) against the staged path; a neighbour table that is not the identity
    must take the staged path."""
    from fgnn_amd import _hip
    from fgnn_amd.mpnn import blocks, mp_conv_residual, mp_conv_type
    nin, nout = width
    B = 37
    g = torch.Generator().manual_seed(N + nin + nout)
    blk = mp_conv_residual(nin, 64, 1, extension=mp_conv_type.NO_EXTENSION, with_residual=False, aggregator='max',
                           nout=nout)
    with torch.no_grad():
        for bn in (blk.conv1[1], blk.mp_conv.bn, blk.conv2[1]):
            C = bn.num_features
            bn.running_mean.copy_(torch.randn(C, generator=g) * 0.1)
            bn.running_var.copy_(torch.rand(C, generator=g) + 0.5)
            bn.weight.copy_(torch.rand(C, generator=g) + 0.5)
            bn.bias.copy_(torch.randn(C, generator=g) * 0.2)
        blk.mp_conv.filters.copy_(torch.randn(64, 64, generator=g) * 0.2)
    blk = blk.to(dev).eval()
    x = torch.randn(B, N, 1, nin, generator=g).bfloat16().to(dev).permute(0, 3, 1, 2)
    if shared:
        idx = torch.arange(N, device=dev).reshape(1, 1, N).expand(B, -1, -1)
        et = (torch.rand(1, 1, 1, N, generator=g) + 0.5).bfloat16().to(dev).expand(B, -1, -1, -1)
    else:
        idx = torch.arange(N, device=dev).reshape(1, 1, N).repeat(B, 1, 1)
        et = (torch.rand(B, 1, N, 1, generator=g) + 0.5).bfloat16().to(dev).permute(0, 3, 1, 2)
    add = torch.randn(B, 1, 1, nout, generator=g).bfloat16().to(dev).permute(0, 3, 1, 2) if with_addend else None
    with torch.no_grad():
        y = blk(x, idx, et, addend=add)
        assert 'mpconv_block_fanin' in _hip.lib().fgnn_last_kernel().decode()
        blocks.FUSE_EVAL_BLOCKS = False
        try:
            ref = blk(x, idx, et, addend=add)
        finally:
            blocks.FUSE_EVAL_BLOCKS = True
        assert y.shape == ref.shape == (B, nout, 1, 1)
        err = float((y.float() - ref.float()).abs().max() / ref.float().abs().max())
        assert err <= 2.0 ** -5, err
        _block_vs_oracle(blk, x, idx.contiguous(), et.contiguous(), add, y)
        if N > 2 and with_addend and shared:
            perm = idx.flip(-1).contiguous()                       # not the identity: must not take the fused kernel
            y2 = blk(x, perm, et, addend=add)
            assert 'mpconv_block_fanin' not in _hip.lib().fgnn_last_kernel().decode()
            blocks.FUSE_EVAL_BLOCKS = False
            try:
                ref2 = blk(x, perm, et, addend=add)
            finally:
                blocks.FUSE_EVAL_BLOCKS = True
            assert torch.equal(y2, ref2)


def test_graph_replay_reproduces_eager_gradients_bitwise(dev):
    """A captured training step (graph.StepGraph: the hyper-factor branch of every layer is a parallel branch of the
    hipGraph, backward included) leaves exactly the gradients of the same step launched eagerly — replayed twice, with
    the inputs changed in place in between."""
    import fgnn_amd
    from fgnn_amd.dp import FlatGradBucket
    from fgnn_amd.graph import StepGraph
    from fgnn_amd.ldpc import synthetic_batch
    torch.manual_seed(99)
    m = fgnn_amd.LDPCModel(2, 6, 4).to(dev).train()
    bucket = FlatGradBucket(m.parameters(), flatten_params=True)
    data = [t.clone() if t.is_floating_point() else t for t in synthetic_batch(384, dev, seed=31, dtype=torch.bfloat16)]
    other = synthetic_batch(384, dev, seed=32, dtype=torch.bfloat16)

    def compute():
        bucket.zero()
        with torch.autocast('cuda', dtype=torch.bfloat16):
            logits, snr = m(*data[:6])
        (torch.nn.functional.binary_cross_entropy_with_logits(logits.float(), data[6]) + 0.1 * snr.float().pow(2).mean()).backward()

    def eager_reference():
        state = {k: v.clone() for k, v in m.state_dict().items()}          # BatchNorm running statistics move per step
        compute()
        torch.cuda.synchronize()
        g = bucket.flat.clone()
        m.load_state_dict(state)
        return g

    g1 = eager_reference()
    state = {k: v.clone() for k, v in m.state_dict().items()}
    graph = StepGraph(compute)                                                  # 2 warm-up steps + capture (one more step)
    m.load_state_dict(state)
    graph.replay()
    torch.cuda.synchronize()
    assert torch.equal(bucket.flat, g1)
    for i in (0, 1, 4, 5, 6):                                                   # new inputs, same static buffers
        data[i].copy_(other[i])
    m.load_state_dict(state)
    g2 = eager_reference()
    assert not torch.equal(g2, g1)
    graph.replay()
    torch.cuda.synchronize()
    assert torch.equal(bucket.flat, g2)


def test_graph_replayed_training_equals_eager_training_bitwise(dev):
    """Three optimizer steps driven by graph replays end in exactly the parameters of three eager steps: the replay
    reads the parameters as they are at that moment (the low-precision weight copies of the library GEMMs are refreshed
    by a captured kernel, not frozen at capture time)."""
    import fgnn_amd
    from fgnn_amd.dp import FlatAdam, FlatGradBucket
    from fgnn_amd.graph import StepGraph
    from fgnn_amd.ldpc import synthetic_batch
    data = synthetic_batch(384, dev, seed=41, dtype=torch.bfloat16)

    def train(use_graph):
        torch.manual_seed(123)
        m = fgnn_amd.LDPCModel(2, 6, 4).to(dev).train()
        bucket = FlatGradBucket(m.parameters(), flatten_params=True)
        opt = FlatAdam(bucket, lr=1e-3, weight_decay=1e-8)

        def compute():
            bucket.zero()
            with torch.autocast('cuda', dtype=torch.bfloat16):
                logits, snr = m(*data[:6])
            (torch.nn.functional.binary_cross_entropy_with_logits(logits.float(), data[6]) + 0.1 * snr.float().pow(2).mean()).backward()

        start = {k: v.clone() for k, v in m.state_dict().items()}
        step = compute
        if use_graph:
            step = StepGraph(compute).replay                  # warm-up + capture advance the BatchNorm buffers: reset
            m.load_state_dict(start)
        for _ in range(3):
            step()
            opt.step()
        torch.cuda.synchronize()
        return bucket.flat_param.detach().clone(), {k: v.clone() for k, v in m.state_dict().items() if 'running' in k}

    pe, be = train(False)
    pg, bg = train(True)
    assert torch.equal(pe, pg)
    assert all(torch.equal(be[k], bg[k]) for k in be)


def test_inference_after_training_sees_the_trained_state(dev):
    """eval -> train -> eval on one model: the second evaluation must use the updated parameters and BatchNorm running
    statistics although kernels and the flat optimizer changed them behind torch's version counters (cached folded
    BatchNorm affines / one-kernel block weights key on pointwise.state_epoch()).  Checked against a fresh copy of the
    model that loads the trained state_dict."""
    import fgnn_amd
    from fgnn_amd.dp import FlatAdam, FlatGradBucket
    from fgnn_amd.graph import StepGraph
    from fgnn_amd.ldpc import synthetic_batch
    data = synthetic_batch(256, dev, seed=51, dtype=torch.bfloat16)
    torch.manual_seed(5)
    m = fgnn_amd.LDPCModel(2, 6, 4).to(dev)
    bucket = FlatGradBucket(m.parameters(), flatten_params=True)
    opt = FlatAdam(bucket, lr=1e-2)

    def infer(model):
        model.eval()
        with torch.no_grad(), torch.autocast('cuda', dtype=torch.bfloat16):
            return model(*data[:6])[0].float().clone()

    def compute():
        bucket.zero()
        with torch.autocast('cuda', dtype=torch.bfloat16):
            logits, snr = m(*data[:6])
        (torch.nn.functional.binary_cross_entropy_with_logits(logits.float(), data[6]) + 0.1 * snr.float().pow(2).mean()).backward()

    y0 = infer(m)
    m.train()
    for _ in range(2):
        compute()
        opt.step()
    y1 = infer(m)
    assert not torch.equal(y1, y0)
    fresh = fgnn_amd.LDPCModel(2, 6, 4).to(dev)
    fresh.load_state_dict(m.state_dict())
    assert torch.equal(infer(fresh), y1)
    # ... also when the training steps are graph replays
    m.train()
    graph = StepGraph(compute)
    for _ in range(2):
        graph.replay()
        opt.step()
    y2 = infer(m)
    fresh.load_state_dict(m.state_dict())
    assert not torch.equal(y2, y1) and torch.equal(infer(fresh), y2)
    m.train()
    graph.replay()
    opt.step()
    y3 = infer(m)
    fresh.load_state_dict(m.state_dict())
    assert torch.equal(infer(fresh), y3) and not torch.equal(y3, y2)


@pytest.mark.parametrize('tag', ['pw', 'hop'])
def test_factor_mpnn_training_steps_are_bitwise_reproducible(tag, dev):
    """BASELINE configs 2 / 5: two runs of the same three f32 training steps of factor_mpnn + its two edge models (same seed,
    same data; train_syn_hop_factor.py:283-303) end in bit-identical parameters and losses: the message operator's kernels
    (csrc/mpconv_fwd_ext.hip, csrc/mpconv_bwd_ext.hip) and everything around them have no order-dependent accumulation."""
    import fgnn_amd
    from fgnn_amd import _hip
    from fgnn_amd.dp import FlatAdam, FlatGradBucket

    hop_dim, pw_idx, pw_ef, hi_idx, hi_ef = H.syn_setup(tag)
    B = 96

    def run():
        torch.manual_seed(77)
        model = fgnn_amd.factor_mpnn(2, [4, hop_dim], O.SYN_DIMS, [16, 16]).to(dev).train()
        # the edge models are 1x1 convolutions: this package's Conv2d subclass (same state_dict; torch / MIOpen's own
        # weight-gradient kernel for a [1, 1, 31, 9] input is not run-to-run reproducible, with or without
        # torch.backends.cudnn.deterministic)
        C = fgnn_amd.mpnn.pointwise.PointwiseConv2d
        em_pw = torch.nn.Sequential(C(3, 64, 1), torch.nn.ReLU(), C(64, 16, 1)).to(dev)
        em_hi = torch.nn.Sequential(C(hi_ef.shape[0], 64, 1), torch.nn.ReLU(), C(64, 16, 1)).to(dev)
        everything = torch.nn.ModuleList([model, em_pw, em_hi])
        bucket = FlatGradBucket(everything.parameters(), flatten_params=True)
        opt = FlatAdam(bucket, lr=3e-3)
        g = torch.Generator().manual_seed(5)
        nf, pws = torch.rand(B, 2, 30, 1, generator=g).to(dev), torch.rand(B, 4, 30, 1, generator=g).to(dev)
        hi = torch.rand(B, hop_dim, 1 if tag == 'pw' else 30, 1, generator=g).to(dev)
        label = torch.randint(0, 2, (B, 30), generator=g).to(dev)
        t = lambda a: torch.from_numpy(a).to(dev)[None]
        losses = []
        for _ in range(3):
            bucket.zero()
            et_pw, et_hi = em_pw(t(pw_ef)), em_hi(t(hi_ef))
            pred, _ = model(nf, [pws, hi], [[t(pw_idx).expand(B, -1, -1), et_pw.expand(B, -1, -1, -1)],
                                           [t(hi_idx).expand(B, -1, -1), et_hi.expand(B, -1, -1, -1)]])
            loss = torch.nn.functional.cross_entropy(pred.squeeze(-1).permute(0, 2, 1).reshape(-1, 2), label.reshape(-1))
            loss.backward()
            opt.step()
            losses.append(float(loss.detach()))
        torch.cuda.synchronize()
        return bucket.flat_param.detach().clone(), losses, bucket.flat.detach().clone()

    pa, la, ga = run()
    pb, lb, gb = run()
    assert la == lb, (la, lb)
    assert torch.equal(ga, gb)
    assert torch.equal(pa, pb)
    assert all(l == l and abs(l) < 1e3 for l in la)               # finite
    assert float((ga != 0).float().mean()) > 0.5                   # the gradients reach (nearly) every parameter


@pytest.mark.parametrize('tag', ['pw', 'hop'])
def test_factor_mpnn_training_reduces_loss(tag, dev):
    """The synthetic-PGM training step as train_syn_hop_factor.py:283-303 runs it — edge models, factor_mpnn, cross entropy,
    gradient-norm clip, Adam — on one fixed batch whose labels are a function of the node potentials: the loss must come down,
    i.e. every gradient of csrc/mpconv_bwd_ext.hip (inputs, filters, bias, shared edge weights) points the right way."""
    import fgnn_amd
    hop_dim, pw_idx, pw_ef, hi_idx, hi_ef = H.syn_setup(tag)
    B = 64
    torch.manual_seed(3)
    model = fgnn_amd.factor_mpnn(2, [4, hop_dim], O.SYN_DIMS, [16, 16]).to(dev).train()
    C = torch.nn.Conv2d
    em_pw = torch.nn.Sequential(C(3, 64, 1), torch.nn.ReLU(), C(64, 16, 1)).to(dev)
    em_hi = torch.nn.Sequential(C(hi_ef.shape[0], 64, 1), torch.nn.ReLU(), C(64, 16, 1)).to(dev)
    params = list(model.parameters()) + list(em_pw.parameters()) + list(em_hi.parameters())
    # lr 1e-3: at the scripts' 3e-3 this batch sits on a plateau for 15-35 steps before it drops, and WHEN depends on the last
    # bits of the gradients (tools/diag_loss.py); at 1e-3 every kernel variant is under 0.2 by step 20
    opt = torch.optim.Adam(params, lr=1e-3)
    g = torch.Generator().manual_seed(11)
    nf = torch.rand(B, 2, 30, 1, generator=g)
    label = (nf[:, 1, :, 0] > nf[:, 0, :, 0]).long().to(dev)             # the better unary potential: learnable from the inputs
    nf, pws = nf.to(dev), torch.rand(B, 4, 30, 1, generator=g).to(dev)
    hi = torch.rand(B, hop_dim, 1 if tag == 'pw' else 30, 1, generator=g).to(dev)
    t = lambda a: torch.from_numpy(a).to(dev)[None]
    losses = []
    for _ in range(30):
        opt.zero_grad()
        et_pw, et_hi = em_pw(t(pw_ef)), em_hi(t(hi_ef))
        pred, _ = model(nf, [pws, hi], [[t(pw_idx).repeat(B, 1, 1), et_pw.repeat(B, 1, 1, 1)],       # the scripts' .repeat form
                                       [t(hi_idx).repeat(B, 1, 1), et_hi.repeat(B, 1, 1, 1)]])
        loss = torch.nn.functional.cross_entropy(pred.squeeze(-1).permute(0, 2, 1).reshape(-1, 2), label.reshape(-1))
        loss.backward()
        torch.nn.utils.clip_grad_norm_(params, 1.0)
        opt.step()
        losses.append(float(loss.detach()))
    assert all(l == l for l in losses), losses
    assert min(losses[-5:]) < 0.5 * losses[0], losses
    assert all(p.grad is not None and bool(torch.isfinite(p.grad).all()) for p in params)


@pytest.mark.parametrize('tag', ['pw', 'hop'])
def test_factor_mpnn_graph_replays_equal_eager_steps(tag, dev):
    """`bench.py --workload syn_*` times hipGraph replays of the training step: each replay must leave the gradients the eager
    step leaves, bit for bit, over several optimizer steps — for the package's own parameters (gradient-sink kernels) AND for
    the plain torch edge models, whose gradients come through the batch-summed edge-type gradient of csrc/mpconv_bwd_ext.hip.
    (A captured hipMemsetAsync in front of that sum ran out of order on replay: finite garbage in exactly those gradients from
    the second replay on, depending on the allocator's layout — nothing of the step is kept alive here, as in bench.py.)"""
    import fgnn_amd
    from fgnn_amd.dp import FlatAdam, FlatGradBucket
    from fgnn_amd.graph import StepGraph

    hop_dim, pw_idx, pw_ef, hi_idx, hi_ef = H.syn_setup(tag)
    B = 256
    torch.manual_seed(21)
    model = fgnn_amd.factor_mpnn(2, [4, hop_dim], O.SYN_DIMS, [16, 16]).to(dev).train()
    C = torch.nn.Conv2d
    em_pw = torch.nn.Sequential(C(3, 64, 1), torch.nn.ReLU(inplace=True), C(64, 16, 1)).to(dev)
    em_hi = torch.nn.Sequential(C(hi_ef.shape[0], 64, 1), torch.nn.ReLU(inplace=True), C(64, 16, 1)).to(dev)
    everything = torch.nn.ModuleList([model, em_pw, em_hi])
    bucket = FlatGradBucket(everything.parameters(), flatten_params=True)
    opt = FlatAdam(bucket, lr=3e-3)
    g = torch.Generator().manual_seed(9)
    nf, pws = torch.rand(B, 2, 30, 1, generator=g).to(dev), torch.rand(B, 4, 30, 1, generator=g).to(dev)
    hi = torch.rand(B, hop_dim, 1 if tag == 'pw' else 30, 1, generator=g).to(dev)
    label = torch.randint(0, 2, (B, 30), generator=g).to(dev)
    t = lambda a: torch.from_numpy(a).to(dev)[None]
    idx_pw, idx_hi, ef_pw, ef_hi = t(pw_idx), t(hi_idx), t(pw_ef), t(hi_ef)

    def compute():
        bucket.zero()
        et_pw, et_hi = em_pw(ef_pw), em_hi(ef_hi)
        pred, _ = model(nf, [pws, hi], [[idx_pw.expand(B, -1, -1), et_pw.expand(B, -1, -1, -1)],
                                       [idx_hi.expand(B, -1, -1), et_hi.expand(B, -1, -1, -1)]])
        torch.nn.functional.cross_entropy(pred.squeeze(-1).permute(0, 2, 1).reshape(-1, 2), label.reshape(-1)).backward()

    prev_cudnn = torch.backends.cudnn.enabled
    torch.backends.cudnn.enabled = False          # as bench.py: no MIOpen find-mode inside the capture
    try:
        graph = StepGraph(compute)
        for step in range(4):
            buffers = [b.clone() for b in everything.buffers()]
            graph.replay()
            torch.cuda.synchronize()
            g_graph = bucket.flat.clone()
            for b, s in zip(everything.buffers(), buffers):      # the eager step starts from the same running statistics
                b.copy_(s)
            compute()
            torch.cuda.synchronize()
            assert bool(torch.isfinite(g_graph).all())
            assert torch.equal(g_graph, bucket.flat), 'step %d: max |graph - eager| %.3e of %.3e' % (
                step, float((g_graph - bucket.flat).abs().max()), float(bucket.flat.abs().max()))
            opt.step()
    finally:
        torch.backends.cudnn.enabled = prev_cudnn


@pytest.mark.parametrize('B', [3, 70, 300])
@pytest.mark.parametrize('hyper_weights', ['ones', 'random'])
def test_one_kernel_factor_layer_vs_per_block_path_and_oracle(B, hyper_weights, dev):
    """a 64 -> 64 `FactorNN` layer as ONE kernel (csrc/factor_layer_fwd.hip) — v2v / f2f maps with their
    InstanceNorms, the four mp_conv_residual blocks, residual and skip-link sums — against (a) the per-block inference path
    of the same model and (b) the f32 ORACLE's FactorNN (factor_mpnn_sp.py:136-168 in the reference's op order).  Three
    64-wide layers, the last with a skip link from the first's output; batches below, around and above the grid size;
    the hyper-factor's edge weights as train_ldpc.py passes them (ones) and arbitrary."""
    import fgnn_amd
    from fgnn_amd import _hip
    from fgnn_amd.ldpc import synthetic_batch
    from fgnn_amd.mpnn import assemblies
    torch.manual_seed(17)
    net = fgnn_amd.mpnn.FactorNN(2, [6, 96], [64, 64, 64, 64], [4, 1], 2, skip_link={2: 0}, ret_high=True, aggregator='max').to(dev)
    with torch.no_grad():                                # BatchNorm affines / statistics that are not the identity
        for mod in net.modules():
            if isinstance(mod, torch.nn.modules.batchnorm._BatchNorm):
                mod.running_mean.normal_(0, 0.2)
                mod.running_var.uniform_(0.5, 1.5)
                mod.weight.uniform_(0.7, 1.3)
                mod.bias.normal_(0, 0.2)
    net.eval()
    node, hop, idx_f2v, idx_v2f = synthetic_batch(B, dev, seed=4, dtype=torch.bfloat16)[:4]
    g = torch.Generator().manual_seed(8)
    et_f2v = torch.randn(B, 96, 3, 4, generator=g).to(dev, torch.bfloat16).permute(0, 3, 1, 2)
    et_v2f = torch.randn(B, 48, 6, 4, generator=g).to(dev, torch.bfloat16).permute(0, 3, 1, 2)
    hyper_in = node[:, 0, :, :].reshape(B, 96, 1, 1)
    hidx_v2f = torch.arange(96, device=dev).reshape(1, 1, 96).expand(B, -1, -1)
    hidx_f2v = torch.zeros(1, 96, 1, dtype=torch.int64, device=dev).expand(B, -1, -1)
    if hyper_weights == 'ones':
        het_v2f, het_f2v = torch.ones(1, 1, 1, 96, device=dev), torch.ones(1, 1, 96, 1, device=dev)
    else:
        het_v2f, het_f2v = torch.randn(1, 1, 1, 96, generator=g).to(dev), torch.randn(1, 1, 96, 1, generator=g).to(dev)
    het_v2f, het_f2v = het_v2f.to(torch.bfloat16).expand(B, -1, -1, -1), het_f2v.to(torch.bfloat16).expand(B, -1, -1, -1)
    args = (node, [hop, hyper_in], [idx_f2v, hidx_f2v], [idx_v2f, hidx_v2f], [et_f2v, het_f2v], [et_v2f, het_v2f])

    def run(fused):
        assemblies.FUSE_EVAL_LAYERS = fused
        try:
            with torch.no_grad(), torch.autocast(device_type='cuda', dtype=torch.bfloat16):
                out, facs = net(*args)
            torch.cuda.synchronize()
            return out.float(), [f.float() for f in facs]
        finally:
            assemblies.FUSE_EVAL_LAYERS = True

    calls = []
    real = assemblies.FactorNN._fused_layer

    def spy(self, L, *a, **k):
        r = real(self, L, *a, **k)
        calls.append((L, r is not None))
        return r
    assemblies.FactorNN._fused_layer = spy
    try:
        y1, f1 = run(True)
    finally:
        assemblies.FactorNN._fused_layer = real
    assert calls == [(0, True), (1, True), (2, True)], calls          # every layer took the one-kernel path
    y0, f0 = run(False)
    for a, b in [(y1, y0), (f1[0], f0[0]), (f1[1], f0[1])]:
        assert torch.isfinite(a).all()
        assert H.rel_err(a, b) <= 2.0 ** -5, H.rel_err(a, b)           # two bf16 pipelines that round at different points
    # the f32 oracle on the same (bf16-rounded) inputs and parameters
    sd = {'main.' + k: v.detach().float().cpu() for k, v in net.state_dict().items()}
    cpu = lambda t: t.detach().cpu().float() if t.is_floating_point() else t.detach().cpu()
    with torch.no_grad():
        yo, fo = O.factor_nn(sd, 'main.', cpu(node), [cpu(hop), cpu(hyper_in)], [cpu(idx_f2v), cpu(hidx_f2v)],
                             [cpu(idx_v2f), cpu(hidx_v2f)], [cpu(et_f2v), cpu(het_f2v)], [cpu(et_v2f), cpu(het_v2f)],
                             dims=[64, 64, 64, 64], netypes=[4, 1], skip_link={2: 0})
    e1, e0 = H.rel_err(y1.cpu(), yo), H.rel_err(y0.cpu(), yo)
    print('one-kernel layers vs oracle %.3e, per-block path vs oracle %.3e (B = %d)' % (e1, e0, B))
    assert e1 <= 2.0 ** -5, e1
    assert H.rel_err(f1[0].cpu(), fo[0]) <= 2.0 ** -5
    assert H.rel_err(f1[1].cpu(), fo[1]) <= 2.0 ** -5


def test_parked_weight_gradients_equal_inline_ones_bitwise(dev, monkeypatch):
    """ops.defer_wgrad parks the weight-gradient kernels of a backward pass and issues them where their stream would wait for
    the other one (at the latest from an engine callback when the pass ends).  Same kernels, same operands, another place in
    stream order: the flat gradient must come out bit-identical to the inline schedule — eagerly, twice, and with the addends'
    gradient routed through its own node or through the fused tail's backward."""
    import fgnn_amd
    from fgnn_amd import ops
    from fgnn_amd.dp import FlatGradBucket
    from fgnn_amd.ldpc import synthetic_batch
    from fgnn_amd.mpnn import blocks
    torch.manual_seed(7)
    m = fgnn_amd.LDPCModel(2, 6, 4).to(dev).train()
    bucket = FlatGradBucket(m.parameters(), flatten_params=True)
    data = synthetic_batch(200, dev, seed=5, dtype=torch.bfloat16)
    state = {k: v.clone() for k, v in m.state_dict().items()}

    def grads():
        m.load_state_dict(state)
        bucket.zero()
        with torch.autocast('cuda', dtype=torch.bfloat16):
            logits, snr = m(*data[:6])
        (torch.nn.functional.binary_cross_entropy_with_logits(logits.float(), data[6]) + 0.1 * snr.float().pow(2).mean()).backward()
        torch.cuda.synchronize()
        assert not any(lst for lst in ops._DEFERRED.values())                  # nothing is left parked after a backward pass
        return bucket.flat.clone()

    monkeypatch.setattr(ops, 'DEFER_WGRAD', False)
    inline = grads()
    assert float(inline.abs().max()) > 0
    monkeypatch.setattr(ops, 'DEFER_WGRAD', True)
    assert torch.equal(grads(), inline) and torch.equal(grads(), inline)
    monkeypatch.setattr(blocks, 'ROUTE_ADDEND_GRADS', False)
    assert torch.equal(grads(), inline)
    monkeypatch.setattr(blocks, 'ROUTE_ADDEND_GRADS', True)
    monkeypatch.setattr(blocks, 'LATE_JOIN', False)
    assert torch.equal(grads(), inline)


def test_parked_launches_of_a_failed_backward_pass_are_dropped(dev, monkeypatch):
    """A backward pass that raises half-way never reaches the end-of-pass callback: what it parked must not land in the next
    pass's sums (FlatGradBucket.zero issues it ahead of the zeroing; a pass of another id arriving first would issue it too —
    parked launches are never silently dropped, a nested pass relies on that) nor keep that pass from queueing its own callback."""
    import fgnn_amd
    from fgnn_amd import ops
    from fgnn_amd.dp import FlatGradBucket
    from fgnn_amd.mpnn.pointwise import PointwiseConv2d

    class Boom(torch.autograd.Function):
        @staticmethod
        def forward(ctx, x):
            return x.view_as(x)

        @staticmethod
        def backward(ctx, g):
            raise RuntimeError('boom')

    torch.manual_seed(3)
    monkeypatch.setattr(ops, 'SIDE_ACTIVE', True)       # parking is only on while an assembly runs its side stream
    a, b = PointwiseConv2d(64, 64).to(dev), PointwiseConv2d(64, 64).to(dev)
    bucket = FlatGradBucket(list(a.parameters()) + list(b.parameters()))
    x = torch.randn(8, 64, 96, 1, device=dev).bfloat16().contiguous(memory_format=torch.channels_last)

    def run(fail):
        bucket.zero()
        with torch.autocast('cuda', dtype=torch.bfloat16):
            h = a(x)
            if fail:
                h = Boom.apply(h)
            y = b(h)
        y.float().sum().backward()
        torch.cuda.synchronize()
        return bucket.flat.clone()

    good = run(False)
    with pytest.raises(RuntimeError, match='boom'):
        run(True)                                       # b's weight gradient was parked, then the pass died
    assert any(lst for lst in ops._DEFERRED.values())
    again = run(False)
    assert torch.equal(again, good)                     # not doubled by the stale launch, and a's / b's own gradients are there
    assert not any(lst for lst in ops._DEFERRED.values())


def test_a_nested_backward_pass_does_not_lose_parked_launches(dev, monkeypatch):
    """a re-entrant backward (checkpointing, a Function that calls backward()) has another graph-task id than the pass
    that parked weight-gradient launches; those must be ISSUED, not cleared.  Gradients with parking on == with parking off."""
    from fgnn_amd import ops
    from fgnn_amd.dp import FlatGradBucket
    from fgnn_amd.mpnn.pointwise import PointwiseConv2d
    torch.manual_seed(4)
    a, b, c = (PointwiseConv2d(64, 64).to(dev) for _ in range(3))
    bucket = FlatGradBucket(list(a.parameters()) + list(b.parameters()) + list(c.parameters()))
    x = torch.randn(8, 64, 96, 1, device=dev).bfloat16().contiguous(memory_format=torch.channels_last)

    class Nested(torch.autograd.Function):              # its backward runs a whole inner pass through c
        @staticmethod
        def forward(ctx, h):
            ctx.save_for_backward(h)
            return h.view_as(h)

        @staticmethod
        def backward(ctx, g):
            (h,) = ctx.saved_tensors
            with torch.enable_grad(), torch.autocast('cuda', dtype=torch.bfloat16):
                c(h.detach()).float().sum().backward()
            return g

    def run():
        bucket.zero()
        with torch.autocast('cuda', dtype=torch.bfloat16):
            y = b(Nested.apply(a(x)))
        y.float().sum().backward()
        torch.cuda.synchronize()
        return bucket.flat.clone()

    plain = run()
    monkeypatch.setattr(ops, 'SIDE_ACTIVE', True)
    parked = run()
    assert not any(lst for lst in ops._DEFERRED.values())
    assert float(plain.abs().max()) > 0 and torch.equal(parked, plain)


def test_recorded_gradient_folds_equal_immediate_ones(dev):
    """csrc/fold_batch.hip: with the parameter gradients going to the flat bucket, every slab fold of a backward pass (the operator's
    filter / bias gradients, the node-wise maps' weight gradients) is recorded and ONE launch at the end of the pass folds them: the
    gradients of the folds launched one by one up to the f32 rounding of a different (fixed) summation order, bit-reproducible run to
    run; nothing stays recorded, and a second pass accumulates on top (two folds into the same accumulators: separate launches)."""
    import fgnn_amd
    from fgnn_amd import _hip, ops
    from fgnn_amd.dp import FlatGradBucket
    from fgnn_amd.ldpc import synthetic_batch
    torch.manual_seed(3)
    m = fgnn_amd.LDPCModel(2, 6, 4).to(dev).train()
    bucket = FlatGradBucket(m.parameters())
    data = synthetic_batch(48, dev, seed=11, dtype=torch.bfloat16)
    state = {k: v.clone() for k, v in m.state_dict().items()}

    def grads(defer, passes=1):
        m.load_state_dict(state)
        bucket.zero()
        ops.DEFER_FOLDS = defer
        try:
            for _ in range(passes):
                with torch.autocast('cuda', dtype=torch.bfloat16):
                    logits, snr = m(*data[:6])
                (torch.nn.functional.binary_cross_entropy_with_logits(logits.float(), data[6]) + 0.1 * snr.float().pow(2).mean()).backward()
                assert _hip.lib().fgnn_fold_pending() == 0 and not ops._FOLD_KEEP
        finally:
            ops.DEFER_FOLDS = True
        torch.cuda.synchronize()
        return bucket.flat.clone()

    a, b = grads(False), grads(True)
    scale = float(a.abs().max())
    assert scale > 0 and float((a - b).abs().max()) <= 1e-5 * scale
    assert torch.equal(b, grads(True))
    a2, b2 = grads(False, passes=2), grads(True, passes=2)
    assert float((a2 - b2).abs().max()) <= 2e-5 * scale and float((b2 - 2 * b).abs().max()) <= 2e-5 * scale


def test_merged_fan_out_gradients_match_the_per_consumer_sums(dev):
    """ops.FanBox: the gradient of a layer state that feeds the node-wise map and one block per factor type is ONE product over the
    consumers' (gz, W) pairs with the residual / skip gradients as addends (csrc/linear_fwd_b16.hip::linear_multi_b16_kernel), not one
    [R, C] tensor per consumer and an n-input sum.  Same gradients up to bf16 rounding (the merged form rounds the sum once), for
    every parameter of the LDPC model and for the input features; the kernel really ran; bit-reproducible."""
    import fgnn_amd
    from fgnn_amd import ops
    from fgnn_amd.dp import FlatGradBucket
    from fgnn_amd.ldpc import synthetic_batch
    torch.manual_seed(4)
    m = fgnn_amd.LDPCModel(2, 6, 4).to(dev).train()
    bucket = FlatGradBucket(m.parameters())
    data = synthetic_batch(64, dev, seed=12, dtype=torch.bfloat16)
    state = {k: v.clone() for k, v in m.state_dict().items()}

    def grads(merge):
        m.load_state_dict(state)
        bucket.zero()
        ops.MERGE_FAN_GRADS = merge
        rec = []
        ops.TIMER = type('T', (), {'run': staticmethod(lambda sym, nb, nf, launch, use_note=True, **kw: (rec.append(sym), launch()))})()
        try:
            with torch.autocast('cuda', dtype=torch.bfloat16):
                logits, snr = m(*data[:6])
            (torch.nn.functional.binary_cross_entropy_with_logits(logits.float(), data[6]) + 0.1 * snr.float().pow(2).mean()).backward()
        finally:
            ops.MERGE_FAN_GRADS = True
            ops.TIMER = None
        torch.cuda.synchronize()
        return bucket.flat.clone(), rec

    (a, ra), (b, rb) = grads(False), grads(True)
    assert 'linear_multi_b16_kernel' not in ra and rb.count('linear_multi_b16_kernel') >= 12 and rb.count('sum_n_kernel') < ra.count('sum_n_kernel')
    fro = float((a - b).norm() / a.norm())
    assert float(a.abs().max()) > 0 and fro <= 3e-2, fro
    assert torch.equal(b, grads(True)[0])


def test_merged_fan_out_weight_gradients_match_the_per_consumer_launches(dev):
    """ops.FanBox: the weight gradients of the maps that consume one layer state (the state's v2v / f2f map, conv1 of every block
    that starts from it) leave through ONE csrc/linear_wgrad_b16.hip launch per state (fgnn_linear_wgrad_multi) instead of one per
    map — same gradients for every parameter of the LDPC model up to the f32 rounding of another summation grid, fewer weight-gradient
    launches, bit-reproducible."""
    import fgnn_amd
    from fgnn_amd import ops
    from fgnn_amd.dp import FlatGradBucket
    from fgnn_amd.ldpc import synthetic_batch
    torch.manual_seed(4)
    m = fgnn_amd.LDPCModel(2, 6, 4).to(dev).train()
    bucket = FlatGradBucket(m.parameters())
    data = synthetic_batch(64, dev, seed=12, dtype=torch.bfloat16)
    state = {k: v.clone() for k, v in m.state_dict().items()}

    def grads(merge):
        m.load_state_dict(state)
        bucket.zero()
        ops.MERGE_FAN_WGRADS = merge
        rec = []
        ops.TIMER = type('T', (), {'run': staticmethod(lambda sym, nb, nf, launch, use_note=True, **kw: (rec.append(sym), launch()))})()
        try:
            with torch.autocast('cuda', dtype=torch.bfloat16):
                logits, snr = m(*data[:6])
            (torch.nn.functional.binary_cross_entropy_with_logits(logits.float(), data[6]) + 0.1 * snr.float().pow(2).mean()).backward()
        finally:
            ops.MERGE_FAN_WGRADS = True
            ops.TIMER = None
        torch.cuda.synchronize()
        return bucket.flat.clone(), rec

    (a, ra), (b, rb) = grads(False), grads(True)
    na, nb = ra.count('linear_wgrad_b16_kernel'), rb.count('linear_wgrad_b16_kernel')
    assert nb <= na - 12, (na, nb)
    assert float(a.abs().max()) > 0 and float((a - b).abs().max()) <= 2e-5 * float(a.abs().max()), float((a - b).abs().max())
    assert torch.equal(b, grads(True)[0])


def test_regressor_head_through_the_map_kernels_matches_the_torch_modules(dev):
    """LDPCModel._regress: the burst-noise regressor (train_ldpc.py:48-54,93) through this package's node-wise map / BatchNorm
    kernels when training on bf16 activations, against the seven torch modules in f32 on the same input: prediction, input
    gradient, every parameter's gradient, BatchNorm1d's running statistics and step counter."""
    import copy
    import fgnn_amd
    from fgnn_amd import ldpc
    torch.manual_seed(5)
    m = fgnn_amd.LDPCModel(2, 6, 4).to(dev).train()
    with torch.no_grad():
        m.nhop_regressor[5].bias.fill_(2.0)      # the closing ReLU away from its kink (a bf16 / f32 sign flip there moves a whole row's gradient)
    ref = copy.deepcopy(m.nhop_regressor).float()
    fro = lambda a, b: float((a.float() - b.float()).norm() / b.float().norm())
    g = torch.Generator().manual_seed(6)
    hop = torch.randn(512, 64, generator=g).to(dev).to(torch.bfloat16)
    gout = torch.randn(512, 1, generator=g).to(dev)
    x1 = hop.clone().requires_grad_(True)
    with torch.autocast('cuda', dtype=torch.bfloat16):
        y1 = m._regress(x1)
    assert y1.dtype == torch.float32 and y1.shape == (512, 1)
    y1.backward(gout)
    x2 = hop.float().requires_grad_(True)
    y2 = ref(x2)
    y2.backward(gout)
    assert float(y2.abs().max()) > 0
    assert float(y2.min()) > 0 and H.rel_err(y1, y2) <= 2e-2
    # (bf16 activations between the maps against f32 ones; ReLU-kink flips of hidden units included: Frobenius norms)
    assert fro(x1.grad, x2.grad) <= 8e-2
    for (k, p1), (_, p2) in zip(m.nhop_regressor.named_parameters(), ref.named_parameters()):
        if k == '0.bias':          # a bias in front of a batch-statistics BatchNorm: its true gradient is 0, both sides hold rounding noise
            assert float(p1.grad.abs().max()) <= 1e-2 and float(p2.grad.abs().max()) <= 1e-5
        else:
            assert fro(p1.grad, p2.grad) <= 8e-2, k
    assert int(m.nhop_regressor[1].num_batches_tracked) == 1
    assert H.rel_err(m.nhop_regressor[1].running_mean, ref[1].running_mean) <= 1e-2
    assert H.rel_err(m.nhop_regressor[1].running_var, ref[1].running_var) <= 1e-2
    # eval mode / f32 activations / the switch: the torch modules
    ldpc._FAST_REGRESSOR = False
    try:
        with torch.autocast('cuda', dtype=torch.bfloat16):
            y3 = m._regress(hop)
    finally:
        ldpc._FAST_REGRESSOR = True
    assert H.rel_err(y3.float(), y2) <= 3e-2
    assert int(m.nhop_regressor[1].num_batches_tracked) == 2


@pytest.mark.parametrize('dtype', [torch.float32, torch.bfloat16])
@pytest.mark.parametrize('B', [1, 7, 4096])
def test_decoding_loss_kernel_vs_torch(B, dtype, dev):
    """ldpc.decoding_loss (train_ldpc.py:222-227) as one launch each way against the torch expression in f32 on the same values:
    the loss, the logits' and the regressor's gradients, under a non-unit upstream gradient; bit-reproducible."""
    from fgnn_amd.ldpc import decoding_loss
    g = torch.Generator().manual_seed(B)
    logits = (torch.randn(B, 48, generator=g) * 4).to(dtype).to(dev)
    pred = torch.rand(B, 1, generator=g).mul(3).to(dev)
    label = torch.randint(0, 2, (B, 48), generator=g).float().to(dev)
    sigma_b = torch.randint(0, 6, (B,), generator=g).float().to(dev)
    l1, p1 = logits.clone().requires_grad_(True), pred.clone().requires_grad_(True)
    loss = decoding_loss(l1, p1, label, sigma_b)
    (loss * 1.5).backward()
    l2, p2 = logits.float().clone().requires_grad_(True), pred.clone().requires_grad_(True)
    ref = (torch.nn.functional.binary_cross_entropy_with_logits(l2.view(-1), label.view(-1)) +
           0.1 * torch.nn.functional.mse_loss(p2.view(-1), torch.pow(10.0, sigma_b / 20).view(-1)))
    (ref * 1.5).backward()
    assert loss.shape == () and abs(float(loss) - float(ref)) <= 1e-5 * max(1.0, abs(float(ref)))
    tol = 1e-6 if dtype == torch.float32 else 2.0 ** -8
    assert float((l1.grad.float() - l2.grad).abs().max()) <= tol * float(l2.grad.abs().max())
    assert float((p1.grad - p2.grad).abs().max()) <= 1e-6 * max(1e-3, float(p2.grad.abs().max()))
    assert torch.equal(decoding_loss(logits, pred, label, sigma_b), loss.detach())


def test_fast_path_switch_for_an_unchanged_script(dev):
    """fgnn_amd.enable_fast_path(): a module composed the way the reference's script composes its model — FactorNN from the
    `lib.model.mpnn` shim + torch `Sequential(Conv2d, ReLU, Conv2d)` edge models (train_ldpc.py:19-99, restated here) — and
    a `torch.optim.Adam(model.parameters())` + `LambdaLR` written as the script writes them get: EdgeMLP edge models around the
    SAME parameters, bf16 autocast inside the model's forward, the flat-bucket one-kernel Adam.  Outputs stay within bf16
    tolerance of the f32 path, the training steps run and move the parameters."""
    import sys, os
    import fgnn_amd
    from fgnn_amd.edge_mlp import EdgeMLP
    from fgnn_amd.fastpath import FastAdam
    from fgnn_amd.ldpc import synthetic_batch
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'factor-graph-neural-network_amd'))
    from lib.model.mpnn import FactorNN                                            # the documented import shim

    class ScriptModel(torch.nn.Module):                                            # composition of train_ldpc.py's LDPCModel
        def __init__(self):
            super().__init__()
            self.main = FactorNN(2, [6, 96], [64, 64, 64, 128, 256, 256, 128, 64, 64], [4, 1], 2,
                                 skip_link={4: 3, 5: 2, 7: 0}, ret_high=True, aggregator='max')
            mk = lambda: torch.nn.Sequential(torch.nn.Conv2d(7, 64, 1), torch.nn.ReLU(inplace=True), torch.nn.Conv2d(64, 4, 1))
            self.emodel_f2v, self.emodel_v2f = mk(), mk()

        def forward(self, node_feature, hop_feature, nn_idx_f2v, nn_idx_v2f, ef_f2v, ef_v2f):
            B = node_feature.shape[0]
            hyper = node_feature[:, 0, :, :].detach().reshape(B, 96, 1, 1)
            ones = lambda *s: torch.ones(*s, device=node_feature.device, dtype=node_feature.dtype)
            res, hops = self.main(node_feature, [hop_feature, hyper],
                                  [nn_idx_f2v, torch.zeros(B, 96, 1, dtype=torch.int64, device=node_feature.device)],
                                  [nn_idx_v2f, torch.arange(96, device=node_feature.device).reshape(1, 1, 96).repeat(B, 1, 1)],
                                  [self.emodel_f2v(ef_f2v), ones(B, 1, 96, 1)], [self.emodel_v2f(ef_v2f), ones(B, 1, 1, 96)])
            return res.reshape(B, 96)[:, :48]

    torch.manual_seed(3)
    m = ScriptModel().to(dev).train()          # batch statistics, as the script trains (a random-init net in eval mode is unnormalised)
    for mod in m.modules():
        if isinstance(mod, torch.nn.modules.batchnorm._BatchNorm):
            mod.momentum = 0.0                 # the comparison calls below leave the running statistics alone
    data = synthetic_batch(64, dev, seed=5, dtype=torch.float32)
    with torch.no_grad():
        ref = m(*data[:6]).float()
        with torch.autocast('cuda', dtype=torch.bfloat16):                         # the same regime written by hand in the script
            ref_bf = m(*[t.to(torch.bfloat16) if t.is_floating_point() else t for t in data[:6]]).float()
    fgnn_amd.enable_fast_path()
    try:
        with torch.no_grad():
            m(*data[:6])                                                           # the hook optimises the module at this call
            out = m(*data[:6])
        assert isinstance(m.emodel_f2v, EdgeMLP) and isinstance(m.emodel_v2f, EdgeMLP)
        assert out.dtype == torch.float32
        e_bf = float((out - ref_bf).abs().max() / ref.abs().max())                 # fused EdgeMLP vs torch's two convolutions, both bf16
        e_32 = float((out - ref).abs().max() / ref.abs().max())
        print('fast path vs hand-written autocast %.3e, vs f32 %.3e (hand-written autocast vs f32 %.3e)'
              % (e_bf, e_32, float((ref_bf - ref).abs().max() / ref.abs().max())))
        assert e_bf <= 3e-2 and e_32 <= 6e-2
        opt = torch.optim.Adam(m.parameters(), lr=1e-3, weight_decay=1e-8)         # as train_ldpc.py:160-169 writes it
        sched = torch.optim.lr_scheduler.LambdaLR(opt, lr_lambda=lambda e: 0.5 ** e)
        assert isinstance(opt, FastAdam)
        assert any(p is m.emodel_f2v[0].weight for p in opt.param_groups[0]['params'])
        before = m.emodel_f2v[0].weight.detach().clone()
        label = (torch.rand(64, 48, device=dev) > 0.5).float()
        for _ in range(2):
            opt.zero_grad()
            loss = torch.nn.functional.binary_cross_entropy_with_logits(m(*data[:6]), label)
            loss.backward()
            opt.step()
        sched.step()
        assert torch.isfinite(loss) and abs(opt.flat.lr - 1e-3) < 1e-12 and abs(opt.param_groups[0]['lr'] - 5e-4) < 1e-12
        assert float((m.emodel_f2v[0].weight - before).abs().max()) > 0
        assert isinstance(torch.optim.Adam([torch.nn.Parameter(torch.zeros(3))], lr=1e-3), torch.optim.Adam.stock)   # CPU parameters: stock
        # checkpoint round trip as train_ldpc.py:179-181,187 does it (model + optimizer state_dict): 2 steps, save, rebuild, load,
        # 1 step == 3 straight steps, bit for bit (the moments and the step count travel in stock Adam's layout)
        import copy, io
        buf = io.BytesIO()
        torch.save({'model': m.state_dict(), 'opt': opt.state_dict()}, buf)
        sd = opt.state_dict()
        assert len(sd['state']) == len(opt.param_groups[0]['params']) and float(sd['state'][0]['step']) == 2.0
        assert float(sd['state'][0]['exp_avg'].abs().max()) > 0

        def third_step(model, optimizer):
            optimizer.zero_grad()
            torch.nn.functional.binary_cross_entropy_with_logits(model(*data[:6]), label).backward()
            optimizer.step()
        third_step(m, opt)
        buf.seek(0)
        ck = torch.load(buf)
        torch.manual_seed(11)
        m2 = ScriptModel().to(dev).train()
        for mod in m2.modules():
            if isinstance(mod, torch.nn.modules.batchnorm._BatchNorm):
                mod.momentum = 0.0
        m2.load_state_dict(ck['model'])
        fgnn_amd.fastpath.fast_path(m2)
        opt2 = torch.optim.Adam(m2.parameters(), lr=1e-3, weight_decay=1e-8)
        assert isinstance(opt2, FastAdam)
        opt2.load_state_dict(ck['opt'])
        assert opt2.flat.t == 2 and abs(opt2.param_groups[0]['lr'] - 5e-4) < 1e-12
        third_step(m2, opt2)
        torch.cuda.synchronize()
        for (k, a), (_, b) in zip(m.state_dict().items(), m2.state_dict().items()):
            assert torch.equal(a, b), k
    finally:
        fgnn_amd.disable_fast_path()
    assert not hasattr(torch.optim.Adam, 'stock')


def test_fast_path_replays_hipgraphs_for_an_unchanged_training_loop(dev):
    """Round 6: `fgnn_amd.enable_fast_path()` also gives an UNCHANGED loop (/root/reference/train_ldpc.py:207-231: zero_grad, model(...),
    the script's own loss lines, backward, optimizer.step, a fresh collated batch every iteration) the hipGraph replay — forward and
    backward captured as two graphs after three eager calls (fastpath.GraphedForward).  The model is composed as the script composes
    it, INCLUDING the tables its forward builds with `.repeat` on every call (train_ldpc.py:77-84): under capture no host read can
    classify those, the verdicts recorded from the eager warm-up do (ops.Verdicts).  Checked: the replayed loop ends bit-identical to
    the same loop with the graphs off; calls the graphs cannot follow (another batch size, eval mode, other tables) run eagerly."""
    import os, sys
    import fgnn_amd
    from fgnn_amd import fastpath
    from fgnn_amd.ldpc import synthetic_batch
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'factor-graph-neural-network_amd'))
    from lib.model.mpnn import FactorNN

    class ScriptModel(torch.nn.Module):                       # train_ldpc.py:19-99, restated
        def __init__(self):
            super().__init__()
            self.main = FactorNN(2, [6, 96], [64, 64, 64, 128, 256, 256, 128, 64, 64], [4, 1], 2,
                                 skip_link={4: 3, 5: 2, 7: 0}, ret_high=True, aggregator='max')
            mk = lambda: torch.nn.Sequential(torch.nn.Conv2d(7, 64, 1), torch.nn.ReLU(inplace=True), torch.nn.Conv2d(64, 4, 1))
            self.emodel_f2v, self.emodel_v2f = mk(), mk()
            frozen = lambda t: torch.nn.Parameter(t, requires_grad=False)
            self.hnn_idx_v2f = frozen(torch.arange(96).reshape(1, 1, 96))
            self.hnn_idx_f2v = frozen(torch.zeros(1, 96, 1, dtype=torch.int64))
            self.hetype_v2f, self.hetype_f2v = frozen(torch.ones(1, 1, 1, 96)), frozen(torch.ones(1, 1, 96, 1))
            self.nhop_regressor = torch.nn.Sequential(torch.nn.Linear(64, 128), torch.nn.BatchNorm1d(128), torch.nn.ReLU(),
                                                      torch.nn.Linear(128, 128), torch.nn.ReLU(), torch.nn.Linear(128, 1), torch.nn.ReLU())

        def forward(self, node_feature, hop_feature, nn_idx_f2v, nn_idx_v2f, efeature_f2v, efeature_v2f):
            etype_f2v, etype_v2f = self.emodel_f2v(efeature_f2v), self.emodel_v2f(efeature_v2f)
            with torch.no_grad():
                bsize = node_feature.shape[0]
                nhop = node_feature[:, 0, :, :].reshape(bsize, 96, 1, 1)
            res, nhops = self.main(node_feature, [hop_feature, nhop],
                                   [nn_idx_f2v, self.hnn_idx_f2v.repeat(bsize, 1, 1)], [nn_idx_v2f, self.hnn_idx_v2f.repeat(bsize, 1, 1)],
                                   [etype_f2v, self.hetype_f2v.repeat(bsize, 1, 1, 1)], [etype_v2f, self.hetype_v2f.repeat(bsize, 1, 1, 1)])
            res = (res + node_feature[:, :1, :, :]).squeeze()
            return res[:, :48].contiguous(), self.nhop_regressor(nhops[1].squeeze())

    B, steps = 256, 9
    batches = []
    for i in range(steps):                                      # what the DataLoader collates: fresh tensors, per-sample table copies
        d = synthetic_batch(B, dev, seed=40 + i, dtype=torch.float32, shared_graph=False)
        batches.append(d)

    def loop(graph_after):
        fastpath.GRAPH_AFTER = graph_after
        fgnn_amd.enable_fast_path()
        try:
            torch.manual_seed(3)
            m = ScriptModel().to(dev).train()
            opt = torch.optim.Adam(m.parameters(), lr=1e-3, weight_decay=1e-8)
            losses = []
            for d in batches:
                opt.zero_grad()
                pred, sb = m(*d[:6])
                loss = torch.nn.functional.binary_cross_entropy_with_logits(pred.view(-1), d[6].view(-1).float())
                sloss = torch.nn.functional.mse_loss(sb.view(-1), torch.pow(10.0, d[7].float() / 20).view(-1))
                (loss + 0.1 * sloss).backward()
                opt.step()
                losses.append(float(loss.detach()))
                assert pred.dtype == torch.float32 and ((pred > 0).long() == d[6]).sum() >= 0
            params = torch.cat([q.detach().reshape(-1).float() for q in m.parameters()])
            bufs = torch.cat([b.detach().reshape(-1).float() for b in m.buffers()])
            return m, opt, losses, params, bufs
        finally:
            fgnn_amd.disable_fast_path()
            fastpath.GRAPH_AFTER = 3

    m0, _, l0, p0, b0 = loop(0)
    assert not isinstance(m0.__dict__.get('forward'), fastpath.GraphedForward)
    m1, opt1, l1, p1, b1 = loop(3)
    gf = m1.__dict__.get('forward')
    assert isinstance(gf, fastpath.GraphedForward) and not gf.failed and gf.cap is not None
    assert gf.replays == steps - 4, gf.replays          # call 1: the global hook optimises the module; 2-4: eager sightings; 5..: replays (the capturing call included)
    assert gf.cap.recorded >= 4 and gf.cap.unused == 0, (gf.cap.recorded, gf.cap.unused)     # the in-forward tables were classified under capture
    print('losses eager', l0, 'graphed', l1)
    assert l0 == l1                                               # bit-identical: the same kernels in the same order
    assert torch.equal(p0, p1) and torch.equal(b0, b1)
    # calls the graphs cannot follow run eagerly: another batch size, eval mode, other neighbour tables
    fgnn_amd.enable_fast_path()
    try:
        before = gf.replays
        small = synthetic_batch(64, dev, seed=77, dtype=torch.float32, shared_graph=False)
        pred, _ = m1(*small[:6])
        assert pred.shape == (64, 48) and gf.replays == before
        m1.eval()
        with torch.no_grad():
            pred, _ = m1(*batches[0][:6])
        assert pred.shape == (B, 48) and gf.replays == before
        m1.train()
        other = list(batches[0][:6])
        other[2] = other[2].flip(2).contiguous()                  # the same graph, neighbours listed in another order: other table contents
        pe, _ = m1(*other)
        assert gf.replays == before and bool(torch.isfinite(pe).all())
        pr, _ = m1(*batches[0][:6])
        assert gf.replays == before + 1
        import copy
        m2 = copy.deepcopy(m1)                                    # graphs do not travel; the copy is an eager module that may capture again
        assert m2.__dict__['forward'].cap is None and m2.__dict__['forward'].module is m2
    finally:
        fgnn_amd.disable_fast_path()
