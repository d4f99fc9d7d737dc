"""GPU, BASELINE.json's full size (batch 4096 codewords, the four operator shapes of the 96.3.963 LDPC model): the
oracle is too slow to check every output there, so the kernels are checked through properties the operator has by
construction (mp_nn.py:115-175: samples are independent; max-product aggregation is positively homogeneous and blind
to the order of a destination's neighbours), all BIT-EXACT, plus the oracle on a random sample of the batch."""
import pytest
import torch

import fgnn_oracle as O
import helpers as H

pytestmark = pytest.mark.gpu

B = 4096
SHAPES = {                      # name: (nin, nou, net, N, M, k)
    'parity_v2f': (64, 64, 4, 96, 48, 6),
    'parity_f2v': (64, 64, 4, 48, 96, 3),
    'hyper_v2f': (64, 64, 1, 96, 1, 96),
    'hyper_f2v': (64, 64, 1, 1, 96, 1),
}


def _problem(name, dtype, dev, seed=0):
    nin, nou, net, N, M, k = SHAPES[name]
    g = torch.Generator().manual_seed(seed + N + k)
    cl = lambda t: t.to(dtype).to(dev).permute(0, 2, 3, 1).contiguous().permute(0, 3, 1, 2)
    x = cl(torch.randn(B, nin, N, 1, generator=g))
    idx = torch.randint(0, N, (B, M, k), generator=g).to(dev)
    et = cl(torch.randn(B, net, M, k, generator=g))
    W = (torch.randn(nin, nou * net, generator=g) * 0.1).to(dev)
    return x, idx, et, W


def _fwd(x, idx, et, W, bias=None, name=None):
    from fgnn_amd import _hip, ops
    nin, nou, net, N, M, k = SHAPES[name]
    y, am = ops.mpconv_forward_raw(x, idx, et, W, bias, nou, net, _hip.EXT_NONE, _hip.AGG_MAX, want_argmax=True)
    return y, am


@pytest.mark.parametrize('name', list(SHAPES))
@pytest.mark.parametrize('dtype', [torch.float32, torch.bfloat16], ids=['f32', 'bf16'])
def test_forward_properties_at_full_batch(name, dtype, dev):
    nin, nou, net, N, M, k = SHAPES[name]
    x, idx, et, W = _problem(name, dtype, dev)
    y, am = _fwd(x, idx, et, W, name=name)
    # samples are independent and the kernel is deterministic: any slice of the batch run alone gives the same bits
    for lo, hi in ((0, 64), (1000, 1003), (4095, 4096)):
        ys, ams = _fwd(x[lo:hi], idx[lo:hi], et[lo:hi], W, name=name)
        assert torch.equal(ys, y[lo:hi]) and torch.equal(ams, am[lo:hi])
    # ... and permuting the samples permutes the outputs
    perm = torch.randperm(B, generator=torch.Generator().manual_seed(1)).to(dev)
    yp, _ = _fwd(x[perm], idx[perm], et[perm], W, name=name)
    assert torch.equal(yp, y[perm])
    # positive homogeneity without bias: scaling the features by 2 scales the messages by exactly 2
    y2, am2 = _fwd(x * 2, idx, et, W, name=name)
    assert torch.equal(y2, y * 2) and torch.equal(am2, am)
    # the max does not care in which order a destination lists its neighbours (the argmax index follows the order)
    if k > 1:
        order = torch.randperm(k, generator=torch.Generator().manual_seed(2)).to(dev)
        eo = et[:, :, :, order].permute(0, 2, 3, 1).contiguous().permute(0, 3, 1, 2)     # same memory layout as et
        yo, amo = _fwd(x, idx[:, :, order].contiguous(), eo, W, name=name)
        assert torch.equal(yo, y)
    # the oracle on a random sample of the batch
    pick = torch.randperm(B, generator=torch.Generator().manual_seed(3))[:24]
    sd = {'filters': W.cpu(), 'bias': torch.zeros(nou)}
    ref = O.mp_conv(sd, '', x[pick.to(dev)].float().cpu(), idx[pick.to(dev)].cpu(), et[pick.to(dev)].float().cpu(), nou=nou,
                    net=net, extension=0, aggregator='max', relu=False)
    tol = 2e-5 if dtype == torch.float32 else 2.0 ** -6
    assert H.rel_err(y[pick.to(dev)].float(), ref) <= tol


@pytest.mark.parametrize('name', ['parity_v2f', 'parity_f2v'])
@pytest.mark.parametrize('dtype', [torch.float32, torch.bfloat16], ids=['f32', 'bf16'])
def test_backward_properties_at_full_batch(name, dtype, dev):
    """Gradients at the full batch: input gradients of a slice of samples equal those of the slice run alone (bit-exact),
    the bias gradient is the plain sum of the output gradient, and the filter gradient is additive over a split of the
    batch (f32 accumulation: to rounding)."""
    from fgnn_amd import _hip, ops
    nin, nou, net, N, M, k = SHAPES[name]
    x, idx, et, W = _problem(name, dtype, dev, seed=5)
    bias = torch.zeros(nou, device=dev)
    g = torch.Generator().manual_seed(9)
    gz = torch.randn(B, M, 1, nou, generator=g).to(dtype).to(dev).permute(0, 3, 1, 2)

    def grads(sl):
        xx = x[sl].detach().clone().requires_grad_(True)
        ee = et[sl].detach().clone().requires_grad_(True)
        ww = W.detach().clone().requires_grad_(True)
        bb = bias.detach().clone().requires_grad_(True)
        z = ops.mpconv(xx, idx[sl], ee, ww, bb, nou, net, _hip.EXT_NONE, _hip.AGG_MAX)
        z.backward(gz[sl])
        return xx.grad, ee.grad, ww.grad, bb.grad

    full = grads(slice(0, B))
    for sl in (slice(0, 32), slice(2048, 2051)):
        part = grads(sl)
        assert torch.equal(part[0], full[0][sl]) and torch.equal(part[1], full[1][sl])
    assert H.rel_err(full[3], gz.float().sum((0, 2, 3))) <= 1e-5
    a, b = grads(slice(0, 2048)), grads(slice(2048, B))
    assert H.rel_err(a[2] + b[2], full[2]) <= 1e-5 and H.rel_err(a[3] + b[3], full[3]) <= 1e-5


@pytest.mark.parametrize('dtype', [torch.float32, torch.bfloat16], ids=['f32', 'bf16'])
def test_ldpc_model_inference_at_full_batch(dtype, dev):
    """The whole decoder (LDPCModel, eval mode: 32 operator calls per forward) on 4096 codewords from the GPU data
    path: every codeword is decoded independently, so slices of the batch run alone reproduce the full run (to rounding
    only: the wide node-wise maps go through rocBLAS, whose tiling — hence summation order — depends on the row count);
    in f32 a random sample of the batch matches the CPU oracle within the north-star 1e-4."""
    import fgnn_amd
    from fgnn_amd.datapath import LdpcDataPath
    m = fgnn_amd.LDPCModel(2, 6, 4, aggregator='max')
    m.load_state_dict(H.fill_state_dict(m.state_dict()))
    m = m.to(dev).eval()
    data = LdpcDataPath(dev).sample(B, seed=12, dtype=dtype)[:6]
    amp = torch.autocast('cuda', dtype=torch.bfloat16, enabled=dtype == torch.bfloat16)
    with torch.no_grad(), amp:
        logits, snr = m(*data)
        assert logits.shape == (B, 48) and torch.isfinite(logits.float()).all()
        for lo, hi in ((0, 128), (4000, 4096)):
            l2, s2 = m(*[t[lo:hi] for t in data])
            tol = 1e-5 if dtype == torch.float32 else 2.0 ** -5
            assert H.rel_err(l2.float(), logits[lo:hi].float()) <= tol and H.rel_err(s2.float(), snr[lo:hi].float()) <= tol
    if dtype == torch.float32:
        pick = torch.randperm(B, generator=torch.Generator().manual_seed(4))[:8].to(dev)
        sd = {k: v.cpu() for k, v in m.state_dict().items()}
        lo_, so_ = O.ldpc_model(sd, *[t[pick].cpu().contiguous() for t in data], training=False)
        assert H.rel_err(logits[pick], lo_) <= 1e-4 and H.rel_err(snr[pick], so_) <= 1e-4


@pytest.mark.parametrize('tag,batch', [('pw', 256), ('hop', 1024), ('hop8', 1024)],
                         ids=['pw_factors_b256', 'degree9_hop_factors_b1024', 'degree8_hop_factors_b1024'])
def test_synthetic_pgm_configs_at_baseline_batch(tag, batch, dev):
    """BASELINE.json configs 2 and 5: `factor_mpnn` on the 30-node synthetic PGMs of train_syn_pw_factor.py /
    train_syn_hop_factor.py (pairwise factors + high-order factors of degree 9 — the script's default — and 8 — what
    BASELINE names; 16 edge types, softmax and max aggregation, DIFF extension) at batch 256 / 1024, fp32, eval mode: HIP
    path within a flat 1e-4 of the CPU oracle on a sample of the batch, and every slice of the batch reproduces the
    full run."""
    import fgnn_amd
    hop_dim, pw_idx, pw_ef, hi_idx, hi_ef = H.syn_setup(tag)
    model = fgnn_amd.factor_mpnn(2, [4, hop_dim], O.SYN_DIMS, [16, 16])
    model.load_state_dict(H.syn_fill(model.state_dict()))
    em_pw, em_hi = H.syn_edge_models(hi_ef)
    g = torch.Generator().manual_seed(batch)
    nfeature = torch.rand(batch, 2, 30, 1, generator=g)          # log-potentials ~ U(0,1) (random_pgm.py:22)
    pws = torch.rand(batch, 4, 30, 1, generator=g)
    hi_feat = torch.rand(batch, hop_dim, 1 if tag == 'pw' else 30, 1, generator=g)      # pw: one chain factor per graph
    model.eval()
    with torch.no_grad():
        et_pw_c, et_hi_c = em_pw(torch.from_numpy(pw_ef)[None]), em_hi(torch.from_numpy(hi_ef)[None])
        md, t = model.to(dev), (lambda a: a.to(dev))
        gs = [[torch.from_numpy(pw_idx).to(dev)[None].expand(batch, -1, -1), t(et_pw_c).expand(batch, -1, -1, -1)],
              [torch.from_numpy(hi_idx).to(dev)[None].expand(batch, -1, -1), t(et_hi_c).expand(batch, -1, -1, -1)]]
        pred, ff = md(t(nfeature), [t(pws), t(hi_feat)], gs)
        lo, hi = batch // 2, batch // 2 + 5
        gs_s = [[a[lo:hi], b[lo:hi]] for a, b in gs]
        pred_s, _ = md(t(nfeature[lo:hi]), [t(pws[lo:hi]), t(hi_feat[lo:hi])], gs_s)
        assert H.rel_err(pred_s, pred[lo:hi]) <= 5e-5                 # library GEMMs tile by row count: rounding only
        pick = torch.randperm(batch, generator=torch.Generator().manual_seed(1))[:6]
        sd = {k: v.cpu() for k, v in md.state_dict().items()}
        gs_o = [[torch.from_numpy(pw_idx)[None].repeat(6, 1, 1), et_pw_c.repeat(6, 1, 1, 1)],
                [torch.from_numpy(hi_idx)[None].repeat(6, 1, 1), et_hi_c.repeat(6, 1, 1, 1)]]
        po, fo = O.factor_mpnn(sd, '', nfeature[pick], [pws[pick], hi_feat[pick]], gs_o, dims=O.SYN_DIMS, netypes=[16, 16],
                               training=False)
    e_pred, e_ff = H.rel_err(pred[pick.to(dev)], po), H.rel_err(ff[1][pick.to(dev)], fo[1])
    print('factor_mpnn %s B=%d vs oracle: pred err %.2e, factor-feature err %.2e' % (tag, batch, e_pred, e_ff))
    assert float(po.abs().max()) > 0.05 and float(fo[1].abs().max()) > 0.05          # not a vanishing output
    assert e_pred <= 1e-4
    assert e_ff <= 1e-4


@pytest.mark.parametrize('tag', ['pw', 'hop'], ids=['pw_factors', 'degree9_hop_factors'])
def test_synthetic_pgm_he_gain_stack_within_its_conditioning(tag, dev):
    """The same 11-layer `factor_mpnn` stacks with the He-gain fill (2.0: per-layer gain above one, closer to trained weights than
    the contracting fill of the fixtures).  There the reference's own f32 evaluation moves by a few 1e-5 against an f64 run of the
    same maths, so the bound is conditioning-aware: max(1e-4, 4 x that distance) — the check that an accuracy regression of the
    three-term bf16-split forward (csrc/mpconv_fwd_ext.hip) on an ill-conditioned stack cannot hide behind the well-conditioned
    fixtures."""
    import fgnn_amd
    hop_dim, pw_idx, pw_ef, hi_idx, hi_ef = H.syn_setup(tag)
    fill = lambda sd: H.fill_state_dict(sd, gain=2.0)
    model = fgnn_amd.factor_mpnn(2, [4, hop_dim], O.SYN_DIMS, [16, 16])
    model.load_state_dict(fill(model.state_dict()))
    C = torch.nn.Conv2d
    em_pw = torch.nn.Sequential(C(3, 64, 1), torch.nn.ReLU(), C(64, 16, 1))
    em_hi = torch.nn.Sequential(C(hi_ef.shape[0], 64, 1), torch.nn.ReLU(), C(64, 16, 1))
    em_pw.load_state_dict(fill(em_pw.state_dict()))
    em_hi.load_state_dict(fill(em_hi.state_dict()))
    batch = 48
    g = torch.Generator().manual_seed(77)
    nfeature = torch.rand(batch, 2, 30, 1, generator=g)
    pws = torch.rand(batch, 4, 30, 1, generator=g)
    hi_feat = torch.rand(batch, hop_dim, 1 if tag == 'pw' else 30, 1, generator=g)
    model.eval()
    with torch.no_grad():
        et_pw_c, et_hi_c = em_pw(torch.from_numpy(pw_ef)[None]), em_hi(torch.from_numpy(hi_ef)[None])
        md, t = model.to(dev), (lambda a: a.to(dev))
        gs = [[torch.from_numpy(pw_idx).to(dev)[None].expand(batch, -1, -1), t(et_pw_c).expand(batch, -1, -1, -1)],
              [torch.from_numpy(hi_idx).to(dev)[None].expand(batch, -1, -1), t(et_hi_c).expand(batch, -1, -1, -1)]]
        pred, ff = md(t(nfeature), [t(pws), t(hi_feat)], gs)
        sd = {k: v.cpu() for k, v in md.state_dict().items()}
        n = 8
        gs_o = [[torch.from_numpy(pw_idx)[None].repeat(n, 1, 1), et_pw_c.repeat(n, 1, 1, 1)],
                [torch.from_numpy(hi_idx)[None].repeat(n, 1, 1), et_hi_c.repeat(n, 1, 1, 1)]]
        po, fo = O.factor_mpnn(sd, '', nfeature[:n], [pws[:n], hi_feat[:n]], gs_o, dims=O.SYN_DIMS, netypes=[16, 16], training=False)
        d64 = lambda a: a.double() if torch.is_floating_point(a) else a
        p64, f64 = O.factor_mpnn({k: d64(v) for k, v in sd.items()}, '', nfeature[:n].double(), [pws[:n].double(), hi_feat[:n].double()],
                                 [[a, b.double()] for a, b in gs_o], dims=O.SYN_DIMS, netypes=[16, 16], training=False)
    cond = max(H.rel_err(po.double(), p64), H.rel_err(fo[1].double(), f64[1]))
    e_pred, e_ff = H.rel_err(pred[:n].cpu().double(), p64), H.rel_err(ff[1][:n].cpu().double(), f64[1])
    print('factor_mpnn %s, He gain: f32 oracle vs f64 %.2e; HIP vs f64: pred %.2e, factor features %.2e' % (tag, cond, e_pred, e_ff))
    tol = max(1e-4, 4.0 * cond)
    assert e_pred <= tol and e_ff <= tol
