"""CPU: the oracle restatement against the golden vectors produced by the REAL reference
(oracle/make_golden.py).  This is what pins the oracle; the GPU tests then compare the HIP
path with the oracle and with the same vectors."""
import numpy as np
import pytest
import torch

import fgnn_oracle as O
import helpers as H

CASES = H.operator_cases()


@pytest.mark.parametrize('c', CASES, ids=[repr(c) for c in CASES])
def test_operator_forward_backward(c):
    sd = c.sd()
    for p in ('filters', 'bias', 'bn.weight', 'bn.bias'):
        if p in sd:
            sd[p].requires_grad_(True)
    x = c.t['x'].clone().requires_grad_(True)
    et = c.t['etype'].clone().requires_grad_(True)
    y = O.mp_conv(sd, '', x, c.t['idx'], et, nou=c.nou, net=c.net, extension=c.ext,
                  aggregator=c.agg, training=(c.bn == 'train'), relu=c.relu)
    assert H.rel_err(y, c.t['y']) <= 1e-6
    y.backward(c.t['gy'])
    assert H.rel_err(x.grad, c.t['gx']) <= 1e-6
    assert H.rel_err(et.grad, c.t['getype']) <= 1e-6
    assert H.rel_err(sd['filters'].grad, c.t['gfilters']) <= 1e-6
    if c.has_bias:
        assert H.rel_err(sd['bias'].grad, c.t['gbias']) <= 1e-6
    if c.bn == 'train':
        assert H.rel_err(sd['bn.running_var'], c.t['post_running_var']) <= 1e-6
        assert H.rel_err(sd['bn.running_mean'], c.t['post_running_mean']) <= 1e-6


def test_residual_block():
    z = H.load('block.npz')
    for row in z['meta']:
        cid, ext, with_res, nout, train, nin, nmed, net = [int(v) for v in row]
        pre = 'b%02d.' % cid
        sd = {k[len(pre) + 3:]: torch.from_numpy(z[k]).clone() for k in z.files
              if k.startswith(pre + 'sd.')}
        g = lambda n: torch.from_numpy(z[pre + n])
        y = O.residual_block(sd, '', g('x'), g('idx'), g('etype'), net=net, extension=ext,
                             aggregator='max', with_residual=bool(with_res), training=bool(train))
        assert H.rel_err(y, g('y')) <= 1e-6, row


def _ldpc_state():
    import fgnn_amd  # only for the parameter names/shapes (construction is CPU-safe)
    m = fgnn_amd.LDPCModel(2, 6, 4, aggregator='max')
    return H.fill_state_dict(m.state_dict())


def test_ldpc_model_full_size():
    z = H.load('ldpc_model.npz')
    inputs = [torch.from_numpy(z['in%d' % i]) for i in range(6)]
    sd = _ldpc_state()
    with torch.no_grad():
        logits, snr = O.ldpc_model(sd, *inputs, training=False)
    assert H.rel_err(logits, torch.from_numpy(z['eval_logits'])) <= 1e-5
    assert H.rel_err(snr, torch.from_numpy(z['eval_snr'])) <= 1e-5
    # train mode (B=16): batch-stat BatchNorm amplifies f32 rounding (f32-vs-f64 of this very
    # model: 2e-4), so cross-process agreement is checked at 1e-3, not 1e-5
    inputs = [torch.from_numpy(z['tin%d' % i]) for i in range(6)]
    sd = _ldpc_state()
    with torch.no_grad():
        logits, snr = O.ldpc_model(sd, *inputs, training=True)
    assert H.rel_err(logits, torch.from_numpy(z['train_logits'])) <= 1e-3
    rv = np.asarray([float(v.double().sum()) for k, v in sorted(sd.items())
                     if k.endswith('running_var')])
    assert np.allclose(rv, z['post_rv_digest'], rtol=1e-3)


@pytest.mark.parametrize('tag', H.SYN_TAGS)
def test_factor_mpnn_full_size(tag):
    import fgnn_amd
    z = H.load('factor_mpnn_%s.npz' % tag)
    hop_dim, pw_idx, pw_ef, hi_idx, hi_ef = H.syn_setup(tag)
    model = fgnn_amd.factor_mpnn(2, [4, hop_dim], O.SYN_DIMS, [16, 16])
    em_pw, em_hi = H.syn_edge_models(hi_ef)
    B = z['nfeature'].shape[0]
    with torch.no_grad():
        et_pw = em_pw(torch.from_numpy(pw_ef)[None]).repeat(B, 1, 1, 1)
        et_hi = em_hi(torch.from_numpy(hi_ef)[None]).repeat(B, 1, 1, 1)
        gs = [[torch.from_numpy(pw_idx)[None].repeat(B, 1, 1), et_pw],
              [torch.from_numpy(hi_idx)[None].repeat(B, 1, 1), et_hi]]
        for mode in ('eval', 'train'):
            sd = H.syn_fill(model.state_dict())
            pred, ff = O.factor_mpnn(sd, '', torch.from_numpy(z['nfeature']),
                                     [torch.from_numpy(z['pws']), torch.from_numpy(z['hi_feat'])],
                                     gs, dims=O.SYN_DIMS, netypes=[16, 16],
                                     training=(mode == 'train'))
            assert H.rel_err(pred, torch.from_numpy(z[mode + '_pred'])) <= 2e-5
            assert H.rel_err(ff[1], torch.from_numpy(z[mode + '_ff1'])) <= 2e-5
    assert float(z['eval_cond']) <= 5e-6          # the fixture is well-conditioned: eval parity is asserted at a flat 1e-4


def test_sequential_config1():
    import fgnn_amd
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
    idx, ef = tables.knn_table(30, 8)
    x = torch.from_numpy(z['x'])
    B = x.shape[0]
    with torch.no_grad():
        et = emodel(torch.from_numpy(ef)[None]).repeat(B, 1, 1, 1)
        for mode in ('eval', 'train'):
            sd = H.fill_state_dict(model.state_dict())
            y = O.fixed_pw_hop_net(sd, x, torch.from_numpy(idx)[None].repeat(B, 1, 1), et,
                                   training=(mode == 'train'))
            assert H.rel_err(y, torch.from_numpy(z[mode + '_y'])) <= 2e-5
