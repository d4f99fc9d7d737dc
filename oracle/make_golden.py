#!/usr/bin/env python3
"""Generate the golden vectors under tests/golden/ from the REAL reference.

TEST INFRASTRUCTURE ONLY — runs in the build container (where
/root/reference exists), never on the GPU box.  It

  1. imports the reference's ``lib/model/mpnn`` package straight from
     /root/reference (nothing is copied) and AST-extracts ``LDPCModel``, the
     ``generate_*_table`` helpers and ``ldpc_graph_structure_generator`` from
     the reference scripts (they cannot be imported whole: they need
     tensorboardX / ad3 / the compiled MNC extension),
  2. runs the reference on seeded inputs,
  3. checks ``oracle/fgnn_oracle.py`` against it (<= 1e-6; fails loudly),
  4. writes inputs + expected outputs (+ gradients) as small ``.npz`` files.

Fixtures are data only: tensors in, tensors out.  Re-run with
``python oracle/make_golden.py`` after changing the case list.
"""
import ast
import os
import sys

import numpy as np
import torch

REF = '/root/reference'
HERE = os.path.dirname(os.path.abspath(__file__))
OUT = os.path.join(os.path.dirname(HERE), 'tests', 'golden')
sys.path.insert(0, HERE)
sys.dont_write_bytecode = True

import fgnn_oracle as O  # noqa: E402

# InstanceNorm2d on the 1-element LDPC hyper-factor tensor [B,C,1,1].
# torch 1.0 (the reference's era) returned exactly 0 = (x-mean)/sqrt(0+eps); torch >= 1.9
# raises in F._verify_spatial_size, and if only that check is patched out torch 2.10's
# batch-norm kernel returns ~1e-4 rounding noise instead of 0.  The harness therefore
# defines the single-element case as the mathematically exact 0 (as the oracle and the
# product do) and otherwise calls the stock implementation.
_stock_instance_norm = torch.nn.functional.instance_norm


def _instance_norm_defined(input, *args, **kwargs):
    if input.dim() == 4 and input.shape[2] * input.shape[3] == 1:
        return torch.zeros_like(input)
    return _stock_instance_norm(input, *args, **kwargs)


torch.nn.functional.instance_norm = _instance_norm_defined
np.int = int                                                   # numpy>=1.24

sys.path.insert(0, os.path.join(REF, 'lib', 'model'))
import mpnn as R  # noqa: E402  (the reference package)


def extract(path, names, namespace):
    """exec selected top-level defs of a reference source file in ``namespace``."""
    src = open(path).read()
    tree = ast.parse(src)
    keep = [n for n in tree.body
            if isinstance(n, (ast.ClassDef, ast.FunctionDef)) and n.name in names]
    assert len(keep) == len(names), (path, names)
    mod = ast.Module(body=keep, type_ignores=[])
    exec(compile(mod, path, 'exec'), namespace)
    return namespace


def np_sd(sd):
    return {k: v.detach().cpu().numpy() for k, v in sd.items()}


def fill_state_dict(sd, gain=2.0):
    """Closed-form deterministic parameters: a function of (sorted key rank,
    flat index) only, so the GPU-box tests can regenerate them without the
    reference.  Mirrored verbatim in tests/helpers.py::fill_state_dict.
    ``gain``: weight matrices are hash values in (-1, 1) times sqrt(gain / fan_in)."""
    out = {}
    for rank, key in enumerate(sorted(sd.keys())):
        t = sd[key]
        if not torch.is_floating_point(t):
            out[key] = t.clone()
            continue
        n = t.numel()
        i = torch.arange(n, dtype=torch.float64)
        # GLSL-style hash in float64: well-spread pseudo-random values in (-1, 1)
        wave = torch.sin(12.9898 * i + 78.233 * rank) * 43758.5453
        wave = 2.0 * (wave - torch.floor(wave)) - 1.0
        if key.endswith('running_var'):
            v = 1.0 + 0.5 * wave.abs()
        elif key.endswith('running_mean'):
            v = 0.1 * wave
        elif t.dim() >= 2:
            fan_in = t.shape[1] if key.endswith('weight') else t.shape[0]
            if t.dim() == 4:
                fan_in = t.shape[1]
            v = wave * (gain / max(fan_in, 1)) ** 0.5
        elif key.endswith('bn.weight') or key.endswith('1.weight'):
            v = 1.0 + 0.1 * wave
        else:
            v = 0.1 * wave
        out[key] = v.reshape(t.shape).to(t.dtype)
    return out


def maxdiff(a, b):
    """max |a-b| relative to max(1, max|a|)."""
    a, b = a.detach(), b.detach()
    if not a.numel():
        return 0.0
    return float((a - b).abs().max()) / max(1.0, float(a.abs().max()))


def to64(obj):
    if torch.is_tensor(obj):
        return obj.detach().double() if obj.is_floating_point() else obj.detach()
    if isinstance(obj, dict):
        return {k: to64(v) for k, v in obj.items()}
    if isinstance(obj, (list, tuple)):
        return [to64(v) for v in obj]
    return obj


def conditioning(ref32, fn64):
    """rel. distance between the reference's f32 result and the same computation in f64:
    how much f32 rounding alone moves this output.  Tests accept the HIP path within a small
    multiple of it (batch-statistics BatchNorm over small batches amplifies rounding)."""
    with torch.no_grad():
        y64 = fn64()
    return maxdiff(ref32.detach().double(), y64)


EXT = {0: R.mp_conv_type.NO_EXTENSION, 1: R.mp_conv_type.ORIG_WITH_NEIGHBOR,
       2: R.mp_conv_type.ORIG_WITH_DIFF}


# --------------------------------------------------------------------------
# 1. operator fixtures
# --------------------------------------------------------------------------
def operator_cases():
    cases = []
    cid = 0
    # (B, nin, nou, net, N, M, k)
    shapes_noext = [(3, 5, 6, 4, 7, 5, 3), (1, 4, 3, 2, 6, 9, 2), (2, 3, 2, 1, 9, 1, 9),
                    (2, 8, 8, 1, 1, 10, 1), (2, 64, 64, 4, 96, 48, 6), (2, 2, 5, 3, 96, 1, 96)]
    shapes_ext = [(3, 5, 6, 4, 7, 7, 3), (1, 2, 7, 16, 6, 6, 8), (2, 6, 2, 3, 10, 10, 1),
                  (2, 64, 64, 16, 60, 60, 2)]
    for ext in (0, 1, 2):
        for agg in ('max', 'softmax', 'mean'):
            for bn in ('off', 'train', 'eval'):
                shapes = shapes_noext if ext == 0 else shapes_ext
                # full cross on the first shape, a thinner cross on the rest
                for si, shp in enumerate(shapes):
                    if si > 0 and not (bn == 'eval' or (bn == 'train' and agg == 'max')):
                        continue
                    big = shp[1] >= 64          # headline-sized shapes: keep few (file size)
                    if big and not ((agg == 'max' and bn == 'eval') or
                                    (agg == 'softmax' and bn == 'eval' and ext == 2)):
                        continue
                    cases.append(dict(id=cid, ext=ext, agg=agg, bn=bn, shape=shp,
                                      bias=(cid % 5 != 4), relu=(cid % 7 != 6)))
                    cid += 1
    return cases


def run_operator_case(c):
    B, nin, nou, net, N, M, k = c['shape']
    g = torch.Generator().manual_seed(1000 + c['id'])
    m = R.mp_conv_v2(nin, nou, net, bias=c['bias'], bn=(c['bn'] != 'off'),
                     extension=EXT[c['ext']],
                     activation_fn='relu' if c['relu'] else None, aggregtor=c['agg'])
    with torch.no_grad():
        m.filters.copy_(torch.randn(m.filters.shape, generator=g) * (1.0 / m.filters.shape[0]) ** 0.5)
        if m.bias is not None:
            m.bias.copy_(torch.randn(nou, generator=g) * 0.3)
        if m.bn is not None:
            m.bn.weight.copy_(1 + 0.3 * torch.randn(nou, generator=g))
            m.bn.bias.copy_(0.3 * torch.randn(nou, generator=g))
            m.bn.running_mean.copy_(0.2 * torch.randn(nou, generator=g))
            m.bn.running_var.copy_(0.5 + torch.rand(nou, generator=g))
    m.train(c['bn'] == 'train')
    sd0 = {k_: v.clone() for k_, v in m.state_dict().items()}
    x = torch.randn(B, nin, N, 1, generator=g, requires_grad=True)
    idx = torch.randint(0, N, (B, M, k), generator=g)
    if c['id'] % 3 == 0 and k > 1:          # force exact ties (duplicate neighbours)
        idx[:, :, -1] = idx[:, :, 0]
    et = torch.randn(B, net, M, k, generator=g, requires_grad=True)
    gy = torch.randn(B, nou, M, 1, generator=g)
    y = m(x, idx, et)
    y.backward(gy)
    # oracle check
    sdo = {k_: v.clone() for k_, v in sd0.items()}
    xo = x.detach().clone().requires_grad_(True)
    eo = et.detach().clone().requires_grad_(True)
    for p in ('filters', 'bias'):
        if p in sdo:
            sdo[p].requires_grad_(True)
    yo = O.mp_conv(sdo, '', xo, idx, eo, nou=nou, net=net, extension=c['ext'],
                   aggregator=c['agg'], training=(c['bn'] == 'train'), relu=c['relu'])
    yo.backward(gy)
    errs = [maxdiff(y, yo), maxdiff(x.grad, xo.grad), maxdiff(et.grad, eo.grad),
            maxdiff(m.filters.grad, sdo['filters'].grad)]
    if c['bn'] == 'train':
        errs.append(maxdiff(m.bn.running_var, sdo['bn.running_var']))
    assert max(errs) <= 1e-6, (c, errs)
    rec = {'x': x, 'idx': idx, 'etype': et, 'gy': gy, 'y': y, 'gx': x.grad,
           'getype': et.grad, 'gfilters': m.filters.grad}
    if m.bias is not None:
        rec['gbias'] = m.bias.grad
    if m.bn is not None:
        rec['gbn_weight'] = m.bn.weight.grad
        rec['gbn_bias'] = m.bn.bias.grad
        rec['post_running_mean'] = m.bn.running_mean
        rec['post_running_var'] = m.bn.running_var
    for k_, v in sd0.items():
        rec['sd.' + k_] = v
    return {k_: v.detach().numpy() for k_, v in rec.items()}, max(errs)


def make_operator():
    cases = operator_cases()
    blob = {}
    worst = 0.0
    meta = []
    for c in cases:
        rec, e = run_operator_case(c)
        worst = max(worst, e)
        for k_, v in rec.items():
            blob['c%03d.%s' % (c['id'], k_)] = v
        meta.append([c['id'], c['ext'], ['max', 'softmax', 'mean'].index(c['agg']),
                     ['off', 'train', 'eval'].index(c['bn']), int(c['bias']), int(c['relu'])]
                    + list(c['shape']))
    blob['meta'] = np.asarray(meta, dtype=np.int64)
    np.savez_compressed(os.path.join(OUT, 'operator.npz'), **blob)
    print('operator.npz: %d cases, oracle-vs-reference worst %.2e' % (len(cases), worst))


# --------------------------------------------------------------------------
# 2. residual-block fixtures
# --------------------------------------------------------------------------
def make_block():
    blob = {}
    meta = []
    worst = 0.0
    cid = 0
    for ext in (0, 2):
        for with_res in (True, False):
            for nout in (None, 10):
                if with_res and nout is not None:
                    continue
                for train in (False, True):
                    g = torch.Generator().manual_seed(2000 + cid)
                    nin, nmed, net, N, k, B = 6, 4, 3, 8, 3, 3
                    m = R.mp_conv_residual(nin, nmed, net, extension=EXT[ext],
                                           with_residual=with_res, aggregator='max', nout=nout)
                    m.load_state_dict(fill_state_dict(m.state_dict()))
                    m.train(train)
                    sd0 = {k_: v.clone() for k_, v in m.state_dict().items()}
                    x = torch.randn(B, nin, N, 1, generator=g, requires_grad=True)
                    idx = torch.randint(0, N, (B, N, k), generator=g)
                    et = torch.randn(B, net, N, k, generator=g, requires_grad=True)
                    y = m(x, idx, et)
                    gy = torch.randn(y.shape, generator=g)
                    y.backward(gy)
                    sdo = {k_: v.clone() for k_, v in sd0.items()}
                    yo = O.residual_block(sdo, '', x.detach(), idx, et.detach(), net=net,
                                          extension=ext, aggregator='max',
                                          with_residual=with_res, training=train)
                    e = maxdiff(y, yo)
                    assert e <= 1e-6, e
                    worst = max(worst, e)
                    pre = 'b%02d.' % cid
                    for k_, v in sd0.items():
                        blob[pre + 'sd.' + k_] = v.numpy()
                    for k_, v in dict(x=x, idx=idx, etype=et, gy=gy, y=y, gx=x.grad,
                                      getype=et.grad,
                                      gfilters=m.mp_conv.filters.grad,
                                      gconv1=m.conv1[0].weight.grad).items():
                        blob[pre + k_] = v.detach().numpy()
                    meta.append([cid, ext, int(with_res), -1 if nout is None else nout,
                                 int(train), nin, nmed, net])
                    cid += 1
    blob['meta'] = np.asarray(meta, dtype=np.int64)
    np.savez_compressed(os.path.join(OUT, 'block.npz'), **blob)
    print('block.npz: %d cases, worst %.2e' % (cid, worst))


# --------------------------------------------------------------------------
# 3. graph tables (restated by the build; pinned here)
# --------------------------------------------------------------------------
def make_tables():
    ns = {'np': np, 'torch': torch}
    extract(os.path.join(REF, 'train_syn_fixed_pw_hop.py'), ['generate_knn_table'], ns)
    knn_idx, knn_ef = ns['generate_knn_table'](30, 8)
    ns2 = {'np': np, 'torch': torch}
    extract(os.path.join(REF, 'train_syn_pw_factor.py'),
            ['generate_pw_factor_table', 'generate_high_factor_table'], ns2)
    pw_idx, pw_ef = ns2['generate_pw_factor_table'](30)
    hi_idx, hi_ef, hi_ff = ns2['generate_high_factor_table'](30, 9)
    ns3 = {'np': np, 'torch': torch}
    extract(os.path.join(REF, 'train_syn_hop_factor.py'), ['generate_high_factor_table'], ns3)
    hop9_idx, hop9_ef = ns3['generate_high_factor_table'](30, 9)
    hop8_idx, hop8_ef = ns3['generate_high_factor_table'](30, 8)
    ns4 = {'np': np, 'os': os, '__file__': os.path.join(REF, 'lib/data/ldpc_dataset.py')}
    extract(os.path.join(REF, 'lib/data/ldpc_dataset.py'), ['ldpc_graph_structure_generator'], ns4)
    gen = ns4['ldpc_graph_structure_generator']()
    y = np.random.RandomState(7).randn(96).astype(np.float32)
    hop, f2v, v2f, ef_f2v, ef_v2f = gen.get_mpnn_sp_structure(y)
    np.savez_compressed(
        os.path.join(OUT, 'tables.npz'),
        knn_idx=knn_idx.numpy(), knn_ef=knn_ef.numpy(),
        pw_idx=pw_idx.numpy(), pw_ef=pw_ef.numpy(),
        hi_idx=hi_idx.numpy(), hi_ef=hi_ef.numpy(), hi_ff=hi_ff.numpy(),
        hop9_idx=hop9_idx.numpy(), hop9_ef=hop9_ef.numpy(),
        hop8_idx=hop8_idx.numpy(), hop8_ef=hop8_ef.numpy(),
        ldpc_f2v=f2v.astype(np.int64), ldpc_v2f=v2f.astype(np.int64),
        ldpc_y=y, ldpc_hop=hop.astype(np.float32),
        ldpc_ef_f2v=ef_f2v.astype(np.float32), ldpc_ef_v2f=ef_v2f.astype(np.float32))
    print('tables.npz written')
    return dict(knn=(knn_idx, knn_ef), pw=(pw_idx, pw_ef), hi=(hi_idx, hi_ef, hi_ff),
                hop9=(hop9_idx, hop9_ef), hop8=(hop8_idx, hop8_ef), gen=gen)


# --------------------------------------------------------------------------
# 4. assemblies, full size, closed-form parameters
# --------------------------------------------------------------------------
def grad_digest(named_params):
    rows = []
    for name, p in sorted(named_params):
        if p.grad is None:
            rows.append([0.0, 0.0])
        else:
            rows.append([float(p.grad.double().sum()), float(p.grad.double().norm())])
    return np.asarray(rows, dtype=np.float64)


def ldpc_inputs(gen, B, seed):
    rs = np.random.RandomState(seed)
    nf, hf, f2v, v2f, ef1, ef2 = [], [], [], [], [], []
    for b in range(B):
        y = rs.randn(96).astype(np.float32)
        hop, a, c, e1, e2 = gen.get_mpnn_sp_structure(y)
        snr = np.float32(rs.randint(0, 5))
        nf.append(np.stack([y, np.full(96, snr, np.float32)], 0)[:, :, None])
        hf.append(hop.astype(np.float32).T[:, :, None])          # [6,48,1]
        f2v.append(a); v2f.append(c)
        ef1.append(np.transpose(e1, (2, 0, 1)))                  # [7,96,3]
        ef2.append(np.transpose(e2, (2, 0, 1)))                  # [7,48,6]
    t = lambda a, dt=np.float32: torch.from_numpy(np.stack(a).astype(dt))
    return (t(nf), t(hf), t(f2v, np.int64), t(v2f, np.int64), t(ef1), t(ef2))


def make_ldpc(tabs):
    ns = {'np': np, 'torch': torch, 'FactorNN': R.FactorNN}
    extract(os.path.join(REF, 'train_ldpc.py'), ['LDPCModel'], ns)
    model = ns['LDPCModel'](2, 6, 4, aggregator='max')
    model.load_state_dict(fill_state_dict(model.state_dict()))
    inputs = ldpc_inputs(tabs['gen'], 4, 11)
    blob = {'in%d' % i: v.numpy() for i, v in enumerate(inputs)}
    # eval
    model.eval()
    with torch.no_grad():
        logits, snr = model(*inputs)
        sdo = {k: v.clone() for k, v in model.state_dict().items()}
        lo, so = O.ldpc_model(sdo, *inputs, training=False)
    e = max(maxdiff(logits, lo), maxdiff(snr, so))
    assert e <= 1e-5, e
    blob['eval_logits'], blob['eval_snr'] = logits.numpy(), snr.numpy()
    blob['eval_cond'] = np.float64(conditioning(
        logits, lambda: O.ldpc_model(to64(sdo), *to64(list(inputs)), training=False)[0]))
    # train fwd + bwd on a larger batch: batch-statistics BatchNorm over 4 codewords is
    # ill-conditioned (f32-vs-f64 of the same model differ by 1.5e-3 at B=4, 2e-4 at B=16)
    inputs = ldpc_inputs(tabs['gen'], 16, 12)
    for i, v in enumerate(inputs):
        blob['tin%d' % i] = v.numpy()
    model.train()
    sdo = {k: v.clone() for k, v in model.state_dict().items()}
    logits, snr = model(*inputs)
    tgt = (torch.arange(16 * 48).reshape(16, 48) % 3 == 0).float()
    loss = torch.nn.functional.binary_cross_entropy_with_logits(logits.view(-1), tgt.view(-1)) \
        + 0.1 * torch.nn.functional.mse_loss(snr.view(-1), torch.ones(16))
    loss.backward()
    lo, so = O.ldpc_model(sdo, *inputs, training=True)
    e2 = max(maxdiff(logits, lo), maxdiff(snr, so))
    assert e2 <= 1e-5, e2
    blob['train_logits'], blob['train_snr'] = logits.detach().numpy(), snr.detach().numpy()
    sd_pre = fill_state_dict(model.state_dict())
    blob['train_cond'] = np.float64(conditioning(
        logits, lambda: O.ldpc_model(to64(sd_pre), *to64(list(inputs)), training=True)[0]))
    blob['train_loss'] = np.float64(loss.item())
    names = [n for n, _ in sorted(model.named_parameters())]
    blob['grad_digest'] = grad_digest(model.named_parameters())
    blob['param_names'] = np.asarray(names)
    blob['sd_keys'] = np.asarray(sorted(model.state_dict().keys()))
    blob['sd_shapes'] = np.asarray([str(tuple(model.state_dict()[k].shape))
                                    for k in sorted(model.state_dict().keys())])
    blob['post_rv_digest'] = np.asarray(
        [float(v.double().sum()) for k, v in sorted(model.state_dict().items())
         if k.endswith('running_var')])
    np.savez_compressed(os.path.join(OUT, 'ldpc_model.npz'), **blob)
    print('ldpc_model.npz: eval err %.2e train err %.2e, %d state entries; cond eval %.2e train %.2e' %
          (e, e2, len(model.state_dict()), blob['eval_cond'], blob['train_cond']))


# The 11-layer synthetic-PGM stack alternates message blocks with InstanceNorm layers; with the unit-gain fill used for
# the other fixtures (gain 2.0, He-style) its EVAL output moves by 3.5e-5 / 5.2e-5 (relative) between the reference's
# own f32 run and an f64 run of the same maths — too close to the 1e-4 north-star tolerance to assert it flat.  A
# slightly contracting fill (gain 0.75: per-layer gain just below one) brings that distance to ~2e-6 with outputs and
# factor features still O(0.2 .. 3), so the eval-mode tests assert 1e-4 with no conditioning allowance.
CONTRACTING_GAIN = 0.75


def make_factor_mpnn(tabs):
    fill = lambda sd: fill_state_dict(sd, gain=CONTRACTING_GAIN)
    for tag, hop_dim, (hidx, hef) in (('pw', 1, tabs['hi'][:2]), ('hop', 9, tabs['hop9']), ('hop8', 8, tabs['hop8'])):
        g = torch.Generator().manual_seed({'pw': 31, 'hop': 32, 'hop8': 33}[tag])
        B = 3
        model = R.factor_mpnn(2, [4, hop_dim], O.SYN_DIMS, [16, 16])
        emodel_pw = torch.nn.Sequential(torch.nn.Conv2d(3, 64, 1), torch.nn.ReLU(inplace=True),
                                        torch.nn.Conv2d(64, 16, 1))
        emodel_hi = torch.nn.Sequential(torch.nn.Conv2d(hef.shape[1], 64, 1),
                                        torch.nn.ReLU(inplace=True), torch.nn.Conv2d(64, 16, 1))
        model.load_state_dict(fill(model.state_dict()))
        emodel_pw.load_state_dict(fill(emodel_pw.state_dict()))
        emodel_hi.load_state_dict(fill(emodel_hi.state_dict()))
        pw_idx, pw_ef = tabs['pw']
        nfeature = torch.rand(B, 2, 30, 1, generator=g)
        pws = torch.rand(B, 4, 30, 1, generator=g)
        if tag == 'pw':
            hi_feat = tabs['hi'][2].repeat(B, 1, 1, 1)
        else:
            hi_feat = torch.rand(B, hop_dim, 30, 1, generator=g)
        blob = dict(nfeature=nfeature.numpy(), pws=pws.numpy(), hi_feat=hi_feat.numpy())
        for mode in ('eval', 'train'):
            model.train(mode == 'train')
            sdo = {k: v.clone() for k, v in model.state_dict().items()}
            et_pw = emodel_pw(pw_ef)
            et_hi = emodel_hi(hef)
            gs = [[pw_idx.repeat(B, 1, 1), et_pw.repeat(B, 1, 1, 1)],
                  [hidx.repeat(B, 1, 1), et_hi.repeat(B, 1, 1, 1)]]
            with torch.set_grad_enabled(mode == 'train'):
                pred, ff = model(nfeature, [pws, hi_feat], gs)
            with torch.no_grad():
                po, fo = O.factor_mpnn(sdo, '', nfeature, [pws, hi_feat],
                                       [[a, b.detach()] for a, b in gs],
                                       dims=O.SYN_DIMS, netypes=[16, 16],
                                       training=(mode == 'train'))
            e = maxdiff(pred, po)
            assert e <= 2e-5, (tag, mode, e)
            blob[mode + '_pred'] = pred.detach().numpy()
            sd_pre = fill(model.state_dict())
            blob[mode + '_cond'] = np.float64(conditioning(
                pred, lambda: O.factor_mpnn(to64(sd_pre), '', to64(nfeature), to64([pws, hi_feat]),
                                            [[a, to64(b)] for a, b in gs], dims=O.SYN_DIMS,
                                            netypes=[16, 16], training=(mode == 'train'))[0]))
            blob[mode + '_ff0'] = ff[0].detach().numpy()
            blob[mode + '_ff1'] = ff[1].detach().numpy()
            if mode == 'train':
                lab = (torch.arange(B * 30) % 2).long()
                loss = torch.nn.functional.cross_entropy(
                    pred.squeeze(-1).permute(0, 2, 1).reshape(-1, 2), lab)
                loss.backward()
                blob['train_loss'] = np.float64(loss.item())
                blob['grad_digest'] = grad_digest(
                    list(model.named_parameters())
                    + [('emodel_pw.' + n, p) for n, p in emodel_pw.named_parameters()]
                    + [('emodel_hi.' + n, p) for n, p in emodel_hi.named_parameters()])
            print('factor_mpnn_%s %s: oracle err %.2e cond %.2e' % (tag, mode, e, blob[mode + '_cond']))
        np.savez_compressed(os.path.join(OUT, 'factor_mpnn_%s.npz' % tag), **blob)


def make_sequential(tabs):
    g = torch.Generator().manual_seed(41)
    B = 4
    T = R.mp_conv_type
    model = R.mp_sequential(
        R.mp_conv_v2(2, 64, 16, extension=T.ORIG_WITH_NEIGHBOR),
        R.mp_conv_residual(64, 64, 16), torch.nn.Conv2d(64, 128, 1),
        torch.nn.BatchNorm2d(128), torch.nn.ReLU(inplace=True),
        R.mp_conv_residual(128, 64, 16), torch.nn.Conv2d(128, 256, 1),
        torch.nn.BatchNorm2d(256), torch.nn.ReLU(inplace=True),
        R.mp_conv_residual(256, 64, 16), torch.nn.Conv2d(256, 128, 1),
        torch.nn.BatchNorm2d(128), torch.nn.ReLU(inplace=True),
        R.mp_conv_residual(128, 64, 16), torch.nn.Conv2d(128, 64, 1),
        torch.nn.BatchNorm2d(64), torch.nn.ReLU(inplace=True),
        R.mp_conv_residual(64, 64, 16), torch.nn.Conv2d(64, 2, 1))
    emodel = torch.nn.Sequential(torch.nn.Conv2d(1, 64, 1), torch.nn.ReLU(inplace=True),
                                 torch.nn.Conv2d(64, 16, 1))
    model.load_state_dict(fill_state_dict(model.state_dict()))
    emodel.load_state_dict(fill_state_dict(emodel.state_dict()))
    idx, ef = tabs['knn']
    x = torch.rand(B, 2, 30, 1, generator=g)
    blob = dict(x=x.numpy())
    for mode in ('eval', 'train'):
        model.train(mode == 'train')
        sdo = {k: v.clone() for k, v in model.state_dict().items()}
        with torch.no_grad():
            et = emodel(ef)
            y = model(x, idx.repeat(B, 1, 1), et.repeat(B, 1, 1, 1))
            yo = O.fixed_pw_hop_net(sdo, x, idx.repeat(B, 1, 1), et.repeat(B, 1, 1, 1),
                                    training=(mode == 'train'))
        e = maxdiff(y, yo)
        assert e <= 2e-5, (mode, e)
        blob[mode + '_y'] = y.numpy()
        sd_pre = fill_state_dict(model.state_dict())
        blob[mode + '_cond'] = np.float64(conditioning(
            y, lambda: O.fixed_pw_hop_net(to64(sd_pre), to64(x), idx.repeat(B, 1, 1),
                                          to64(et.repeat(B, 1, 1, 1)), training=(mode == 'train'))))
        print('mp_sequential cfg1 %s: oracle err %.2e cond %.2e' % (mode, e, blob[mode + '_cond']))
    np.savez_compressed(os.path.join(OUT, 'sequential_cfg1.npz'), **blob)


if __name__ == '__main__':
    os.makedirs(OUT, exist_ok=True)
    torch.manual_seed(0)
    torch.set_num_threads(8)
    only = set(sys.argv[1:])                  # e.g. `make_golden.py factor_mpnn`: regenerate one family
    want = lambda name: not only or name in only
    if want('operator'):
        make_operator()
    if want('block'):
        make_block()
    tabs = make_tables()
    if want('ldpc'):
        make_ldpc(tabs)
    if want('factor_mpnn'):
        make_factor_mpnn(tabs)
    if want('sequential'):
        make_sequential(tabs)
    print('done ->', OUT)
