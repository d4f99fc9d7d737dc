"""Shared test utilities (no reference access: everything comes from tests/golden/)."""
import os

import numpy as np
import torch

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden')
CONTRACTING_GAIN = 0.75
AGGS = ['max', 'softmax', 'mean']
BNS = ['off', 'train', 'eval']


def load(name):
    return np.load(os.path.join(GOLDEN, name), allow_pickle=False)


def fill_state_dict(sd, gain=2.0):
    """Closed-form parameters — identical to oracle/make_golden.py::fill_state_dict, which
    produced the full-size assembly fixtures.  ``gain``: weights are hash values in (-1, 1) times sqrt(gain / fan_in);
    2.0 (LDPCModel, config 1) or CONTRACTING_GAIN (the factor_mpnn fixtures: see make_golden.py)."""
    out = {}
    for rank, key in enumerate(sorted(sd.keys())):
        t = sd[key]
        if not torch.is_floating_point(t):
            out[key] = t.clone()
            continue
        i = torch.arange(t.numel(), dtype=torch.float64)
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


class OpCase:
    """One record of tests/golden/operator.npz."""

    def __init__(self, z, row):
        (self.id, self.ext, agg, bn, bias, relu, self.B, self.nin, self.nou, self.net,
         self.N, self.M, self.k) = [int(v) for v in row]
        self.agg, self.bn, self.has_bias, self.relu = AGGS[agg], BNS[bn], bool(bias), bool(relu)
        pre = 'c%03d.' % self.id
        self.t = {k[len(pre):]: torch.from_numpy(z[k]) for k in z.files if k.startswith(pre)}

    def sd(self):
        return {k[3:]: v.clone() for k, v in self.t.items() if k.startswith('sd.')}

    def __repr__(self):
        return 'case%d(ext=%d,%s,bn=%s,B%d nin%d nou%d net%d N%d M%d k%d)' % (
            self.id, self.ext, self.agg, self.bn, self.B, self.nin, self.nou, self.net,
            self.N, self.M, self.k)


def operator_cases():
    z = load('operator.npz')
    return [OpCase(z, row) for row in z['meta']]


def rel_err(a, b):
    a, b = a.detach().double().cpu(), b.detach().double().cpu()
    return float((a - b).abs().max()) / max(1.0, float(b.abs().max())) if a.numel() else 0.0


SYN_TAGS = ['pw', 'hop', 'hop8']          # train_syn_pw_factor.py; train_syn_hop_factor.py at --hop_order 9 and 8


def syn_fill(sd):
    """The factor_mpnn fixtures' parameters: the closed-form fill at the contracting gain (make_golden.py)."""
    return fill_state_dict(sd, gain=CONTRACTING_GAIN)


def syn_setup(tag):
    """(hop_dim, pw_idx, pw_ef, hi_idx, hi_ef) of one synthetic-PGM configuration, from fgnn_amd.tables."""
    from fgnn_amd import tables
    pw_idx, pw_ef = tables.pw_factor_table(30)
    if tag == 'pw':
        hi_idx, hi_ef, _ = tables.chain_high_table(30, 9)
        return 1, pw_idx, pw_ef, hi_idx, hi_ef
    order = 9 if tag == 'hop' else 8
    hi_idx, hi_ef = tables.ring_hop_table(30, order)
    return order, pw_idx, pw_ef, hi_idx, hi_ef


def syn_edge_models(hi_ef):
    C = torch.nn.Conv2d
    em_pw = torch.nn.Sequential(C(3, 64, 1), torch.nn.ReLU(), C(64, 16, 1))
    em_hi = torch.nn.Sequential(C(hi_ef.shape[0], 64, 1), torch.nn.ReLU(), C(64, 16, 1))
    em_pw.load_state_dict(syn_fill(em_pw.state_dict()))
    em_hi.load_state_dict(syn_fill(em_hi.state_dict()))
    return em_pw, em_hi


def bn_spec_for(C, dev, seed=0):
    """A ``pointwise.bn_spec`` tuple (gamma, beta, running_mean, running_var, num_batches_tracked, momentum, eps) with fresh buffers:
    what a statistics-producing launch needs to finalise the BatchNorm behind it."""
    import torch
    g = torch.Generator().manual_seed(seed)
    return ((torch.rand(C, generator=g) + 0.5).to(dev), (torch.randn(C, generator=g) * 0.3).to(dev), torch.zeros(C, device=dev),
            torch.ones(C, device=dev), torch.zeros((), device=dev, dtype=torch.int64), 0.1, 1e-5)
