"""Shared test utilities (no reference access: everything comes from tests/golden/)."""
import os

import numpy as np
import torch

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden')
AGGS = ['max', 'softmax', 'mean']
BNS = ['off', 'train', 'eval']


def load(name):
    return np.load(os.path.join(GOLDEN, name), allow_pickle=False)


def fill_state_dict(sd):
    """Closed-form parameters — identical to oracle/make_golden.py::fill_state_dict, which
    produced the full-size assembly fixtures."""
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
            v = wave * (2.0 / max(fan_in, 1)) ** 0.5
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
