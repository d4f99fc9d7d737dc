"""The LDPC data path in front of the decoder, per batch on the GPU.

The reference builds every training item on the host: ``gen_data_item`` (/root/reference/lib/data/ldpc.py:7-30)
calls the pybind11 MNC library for the GF(2) encode ``s2t`` and the channel ``t2y``
(lib/data/MNC/MNC_py.cpp:22-108), then ``ContinousCodesSP.__getitem__``
(lib/data/ldpc_dataset.py:222-236) assembles the model inputs with numpy takes.  ``LdpcDataPath`` produces a whole
batch of the same eight arrays with two kernels (csrc/ldpc_datapath.hip) in the dtype the model kernels read.
No CPU fallback: the arrays are produced where they are consumed.
"""
import os

import numpy as np
import torch

from . import _hip
from .tables import _DATA, LdpcGraph


class LdpcDataPath:
    """96.3.963 code: 48 message bits, 48 parity bits (codeword = [s | G s]), 3 checks per variable, 6 variables
    per check.  ``sample`` mirrors ``ContinousCodesSP.__getitem__`` over a batch."""
    K, P = 48, 48
    sigma_b_choices = (0, 1, 2, 3, 4, 5)            # ldpc_dataset.py:212
    snr_db_choices = (0, 1, 2, 3, 4)                # ldpc_dataset.py:216

    def __init__(self, device):
        self.device = torch.device(device)
        if self.device.type != 'cuda':
            raise RuntimeError('LdpcDataPath runs on a ROCm device (no CPU fallback)')
        g = LdpcGraph()
        z = np.load(os.path.join(_DATA, 'ldpc_96_3_963_G.npz'))
        G = z['G'].astype(np.uint64)                                                            # [P, K]
        self._decode_tables = self._incidence_tables(z['A2_nlist'], 48)
        masks = (G << np.arange(self.K, dtype=np.uint64)[None, :]).sum(1).astype(np.uint64)
        self.gmask = torch.from_numpy(masks.view(np.int64)).to(self.device)
        self.var_to_factors = torch.from_numpy(g.var_to_factors.astype(np.int32)).to(self.device)
        self.factor_to_vars = torch.from_numpy(g.factor_to_vars.astype(np.int32)).to(self.device)
        self.nn_idx_f2v = torch.from_numpy(g.var_to_factors).to(self.device)
        self.nn_idx_v2f = torch.from_numpy(g.factor_to_vars).to(self.device)

    def encode(self, s):
        """s [B,48] (any integer / bool dtype, values 0/1) -> codewords [B,96] uint8 = [s | G s mod 2]."""
        if s.dim() != 2 or s.shape[1] != self.K:
            raise ValueError('messages must be [B, %d], got %s' % (self.K, tuple(s.shape)))
        s = s.to(self.device, torch.uint8).contiguous()
        B = s.shape[0]
        cw = torch.empty((B, self.K + self.P), device=self.device, dtype=torch.uint8)
        _hip.check(_hip.lib().fgnn_ldpc_encode(_hip._ptr(s), _hip._ptr(self.gmask), B, self.K, self.P, _hip._ptr(cw),
                                               _hip.stream_ptr()))
        return cw

    def channel_features(self, cw, snr_db, sigma_b, burst_prob=0.05, noise=None, generator=None,
                         dtype=torch.float32, kernel_rng=None):
        """Received words + model inputs for codewords cw [B,96].  ``noise`` = (z1, u, z2) [B,96] f32 draws
        (standard normal, uniform [0,1), standard normal); drawn from ``generator`` when None.  ``kernel_rng`` = (seed,
        offset): no noise tensors at all — the kernel draws them itself (Philox4x32-10 keyed by seed, counter = (bit index,
        offset); csrc/ldpc_datapath.hip, restated in oracle/fgnn_oracle.py::philox_channel_draws).
        Returns (y [B,96] f32, node_feature [B,2,96,1], hop_feature [B,6,48,1], efeature_f2v [B,7,96,3],
        efeature_v2f [B,7,48,6])."""
        if dtype not in (torch.float32, torch.bfloat16):
            raise ValueError('dtype must be float32 or bfloat16')
        B, N = cw.shape
        dev = self.device
        cw = cw.to(dev, torch.uint8).contiguous()
        snr_db = snr_db.to(dev, torch.float32).contiguous()
        sigma_b = sigma_b.to(dev, torch.float32).contiguous()
        if kernel_rng is not None:
            z1 = u = z2 = None
        elif noise is None:
            z1 = torch.randn((B, N), device=dev, generator=generator)
            u = torch.rand((B, N), device=dev, generator=generator)
            z2 = torch.randn((B, N), device=dev, generator=generator)
        else:
            z1, u, z2 = [t.to(dev, torch.float32).contiguous() for t in noise]
        y = torch.empty((B, N), device=dev, dtype=torch.float32)
        node = torch.empty((B, 2, 96, 1), device=dev, dtype=dtype)
        hop = torch.empty((B, 6, 48, 1), device=dev, dtype=dtype)
        ef_f2v = torch.empty((B, 7, 96, 3), device=dev, dtype=dtype)
        ef_v2f = torch.empty((B, 7, 48, 6), device=dev, dtype=dtype)
        P = _hip._ptr
        if kernel_rng is not None:
            seed, offset = (int(v) & 0xFFFFFFFFFFFFFFFF for v in kernel_rng)
            _hip.check(_hip.lib().fgnn_ldpc_channel_features_rng(
                P(cw), P(snr_db), P(sigma_b), float(burst_prob), seed, offset, P(self.var_to_factors), P(self.factor_to_vars),
                B, 96, 48, 3, 6, _hip.dtype_code(node), P(y), P(node), P(hop), P(ef_f2v), P(ef_v2f), _hip.stream_ptr()))
            return y, node, hop, ef_f2v, ef_v2f
        _hip.check(_hip.lib().fgnn_ldpc_channel_features(
            P(cw), P(snr_db), P(sigma_b), float(burst_prob), P(z1), P(u), P(z2), P(self.var_to_factors),
            P(self.factor_to_vars), B, 96, 48, 3, 6, _hip.dtype_code(node), P(y), P(node), P(hop), P(ef_f2v), P(ef_v2f),
            _hip.stream_ptr()))
        return y, node, hop, ef_f2v, ef_v2f

    def _incidence_tables(self, nlist, nchk):
        """alist column lists (each variable's checks in file order, -1 = padding) -> the device tables of
        `fgnn_ldpc_decode`: col_ptr, row_ptr, row_edge, row_var."""
        cols = [[int(m) for m in row if m >= 0] for row in nlist]
        col_ptr = np.concatenate([[0], np.cumsum([len(c) for c in cols])]).astype(np.int32)
        rows = [[] for _ in range(nchk)]
        for n, c in enumerate(cols):
            for u, m in enumerate(c):
                rows[m].append((int(col_ptr[n]) + u, n))
        if max(len(c) for c in cols) > 16 or max(len(r) for r in rows) > 16:
            raise ValueError('at most 16 checks per variable / variables per check')
        row_ptr = np.concatenate([[0], np.cumsum([len(r) for r in rows])]).astype(np.int32)
        row_edge = np.array([e for r in rows for e, _ in r], np.int32)
        row_var = np.array([n for r in rows for _, n in r], np.int32)
        dev = lambda a: torch.from_numpy(a).to(self.device)
        return dev(col_ptr), dev(row_ptr), dev(row_edge), dev(row_var), len(cols), nchk, int(col_ptr[-1])

    def bit_prior(self, y, snr_db):
        """`y2b` (MNC_py.cpp:104-108): P(bit = 1 | y) = 1 / (1 + exp(-2 gcx y)), float64."""
        gcx = torch.pow(10.0, snr_db.to(self.device, torch.float64) / 20.0)[:, None]
        return 1.0 / (1.0 + torch.exp(-2.0 * gcx * y.to(self.device, torch.float64)))

    def decode(self, bias, loops=100, want_posteriors=False):
        """The reference's sum-product baseline `zb2x(bias, 48, 48, A2, 1, loops)` (lib/data/ldpc.py:18-24) for a
        batch: bias [B,96] float64 = P(bit = 1) (see ``bit_prior``).  Returns (x [B,96] uint8 hard decisions — the
        message is x[:, :48] —, violated checks [B] int32, iterations [B] int32[, q1 [B,96] float64])."""
        col_ptr, row_ptr, row_edge, row_var, N, M, E = self._decode_tables
        if bias.dim() != 2 or bias.shape[1] != N:
            raise ValueError('bias must be [B, %d], got %s' % (N, tuple(bias.shape)))
        bias = bias.to(self.device, torch.float64).contiguous()
        B = bias.shape[0]
        x = torch.empty((B, N), device=self.device, dtype=torch.uint8)
        q1 = torch.empty((B, N), device=self.device, dtype=torch.float64) if want_posteriors else None
        viol = torch.empty((B,), device=self.device, dtype=torch.int32)
        iters = torch.empty((B,), device=self.device, dtype=torch.int32)
        P = _hip._ptr
        _hip.check(_hip.lib().fgnn_ldpc_decode(P(bias), P(col_ptr), P(row_ptr), P(row_edge), P(row_var), B, N, M, E,
                                               int(loops), P(x), P(q1), P(viol), P(iters), _hip.stream_ptr()))
        return (x, viol, iters, q1) if want_posteriors else (x, viol, iters)

    def sample(self, B, seed=0, dtype=torch.float32, snr_db=None, burst_prob=0.05, kernel_rng=False, step=0):
        """A batch of B training items (ldpc_dataset.py:222-236): random messages, encoded, sent through the
        channel at a per-item SNR drawn from ``snr_db_choices`` (or the fixed ``snr_db``) and burst level from
        ``sigma_b_choices``.  Returns (node_feature, hop_feature, nn_idx_f2v [B,96,3], nn_idx_v2f [B,48,6],
        efeature_f2v, efeature_v2f, label [B,96] int64 = the transmitted codeword, sigma_b [B])."""
        gen = torch.Generator(device=self.device).manual_seed(seed)
        s = torch.randint(0, 2, (B, self.K), device=self.device, generator=gen, dtype=torch.uint8)
        cw = self.encode(s)
        pick = lambda choices: torch.tensor(choices, device=self.device, dtype=torch.float32)[
            torch.randint(0, len(choices), (B,), device=self.device, generator=gen)]
        snr = pick(self.snr_db_choices) if snr_db is None else torch.full((B,), float(snr_db), device=self.device)
        sigma_b = pick(self.sigma_b_choices)
        # kernel_rng: the channel's draws are made inside the feature kernel from (seed, step) instead of three noise tensors
        _, node, hop, ef_f2v, ef_v2f = self.channel_features(cw, snr, sigma_b, burst_prob, generator=gen, dtype=dtype,
                                                             kernel_rng=(seed, step) if kernel_rng else None)
        return (node, hop, self.nn_idx_f2v.unsqueeze(0).expand(B, -1, -1), self.nn_idx_v2f.unsqueeze(0).expand(B, -1, -1),
                ef_f2v, ef_v2f, cw.long(), sigma_b)
