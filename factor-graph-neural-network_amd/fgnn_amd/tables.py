"""Static factor-graph tables of the four reference workloads (incidence + raw edge features).

Restated from the reference's generators and pinned bit-for-bit by tests/golden/tables.npz:
  knn_table            train_syn_fixed_pw_hop.py:86-101   (config 1; fills 7 of 8 slots — kept)
  pw_factor_table      train_syn_pw_factor.py:113-133
  chain_high_table     train_syn_pw_factor.py:136-156     (one extra all-self "factor" row)
  ring_hop_table       train_syn_hop_factor.py:135-151
  LdpcGraph            lib/data/ldpc_dataset.py:11-106    (MacKay 96.3.963 code, alist incidence)
All functions return numpy arrays without the leading batch axis.
"""
import os

import numpy as np

_DATA = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'data')


def _clamp(v, n):
    return min(max(v, 0), n - 1)


def knn_table(n, k):
    """nn_idx [n,k] int64, efeature [1,n,k]: chain neighbours i-h..i-1, i+1..i+h-1 (h=k//2)."""
    idx = np.zeros((n, k), np.int64)
    ef = np.zeros((1, n, k), np.float32)
    h = k // 2
    for i in range(n):
        offsets = list(range(-h, 0)) + list(range(1, h))
        for slot, off in enumerate(offsets):
            j = _clamp(i + off, n)
            idx[i, slot] = j
            ef[0, i, slot] = i - j
    return idx, ef


def pw_factor_table(n):
    """Variables 0..n-1 and pairwise factors n..2n-1 on a ring; k = 2.
    Variable i listens to factors (i-1)%n and i; factor i listens to variables i and (i+1)%n."""
    idx = np.zeros((2 * n, 2), np.int64)
    ef = np.zeros((3, 2 * n, 2), np.float32)
    for i in range(n):
        for slot, f in enumerate(((i - 1) % n, i)):
            idx[i, slot] = n + f
            ef[0, i, slot] = 1
            ef[2, i, slot] = (i - f + 0.5) * 2
        for slot, v in enumerate((i, (i + 1) % n)):
            idx[n + i, slot] = v
            ef[1, n + i, slot] = 1
            ef[2, n + i, slot] = (i - v + 0.5) * 2
    return idx, ef


def chain_high_table(n, k):
    """nn_idx [n+1,k'], efeature [1,n+1,k'], factor_feature [1,1,1] with k' = k | 1.
    Row i < n lists the clamped window i-h..i+h-1 (last slot stays 0); row n points at itself."""
    k = k + 1 if k % 2 == 0 else k
    idx = np.zeros((n + 1, k), np.int64)
    ef = np.zeros((1, n + 1, k), np.float32)
    h = k // 2
    for i in range(n):
        for slot, off in enumerate(range(-h, h)):
            j = _clamp(i + off, n)
            idx[i, slot] = j
            ef[0, i, slot] = i - j
    idx[n, :] = n
    return idx, ef, np.zeros((1, 1, 1), np.float32)


def ring_hop_table(n, k):
    """Variables 0..n-1 and order-k hop factors n..2n-1 on a ring: nn_idx [2n,k], efeature [2,2n,k]."""
    idx = np.zeros((2 * n, k), np.int64)
    ef = np.zeros((2, 2 * n, k), np.float32)
    h = k >> 1
    for i in range(n):
        for slot in range(k):
            v = (i + slot - h + n) % n
            idx[i, slot] = v + n
            ef[0, i, slot] = 1
            idx[n + i, slot] = v
            ef[1, n + i, slot] = 1
    return idx, ef


class LdpcGraph:
    """Incidence of the (96,48) rate-1/2 regular LDPC code 96.3.963: every variable sits in 3
    parity checks, every check touches 6 variables (288 edges)."""
    n_var, n_chk, deg_var, deg_chk = 96, 48, 3, 6

    def __init__(self):
        z = np.load(os.path.join(_DATA, 'ldpc_96_3_963.npz'))
        self.var_to_factors = z['var_to_factors'].astype(np.int64)     # [96,3] == nn_idx_f2v
        self.factor_to_vars = z['factor_to_vars'].astype(np.int64)     # [48,6] == nn_idx_v2f

    def features(self, y, snr_db):
        """Per-codeword model inputs from a received word y[96] (ldpc_dataset.py:92-106,222-236).
        Returns node_feature [2,96,1], hop_feature [6,48,1], efeature_f2v [7,96,3],
        efeature_v2f [7,48,6]."""
        y = np.asarray(y, np.float32)
        hop = y[self.factor_to_vars]                                   # [48,6]
        ef_f2v = np.concatenate([hop[self.var_to_factors],             # [96,3,6]
                                 np.broadcast_to(y[:, None, None], (96, 3, 1))], axis=2)
        ef_v2f = np.concatenate([np.broadcast_to(hop[:, None, :], (48, 6, 6)),
                                 hop[:, :, None]], axis=2)
        node = np.stack([y, np.full(96, snr_db, np.float32)], 0)[:, :, None]
        return (node.astype(np.float32), hop.T[:, :, None].astype(np.float32),
                np.ascontiguousarray(ef_f2v.transpose(2, 0, 1), np.float32),
                np.ascontiguousarray(ef_v2f.transpose(2, 0, 1), np.float32))
