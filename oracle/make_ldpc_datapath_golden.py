"""TEST INFRASTRUCTURE.  Generates tests/golden/ldpc_datapath.npz and the packaged generator matrix
factor-graph-neural-network_amd/fgnn_amd/data/ldpc_96_3_963_G.npz from the REFERENCE itself:

  * G (48x48 over GF(2)) is read from /root/reference/ldpc_codes/96.3.963/G by the reference's own
    `mod2mat_read` (lib/data/MNC/radford/mod2mat.cpp:140) through oracle/_ref/libmod2mat_ref.so;
  * 64 seeded 48-bit messages are encoded by the reference's `mod2mat_multiply` exactly as `s2t(..., smn=True)`
    does (lib/data/MNC/MNC_py.cpp:22-83): codeword = [s | G s];
  * H_A2 is the parity-check matrix of ldpc_codes/96.3.963/A2, the one the reference pairs with this G (its
    sum-product baseline decodes with it): every encoded word has zero syndrome under it.  (The regular 96.3.963
    incidence lists the FGNN runs on have rank 46; A2 patches three rows to make the code systematic.)
  * the reference's sum-product baseline decoder (`zb2x` -> `bndecode`, lib/data/MNC/bnd/bnd.cpp) is compiled from
    its own sources into oracle/_ref/libbnd_ref.so and run on 96 received words: hard decisions, pseudo-posteriors
    (float64, bit-for-bit), violated checks and iteration counts are stored, and the pure-Python restatement
    `ldpc_sum_product` is asserted IDENTICAL to it on all of them;
  * the channel `t2y` (MNC_py.cpp:86-102): its ARITHMETIC is stored here for fixed seeded noise draws via the float64
    restatement in oracle/fgnn_oracle.py (`ldpc_channel`); its random stream is pinned separately, against the reference's own
    module, by oracle/make_t2y_golden.py.

Run here (needs /root/reference and `sh oracle/build_ref.sh`); the outputs are data and are committed.
"""
import ctypes
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'oracle'))
import fgnn_oracle as O                                                    # noqa: E402

REF = os.environ.get('FGNN_REFERENCE', '/root/reference')
GFILE = os.path.join(REF, 'ldpc_codes/96.3.963/G').encode()


def main():
    L = ctypes.CDLL(os.path.join(ROOT, 'oracle/_ref/libmod2mat_ref.so'))
    r, c = ctypes.c_int(), ctypes.c_int()
    assert L.ref_G_dims(GFILE, ctypes.byref(r), ctypes.byref(c)) == 0 and (r.value, c.value) == (48, 48)
    G = np.zeros((48, 48), np.uint8)
    assert L.ref_G_bits(GFILE, G.ctypes.data_as(ctypes.c_void_p)) == 0
    rng = np.random.default_rng(963)
    s = rng.integers(0, 2, (64, 48)).astype(np.uint8)
    s[0] = 0
    s[1] = 1
    s[2:50] = np.eye(48, dtype=np.uint8)                                   # unit messages: the columns of G
    cw = np.zeros((64, 96), np.uint8)
    for i in range(64):
        assert L.ref_encode(GFILE, s[i].ctypes.data_as(ctypes.c_void_p), 48, 48,
                            cw[i].ctypes.data_as(ctypes.c_void_p)) == 0
    assert np.array_equal(O.ldpc_encode(G, s), cw), 'numpy restatement disagrees with the reference encoder'
    # the parity-check matrix the reference decodes these codewords with (`zb2x(..., Afile=.../A2)`,
    # lib/data/ldpc.py:20-24): alist text, variables' check lists 1-based, 0 = padding
    rows = open(os.path.join(REF, 'ldpc_codes/96.3.963/A2')).read().split('\n')
    n, m = map(int, rows[0].split())
    H_A2 = np.zeros((m, n), np.uint8)
    for v in range(n):
        for f in map(int, rows[4 + v].split()):
            if f > 0:
                H_A2[f - 1, v] = 1
    assert not ((cw.astype(np.int64) @ H_A2.T.astype(np.int64)) % 2).any()
    A2_nlist = np.full((n, 4), -1, np.int32)           # each variable's checks in the file's order (the decoder's u order)
    for v in range(n):
        for u, f in enumerate(map(int, rows[4 + v].split())):
            if f > 0:
                A2_nlist[v, u] = f - 1
    snr_db = rng.integers(0, 5, 64).astype(np.float64)
    sigma_b = rng.integers(0, 6, 64).astype(np.float64)
    z1, u, z2 = rng.standard_normal((64, 96)), rng.random((64, 96)), rng.standard_normal((64, 96))
    y = O.ldpc_channel(cw, snr_db, sigma_b, 0.05, z1, u, z2)
    # the reference's sum-product baseline (`zb2x(y2b(y), 48, 48, A2, 1, 100)`, lib/data/ldpc.py:18-24) on these 64
    # received words and on 32 burst-free ones at 4 dB (so that early successes are covered), by its compiled code
    Lb = ctypes.CDLL(os.path.join(ROOT, 'oracle/_ref/libbnd_ref.so'))
    Lb.ref_bnd_decode.restype = ctypes.c_int
    afile = os.path.join(REF, 'ldpc_codes/96.3.963/A2').encode()
    y_easy = O.ldpc_channel(cw[:32], np.full(32, 4.0), np.zeros(32), 0.05, z1[:32], u[:32], z2[:32])
    dec_bias = np.concatenate([O.ldpc_bit_prior(y, snr_db), O.ldpc_bit_prior(y_easy, np.full(32, 4.0))])
    dec_x = np.zeros((96, 96), np.uint8)
    dec_q1 = np.zeros((96, 96), np.float64)
    dec_viol = np.zeros(96, np.int32)
    dec_loops = np.zeros(96, np.int32)
    for i in range(96):
        loops = ctypes.c_int()
        b = np.ascontiguousarray(dec_bias[i])
        dec_viol[i] = Lb.ref_bnd_decode(afile, b.ctypes.data_as(ctypes.c_void_p), 48, 48, 100,
                                        dec_x[i].ctypes.data_as(ctypes.c_void_p), dec_q1[i].ctypes.data_as(ctypes.c_void_p),
                                        ctypes.byref(loops))
        dec_loops[i] = loops.value
        xo, qo, vo, lo = O.ldpc_sum_product(A2_nlist, 48, b)
        assert np.array_equal(xo, dec_x[i]) and np.array_equal(qo, dec_q1[i]) and vo == dec_viol[i] and lo == loops.value, \
            'restated decoder disagrees with the compiled reference on word %d' % i
    print('decoder: %d of 96 words decoded, iterations %s' % (int((dec_viol == 0).sum()), np.bincount(dec_loops)[:8]))
    np.savez_compressed(os.path.join(ROOT, 'tests/golden/ldpc_datapath.npz'), G=G, s=s, codewords=cw,
                        H_A2=H_A2, A2_nlist=A2_nlist, dec_bias=dec_bias, dec_x=dec_x, dec_q1=dec_q1, dec_viol=dec_viol,
                        dec_loops=dec_loops,
                        snr_db=snr_db, sigma_b=sigma_b, z1=z1, u=u, z2=z2, y=y)
    np.savez_compressed(os.path.join(ROOT, 'factor-graph-neural-network_amd/fgnn_amd/data/ldpc_96_3_963_G.npz'), G=G,
                        A2_nlist=A2_nlist)
    print('wrote golden + packaged G; ones in G:', int(G.sum()))


if __name__ == '__main__':
    main()
