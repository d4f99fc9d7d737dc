"""CPU: the oracle's restatement of the reference's LDPC data path (encode `s2t`, channel `t2y`) against vectors
produced by the REFERENCE's own compiled GF(2) code (oracle/make_ldpc_datapath_golden.py) and against the code's
parity-check structure."""
import ctypes
import os

import numpy as np
import pytest

import fgnn_oracle as O
import helpers as H

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _graph():
    from fgnn_amd.tables import LdpcGraph
    return LdpcGraph()


def test_oracle_encode_matches_reference_vectors():
    z = H.load('ldpc_datapath.npz')
    assert z['G'].shape == (48, 48) and z['codewords'].shape == (64, 96)
    assert np.array_equal(O.ldpc_encode(z['G'], z['s']), z['codewords'])
    # unit messages (rows 2..49 of the fixture) read out the columns of G
    assert np.array_equal(z['codewords'][2:50, 48:], z['G'].T)
    assert not z['codewords'][0].any()


def test_packaged_generator_matrix_is_the_reference_one():
    from fgnn_amd.tables import _DATA
    G = np.load(os.path.join(_DATA, 'ldpc_96_3_963_G.npz'))['G']
    assert np.array_equal(G, H.load('ldpc_datapath.npz')['G'])


def _check_codewords(cw, z):
    """Zero syndrome under A2, the parity-check matrix the reference pairs with its G file; under the regular
    96.3.963 incidence lists the model runs on (rank 46) all checks hold except that checks 45 and 47 — the rows A2
    patches — may fail TOGETHER."""
    assert not ((cw.astype(np.int64) @ z['H_A2'].T.astype(np.int64)) % 2).any()
    syn = cw[:, _graph().factor_to_vars].sum(2) % 2
    assert not np.delete(syn, [45, 47], axis=1).any() and np.array_equal(syn[:, 45], syn[:, 47])


def test_every_codeword_satisfies_the_parity_checks():
    z = H.load('ldpc_datapath.npz')
    _check_codewords(z['codewords'], z)
    s = np.random.default_rng(5).integers(0, 2, (2000, 48))
    cw = O.ldpc_encode(z['G'], s)
    _check_codewords(cw, z)
    # linearity over GF(2)
    assert np.array_equal(O.ldpc_encode(z['G'], s[:1000] ^ s[1000:]), cw[:1000] ^ cw[1000:])


def test_oracle_channel_restatement():
    z = H.load('ldpc_datapath.npz')
    y = O.ldpc_channel(z['codewords'], z['snr_db'], z['sigma_b'], 0.05, z['z1'], z['u'], z['z2'])
    assert np.array_equal(y, z['y'])
    gcx = 10.0 ** (z['snr_db'] / 20.0)
    quiet = O.ldpc_channel(z['codewords'], z['snr_db'], np.zeros(64), 0.05, np.zeros((64, 96)), z['u'], z['z2'])
    assert np.allclose(quiet, (2.0 * z['codewords'] - 1.0) * gcx[:, None])          # bit 1 -> +gcx, bit 0 -> -gcx
    no_burst = O.ldpc_channel(z['codewords'], z['snr_db'], z['sigma_b'], 0.0, z['z1'], z['u'], z['z2'])
    assert np.array_equal(no_burst, 2.0 * gcx[:, None] * (z['codewords'] - 0.5) + z['z1'])
    hit = (z['u'] < 0.05) & (z['sigma_b'][:, None] >= 1e-20)
    assert hit.any() and np.array_equal(y != no_burst, hit & (z['z2'] != 0))


def test_oracle_encode_matches_compiled_reference_when_present():
    lib = os.path.join(ROOT, 'oracle', '_ref', 'libmod2mat_ref.so')
    gfile = '/root/reference/ldpc_codes/96.3.963/G'
    if not (os.path.exists(lib) and os.path.exists(gfile)):
        pytest.skip('reference build (oracle/build_ref.sh) or /root/reference not present')
    L = ctypes.CDLL(lib)
    G = H.load('ldpc_datapath.npz')['G']
    s = np.random.default_rng(77).integers(0, 2, (32, 48)).astype(np.uint8)
    out = np.zeros(96, np.uint8)
    for row in s:
        assert L.ref_encode(gfile.encode(), row.ctypes.data_as(ctypes.c_void_p), 48, 48,
                            out.ctypes.data_as(ctypes.c_void_p)) == 0
        assert np.array_equal(out, O.ldpc_encode(G, row))


def test_oracle_decoder_matches_compiled_reference_vectors():
    """`ldpc_sum_product` (the restated `bndecode`) against outputs of the reference's own compiled decoder on 96
    received words: hard decisions, float64 pseudo-posteriors bit for bit, violated checks, iteration counts."""
    z = H.load('ldpc_datapath.npz')
    assert (z['dec_viol'] == 0).sum() >= 40 and (z['dec_viol'] > 0).sum() >= 20 and len(set(z['dec_loops'])) >= 5
    for i in list(range(0, 96, 7)) + [53, 55, 59]:
        x, q1, viol, it = O.ldpc_sum_product(z['A2_nlist'], 48, z['dec_bias'][i])
        assert np.array_equal(x, z['dec_x'][i]) and np.array_equal(q1, z['dec_q1'][i])
        assert viol == z['dec_viol'][i] and it == z['dec_loops'][i]
    # a decoded word is a codeword of A2, and on the burst-free words it is the transmitted one
    ok = z['dec_viol'] == 0
    assert not ((z['dec_x'][ok].astype(np.int64) @ z['H_A2'].T.astype(np.int64)) % 2).any()
    easy = np.arange(64, 96)[ok[64:]]
    assert len(easy) >= 24 and np.array_equal(z['dec_x'][easy], z['codewords'][easy - 64])
    assert np.allclose(O.ldpc_bit_prior(z['y'], z['snr_db']), z['dec_bias'][:64], rtol=0, atol=0)


def test_philox_restatement_known_answers():
    """Random123's published known-answer vectors for Philox4x32-10 (the generator of fgnn_ldpc_channel_features_rng)."""
    import numpy as np
    kat = [((0, 0, 0, 0), (0, 0), (0x6627e8d5, 0xe169c58d, 0xbc57ac4c, 0x9b00dbd8)),
           ((0xffffffff,) * 4, (0xffffffff, 0xffffffff), (0x408f276d, 0x41c83b0e, 0xa20bc7c6, 0x6d5451fd)),
           ((0x243f6a88, 0x85a308d3, 0x13198a2e, 0x03707344), (0xa4093822, 0x299f31d0), (0xd16cfe09, 0x94fdcceb, 0x5001e420, 0x24126ea1))]
    for ctr, key, want in kat:
        got = O.philox4x32(np.array([ctr], dtype=np.uint32), key)[0]
        assert tuple(int(v) for v in got) == want
    z1, u, z2 = O.philox_channel_draws(100000, 99, 3)
    assert abs(float(z1.mean())) < 0.01 and abs(float(z1.std()) - 1.0) < 0.01 and abs(float(z2.std()) - 1.0) < 0.01
    assert 0.0 <= float(u.min()) and float(u.max()) < 1.0 and abs(float(u.mean()) - 0.5) < 0.005
    assert abs(float(np.corrcoef(z1, z2)[0, 1])) < 0.01


def _t2y_streams():
    z = H.load('ldpc_t2y_stream.npz')
    start = 0
    for sd in dict.fromkeys(int(s) for s in z['seed']):                     # the calls of a seed continue one stream
        rows = [i for i in range(len(z['seed'])) if int(z['seed'][i]) == sd]
        assert rows == list(range(start, start + len(rows)))
        start += len(rows)
        yield z, sd, rows


def test_oracle_replays_the_reference_t2y_stream_bit_for_bit():
    """`init_seed(s)` + successive `t2y` calls of the reference's own compiled module (oracle/make_t2y_golden.py): the restated
    generator (mt19937 -> libstdc++ canonical doubles -> polar normals, a fresh distribution per xtensor call) reproduces every
    received word exactly, across calls, with and without burst draws, for even and odd lengths."""
    seen = set()
    for z, sd, rows in _t2y_streams():
        stream = O.XtensorStream(sd)
        for i in rows:
            n = int(z['length'][i])
            y = O.ldpc_channel_stream(z['t'][i, :n], z['snr_db'][i], z['sigma_b'][i], z['rho'][i], stream)
            assert np.array_equal(y, z['y'][i, :n]), (sd, i)
            assert np.abs(O.ldpc_bit_prior(y[None], z['snr_db'][i:i + 1])[0] - z['prior'][i, :n]).max() <= 2.0 ** -52
            seen.add((n % 2, bool(z['sigma_b'][i] >= 1e-20), float(z['rho'][i])))
    assert {(0, False), (0, True), (1, True)} <= {(a, b) for a, b, _ in seen} and {0.0, 0.05, 0.5, 1.0} <= {r for _, _, r in seen}


def test_stream_draws_fed_to_the_explicit_channel_give_the_reference_words():
    """The product's channel takes its draws as inputs (`ldpc_channel` = fgnn_ldpc_channel_features' arithmetic): with the
    draws the reference's stream made, it returns the reference's words."""
    for z, sd, rows in _t2y_streams():
        stream = O.XtensorStream(sd)
        for i in rows:
            n = int(z['length'][i])
            y, z1, u, z2 = O.ldpc_channel_stream(z['t'][i, :n], z['snr_db'][i], z['sigma_b'][i], z['rho'][i], stream, return_draws=True)
            again = O.ldpc_channel(z['t'][i:i + 1, :n], z['snr_db'][i:i + 1], z['sigma_b'][i:i + 1], z['rho'][i], z1[None], u[None], z2[None])[0]
            assert np.abs(again - y).max() <= 2.0 ** -50 * np.abs(y).max()       # numpy's vector pow vs libm's: 1 ulp of gcx


def test_whole_items_chain_like_gen_data_item():
    """s -> s2t -> t2y -> y2b -> zb2x of the reference module, item after item on one stream (lib/data/ldpc.py:7-30)."""
    z, fix = H.load('ldpc_t2y_stream.npz'), H.load('ldpc_datapath.npz')
    stream = O.XtensorStream(int(z['item_seed']))
    wrong = 0
    for i in range(len(z['item_s'])):
        t = O.ldpc_encode(fix['G'], z['item_s'][i])
        assert np.array_equal(t, z['item_t'][i])
        y = O.ldpc_channel_stream(t, z['item_snr_db'][i], z['item_sigma_b'][i], 0.05, stream)
        assert np.array_equal(y, z['item_y'][i])
        if i % 4 == 0:
            x, _, viol, _ = O.ldpc_sum_product(fix['A2_nlist'], 48, z['item_prior'][i])
            assert np.array_equal(x[:48], z['item_x'][i])
            wrong += int(viol > 0)
    assert 0 < (z['item_x'] != z['item_s']).mean() < 0.2


def test_xtensor_stream_statistics():
    st = O.XtensorStream(5)
    zs = np.array(st.randn(20001))
    us = np.array([st.rand1() for _ in range(20000)])
    assert abs(zs.mean()) < 0.03 and abs(zs.std() - 1.0) < 0.03 and 0.0 <= us.min() and us.max() < 1.0 and abs(us.mean() - 0.5) < 0.01
