"""GPU: csrc/ldpc_datapath.hip (encode + channel + model inputs for a whole batch) through the C ABI against the
oracle and the reference-generated vectors — bit-exact for the GF(2) work, f32 rounding for the channel, exact
copies for the gathered features."""
import numpy as np
import pytest
import torch

import fgnn_oracle as O
import helpers as H

pytestmark = pytest.mark.gpu


@pytest.fixture(scope='module')
def path(dev):
    from fgnn_amd.datapath import LdpcDataPath
    return LdpcDataPath(dev)


def test_encode_matches_reference_vectors(path):
    from fgnn_amd import _hip
    z = H.load('ldpc_datapath.npz')
    cw = path.encode(torch.from_numpy(z['s']))
    assert _hip.lib().fgnn_last_kernel().decode() == 'ldpc_encode_kernel'
    assert cw.dtype == torch.uint8 and np.array_equal(cw.cpu().numpy(), z['codewords'])


@pytest.mark.parametrize('B', [1, 63, 64, 65, 4096, 50001])
def test_encode_bit_exact_vs_oracle_and_parity_checks(B, path):
    z = H.load('ldpc_datapath.npz')
    s = np.random.default_rng(B).integers(0, 2, (B, 48)).astype(np.uint8)
    cw = path.encode(torch.from_numpy(s)).cpu().numpy()
    assert np.array_equal(cw, O.ldpc_encode(z['G'], s))
    # size-independent property: zero syndrome under the parity-check matrix the reference pairs with its G (A2)
    assert not ((cw.astype(np.int64) @ z['H_A2'].T.astype(np.int64)) % 2).any()


def test_encode_rejects_bad_shapes(path):
    with pytest.raises(ValueError):
        path.encode(torch.zeros(4, 47, dtype=torch.uint8))
    assert path.encode(torch.zeros(0, 48, dtype=torch.uint8)).shape == (0, 96)


@pytest.mark.parametrize('dtype', [torch.float32, torch.bfloat16], ids=['f32', 'bf16'])
def test_channel_and_features_match_oracle(dtype, path):
    from fgnn_amd.tables import LdpcGraph
    z = H.load('ldpc_datapath.npz')
    t = lambda k: torch.from_numpy(z[k])
    y, node, hop, e1, e2 = path.channel_features(t('codewords'), t('snr_db'), t('sigma_b'), 0.05,
                                                 noise=(t('z1'), t('u'), t('z2')), dtype=dtype)
    assert y.dtype == torch.float32 and node.dtype == dtype
    # channel: float64 reference arithmetic vs f32 on the device
    assert np.abs(y.cpu().numpy() - z['y']).max() <= 4e-6 * np.abs(z['y']).max()
    # model inputs: pure gathers of the device's own y (ldpc_dataset.py:92-106,222-236), rounded once to `dtype`
    g = LdpcGraph()
    yh = y.cpu().numpy()
    for b in range(64):
        ref = g.features(yh[b], z['snr_db'][b])
        for got, want in zip((node[b], hop[b], e1[b], e2[b]), ref):
            assert torch.equal(got.cpu(), torch.from_numpy(want).to(dtype))


def test_channel_without_noise_is_bpsk_at_the_snr_gain(path):
    z = H.load('ldpc_datapath.npz')
    B = 64
    zero = torch.zeros(B, 96)
    y = path.channel_features(torch.from_numpy(z['codewords']), torch.from_numpy(z['snr_db']), torch.zeros(B), 0.05,
                              noise=(zero, zero, zero))[0].cpu().numpy()
    gcx = 10.0 ** (z['snr_db'] / 20.0)
    assert np.allclose(y, (2.0 * z['codewords'] - 1.0) * gcx[:, None], rtol=2e-6)


def test_sample_is_a_training_batch_the_model_accepts(path, dev):
    import fgnn_amd
    a = path.sample(256, seed=3, dtype=torch.bfloat16)
    b = path.sample(256, seed=3, dtype=torch.bfloat16)
    c = path.sample(256, seed=4, dtype=torch.bfloat16)
    assert all(torch.equal(x, y) for x, y in zip(a, b)) and not torch.equal(a[6], c[6])
    node, hop, i1, i2, e1, e2, label, sigma_b = a
    assert node.shape == (256, 2, 96, 1) and hop.shape == (256, 6, 48, 1) and e1.shape == (256, 7, 96, 3)
    assert e2.shape == (256, 7, 48, 6) and i1.shape == (256, 96, 3) and i2.shape == (256, 48, 6)
    assert label.dtype == torch.int64 and label.shape == (256, 96) and set(label.unique().tolist()) <= {0, 1}
    assert set(sigma_b.unique().tolist()) <= {0., 1., 2., 3., 4., 5.}
    # received words lean towards their bits: sign(y) recovers most of the codeword at 0..4 dB
    agree = ((node[:, 0, :, 0].float() > 0).long() == label).float().mean().item()
    assert 0.80 < agree < 1.0
    m = fgnn_amd.LDPCModel(2, 6, 4).to(dev).eval()
    with torch.no_grad(), torch.autocast('cuda', dtype=torch.bfloat16):
        logits, snr = m(node, hop, i1, i2, e1, e2)
    assert logits.shape == (256, 48) and torch.isfinite(logits.float()).all()


def test_sum_product_decoder_matches_compiled_reference_bit_for_bit(path):
    """csrc/ldpc_decode.hip against the outputs of the reference's own compiled `bndecode` (fixture made by
    oracle/make_ldpc_datapath_golden.py): hard decisions, float64 pseudo-posteriors, violated checks and iteration
    counts are EQUAL on all 96 words (successes after 1..97 iterations and failures at the 100-iteration cap)."""
    from fgnn_amd import _hip
    z = H.load('ldpc_datapath.npz')
    x, viol, iters, q1 = path.decode(torch.from_numpy(z['dec_bias']), loops=100, want_posteriors=True)
    assert _hip.lib().fgnn_last_kernel().decode() == 'ldpc_decode_kernel'
    assert np.array_equal(viol.cpu().numpy(), z['dec_viol']) and np.array_equal(iters.cpu().numpy(), z['dec_loops'])
    assert np.array_equal(x.cpu().numpy(), z['dec_x'])
    assert np.array_equal(q1.cpu().numpy(), z['dec_q1'])


@pytest.mark.parametrize('B', [1, 4096])
def test_sum_product_decoder_vs_oracle_and_properties(B, path):
    z = H.load('ldpc_datapath.npz')
    rng = np.random.default_rng(B)
    s = rng.integers(0, 2, (B, 48)).astype(np.uint8)
    cw = O.ldpc_encode(z['G'], s)
    snr = rng.integers(2, 5, B).astype(np.float64)
    y = O.ldpc_channel(cw, snr, np.zeros(B), 0.0, rng.standard_normal((B, 96)), rng.random((B, 96)), rng.standard_normal((B, 96)))
    bias = O.ldpc_bit_prior(y, snr)
    x, viol, iters = path.decode(torch.from_numpy(bias), loops=50)
    x, viol, iters = x.cpu().numpy(), viol.cpu().numpy(), iters.cpu().numpy()
    for i in range(0, B, max(1, B // 12)):                       # the (slow, pure-Python) oracle on a sample
        xo, _, vo, io = O.ldpc_sum_product(z['A2_nlist'], 48, bias[i], loops=50)
        assert np.array_equal(x[i], xo) and viol[i] == vo and iters[i] == io
    # size-independent properties: zero reported violations <=> zero syndrome under A2; most words are recovered
    syn = ((x.astype(np.int64) @ z['H_A2'].T.astype(np.int64)) % 2).sum(1)
    assert np.array_equal(syn, viol)
    ok = viol == 0
    assert ok.mean() > (0.5 if B > 1 else -1) and (x[ok] == cw[ok]).mean() > 0.999
    assert iters.min() >= 1 and iters.max() <= 50 and np.all(iters[~ok] == 50)
    # the decoder must beat hard decisions on y
    if B > 1:
        assert (x != cw).mean() < ((y > 0) != cw).mean()


def test_decode_from_received_words(path):
    """End to end on the device: sample -> y2b -> decode."""
    node, hop, i1, i2, e1, e2, label, sigma_b = path.sample(512, seed=8, snr_db=4.0)
    y = node[:, 0, :, 0]
    bias = path.bit_prior(y, torch.full((512,), 4.0))
    x, viol, iters = path.decode(bias)
    ok = (viol == 0)
    assert ok.float().mean().item() > 0.3 and (x[ok].long() == label[ok]).float().mean().item() > 0.98


def test_kernel_rng_channel_matches_the_philox_restatement(path):
    """fgnn_ldpc_channel_features_rng: the channel's draws come from Philox4x32-10 inside the kernel.  The oracle restates the
    generator (pinned by the Random123 known-answer vectors in the CPU suite) and the Box-Muller map; the received words must
    agree to f32 rounding of the transcendental functions, the burst decisions (u < rho) exactly."""
    z = H.load('ldpc_datapath.npz')
    B, seed, step, rho = 64, 0x1234567887654321, 5, 0.05
    cw, snr, sb = z['codewords'], z['snr_db'], z['sigma_b']
    y = path.channel_features(torch.from_numpy(cw), torch.from_numpy(snr), torch.from_numpy(sb), rho, kernel_rng=(seed, step))[0]
    z1, u, z2 = (a.reshape(B, 96) for a in O.philox_channel_draws(B * 96, seed, step))
    ref = O.ldpc_channel(cw, snr, sb, rho, z1.astype(np.float64), u, z2.astype(np.float64))
    err = np.abs(y.cpu().numpy() - ref)
    assert err.max() <= 2e-5 * np.abs(ref).max(), err.max()
    # reproducible from (seed, step) whatever the batch is cut into: the first 16 codewords alone give the same words
    y16 = path.channel_features(torch.from_numpy(cw[:16]), torch.from_numpy(snr[:16]), torch.from_numpy(sb[:16]), rho,
                                kernel_rng=(seed, step))[0]
    assert torch.equal(y16, y[:16])
    # another step, another stream
    y2 = path.channel_features(torch.from_numpy(cw), torch.from_numpy(snr), torch.from_numpy(sb), rho, kernel_rng=(seed, step + 1))[0]
    assert float((y2 - y).abs().min()) > 0.0 or float((y2 != y).float().mean()) > 0.99


def test_kernel_rng_statistics(path, dev):
    """40 000 codewords: unit-variance Gaussian noise around the BPSK points, bursts on a fraction rho of the bits."""
    B, rho = 40000, 0.05
    cw = torch.zeros(B, 96, dtype=torch.uint8)
    snr = torch.zeros(B)                                   # gcx = 1
    y0 = path.channel_features(cw, snr, torch.zeros(B), rho, kernel_rng=(7, 0))[0]        # no bursts: y = -1 + z1
    n = (y0 + 1.0).double()
    assert abs(float(n.mean())) < 2e-3 and abs(float(n.var()) - 1.0) < 5e-3
    assert abs(float((n ** 4).mean()) - 3.0) < 5e-2                                      # kurtosis of a normal
    yb = path.channel_features(cw, snr, torch.full((B,), 10.0), rho, kernel_rng=(7, 0))[0]   # bursts of sigma 10 where u < rho
    hit = (yb != y0).double().mean()
    assert abs(float(hit) - rho) < 1e-3
    extra = (yb - y0)[yb != y0].double()
    assert abs(float(extra.mean())) < 0.1 and abs(float(extra.std()) - 10.0) < 0.1


def test_sample_with_kernel_rng_is_a_training_batch(path, dev):
    a = path.sample(32, seed=3, dtype=torch.bfloat16, kernel_rng=True, step=0)
    b = path.sample(32, seed=3, dtype=torch.bfloat16, kernel_rng=True, step=0)
    c = path.sample(32, seed=3, dtype=torch.bfloat16, kernel_rng=True, step=1)
    assert all(torch.equal(x, y) for x, y in zip(a, b))
    assert not torch.equal(a[0], c[0]) and torch.equal(a[6], c[6])       # new noise, same codewords


def test_channel_fed_the_reference_stream_draws_gives_the_reference_words(path):
    """tests/golden/ldpc_t2y_stream.npz holds words the reference's own `t2y` returned after `init_seed` (draws included).  The
    draws its generator made, replayed by the oracle's XtensorStream, go into fgnn_ldpc_channel_features as its noise inputs:
    the device's f32 channel lands on the reference's float64 words."""
    z = H.load('ldpc_t2y_stream.npz')
    rows, draws, stream, seed = [], [], None, None
    for i in range(len(z['seed'])):
        if int(z['seed'][i]) != seed:
            seed = int(z['seed'][i])
            stream = O.XtensorStream(seed)
        n = int(z['length'][i])
        y, z1, u, z2 = O.ldpc_channel_stream(z['t'][i, :n], z['snr_db'][i], z['sigma_b'][i], z['rho'][i], stream, return_draws=True)
        assert np.array_equal(y, z['y'][i, :n])
        if n == 96:
            rows.append(i), draws.append((z1, u, z2))
    assert len(rows) >= 30
    for rho in sorted(set(z['rho'][rows])):
        sel = [k for k, i in enumerate(rows) if z['rho'][i] == rho]
        idx = [rows[k] for k in sel]
        noise = tuple(torch.from_numpy(np.stack([draws[k][j] for k in sel])) for j in range(3))
        y = path.channel_features(torch.from_numpy(z['t'][idx]), torch.from_numpy(z['snr_db'][idx]), torch.from_numpy(z['sigma_b'][idx]),
                                  float(rho), noise=noise)[0].cpu().numpy()
        assert np.abs(y - z['y'][idx]).max() <= 4e-6 * np.abs(z['y'][idx]).max()
