"""TEST INFRASTRUCTURE.  Generates tests/golden/ldpc_t2y_stream.npz from the REFERENCE's own Python module `MNC`
(lib/data/MNC/MNC_py.cpp), compiled where it lies by oracle/build_ref.sh into oracle/_ref/MNC*.so:

  * `init_seed(seed)` followed by successive `t2y(t, snr_db, sigma_b, rho)` calls (MNC_py.cpp:86-102,185-187): the received
    words INCLUDING the draws of xtensor's process-wide generator.  The calls of one seed continue one stream, as the items
    of a worker do in the reference (`data_generate/ldpc.py:35-40`: one `init_seed(os.getpid())`, then item after item).
    Covered: sigma_b = 0 (no burst draws at all), rho = 1 (every bit draws a burst), odd lengths (a kept normal is dropped
    with its distribution object), the SNR / burst grid of `lib/data/ldpc_dataset.py`;
  * `y2b(y, snr_db)` of those words (MNC_py.cpp:104-108);
  * whole items the way `gen_data_item` chains the module (lib/data/ldpc.py:7-30): s -> s2t -> t2y -> y2b -> zb2x.

The restatements in oracle/fgnn_oracle.py (`XtensorStream`, `ldpc_channel_stream`, `ldpc_encode`, `ldpc_bit_prior`,
`ldpc_sum_product`) are asserted against every vector before anything is written: t2y bit-for-bit.

Run here (needs /root/reference and `sh oracle/build_ref.sh`); the output is data and is committed.
"""
import contextlib
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'oracle'))
sys.path.insert(0, os.path.join(ROOT, 'oracle', '_ref'))
import fgnn_oracle as O                                                    # noqa: E402
import MNC                                                                 # noqa: E402  (the reference's module)

REF = os.environ.get('FGNN_REFERENCE', '/root/reference')
SEEDS = (0, 1, 963, 40961, 65536, 2 ** 31 - 1)      # train_ldpc.py:147 seeds with `seed % 65537`; workers with their pid


@contextlib.contextmanager
def quiet_stdout():
    """`zb2x` prints the block length to the C++ stdout on every call (MNC_py.cpp:165)."""
    sys.stdout.flush()
    keep, null = os.dup(1), os.open(os.devnull, os.O_WRONLY)
    os.dup2(null, 1)
    try:
        yield
    finally:
        os.dup2(keep, 1)
        os.close(null)
        os.close(keep)


def main():
    rng = np.random.default_rng(2963)
    calls = 8
    seed, length, snr, sb, rho = [], [], [], [], []
    tbits = np.zeros((len(SEEDS) * calls, 96), np.int64)
    y = np.zeros((len(SEEDS) * calls, 96), np.float64)
    prior = np.zeros_like(y)
    row = 0
    for sd in SEEDS:
        MNC.init_seed(sd)
        stream = O.XtensorStream(sd)
        for c in range(calls):
            n = 96 if c < 6 else 95 - 2 * (c - 6)                          # the last two calls of a stream: 95 and 93 bits
            t = rng.integers(0, 2, n).astype(np.int64)
            a = float(rng.integers(0, 5))                                   # ldpc_dataset.py: snr_db in {0..4}
            b = (0.0, 0.0, 1.0, 2.0, 3.0, 4.0, 5.0, 1e-21)[int(rng.integers(0, 8))]   # sigma_b in {0..5}; 1e-21 < the 1e-20 gate
            r = (0.05, 0.05, 0.05, 0.5, 1.0, 0.0)[int(rng.integers(0, 6))]
            out = np.asarray(MNC.t2y(t, a, b, r))
            mine = O.ldpc_channel_stream(t, a, b, r, stream)
            assert np.array_equal(out, mine), 'restated t2y stream differs from the reference: seed %d call %d' % (sd, c)
            z = np.asarray(MNC.y2b(out, a))
            assert np.abs(z - O.ldpc_bit_prior(out[None], np.array([a]))[0]).max() <= 2.0 ** -52
            seed.append(sd), length.append(n), snr.append(a), sb.append(b), rho.append(r)
            tbits[row, :n], y[row, :n], prior[row, :n] = t, out, z
            row += 1
    # whole items, chained as gen_data_item does
    gfile = os.path.join(REF, 'ldpc_codes/96.3.963/G')
    afile = os.path.join(REF, 'ldpc_codes/96.3.963/A2')
    fix = np.load(os.path.join(ROOT, 'tests/golden/ldpc_datapath.npz'))
    items = 24
    MNC.init_seed(4242)
    stream = O.XtensorStream(4242)
    it_s = rng.integers(0, 2, (items, 48)).astype(np.int64)
    it_snr = rng.integers(0, 5, items).astype(np.float64)
    it_sb = rng.integers(0, 6, items).astype(np.float64)
    it_t = np.zeros((items, 96), np.int64)
    it_y = np.zeros((items, 96), np.float64)
    it_z = np.zeros((items, 96), np.float64)
    it_x = np.zeros((items, 48), np.int64)
    for i in range(items):
        it_t[i] = MNC.s2t(it_s[i], 48, 48, gfile, True)
        it_y[i] = MNC.t2y(it_t[i], it_snr[i], it_sb[i], 0.05)
        it_z[i] = MNC.y2b(it_y[i], it_snr[i])
        with quiet_stdout():
            it_x[i] = MNC.zb2x(it_z[i], 48, 48, afile, 1, 100)
        assert np.array_equal(O.ldpc_encode(fix['G'], it_s[i]), it_t[i])
        assert np.array_equal(O.ldpc_channel_stream(it_t[i], it_snr[i], it_sb[i], 0.05, stream), it_y[i])
        x, _, _, _ = O.ldpc_sum_product(fix['A2_nlist'], 48, it_z[i])
        assert np.array_equal(x[:48], it_x[i]), 'restated decoder differs from zb2x on item %d' % i
    errs = (it_x != it_s).mean(1)
    np.savez_compressed(os.path.join(ROOT, 'tests/golden/ldpc_t2y_stream.npz'),
                        seed=np.array(seed, np.int64), length=np.array(length, np.int32), snr_db=np.array(snr), sigma_b=np.array(sb),
                        rho=np.array(rho), t=tbits.astype(np.uint8), y=y, prior=prior,
                        item_seed=np.int64(4242), item_s=it_s.astype(np.uint8), item_snr_db=it_snr, item_sigma_b=it_sb,
                        item_t=it_t.astype(np.uint8), item_y=it_y, item_prior=it_z, item_x=it_x.astype(np.uint8))
    print('wrote tests/golden/ldpc_t2y_stream.npz: %d t2y calls over %d seeds (bit-for-bit), %d chained items, '
          'sum-product bit error rate of the items %.4f' % (row, len(SEEDS), items, errs.mean()))


if __name__ == '__main__':
    main()
