"""CPU: host-side logic — tables vs the reference's generators (golden), state_dict surface,
constructor error behaviour, the C-ABI library loads and exports every declared symbol."""
import ctypes
import os
import re

import numpy as np
import pytest
import torch

import helpers as H

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_tables_match_reference_generators():
    from fgnn_amd import tables
    t = H.load('tables.npz')
    a, b = tables.knn_table(30, 8)
    assert (a == t['knn_idx'][0]).all() and (b == t['knn_ef'][0]).all()
    assert (a[:, 7] == 0).all()                 # the 7-of-8 quirk is reproduced, not fixed
    a, b = tables.pw_factor_table(30)
    assert (a == t['pw_idx'][0]).all() and (b == t['pw_ef'][0]).all()
    a, b, c = tables.chain_high_table(30, 9)
    assert (a == t['hi_idx'][0]).all() and (b == t['hi_ef'][0]).all() and (c == t['hi_ff'][0]).all()
    for k in (8, 9):
        a, b = tables.ring_hop_table(30, k)
        assert (a == t['hop%d_idx' % k][0]).all() and (b == t['hop%d_ef' % k][0]).all()


def test_ldpc_graph_matches_alist_incidence():
    from fgnn_amd import tables
    t = H.load('tables.npz')
    g = tables.LdpcGraph()
    assert (g.var_to_factors == t['ldpc_f2v']).all() and (g.factor_to_vars == t['ldpc_v2f']).all()
    # the two incidence tables are transposes of each other: 288 edges, degrees 3 and 6
    edges = {(v, f) for v in range(96) for f in g.var_to_factors[v]}
    assert edges == {(v, f) for f in range(48) for v in g.factor_to_vars[f]} and len(edges) == 288
    n, h, e1, e2 = g.features(t['ldpc_y'], 2.0)
    assert np.allclose(h[:, :, 0].T, t['ldpc_hop'])
    assert np.allclose(e1.transpose(1, 2, 0), t['ldpc_ef_f2v'])
    assert np.allclose(e2.transpose(1, 2, 0), t['ldpc_ef_v2f'])


def test_ldpc_model_state_dict_surface():
    import fgnn_amd
    z = H.load('ldpc_model.npz')
    m = fgnn_amd.LDPCModel(2, 6, 4, aggregator='max')
    keys = sorted(m.state_dict().keys())
    assert keys == list(z['sd_keys']) and len(keys) == 651
    assert [str(tuple(m.state_dict()[k].shape)) for k in keys] == list(z['sd_shapes'])
    from fgnn_amd.ldpc import MESSAGES_PER_CODEWORD
    assert MESSAGES_PER_CODEWORD == 6144


def test_operator_constructor_contract():
    from fgnn_amd.mpnn import base_mp_nn, mp_conv_residual, mp_conv_type, mp_conv_v2
    m = mp_conv_v2(5, 7, 3)                      # defaults: ORIG_WITH_DIFF, 'softmax' (sic kwarg)
    assert isinstance(m, base_mp_nn) and m.is_mp_nn
    assert m.extension == mp_conv_type.ORIG_WITH_DIFF and m.aggregtor == 'softmax'
    assert tuple(m.filters.shape) == (10, 21) and tuple(m.bias.shape) == (7,)
    assert float(m.filters.abs().max()) <= 0.01 and 0 <= float(m.bias.min()) and float(m.bias.max()) <= 0.05
    assert sorted(m.state_dict()) == sorted(['filters', 'bias', 'bn.weight', 'bn.bias', 'bn.running_mean',
                                             'bn.running_var', 'bn.num_batches_tracked'])
    with pytest.raises(ValueError, match='extension must one of mp_conv_type'):
        mp_conv_v2(2, 2, 1, extension=7)
    r = mp_conv_residual(8, 4, 3, nout=10)
    assert r.mp_conv.nin == 4 and r.mp_conv.aggregtor == 'max' and r.conv2[0].out_channels == 10


def test_operator_refuses_cpu_tensors():
    from fgnn_amd import _hip
    from fgnn_amd.mpnn import mp_conv_type, mp_conv_v2
    m = mp_conv_v2(2, 2, 1, extension=mp_conv_type.NO_EXTENSION, aggregtor='max')
    with pytest.raises(_hip.FgnnHipError, match='no CPU fallback'):
        m(torch.zeros(1, 2, 3, 1), torch.zeros(1, 3, 2, dtype=torch.int64), torch.zeros(1, 1, 3, 2))


def test_c_abi_exports_every_declared_symbol():
    from fgnn_amd import _hip
    header = open(os.path.join(ROOT, 'include', 'fgnn_hip.h')).read()
    declared = set(re.findall(r'\b(fgnn_[a-z_]+)\s*\(', header))
    assert declared == set(_hip.EXPORTS), declared ^ set(_hip.EXPORTS)
    L = ctypes.CDLL(_hip.LIB_PATH)
    for name in declared:
        assert hasattr(L, name), name
    assert _hip.lib().fgnn_abi_version() >= 1


def test_descriptor_checks_without_a_gpu():
    """Argument validation happens before any device work, so it is testable here."""
    from fgnn_amd import _hip
    L = _hip.lib()
    d = _hip.MPConvDesc()
    d.B, d.nin, d.nou, d.net, d.N, d.M, d.k = 4, 64, 64, 4, 96, 48, 6
    d.ext, d.agg, d.dtype = 0, 0, 0
    assert 0 < L.fgnn_mpconv_forward_lds_bytes(ctypes.byref(d)) <= 160 * 1024
    # LDPC parity V->F call, fp32, per-sample int64 indices: 43 776 B per codeword + weights
    d.x_sb, d.idx_sb, d.et_sb = 64 * 96, 288, 4 * 288
    assert L.fgnn_mpconv_algorithmic_bytes(ctypes.byref(d)) == 4 * 43776 + 4 * (64 * 256 + 5 * 64)
    d.ext, d.M = 2, 48                           # extension needs N == M
    assert L.fgnn_mpconv_forward_lds_bytes(ctypes.byref(d)) == -1
    assert b'N == M' in L.fgnn_last_error()
    d.ext, d.k = 0, 300
    assert L.fgnn_mpconv_forward_lds_bytes(ctypes.byref(d)) == -1


def test_side_entry_points_validate_their_arguments_without_a_gpu():
    """Size / pointer checks of the entry points either side of the operator run before any device work: unsupported
    shapes come back as FGNN_EUNSUPPORTED (-3), missing buffers as FGNN_EINVAL (-1), with a message."""
    from fgnn_amd import _hip
    L = _hip.lib()
    one = ctypes.c_void_p(16)                      # a non-NULL pointer that is never dereferenced on these paths
    assert L.fgnn_edge_mlp_forward(one, 0, 0, 0, one, one, one, one, one, 8, 288, 9, 4, None) == _hip.EUNSUPPORTED
    assert b'Cin=9' in L.fgnn_last_error()
    assert L.fgnn_edge_mlp_forward(one, 0, 0, 0, one, one, one, one, one, 8, 288, 7, 5, None) == _hip.EUNSUPPORTED
    assert L.fgnn_edge_mlp_forward(None, 0, 0, 0, one, one, one, one, one, 8, 288, 7, 4, None) == -1
    assert L.fgnn_edge_mlp_workspace_bytes(4096, 288) > 0
    assert L.fgnn_ldpc_encode(one, one, 4, 65, 48, one, None) == _hip.EUNSUPPORTED
    assert L.fgnn_ldpc_encode(None, None, 0, 48, 48, None, None) == 0          # empty batch: nothing to do
    assert L.fgnn_ldpc_encode(None, one, 4, 48, 48, one, None) == -1
    assert L.fgnn_ldpc_decode(one, one, one, one, one, 4, 2000, 48, 291, 100, one, None, one, one, None) == _hip.EUNSUPPORTED
    assert L.fgnn_ldpc_decode(None, one, one, one, one, 4, 96, 48, 291, 100, one, None, one, one, None) == -1
    assert L.fgnn_ldpc_channel_features(one, one, one, 0.05, one, one, one, one, one, 4, 4096, 48, 3, 6, 0, one, one, one,
                                        one, one, None) == _hip.EUNSUPPORTED
    assert L.fgnn_bn_supported(393216, 64, 1) == 1 and L.fgnn_bn_supported(393216, 60, 1) == 0
    assert L.fgnn_linear_forward_partials(393216, 64, 64) > 0 and L.fgnn_linear_forward_partials(393216, 96, 64) <= 0
    assert L.fgnn_abi_version() == _hip.ABI_VERSION
    # the one-kernel FactorNN layer: packed-parameter count = two 64 x 64 maps + two parity blocks (64 x 256
    # filters) + two hyper-factor blocks (64 x 64 filters); NULL buffers / misaligned operands / empty batches before any launch
    blk = lambda ncol: 64 * 64 + 2 * 64 + 64 * ncol + 2 * 64 + 64 * 64 + 2 * 64
    assert L.fgnn_factor_layer_param_count() == 2 * 64 * 64 + 2 * blk(256) + 2 * blk(64)
    big = ctypes.c_void_p(4096)
    layer = lambda var, B=4: L.fgnn_factor_layer_forward(B, var, big, big, None, None, None, big, 6, 1, big, 3, 1, big, 1152, big, 1152,
                                                         None, None, big, 1, 0.01, big, big, big, None)
    assert layer(None) == -1 and b'null pointer' in L.fgnn_last_error()
    assert layer(ctypes.c_void_p(4100)) == _hip.EUNSUPPORTED and b'misaligned' in L.fgnn_last_error()
    assert layer(big, B=0) == 0
    # the training-mode fused block tail (csrc/block_tail.hip): family = Cout in {64, 128, 256}, 16-byte aligned
    # operands; partial rows fit the BatchNorm workspace the finalisers fold (<= 1024)
    assert 1 <= L.fgnn_block_tail_partials(393216, 256) <= 1024 and 1 <= L.fgnn_block_tail_backward_partials(393216, 64) <= 1024
    assert L.fgnn_block_tail_partials(393216, 96) == 0 and L.fgnn_block_tail_partials(0, 64) == 0
    assert L.fgnn_block_tail_partials(5, 64) == 1
    fin = _hip.BnFinal()
    fin.mean = fin.invstd = fin.scale = fin.shift = 4096
    fin.count = 4096
    fref = ctypes.byref(fin)
    stats = lambda e, part, Cout=64, f=fref: L.fgnn_block_tail_stats(e, big, big, 0.0, big, None, 4096, Cout, part, f, big, None)
    assert stats(None, big) == -1 and b'null pointer' in L.fgnn_last_error()
    assert stats(big, None) == -1 and stats(big, big, f=None) == -1
    assert stats(big, big, Cout=96) == _hip.EUNSUPPORTED and stats(ctypes.c_void_p(4100), big) == _hip.EUNSUPPORTED
    fin.count = 4095
    assert stats(big, big) == -1 and b'count' in L.fgnn_last_error()          # the statistics run over exactly R rows
    fin.count = 4096
    apply = lambda out, add0=None, per=None: L.fgnn_block_tail_apply(big, big, big, 0.0, big, None, big, big, 0.01, add0, None, None, per,
                                                                     out, None, 4096, 128, None)
    assert apply(None) == -1 and apply(ctypes.c_void_p(4104)) == _hip.EUNSUPPORTED
    assert apply(big, add0=ctypes.c_void_p(4104)) == _hip.EUNSUPPORTED and b'misaligned' in L.fgnn_last_error()
    assert apply(big, add0=big, per=(ctypes.c_int32 * 3)(0, 1, 1)) == -1 and b'period' in L.fgnn_last_error()
    bwd = lambda gout, ws, nbytes, dsum=None: L.fgnn_block_tail_backward(big, big, big, 0.0, big, None, big, big, big, big, big, 0.01, gout,
                                                                         big, big, None, None, None, None, None, None, dsum, 4096, 256,
                                                                         ws, nbytes, big, None)
    assert bwd(None, big, 1 << 30) == -1
    assert bwd(big, big, 1024) == -1 and b'workspace too small' in L.fgnn_last_error()
    assert bwd(big, big, 1 << 30, dsum=big) == -1 and b'mean2' in L.fgnn_last_error()
    # the BatchNorm entry points take ONE description of what to finalise (fgnn_bn_final) and check it before any launch
    assert L.fgnn_bn_finalize(big, 2000, 64, fref, None) == -1 and b'bad sizes' in L.fgnn_last_error()
    assert L.fgnn_bn_finalize(big, 16, 64, None, None) == -1
    fin.population = 5
    assert L.fgnn_bn_finalize(big, 16, 64, fref, None) == -1 and b'row counts' in L.fgnn_last_error()
    fin.population = 0
    assert L.fgnn_bn_stats(big, 4000, 64, 1, fref, big, 1 << 30, big, None) == -1 and b'count' in L.fgnn_last_error()
    assert L.fgnn_bn_backward_apply(big, big, big, 4096, 64, 1, big, big, big, big, 0.0, None, None) == -1
    assert L.fgnn_linear_forward(big, big, None, big, 4096, 64, 64, None, fref, big, 0, None) == -1 and b'stats_partials' in L.fgnn_last_error()
    assert L.fgnn_node_sum(big, big, 16, 96, 60, 1, None) == _hip.EUNSUPPORTED and L.fgnn_node_sum(None, big, 16, 96, 64, 1, None) == -1
    assert L.fgnn_node_sum(big, big, 0, 96, 64, 1, None) == 0
    assert L.fgnn_bn_apply(big, big, 4096, 64, 1, big, big, 0.0, big, None, None, (ctypes.c_int32 * 3)(0, 1, 1), None) == -1


def test_flat_adam_matches_torch_adam():
    """dp.FlatAdam on flattened parameters == torch.optim.Adam on the separate tensors (same update rule,
    same op order up to the multi-tensor batching), including weight decay."""
    import torch
    from fgnn_amd.dp import FlatAdam, FlatGradBucket
    torch.manual_seed(0)

    def make():
        torch.manual_seed(3)
        return torch.nn.Sequential(torch.nn.Linear(5, 7), torch.nn.Tanh(), torch.nn.Linear(7, 3))

    a, b = make(), make()
    x, y = torch.randn(11, 5), torch.randn(11, 3)
    ref = torch.optim.Adam(a.parameters(), lr=1e-2, weight_decay=1e-3)
    bucket = FlatGradBucket(b.parameters(), flatten_params=True)
    opt = FlatAdam(bucket, lr=1e-2, weight_decay=1e-3)
    for _ in range(6):
        ref.zero_grad()
        torch.nn.functional.mse_loss(a(x), y).backward()
        ref.step()
        bucket.zero()
        torch.nn.functional.mse_loss(b(x), y).backward()
        opt.step()
    for pa, pb in zip(a.parameters(), b.parameters()):
        assert torch.allclose(pa, pb, rtol=1e-5, atol=1e-7)
    # the module's parameters are views of the flat buffer and state_dict round-trips
    assert all(p.data_ptr() >= bucket.flat_param.data_ptr() for p in b.parameters())
    b.load_state_dict(a.state_dict())
    assert torch.equal(bucket.flat_param[:35].view(7, 5), a[0].weight)


def test_fast_adam_checkpoint_round_trip_and_stock_compatibility():
    """fastpath.FastAdam persists its moments and step count in stock Adam's state_dict layout
    (/root/reference/train_ldpc.py:179-181,187 saves and restores ``optimizer.state_dict()``):
    2 steps -> save -> rebuild -> load -> 1 step  ==  3 straight steps, bit for bit; a stock torch.optim.Adam checkpoint
    resumes in FastAdam and a FastAdam checkpoint resumes in the stock class."""
    import copy
    import io
    import torch
    from fgnn_amd.fastpath import FastAdam

    def make():
        torch.manual_seed(3)
        return torch.nn.Sequential(torch.nn.Linear(5, 7), torch.nn.Tanh(), torch.nn.Linear(7, 3))

    torch.manual_seed(0)
    x, y = torch.randn(11, 5), torch.randn(11, 3)

    def steps(model, opt, n):
        for _ in range(n):
            opt.zero_grad()
            torch.nn.functional.mse_loss(model(x), y).backward()
            opt.step()

    kw = dict(lr=1e-2, weight_decay=1e-3)
    straight = make()
    steps(straight, FastAdam(straight.parameters(), **kw), 3)

    first = make()
    opt = FastAdam(first.parameters(), **kw)
    assert opt.state_dict()['state'] == {}                      # nothing to save before the first step (as the stock class)
    steps(first, opt, 2)
    buf = io.BytesIO()
    torch.save({'model': first.state_dict(), 'opt': opt.state_dict()}, buf)
    sd = opt.state_dict()
    assert sorted(sd['state']) == [0, 1, 2, 3] and set(sd['state'][0]) == {'step', 'exp_avg', 'exp_avg_sq'}
    assert float(sd['state'][0]['step']) == 2.0 and sd['state'][0]['exp_avg'].shape == first[0].weight.shape
    buf.seek(0)
    ck = torch.load(buf)
    resumed = make()
    resumed.load_state_dict(ck['model'])
    opt2 = FastAdam(resumed.parameters(), lr=5.0)               # (the saved hyper-parameters win, as in the stock class)
    opt2.load_state_dict(ck['opt'])
    assert opt2.param_groups[0]['lr'] == 1e-2 and opt2.flat.t == 2
    steps(resumed, opt2, 1)
    for a, b in zip(straight.parameters(), resumed.parameters()):
        assert torch.equal(a, b)

    # stock -> fast and fast -> stock
    sm = make()
    so = torch.optim.Adam(sm.parameters(), **kw)
    steps(sm, so, 2)
    fm = make()
    fm.load_state_dict(sm.state_dict())
    fo = FastAdam(fm.parameters(), **kw)
    fo.load_state_dict(copy.deepcopy(so.state_dict()))
    steps(sm, so, 1)
    steps(fm, fo, 1)
    for a, b in zip(sm.parameters(), fm.parameters()):
        assert torch.allclose(a, b, rtol=1e-5, atol=1e-7)
    back = make()
    back.load_state_dict(first.state_dict())
    bo = torch.optim.Adam(back.parameters(), **kw)
    bo.load_state_dict(ck['opt'])
    steps(back, bo, 1)
    for a, b in zip(straight.parameters(), back.parameters()):
        assert torch.allclose(a, b, rtol=1e-5, atol=1e-7)


def test_fast_path_keeps_adam_a_class_and_modules_copyable():
    """enable_fast_path(): ``torch.optim.Adam`` stays a TYPE (subclassing and isinstance keep working), frozen parameters keep
    the stock optimizer, and an optimised module is deepcopy-able and picklable (hooks, not a replaced ``forward``)."""
    import copy
    import pickle
    import torch
    from fgnn_amd import fastpath
    stock = torch.optim.Adam
    fastpath.enable_fast_path()
    try:
        assert isinstance(torch.optim.Adam, type) and torch.optim.Adam is not stock

        class Mine(torch.optim.Adam):
            pass
        w = torch.nn.Parameter(torch.zeros(3))
        assert isinstance(Mine([w]), stock)
        cpu = torch.optim.Adam([w], lr=0.1)
        assert type(cpu) is stock and isinstance(cpu, torch.optim.Adam)          # CPU parameters: the stock class
        assert isinstance(fastpath.FastAdam([torch.nn.Parameter(torch.zeros(2))]), torch.optim.Adam)
        frozen = torch.nn.Parameter(torch.zeros(3), requires_grad=False)
        assert not fastpath._wants_fast_adam([w, frozen], False, {})
        m = torch.nn.Sequential(torch.nn.Linear(2, 2))
        fastpath.fast_path(m)
        m2 = copy.deepcopy(m)
        m3 = pickle.loads(pickle.dumps(m))
        assert torch.equal(m2(torch.ones(1, 2)), m(torch.ones(1, 2))) and torch.equal(m3[0].weight, m[0].weight)
    finally:
        fastpath.disable_fast_path()
    assert torch.optim.Adam is stock


def test_low_precision_weight_copies_are_refreshed_in_place():
    """pointwise.cast_cached: a cached bf16 copy keeps its storage for life (a captured hipGraph may hold the address)
    and is re-cast in place when the source changed or after invalidate_casts(); parameters moved into one flat buffer
    (dp.flatten_parameters) get views of ONE mirror refreshed as a whole."""
    from fgnn_amd.dp import flatten_parameters
    from fgnn_amd.mpnn import pointwise
    w = torch.nn.Parameter(torch.randn(8, 4))
    c1 = pointwise.cast_cached(w, torch.bfloat16)
    assert c1.dtype == torch.bfloat16 and torch.equal(c1, w.detach().bfloat16())
    assert pointwise.cast_cached(w, torch.bfloat16) is c1 and pointwise.cast_cached(w, torch.float32) is w
    ptr = c1.data_ptr()
    with torch.no_grad():
        w.mul_(2)                                             # version bump
    c2 = pointwise.cast_cached(w, torch.bfloat16)
    assert c2 is c1 and c2.data_ptr() == ptr and torch.equal(c2, w.detach().bfloat16())
    w.data.add_(1)                                            # no version bump: needs the explicit invalidation
    pointwise.invalidate_casts()
    c3 = pointwise.cast_cached(w, torch.bfloat16)
    assert c3 is c1 and torch.equal(c3, w.detach().bfloat16())
    # flat mirror
    lin = torch.nn.Sequential(torch.nn.Linear(4, 3), torch.nn.Linear(3, 2))
    flat = flatten_parameters(lin.parameters())
    ps = list(lin.parameters())
    # Linear(4,3), Linear(3,2): 12 + 3 (+1 pad) + 6 (+2) + 2 (+2) elements — every parameter starts on a 16-byte boundary
    assert flat.numel() == 28 and [(p.data_ptr() - flat.data_ptr()) // 4 for p in ps] == [0, 12, 16, 24]
    assert float(flat[15]) == 0 and float(flat[22:24].abs().sum()) == 0 and float(flat[26:].abs().sum()) == 0
    m0 = pointwise.cast_cached(ps[0], torch.bfloat16)
    m2 = pointwise.cast_cached(ps[2], torch.bfloat16)
    assert m0.shape == ps[0].shape and torch.equal(m0, ps[0].detach().bfloat16()) and torch.equal(m2, ps[2].detach().bfloat16())
    assert m2.data_ptr() - m0.data_ptr() == (ps[2].data_ptr() - ps[0].data_ptr()) // 2      # views of one mirror
    with torch.no_grad():
        flat.mul_(3)                                          # what a flat optimizer does
    pointwise.invalidate_casts()
    m0b = pointwise.cast_cached(ps[0], torch.bfloat16)
    assert m0b.data_ptr() == m0.data_ptr() and torch.equal(m0b, ps[0].detach().bfloat16()) and torch.equal(m2, ps[2].detach().bfloat16())
    lin.load_state_dict({k: v * 0 + 1 for k, v in lin.state_dict().items()})             # writes through the parameters
    assert torch.equal(pointwise.cast_cached(ps[1], torch.bfloat16), torch.ones_like(ps[1]).bfloat16())


# ---------------------------------------------------------------------------------------------
# drop-in shim: the import lines of the reference's four scripts, executed against lib.model.mpnn
# ---------------------------------------------------------------------------------------------
SCRIPT_IMPORT_LINES = [
    # train_ldpc.py:13
    'from lib.model.mpnn import factor_mpnn, FactorNN',
    # train_syn_fixed_pw_hop.py:9
    'from lib.model.mpnn import mp_sequential, mp_conv_residual, mp_conv_type, mp_conv_v2, global_pooling',
    # train_syn_hop_factor.py:8, train_syn_pw_factor.py:8
    'from lib.model.mpnn import factor_mpnn',
    # lib/model/mpnn/__init__.py:1-7, the package's full export list
    'from lib.model.mpnn import (mp_conv_v2, mp_conv_type, mp_conv_residual, mp_ensemble, mp_sequential, '
    'global_pooling, factor_mpnn, FactorNN)',
]


@pytest.mark.parametrize('line', SCRIPT_IMPORT_LINES)
def test_reference_script_import_lines_resolve_through_the_shim(line):
    import fgnn_amd.mpnn as native
    scope = {}
    exec(line, scope)                                   # noqa: S102 — the literal import statement of the script
    names = [n for n in scope if not n.startswith('__')]
    assert names
    for n in names:
        assert scope[n] is getattr(native, n), n        # the MI355X-native class, not a stand-in


def test_global_pooling_and_ensemble_follow_the_reference_contract():
    """pooling.py:11-47 / ensemble.py:8-19: pooled feature of the INPUT broadcast to every node and concatenated on
    the channel axis; the ensemble concatenates two graph branches.  (Glue modules: torch ops, CPU-testable.)"""
    from fgnn_amd.mpnn import base_mp_nn, global_pooling, mp_ensemble, parallel_net, identity_module
    torch.manual_seed(0)
    x = torch.randn(3, 5, 7, 1)
    idx, et = torch.zeros(3, 7, 2, dtype=torch.int64), torch.zeros(3, 1, 7, 2)
    gp = global_pooling()
    assert isinstance(gp, base_mp_nn) and gp.is_mp_nn
    y = gp(x, idx, et)
    assert y.shape == (3, 10, 7, 1) and torch.equal(y[:, :5], x)
    assert torch.equal(y[:, 5:], x.max(dim=2, keepdim=True)[0].repeat(1, 1, 7, 1))

    class Twice(base_mp_nn):
        def forward(self, x, nn_idx, etype):
            return 2 * x

    gmap = torch.nn.Conv2d(5, 4, 1)
    gp = global_pooling(orig_mapper=Twice(), gfeature_mapper=gmap)
    assert sorted(k.split('.')[0] for k in gp.state_dict()) == ['gfeature_mapper', 'gfeature_mapper']
    y = gp(x, idx, et)
    assert torch.equal(y[:, :5], 2 * x)
    assert torch.allclose(y[:, 5:], gmap(x.max(dim=2, keepdim=True)[0]).repeat(1, 1, 7, 1))
    ens = mp_ensemble(Twice(), Twice(), torch.nn.Conv2d(10, 3, 1))
    assert ens(x, idx, et, idx, et).shape == (3, 3, 7, 1)
    par = parallel_net(Twice(), identity_module())
    assert torch.equal(par(x, idx, et), 3 * x)


def test_callable_and_none_aggregators_are_accepted():
    """mp_nn.py:89-90: any callable (or None) is a legal aggregator; construction must not raise."""
    from fgnn_amd.mpnn import mp_conv_type, mp_conv_v2
    m = mp_conv_v2(4, 4, 2, extension=mp_conv_type.NO_EXTENSION, aggregtor=lambda e: e.sum(dim=3, keepdim=True))
    assert callable(m.aggregtor)
    assert mp_conv_v2(4, 4, 2, aggregtor=None).aggregtor is None
    with pytest.raises(ValueError):
        mp_conv_v2(4, 4, 2, aggregtor=3)


def test_gradient_sink_is_opt_in():
    """ops.grad_sink: kernels add into ``param.grad`` only for parameters a FlatGradBucket (or enable_grad_sink) opted in;
    a bare module with a leftover dense .grad gets ordinary returned gradients (hooks, autograd.grad keep working)."""
    from fgnn_amd import ops
    from fgnn_amd.dp import FlatGradBucket
    lin = torch.nn.Linear(3, 2)
    lin.weight.grad = torch.zeros_like(lin.weight)
    assert ops.grad_sink(lin.weight) is None
    bucket = FlatGradBucket(lin.parameters())
    assert ops.grad_sink(lin.weight) is lin.weight.grad and lin.weight.grad.data_ptr() == bucket.flat.data_ptr()
    ops.enable_grad_sink([lin.weight], False)
    assert ops.grad_sink(lin.weight) is None and ops.grad_sink(lin.bias) is lin.bias.grad


def test_pointwise_map_outside_the_hip_path_keeps_weight_gradients():
    """PointwiseConv2d on a dtype / device the hand-written path does not take (here: float64 on CPU) must stay
    differentiable w.r.t. its parameters — the cached low-precision copies are detached and only for no-grad paths."""
    from fgnn_amd.mpnn import PointwiseConv2d
    torch.manual_seed(0)
    m = PointwiseConv2d(3, 2).double()
    x = torch.randn(2, 3, 5, 1, dtype=torch.float64, requires_grad=True)
    m(x).square().sum().backward()
    assert m.weight.grad is not None and float(m.weight.grad.abs().sum()) > 0 and m.bias.grad is not None
    ref = torch.nn.Conv2d(3, 2, 1).double()
    ref.load_state_dict(m.state_dict())
    ref(x.detach()).square().sum().backward()
    assert torch.allclose(m.weight.grad, ref.weight.grad) and torch.allclose(m.bias.grad, ref.bias.grad)
    with torch.no_grad():                               # no-grad path: cached copy, same values
        assert torch.allclose(m(x), ref(x))


def test_bench_gpus_n_never_degrades_to_one_rank():
    """`bench.py --gpus N` without a launcher starts its own N ranks; with fewer than N devices it must refuse (this box
    has none) rather than print a 1-rank line, and under a launcher WORLD_SIZE must equal --gpus."""
    import subprocess
    import sys
    bench = os.path.join(ROOT, 'bench.py')
    env = {k: v for k, v in os.environ.items() if k not in ('WORLD_SIZE', 'RANK', 'LOCAL_RANK', 'FGNN_BENCH_DEVICE')}
    if torch.cuda.is_available() and torch.cuda.device_count() >= 2:
        pytest.skip('a multi-GPU node: the refusal cannot be provoked')
    r = subprocess.run([sys.executable, bench, '--gpus', '2', '--steps', '1', '--warmup', '0'], env=env,
                       capture_output=True, text=True, timeout=300)
    assert r.returncode != 0 and 'one rank per GPU needs 2' in r.stderr and r.stdout.strip() == ''
    r = subprocess.run([sys.executable, bench, '--gpus', '1', '--steps', '1', '--warmup', '0'],
                       env=dict(env, WORLD_SIZE='2', RANK='0', LOCAL_RANK='0'), capture_output=True, text=True, timeout=300)
    assert r.returncode != 0 and 'WORLD_SIZE=2' in r.stderr and r.stdout.strip() == ''


def test_decoding_loss_on_the_cpu_is_the_reference_expression():
    """ldpc.decoding_loss off the GPU: train_ldpc.py:222-227's two torch calls, differentiable."""
    from fgnn_amd.ldpc import decoding_loss
    g = torch.Generator().manual_seed(0)
    logits = torch.randn(5, 48, generator=g, requires_grad=True)
    pred = torch.rand(5, 1, generator=g, requires_grad=True)
    label = torch.randint(0, 2, (5, 48), generator=g).float()
    sigma_b = torch.randint(0, 6, (5,), generator=g).float()
    loss = decoding_loss(logits, pred, label, sigma_b)
    ref = (torch.nn.functional.binary_cross_entropy_with_logits(logits.view(-1), label.view(-1)) +
           0.1 * torch.nn.functional.mse_loss(pred.view(-1), torch.pow(10.0, sigma_b / 20).view(-1)))
    assert torch.equal(loss, ref)
    loss.backward()
    assert logits.grad is not None and pred.grad is not None


def test_cpu_gradient_bucket_never_needs_the_hip_library(monkeypatch):
    """Round-5 advisory: FlatGradBucket.zero() runs every step; on a CPU / gloo bucket it must not load libfgnn_hip.so (hosts of
    the data-parallel CPU path may not have it)."""
    from fgnn_amd import _hip, dp

    def boom():
        raise _hip.FgnnHipError('libfgnn_hip.so must not be loaded by a CPU bucket')
    monkeypatch.setattr(_hip, 'lib', boom)
    lin = torch.nn.Linear(4, 3)
    bucket = dp.FlatGradBucket(lin.parameters())
    lin(torch.randn(2, 4)).sum().backward()
    assert float(bucket.flat.abs().sum()) > 0
    bucket.zero()
    assert float(bucket.flat.abs().sum()) == 0.0


def test_fan_box_drops_the_deposits_of_another_backward_pass():
    """Round-5 advisory: deposits left behind by a pass that cut the fan-out node off must not be multiplied into a later pass."""
    from fgnn_amd import ops
    box = ops.FanBox.__new__(ops.FanBox)
    box.slots, box.wg, box.task, box._ph = ['stale', None, None], ['stale-job', None, None], 12345, None
    box._enter_pass()                       # (outside any backward pass the current task id is -1: another pass)
    assert box.slots == [None, None, None] and box.wg == [None, None, None] and box.task == -1
    box.slots[0] = 'mine'
    box._enter_pass()                       # same pass: kept
    assert box.slots[0] == 'mine'
