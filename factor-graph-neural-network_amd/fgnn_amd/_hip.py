"""ctypes binding of libfgnn_hip.so (C ABI declared in include/fgnn_hip.h).

PyTorch is only the owner of device memory and streams here: every call passes raw
device pointers, element strides and the current HIP stream.  There is no CPU or
eager-PyTorch fallback: if the shared library is missing the import of the operator
fails loudly (build it with ``python __graft_entry__.py`` or ``make -C csrc``).
"""
import ctypes
import os

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get('FGNN_HIP_LIB') or os.path.join(_HERE, 'libfgnn_hip.so')      # FGNN_HIP_LIB: a tuning build

EXT_NONE, EXT_NEIGHBOR, EXT_DIFF = 0, 1, 2
DESC_GETYPE_REDUCED = 0x10000
DESC_IDENTITY_LIST = 0x20000       # forward: the one-destination call's neighbour table is idx[j] == j
AGG_MAX, AGG_LSE, AGG_MEAN = 0, 1, 2
F32, BF16 = 0, 1
ABI_VERSION = 13             # include/fgnn_hip.h: FGNN_ABI_VERSION (checked before any symbol is bound)
EUNSUPPORTED = -3            # FGNN_EUNSUPPORTED: shape outside a kernel's family (callers fall back)

AGG_CODES = {'max': AGG_MAX, 'softmax': AGG_LSE, 'mean': AGG_MEAN}

EXPORTS = ('fgnn_mpconv_forward', 'fgnn_mpconv_backward', 'fgnn_mpconv_forward_lds_bytes',
           'fgnn_mpconv_backward_workspace_bytes', 'fgnn_mpconv_backward_reduces_getype', 'fgnn_linear_wgrad', 'fgnn_linear_wgrad_workspace_bytes',
           'fgnn_linear_wgrad_multi', 'fgnn_linear_wgrad_multi_workspace_bytes',
           'fgnn_instnorm_forward', 'fgnn_instnorm_backward', 'fgnn_instnorm_dot_forward', 'fgnn_instnorm_dot_backward',
           'fgnn_instnorm_dot_workspace_bytes', 'fgnn_bn_supported', 'fgnn_bn_workspace_bytes',
           'fgnn_bn_stats', 'fgnn_bn_finalize', 'fgnn_bn_apply', 'fgnn_bn_backward',
           'fgnn_linear_forward', 'fgnn_linear_forward_partials', 'fgnn_linear_instnorm_forward', 'fgnn_sum_n', 'fgnn_concat_pair', 'fgnn_concat_rows', 'fgnn_flat_adam', 'fgnn_flat_adam_dev', 'fgnn_edge_mlp_forward', 'fgnn_edge_mlp_workspace_bytes', 'fgnn_edge_mlp_backward', 'fgnn_ldpc_encode', 'fgnn_ldpc_channel_features', 'fgnn_ldpc_channel_features_rng', 'fgnn_ldpc_decode', 'fgnn_ldpc_loss_forward', 'fgnn_ldpc_loss_backward', 'fgnn_ldpc_loss_workspace_bytes', 'fgnn_linear_multi_supported', 'fgnn_linear_multi_forward', 'fgnn_fold_defer', 'fgnn_fold_pending', 'fgnn_fold_discard', 'fgnn_fold_flush', 'fgnn_mpconv_block_forward', 'fgnn_mpconv_block_forward_rows', 'fgnn_mpconv_block_forward_fanout', 'fgnn_mpconv_block_forward_fanin', 'fgnn_factor_layer_forward', 'fgnn_factor_layer_param_count', 'fgnn_mpconv_forward_stats', 'fgnn_mpconv_forward_stats_partials', 'fgnn_block_tail_partials', 'fgnn_block_tail_stats', 'fgnn_block_tail_apply', 'fgnn_block_tail_backward', 'fgnn_block_tail_backward_partials', 'fgnn_block_tail_moments_bytes', 'fgnn_block_tail_backward_moments', 'fgnn_block_tail_wgrad_finish', 'fgnn_bn_backward_apply', 'fgnn_block_head_backward', 'fgnn_node_sum', 'fgnn_set_inkernel_finalisers', 'fgnn_mpconv_backward_tables_bytes', 'fgnn_mpconv_backward_tables', 'fgnn_mpconv_backward_with_tables',
           'fgnn_mpconv_algorithmic_bytes', 'fgnn_mpconv_forward_addends', 'fgnn_last_error', 'fgnn_last_kernel', 'fgnn_stamp', 'fgnn_spin', 'fgnn_set_ext_backward_pieces', 'fgnn_abi_version')


class MPConvDesc(ctypes.Structure):
    _fields_ = [(n, ctypes.c_int32) for n in
                ('B', 'nin', 'nou', 'net', 'N', 'M', 'k', 'ext', 'agg', 'dtype', 'relu', 'reserved')] + \
               [(n, ctypes.c_int64) for n in
                ('x_sb', 'x_sc', 'x_sn', 'idx_sb', 'idx_sm', 'idx_sk',
                 'et_sb', 'et_se', 'et_sm', 'et_sk', 'y_sb', 'y_sc', 'y_sm')]


class BnFinal(ctypes.Structure):
    """include/fgnn_hip.h: fgnn_bn_final — what a statistics-producing launch finalises in its last workgroup."""
    _fields_ = [(n, ctypes.c_void_p) for n in ('gamma', 'beta', 'running_mean', 'running_var', 'num_batches_tracked',
                                               'mean', 'invstd', 'scale', 'shift', 'shift_k')] + \
               [('count', ctypes.c_int64), ('population', ctypes.c_int64), ('momentum', ctypes.c_float), ('eps', ctypes.c_float)]


FOLD_SCRATCH_BYTES = 512 + 64 * 512 * 8       # include/fgnn_hip.h: FGNN_FOLD_SCRATCH_BYTES


def bn_final(stats, gamma, beta, running_mean, running_var, num_batches_tracked, momentum, eps, count, population=0):
    """fgnn_bn_final for a BatchNorm whose outputs go to ``stats`` [4, C] f32 (rows: mean, invstd, scale, shift).  The caller keeps
    the tensors alive until the launch is enqueued."""
    f = BnFinal()
    dp = lambda t: None if t is None else t.data_ptr()
    f.gamma, f.beta = dp(gamma), dp(beta)
    f.running_mean, f.running_var, f.num_batches_tracked = dp(running_mean), dp(running_var), dp(num_batches_tracked)
    f.mean, f.invstd, f.scale, f.shift = (stats[i].data_ptr() for i in range(4))
    f.shift_k = None
    f.count, f.population = int(count), int(population)
    f.momentum, f.eps = float(momentum), float(eps)
    return f


class FgnnHipError(RuntimeError):
    pass


_lib = None


def lib():
    """Load libfgnn_hip.so once; raise (never fall back) when it is absent."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise FgnnHipError(
            'libfgnn_hip.so not found at %s — the FGNN message operator has no CPU/eager '
            'fallback; build it with `python __graft_entry__.py` (hipcc --offload-arch=gfx950).'
            % LIB_PATH)
    L = ctypes.CDLL(LIB_PATH)
    try:
        L.fgnn_abi_version.restype = ctypes.c_int
        have = int(L.fgnn_abi_version())
    except AttributeError:
        have = None
    if have != ABI_VERSION:
        raise FgnnHipError('%s speaks C-ABI version %s, this package binds version %d (include/fgnn_hip.h): rebuild it '
                           'with `python __graft_entry__.py`' % (LIB_PATH, have, ABI_VERSION))
    vp, dp = ctypes.c_void_p, ctypes.POINTER(MPConvDesc)
    L.fgnn_mpconv_forward.restype = ctypes.c_int
    L.fgnn_mpconv_forward.argtypes = [dp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp]
    L.fgnn_mpconv_forward_addends.restype = ctypes.c_int
    L.fgnn_mpconv_forward_addends.argtypes = [dp] + [vp] * 12
    L.fgnn_mpconv_forward_stats.restype = ctypes.c_int
    fp = ctypes.POINTER(BnFinal)
    L.fgnn_mpconv_forward_stats.argtypes = [dp] + [vp] * 8 + [fp, vp, vp]
    L.fgnn_mpconv_forward_stats_partials.restype = ctypes.c_int
    L.fgnn_mpconv_forward_stats_partials.argtypes = [dp]
    L.fgnn_mpconv_backward.restype = ctypes.c_int
    L.fgnn_mpconv_backward.argtypes = [dp] + [vp] * 12 + [ctypes.c_int64, vp]
    L.fgnn_mpconv_backward_with_tables.restype = ctypes.c_int
    L.fgnn_mpconv_backward_with_tables.argtypes = [dp] + [vp] * 12 + [ctypes.c_int64, vp, vp]
    L.fgnn_mpconv_backward_tables_bytes.restype = ctypes.c_int64
    L.fgnn_mpconv_backward_tables_bytes.argtypes = [dp]
    L.fgnn_mpconv_backward_tables.restype = ctypes.c_int
    L.fgnn_mpconv_backward_tables.argtypes = [dp, vp, vp, vp]
    L.fgnn_mpconv_backward_reduces_getype.restype = ctypes.c_int
    L.fgnn_mpconv_backward_reduces_getype.argtypes = [dp]
    L.fgnn_mpconv_backward_workspace_bytes.restype = ctypes.c_int64
    L.fgnn_mpconv_backward_workspace_bytes.argtypes = [dp]
    L.fgnn_mpconv_forward_lds_bytes.restype = ctypes.c_int64
    L.fgnn_mpconv_forward_lds_bytes.argtypes = [dp]
    L.fgnn_mpconv_algorithmic_bytes.restype = ctypes.c_int64
    L.fgnn_mpconv_algorithmic_bytes.argtypes = [dp]
    i32, i64 = ctypes.c_int32, ctypes.c_int64
    L.fgnn_linear_wgrad.restype = ctypes.c_int
    L.fgnn_linear_wgrad.argtypes = [vp, vp, i64, i32, i32, i32, vp, vp, vp, i64, vp]
    L.fgnn_linear_wgrad_workspace_bytes.restype = i64
    L.fgnn_linear_wgrad_workspace_bytes.argtypes = [i64, i32, i32]
    L.fgnn_linear_wgrad_multi.restype = ctypes.c_int
    L.fgnn_linear_wgrad_multi.argtypes = [vp, i64, i32, i32, vp, vp, vp, vp, vp, i64, vp]
    L.fgnn_linear_wgrad_multi_workspace_bytes.restype = i64
    L.fgnn_linear_wgrad_multi_workspace_bytes.argtypes = [i64, i32, i32, vp]
    L.fgnn_instnorm_forward.restype = ctypes.c_int
    L.fgnn_instnorm_forward.argtypes = [vp, vp, i32, i32, i32, i32, i32, vp]
    L.fgnn_instnorm_backward.restype = ctypes.c_int
    L.fgnn_instnorm_backward.argtypes = [vp, vp, vp, i32, i32, i32, i32, i32, vp]
    L.fgnn_linear_multi_supported.restype = ctypes.c_int
    L.fgnn_linear_multi_supported.argtypes = [i64, vp, i32]
    L.fgnn_linear_multi_forward.restype = ctypes.c_int
    L.fgnn_linear_multi_forward.argtypes = [vp, vp, vp, vp, vp, i64, i32, vp]
    L.fgnn_fold_defer.restype = ctypes.c_int
    L.fgnn_fold_defer.argtypes = [i32]
    L.fgnn_fold_pending.restype = ctypes.c_int
    L.fgnn_fold_pending.argtypes = []
    L.fgnn_fold_discard.restype = None
    L.fgnn_fold_discard.argtypes = []
    L.fgnn_fold_flush.restype = ctypes.c_int
    L.fgnn_fold_flush.argtypes = [vp]
    L.fgnn_instnorm_dot_forward.restype = ctypes.c_int
    L.fgnn_instnorm_dot_forward.argtypes = [vp, vp, vp, vp, i32, i32, i32, i32, vp]
    L.fgnn_instnorm_dot_backward.restype = ctypes.c_int
    L.fgnn_instnorm_dot_backward.argtypes = [vp, vp, vp, vp, vp, vp, i32, i32, i32, i32, vp, i64, vp]
    L.fgnn_instnorm_dot_workspace_bytes.restype = i64
    L.fgnn_instnorm_dot_workspace_bytes.argtypes = [i32]
    f32 = ctypes.c_float
    L.fgnn_bn_supported.restype = ctypes.c_int
    L.fgnn_bn_supported.argtypes = [i64, i32, i32]
    L.fgnn_bn_workspace_bytes.restype = i64
    L.fgnn_bn_workspace_bytes.argtypes = [i64, i32]
    L.fgnn_bn_stats.restype = ctypes.c_int
    L.fgnn_bn_stats.argtypes = [vp, i64, i32, i32, fp, vp, i64, vp, vp]
    L.fgnn_bn_finalize.restype = ctypes.c_int
    L.fgnn_bn_finalize.argtypes = [vp, i32, i32, fp, vp]
    L.fgnn_linear_forward.restype = ctypes.c_int
    L.fgnn_linear_forward.argtypes = [vp, vp, vp, vp, i64, i32, i32, vp, fp, vp, i32, vp]
    L.fgnn_set_inkernel_finalisers.restype = ctypes.c_int
    L.fgnn_set_inkernel_finalisers.argtypes = [i32]
    L.fgnn_node_sum.restype = ctypes.c_int
    L.fgnn_node_sum.argtypes = [vp, vp, i64, i32, i32, i32, vp]
    L.fgnn_linear_instnorm_forward.restype = ctypes.c_int
    L.fgnn_linear_instnorm_forward.argtypes = [vp, vp, vp, vp, vp, i32, i32, i32, i32, i32, f32, vp]
    L.fgnn_ldpc_loss_forward.restype = ctypes.c_int
    L.fgnn_ldpc_loss_forward.argtypes = [vp, vp, vp, vp, i64, i32, i32, f32, vp, vp, i64, vp]
    L.fgnn_ldpc_loss_workspace_bytes.restype = i64
    L.fgnn_ldpc_loss_workspace_bytes.argtypes = []
    L.fgnn_ldpc_loss_backward.restype = ctypes.c_int
    L.fgnn_ldpc_loss_backward.argtypes = [vp, vp, vp, vp, vp, i64, i32, i32, f32, vp, vp, vp]
    L.fgnn_linear_forward_partials.restype = ctypes.c_int
    L.fgnn_linear_forward_partials.argtypes = [i64, i32, i32]
    L.fgnn_mpconv_block_forward.restype = ctypes.c_int
    L.fgnn_mpconv_block_forward.argtypes = [dp] + [vp] * 12 + [f32, i32, i32, vp, vp, vp, vp, vp]
    L.fgnn_mpconv_block_forward_rows.restype = ctypes.c_int
    L.fgnn_mpconv_block_forward_rows.argtypes = [dp] + [vp] * 12 + [f32, i32, i32, vp, vp, vp, i32, vp, vp]
    L.fgnn_mpconv_block_forward_fanout.restype = ctypes.c_int
    L.fgnn_mpconv_block_forward_fanout.argtypes = [dp] + [vp] * 11 + [f32, i32, i32, vp, vp, vp, vp, vp]
    L.fgnn_mpconv_block_forward_fanin.restype = ctypes.c_int
    L.fgnn_mpconv_block_forward_fanin.argtypes = [dp] + [vp] * 11 + [f32, i32, i32, vp, vp, vp, vp, vp]
    L.fgnn_factor_layer_param_count.restype = ctypes.c_int64
    L.fgnn_factor_layer_param_count.argtypes = []
    L.fgnn_factor_layer_forward.restype = ctypes.c_int
    L.fgnn_factor_layer_forward.argtypes = [i32] + [vp] * 6 + [vp, i32, i32, vp, i32, i32, vp, i64, vp, i64, vp, vp, vp, i32, f32,
                                            vp, vp, vp, vp]
    L.fgnn_flat_adam.restype = ctypes.c_int
    L.fgnn_flat_adam.argtypes = [vp, vp, vp, vp, vp, i64, f32, f32, f32, f32, f32, f32, i64, vp]
    L.fgnn_flat_adam_dev.restype = ctypes.c_int
    L.fgnn_flat_adam_dev.argtypes = [vp, vp, vp, vp, vp, i64, vp, f32, f32, f32, f32, f32, vp, vp, vp]
    L.fgnn_sum_n.restype = ctypes.c_int
    L.fgnn_sum_n.argtypes = [vp, i32, i64, i32, vp, vp]
    L.fgnn_concat_pair.restype = ctypes.c_int
    L.fgnn_concat_pair.argtypes = [vp, vp, vp] + [i64] * 8 + [vp]
    L.fgnn_concat_rows.restype = ctypes.c_int
    L.fgnn_concat_rows.argtypes = [vp, vp, vp] + [i64] * 8 + [vp]
    L.fgnn_ldpc_decode.restype = ctypes.c_int
    L.fgnn_ldpc_decode.argtypes = [vp, vp, vp, vp, vp, i64, i32, i32, i32, i32, vp, vp, vp, vp, vp]
    L.fgnn_ldpc_encode.restype = ctypes.c_int
    L.fgnn_ldpc_encode.argtypes = [vp, vp, i64, i32, i32, vp, vp]
    L.fgnn_ldpc_channel_features.restype = ctypes.c_int
    L.fgnn_ldpc_channel_features.argtypes = [vp, vp, vp, f32, vp, vp, vp, vp, vp, i64, i32, i32, i32, i32, i32, vp, vp, vp,
                                             vp, vp, vp]
    L.fgnn_ldpc_channel_features_rng.restype = ctypes.c_int
    L.fgnn_ldpc_channel_features_rng.argtypes = [vp, vp, vp, f32, ctypes.c_uint64, ctypes.c_uint64, vp, vp, i64, i32, i32, i32, i32,
                                                 i32, vp, vp, vp, vp, vp, vp]
    L.fgnn_edge_mlp_forward.restype = ctypes.c_int
    L.fgnn_edge_mlp_forward.argtypes = [vp, i64, i64, i64, vp, vp, vp, vp, vp, i64, i32, i32, i32, vp]
    L.fgnn_edge_mlp_workspace_bytes.restype = i64
    L.fgnn_edge_mlp_workspace_bytes.argtypes = [i64, i32]
    L.fgnn_edge_mlp_backward.restype = ctypes.c_int
    L.fgnn_edge_mlp_backward.argtypes = [vp, i64, i64, i64, vp, i64, i64, i64, vp, vp, vp, i64, i32, i32, i32, vp, vp, vp, vp, vp, i64, vp]
    L.fgnn_block_tail_partials.restype = ctypes.c_int
    L.fgnn_block_tail_partials.argtypes = [i64, i32]
    L.fgnn_block_tail_stats.restype = ctypes.c_int
    L.fgnn_block_tail_stats.argtypes = [vp, vp, vp, f32, vp, vp, i64, i32, vp, fp, vp, vp]
    L.fgnn_block_tail_apply.restype = ctypes.c_int
    L.fgnn_block_tail_apply.argtypes = [vp, vp, vp, f32, vp, vp, vp, vp, f32, vp, vp, vp, vp, vp, vp, i64, i32, vp]
    L.fgnn_block_tail_backward.restype = ctypes.c_int
    L.fgnn_block_tail_backward.argtypes = [vp, vp, vp, f32, vp, vp, vp, vp, vp, vp, vp, f32, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp,
                                           i64, i32, vp, i64, vp, vp]
    L.fgnn_block_tail_moments_bytes.restype = i64
    L.fgnn_block_tail_moments_bytes.argtypes = [i64, i32]
    L.fgnn_block_tail_backward_moments.restype = ctypes.c_int
    L.fgnn_block_tail_backward_moments.argtypes = [vp, vp, vp, f32, vp, vp, vp, vp, vp, vp, vp, f32, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp,
                                                   i64, i32, vp, i64, vp, vp, i64, vp]
    L.fgnn_block_tail_wgrad_finish.restype = ctypes.c_int
    L.fgnn_block_tail_wgrad_finish.argtypes = [vp, i64, i64, i32, vp, vp, vp, vp, vp]
    L.fgnn_block_tail_backward_partials.restype = ctypes.c_int
    L.fgnn_block_tail_backward_partials.argtypes = [i64, i32]
    L.fgnn_block_head_backward.restype = ctypes.c_int
    L.fgnn_block_head_backward.argtypes = [vp, vp, vp, vp, vp, vp, f32, vp, vp, vp, vp, vp, i64, i32, vp, i64, vp, vp]
    L.fgnn_bn_backward_apply.restype = ctypes.c_int
    L.fgnn_bn_backward_apply.argtypes = [vp, vp, vp, i64, i32, i32, vp, vp, vp, vp, f32, vp, vp]
    L.fgnn_bn_apply.restype = ctypes.c_int
    L.fgnn_bn_apply.argtypes = [vp, vp, i64, i32, i32, vp, vp, f32, vp, vp, vp, vp, vp]
    L.fgnn_bn_backward.restype = ctypes.c_int
    L.fgnn_bn_backward.argtypes = [vp, vp, vp, i64, i32, i32, vp, vp, vp, vp, f32, vp, vp, vp, i64, vp, vp]
    L.fgnn_set_ext_backward_pieces.restype = ctypes.c_int
    L.fgnn_set_ext_backward_pieces.argtypes = [ctypes.c_int]
    L.fgnn_spin.restype = ctypes.c_int
    L.fgnn_spin.argtypes = [ctypes.c_int64, vp]
    L.fgnn_stamp.restype = ctypes.c_int
    L.fgnn_stamp.argtypes = [vp, vp]
    L.fgnn_last_error.restype = ctypes.c_char_p
    L.fgnn_last_kernel.restype = ctypes.c_char_p
    L.fgnn_abi_version.restype = ctypes.c_int
    _lib = L
    return L


def check(rc):
    if rc != 0:
        raise FgnnHipError('libfgnn_hip: %s (code %d)' % (lib().fgnn_last_error().decode(), rc))


def _ptr(t):
    return None if t is None else ctypes.c_void_p(t.data_ptr())


def stream_ptr():
    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


def dtype_code(t):
    if t.dtype == torch.float32:
        return F32
    if t.dtype == torch.bfloat16:
        return BF16
    raise FgnnHipError('fgnn_amd supports float32 and bfloat16 tensors, got %s' % t.dtype)


def make_desc(x, nn_idx, etype, nou, net, ext, agg, relu, y=None):
    """Describe one operator call from tensor views (no copies).

    x [B,nin,N,1] (any strides), nn_idx [B,M,k] int64, etype [B,net,M,k].  A batch
    stride of 0 (``expand``-ed tensors) marks a graph / edge weights shared by the batch.
    """
    B, nin, N = x.shape[0], x.shape[1], x.shape[2]
    M, k = nn_idx.shape[1], nn_idx.shape[2]
    d = MPConvDesc()
    d.B, d.nin, d.nou, d.net, d.N, d.M, d.k = B, nin, nou, net, N, M, k
    d.ext, d.agg, d.dtype, d.relu = ext, agg, dtype_code(x), int(bool(relu))
    d.x_sb, d.x_sc, d.x_sn = x.stride(0), x.stride(1), x.stride(2)
    d.idx_sb, d.idx_sm, d.idx_sk = nn_idx.stride(0), nn_idx.stride(1), nn_idx.stride(2)
    d.et_sb, d.et_se, d.et_sm, d.et_sk = (etype.stride(0), etype.stride(1),
                                          etype.stride(2), etype.stride(3))
    if y is not None:
        d.y_sb, d.y_sc, d.y_sm = y.stride(0), y.stride(1), y.stride(2)
    return d
