"""``mp_conv_v2`` — the FGNN Variable->Factor / Factor->Variable message operator.

Drop-in for the reference class of the same name
(/root/reference/lib/model/mpnn/mp_nn.py:13-175): same constructor signature (including
the misspelt ``aggregtor`` keyword), same parameter / buffer names and shapes
(``filters [R, nou*net]`` with column ``o*net+e``, ``bias [nou]``, ``bn.*``) so reference
checkpoints load, same ``forward(x, nn_idx, etype) -> [B, nou, M, 1]``.

Unlike the reference, which chains ~11 ATen ops per call, ``forward`` is ONE fused HIP
kernel on gfx950 (csrc/mpconv_fwd.hip) plus — in training mode only — the batch-statistics
BatchNorm.  There is no CPU path: CPU tensors raise.
"""
from enum import Enum

import torch

from .. import _hip, ops
from .pointwise import add_all


class mp_conv_type(Enum):       # mp_nn.py:7-10
    NO_EXTENSION = 0
    ORIG_WITH_NEIGHBOR = 1
    ORIG_WITH_DIFF = 2


class base_mp_nn(torch.nn.Module):
    """Marker base: callers dispatch ``m(x, nn_idx, etype)`` vs ``m(x)`` on
    ``isinstance(m, base_mp_nn)`` (base_model.py:4-16)."""
    NO_EXTENSION = 0
    ORIG_WITH_NEIGHBOR = 1
    ORIG_WITH_DIFF = 2

    def __init__(self):
        super().__init__()
        self.is_mp_nn = True


_EXT_CODE = {mp_conv_type.NO_EXTENSION: _hip.EXT_NONE,
             mp_conv_type.ORIG_WITH_NEIGHBOR: _hip.EXT_NEIGHBOR,
             mp_conv_type.ORIG_WITH_DIFF: _hip.EXT_DIFF}


def fold_batchnorm(bn):
    """Eval-mode BatchNorm as a per-channel affine (scale, shift); recomputed only when a parameter / buffer changed (the
    same key and storage-preserving refresh as BatchNormAct2d._folded: an inference hipGraph keeps its addresses)."""
    from .pointwise import refresh_in_place, state_epoch
    key = (bn.weight._version, bn.bias._version, bn.running_mean._version, bn.running_var._version, bn.weight.device,
           bn.weight.dtype, state_epoch())
    if getattr(bn, '_fgnn_fold_key', None) != key:
        with torch.no_grad():
            scale = bn.weight.detach() * torch.rsqrt(bn.running_var + bn.eps)
            shift = bn.bias.detach() - bn.running_mean * scale
        bn._fgnn_fold = refresh_in_place(getattr(bn, '_fgnn_fold', None), (scale, shift))
        bn._fgnn_fold_key = key
    return bn._fgnn_fold


class mp_conv_v2(base_mp_nn):
    def __init__(self, nin, nou, nedge_types, bias=True, bn=True,
                 extension=mp_conv_type.ORIG_WITH_DIFF, activation_fn='relu',
                 aggregtor='softmax'):
        super().__init__()
        if extension not in _EXT_CODE:
            raise ValueError("extension must one of mp_conv_type")
        self.nin, self.nou, self.nedge_types = nin, nou, nedge_types
        self.extension = extension
        rows = nin if extension == mp_conv_type.NO_EXTENSION else 2 * nin
        self.filters = torch.nn.Parameter(
            torch.empty(rows, nou * nedge_types, dtype=torch.float32).uniform_(-0.01, 0.01))
        self.bias = torch.nn.Parameter(torch.empty(nou).uniform_(0, 0.05)) if bias else None
        from .pointwise import BatchNormAct2d
        # slope 1.0 = plain BatchNorm2d; forward() asks for the fused ReLU per call when the activation is ReLU
        self.bn = BatchNormAct2d(nou, slope=1.0) if bn else None
        if isinstance(activation_fn, torch.nn.Module):
            self.activation_fn = activation_fn
        elif activation_fn == 'relu':
            self.activation_fn = torch.nn.ReLU(inplace=True)
        else:
            self.activation_fn = None
        if isinstance(aggregtor, str):
            print('aggregator = ', aggregtor)           # the reference announces it too
            if aggregtor not in _hip.AGG_CODES:
                raise ValueError("aggregator must be 'max', 'softmax' or 'mean', got %r" % aggregtor)
        elif aggregtor is not None and not callable(aggregtor):
            raise ValueError('aggregtor must be a string, a callable on [B, nou, M, k] or None, got %r' % (aggregtor,))
        self.aggregtor = aggregtor

    def extra_repr(self):
        return 'nin=%d, nou=%d, nedge_types=%d, %s, aggregtor=%s' % (
            self.nin, self.nou, self.nedge_types, self.extension.name, self.aggregtor)

    def forward(self, x, nn_idx, etype, addend=None, population_mult=1):
        """``addend``: optional tensor of the output's shape (or a list of them) added after the activation (fused
        into the BatchNorm kernel when training with the plain ReLU).  ``population_mult``: see BatchNormAct2d.forward (set by
        callers that run the operator on ONE row per sample in place of m identical ones)."""
        x, etype = ops.autocast_operands(x, etype)
        if not isinstance(self.aggregtor, str):
            return self._forward_custom_aggregator(x, nn_idx, etype, addend)
        ext, agg = _EXT_CODE[self.extension], _hip.AGG_CODES[self.aggregtor]
        if ext != _hip.EXT_NONE and x.dim() == 4 and x.shape[1] > 1 and x.stride(1) != 1 and x.is_cuda:
            # the synthetic-PGM kernels (csrc/mpconv_*_ext.hip) read node rows of channels: one transposing copy of x
            x = x.contiguous(memory_format=torch.channels_last)
        needs_grad = torch.is_grad_enabled() and (
            x.requires_grad or etype.requires_grad or self.filters.requires_grad)
        bn_batch_stats = self.bn is not None and (self.bn.training or self.bn.running_mean is None)
        plain_relu = isinstance(self.activation_fn, torch.nn.ReLU)
        if not needs_grad and not bn_batch_stats:
            # inference: bias + folded BatchNorm + ReLU ride in the kernel epilogue
            scale = shift = None
            if self.bn is not None:
                scale, shift = fold_batchnorm(self.bn)
            if etype.shape[0] == 1 and x.shape[0] > 1:      # shared edge weights handed over un-expanded
                etype = etype.expand(x.shape[0], -1, -1, -1)
            from .pointwise import as_addends
            adds = as_addends(addend() if callable(addend) else addend)
            # the caller's running sums ride in the kernel's epilogue (third-generation parity kernels) or join in ONE n-input pass
            # behind it (csrc/sum_n.hip) rather than one elementwise add each
            in_kernel = plain_relu and scale is not None and 0 < len(adds) <= 3
            y, _ = ops.mpconv_forward_raw(x, nn_idx, etype, self.filters, self.bias, self.nou,
                                          self.nedge_types, ext, agg, post_scale=scale,
                                          post_shift=shift, relu=plain_relu, addends=adds if in_kernel else None)
            if in_kernel:
                return y
            if self.activation_fn is not None and not plain_relu:
                y = self.activation_fn(y)
            return ops.add_n([y] + adds)
        # ONE source node feeding every destination through identical edges (the LDPC hyper-factor -> variables call of the plain
        # layers, factor_mpnn_sp.py:88-91): every destination receives the same message, so the operator, its BatchNorm and its ReLU
        # run on one row per sample and the result travels as a broadcast (see mp_conv_residual.forward)
        from .pointwise import BatchNormAct2d, bn_spec
        M = 0
        if (population_mult == 1 and ext == _hip.EXT_NONE and addend is None and needs_grad and bn_batch_stats and plain_relu
                and isinstance(self.bn, BatchNormAct2d) and x.shape[0] > 1):
            M = ops.single_source_fanout(x, nn_idx, etype)
        if M:
            B, C = x.shape[:2]
            one = x.reshape(B, 1, 1, C).permute(0, 3, 1, 2)
            return ops.broadcast_nodes(self.forward(one, nn_idx[:, :1, :], etype[:, :, :1, :], None, M), M)
        # a training-mode BatchNorm right behind the operator is finalised by the operator's own launch where it has the epilogue
        spec = bn_spec(self.bn) if (self.bn is not None and isinstance(self.bn, BatchNormAct2d) and population_mult == 1) else None
        z = ops.mpconv(x, nn_idx, etype, self.filters, self.bias, self.nou, self.nedge_types, ext, agg, bn=spec)
        if callable(addend):
            addend = addend()
        if self.bn is not None:
            if plain_relu:                      # BatchNorm + ReLU (+ addend) in one fused kernel pair
                return self.bn(z, addend=addend, slope=0.0, population_mult=population_mult)
            z = self.bn(z, population_mult=population_mult)
        if self.activation_fn is not None:
            z = self.activation_fn(z)
        return add_all(z, addend)

    def edge_messages(self, x, nn_idx, etype):
        """Un-aggregated messages E [B, nou, M, k] (mp_nn.py:127-160) through the HIP operator: every edge becomes
        its own single-neighbour destination (an aggregate over one neighbour is the message itself).  The
        self / neighbour split of the two extensions turns into two such calls on slices of ``filters``:
        NEIGHBOR  E = et.(x_i W_top + x_j W_bot);   DIFF  E = et.(x_i (W_top + W_bot) - x_j W_bot)."""
        B, M, k = nn_idx.shape
        net, nou, nin = self.nedge_types, self.nou, self.nin
        flat_idx = nn_idx.reshape(B, M * k, 1)
        flat_et = etype.reshape(B, net, M * k, 1)
        one = lambda idx, W: ops.mpconv(x, idx, flat_et, W, None, nou, net, _hip.EXT_NONE, _hip.AGG_MAX)
        if self.extension == mp_conv_type.NO_EXTENSION:
            e = one(flat_idx, self.filters)
        else:
            own = torch.arange(M, device=x.device, dtype=nn_idx.dtype).repeat_interleave(k).reshape(1, M * k, 1)
            own = own.expand(B, -1, -1)
            top, bot = self.filters[:nin], self.filters[nin:]
            if self.extension == mp_conv_type.ORIG_WITH_NEIGHBOR:
                e = one(own, top) + one(flat_idx, bot)
            else:
                e = one(own, top + bot) - one(flat_idx, bot)
        return e.reshape(B, nou, M, k)

    def _forward_custom_aggregator(self, x, nn_idx, etype, addend=None):
        """``aggregtor`` given as a callable on [B, nou, M, k], or None = no aggregation (mp_nn.py:89-90,162-163): the
        messages come from the HIP operator, the user's reduction and the epilogue are torch ops."""
        z = self.edge_messages(x, nn_idx, etype)
        if self.aggregtor is not None:
            z = self.aggregtor(z)
        if self.bias is not None:
            z = z + self.bias.view(1, self.nou, 1, 1).to(z.dtype)
        if self.bn is not None:
            z = self.bn(z)
        if self.activation_fn is not None:
            z = self.activation_fn(z)
        return add_all(z, addend() if callable(addend) else addend)
