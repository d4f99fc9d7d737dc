"""Blocks around the message operator: ``mp_conv_residual`` and the node-wise 1x1 maps.

Mirrors (names, constructor signatures, state_dict keys) of
  mp_conv_residual                   /root/reference/lib/model/mpnn/mp_nn_residual.py:7-56
  iid_mapping / _bn / _in            /root/reference/lib/model/mpnn/base_model.py:43-90
  max_pool_layer, flatten            /root/reference/lib/model/mpnn/base_model.py:19-40
"""
import torch

from .message_op import base_mp_nn, mp_conv_type, mp_conv_v2
from .. import ops
from .pointwise import BatchNormAct2d, NodeInstanceNorm, PointwiseConv2d, as_addends, refresh_in_place
from .pointwise import state_epoch as pointwise_state_epoch


def _conv_norm_act(cin, cout, norm, act, bias=True):
    layers = [PointwiseConv2d(cin, cout, 1, bias=bias)]
    if norm is not None:
        layers.append(norm)
    layers.append(act)
    return torch.nn.Sequential(*layers)


FUSE_EVAL_BLOCKS = True      # inference: run a 64-wide bf16 mp_conv_residual as one kernel when possible


class iid_mapping(torch.nn.Module):
    def __init__(self, nin, nout, bias=True):
        super().__init__()
        self.main = _conv_norm_act(nin, nout, None, torch.nn.LeakyReLU(), bias)

    def forward(self, x):
        return self.main(x)


class iid_mapping_bn(torch.nn.Module):
    def __init__(self, nin, nout, bias=True, bn=True):
        super().__init__()
        # BatchNorm and its ReLU run fused; Identity keeps the Sequential's indices (main.0 conv, main.1 bn)
        self.main = _conv_norm_act(nin, nout, BatchNormAct2d(nout, slope=0.0), torch.nn.Identity(), bias)

    def forward(self, x):
        return self.main[1](self.main[0](x, want_stats=self.training))


class iid_mapping_in(torch.nn.Module):
    def __init__(self, nin, nout, bias=True):
        super().__init__()
        # InstanceNorm and the ReLU behind it run as one kernel; Identity keeps the Sequential's indices
        self.main = _conv_norm_act(nin, nout, NodeInstanceNorm(relu=True), torch.nn.Identity(), bias)

    def forward(self, x):
        return self.main(x)


class max_pool_layer(torch.nn.Module):
    def __init__(self, dim=2):
        super().__init__()
        self.dim = dim

    def forward(self, input):
        return input.max(dim=self.dim, keepdim=True)[0]


class flatten(torch.nn.Module):
    def forward(self, input):
        return input.view(input.size(0), -1)


def _is_identity_list(nn_idx):
    """nn_idx [B, 1, k] lists nodes 0..k-1 in order for every sample (the hyper-factor's neighbour table).  Checked on
    the device once per table — the fused fan-in block does not read the table at all.  The verdict is remembered ON the
    tensor that owns the memory (the view's base, e.g. LDPCModel's frozen `hnn_idx_v2f` behind its per-call `expand`),
    keyed by version and view geometry, so it can never outlive or be confused with another table."""
    owner = nn_idx._base if nn_idx._base is not None else nn_idx
    key = (nn_idx._version, nn_idx.storage_offset(), tuple(nn_idx.shape), tuple(nn_idx.stride()))
    memo = getattr(owner, '_fgnn_identity_list', None)
    if memo is None or memo[0] != key:
        k = nn_idx.shape[-1]
        hit = bool((nn_idx == torch.arange(k, device=nn_idx.device, dtype=nn_idx.dtype)).all().item())
        memo = (key, hit)
        owner._fgnn_identity_list = memo
    return memo[1]


class mp_conv_residual(base_mp_nn):
    """Bottleneck: 1x1 conv+BN+LeakyReLU on the sources -> message operator ->
    1x1 conv+BN+LeakyReLU on the destinations (+ input when ``with_residual``)."""

    def __init__(self, nin, nmed, netype, extension=mp_conv_type.ORIG_WITH_DIFF,
                 with_residual=True, with_hop=False, aggregator='max', nout=None):
        super().__init__()
        nout = nin if nout is None else nout
        # conv + (BatchNorm + LeakyReLU fused) ; Identity keeps the reference's Sequential indices
        self.conv1 = _conv_norm_act(nin, nmed, BatchNormAct2d(nmed, slope=0.01), torch.nn.Identity())
        self.mp_conv = mp_conv_v2(nmed, nmed, netype, extension=extension, aggregtor=aggregator)
        self.conv2 = _conv_norm_act(nmed, nout, BatchNormAct2d(nout, slope=0.01), torch.nn.Identity())
        self.with_residual = with_residual
        self.with_hop = with_hop

    def folded_for_inference(self, device):
        """(W1 [64][nin], s1, t1, filters, s2, t2, W2 [nout][64], s3, t3) as f32: conv weights, operator filters and the three
        eval-mode BatchNorms (with the conv / operator biases) folded into per-channel affines — what the one-kernel block
        and the one-kernel layer (csrc/factor_layer_fwd.hip) take.  Cached; refreshed IN PLACE when a parameter, a buffer or
        the package's state epoch moved (a captured inference graph holds these addresses)."""
        mp = self.mp_conv
        bn1, bn2, bn3 = self.conv1[1], mp.bn, self.conv2[1]
        nin, nout = self.conv1[0].in_channels, self.conv2[0].out_channels
        key = tuple(t._version for t in (self.conv1[0].weight, self.conv1[0].bias, bn1.weight, bn1.bias, bn1.running_mean,
                                         bn1.running_var, mp.filters, mp.bias, bn2.weight, bn2.bias, bn2.running_mean,
                                         bn2.running_var, self.conv2[0].weight, self.conv2[0].bias, bn3.weight, bn3.bias,
                                         bn3.running_mean, bn3.running_var)) + (device, pointwise_state_epoch())
        if getattr(self, '_fuse_key', None) != key:
            def fold(bn, bias):
                s = bn.weight.float() * torch.rsqrt(bn.running_var.float() + bn.eps)
                t = bn.bias.float() - bn.running_mean.float() * s
                return s.contiguous(), (t + (bias.float() * s if bias is not None else 0)).contiguous()
            s1, t1 = fold(bn1, self.conv1[0].bias)
            s2, t2 = fold(bn2, mp.bias)
            s3, t3 = fold(bn3, self.conv2[0].bias)
            self._fuse = refresh_in_place(getattr(self, '_fuse', None), (
                self.conv1[0].weight.detach().float().reshape(64, nin), s1, t1,
                mp.filters.detach().float(), s2, t2,
                self.conv2[0].weight.detach().float().reshape(nout, 64), s3, t3))
            self._fuse_key = key
        return self._fuse

    def fusable_for_inference(self):
        """The block is of the family the one-kernel inference paths fold: nmed 64, max aggregation, no extension, operator
        bias + BatchNorm + ReLU, BatchNorm + LeakyReLU (one slope) behind both 1x1 maps, no residual of its own."""
        mp = self.mp_conv
        bn1, bn2, bn3 = self.conv1[1], mp.bn, self.conv2[1]
        return (not self.with_residual and mp.nin == 64 and mp.nou == 64 and mp.nedge_types in (1, 4) and mp.aggregtor == 'max'
                and mp.extension == mp_conv_type.NO_EXTENSION and bn2 is not None and mp.bias is not None
                and isinstance(mp.activation_fn, torch.nn.ReLU) and isinstance(bn1, BatchNormAct2d)
                and isinstance(bn3, BatchNormAct2d) and bn1.slope == bn3.slope)

    def _fused_eval(self, x, nn_idx, etype, addend):
        """Inference: the whole block as ONE kernel (csrc/mpconv_block_fwd.hip) when it is the 64-wide bf16
        parity-check shape; None otherwise."""
        import ctypes
        from .. import _hip, ops
        mp = self.mp_conv
        bn1, bn2, bn3 = self.conv1[1], mp.bn, self.conv2[1]
        if (not FUSE_EVAL_BLOCKS or self.with_residual or self.training or torch.is_grad_enabled() or not x.is_cuda
                or x.dtype != torch.bfloat16 or x.shape[1] not in (64, 128, 256) or mp.nin != 64 or mp.nou != 64
                or mp.nedge_types not in (1, 4) or mp.aggregtor != 'max' or mp.extension != mp_conv_type.NO_EXTENSION
                or bn2 is None or mp.bias is None or not isinstance(mp.activation_fn, torch.nn.ReLU)
                or self.conv2[0].out_channels not in (64, 128, 256)
                or not isinstance(bn1, BatchNormAct2d) or not isinstance(bn3, BatchNormAct2d)
                or bn1.slope != bn3.slope or etype.dtype != torch.bfloat16):
            return None
        B, nin, N, _ = x.shape
        nout = self.conv2[0].out_channels
        M, k = nn_idx.shape[1:]
        fanout = mp.nedge_types == 1 and N == 1 and k == 1          # the hyper-factor -> variables call
        fanin = mp.nedge_types == 1 and M == 1 and k == N and N > 1 and _is_identity_list(nn_idx)   # variables -> it
        if not (fanout or fanin) and not (mp.nedge_types == 4 and k in (3, 6)):
            return None
        xr = x.permute(0, 2, 3, 1)
        et = etype.permute(0, 2, 3, 1)                                   # [B, M, k, net]
        if not xr.is_contiguous() or not (et.is_contiguous() or (etype.stride(0) == 0 and et[0].is_contiguous())):
            return None
        if addend is not None:
            ar = addend.permute(0, 2, 3, 1)
            if addend.dtype != x.dtype or not ar.is_contiguous():
                return None
        W1, s1, t1, F, s2, t2, W2, s3, t3 = self.folded_for_inference(x.device)
        y = torch.empty((B, M, 1, nout), device=x.device, dtype=x.dtype).permute(0, 3, 1, 2)
        d = _hip.make_desc(x, nn_idx, etype, 64, mp.nedge_types, _hip.EXT_NONE, _hip.AGG_MAX, True, y)
        d.nin = 64                      # the inner operator's width; x / y strides stay the block's
        P = _hip._ptr
        if fanin:
            rc = _hip.lib().fgnn_mpconv_block_forward_fanin(ctypes.byref(d), P(x), P(etype), P(W1), P(s1), P(t1), P(F),
                                                            P(s2), P(t2), P(W2), P(s3), P(t3), float(bn1.slope), nin, nout,
                                                            P(addend), P(y), _hip.stream_ptr())
            if rc == _hip.EUNSUPPORTED:
                return None
            _hip.check(rc)
            return y
        if fanout:
            rc = _hip.lib().fgnn_mpconv_block_forward_fanout(ctypes.byref(d), P(x), P(etype), P(W1), P(s1), P(t1), P(F),
                                                             P(s2), P(t2), P(W2), P(s3), P(t3), float(bn1.slope), nin, nout,
                                                             P(addend), P(y), _hip.stream_ptr())
            if rc == _hip.EUNSUPPORTED:
                return None
            _hip.check(rc)
            return y
        rc = _hip.lib().fgnn_mpconv_block_forward(ctypes.byref(d), P(x), P(nn_idx), P(etype), P(W1), P(s1), P(t1), P(F),
                                                  P(s2), P(t2), P(W2), P(s3), P(t3), float(bn1.slope), nin, nout, P(addend),
                                                  P(y),
                                                  _hip.stream_ptr())
        if rc == _hip.EUNSUPPORTED:
            return None
        _hip.check(rc)
        return y

    def forward(self, node_feature, nn_idx, etype, addend=None):
        """``addend`` (optional, the caller's running sum of the same shape as the output — a tensor, a list of tensors,
        or a callable returning either, evaluated right before it is consumed) is added by conv2's fused
        BatchNorm+activation kernel instead of a separate elementwise pass."""
        nn_idx = ops.shared_graph_view(nn_idx)
        staged = self.training and torch.is_grad_enabled()
        if callable(addend) and not staged:
            addend = addend()                                            # the one-kernel block needs it up front
        if isinstance(addend, (list, tuple)):
            addend = as_addends(addend)
            if not staged:                                               # ... and takes one addend
                addend = ops.add_n(addend) if addend else None
        if not staged:
            y = self._fused_eval(node_feature, nn_idx, etype, addend)
            if y is not None:
                return y
        fuse = self.training            # BatchNorm statistics ride in the 1x1 map's epilogue when training
        h = self.conv1[1](self.conv1[0](node_feature, want_stats=fuse))
        h = self.mp_conv(h, nn_idx, etype)
        h = self.conv2[0](h, want_stats=fuse)
        if callable(addend):            # produced on another stream: asked for (and waited on) only where it is consumed
            addend = addend()
        h = self.conv2[1](h, addend=addend)
        return h + node_feature if self.with_residual else h
