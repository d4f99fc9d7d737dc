"""Blocks around the message operator: ``mp_conv_residual`` and the node-wise 1x1 maps.

Mirrors (names, constructor signatures, state_dict keys) of
  mp_conv_residual                   /root/reference/lib/model/mpnn/mp_nn_residual.py:7-56
  iid_mapping / _bn / _in            /root/reference/lib/model/mpnn/base_model.py:43-90
  max_pool_layer, flatten            /root/reference/lib/model/mpnn/base_model.py:19-40
"""
import torch

from .message_op import base_mp_nn, mp_conv_type, mp_conv_v2
from .pointwise import BatchNormAct2d, NodeInstanceNorm, PointwiseConv2d


def _conv_norm_act(cin, cout, norm, act, bias=True):
    layers = [PointwiseConv2d(cin, cout, 1, bias=bias)]
    if norm is not None:
        layers.append(norm)
    layers.append(act)
    return torch.nn.Sequential(*layers)


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

    def forward(self, node_feature, nn_idx, etype, addend=None):
        """``addend`` (optional, the caller's running sum of the same shape as the output) is added by conv2's
        fused BatchNorm+activation kernel instead of a separate elementwise pass."""
        fuse = self.training            # BatchNorm statistics ride in the 1x1 map's epilogue when training
        h = self.conv1[1](self.conv1[0](node_feature, want_stats=fuse))
        h = self.mp_conv(h, nn_idx, etype)
        h = self.conv2[1](self.conv2[0](h, want_stats=fuse), addend=addend)
        return h + node_feature if self.with_residual else h
