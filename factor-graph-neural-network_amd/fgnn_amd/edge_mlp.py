"""The edge-type MLP in front of the message operator.

``EdgeMLP`` is the ``Sequential(Conv2d(cin,64,1), ReLU, Conv2d(64,net,1))`` the reference builds as
``emodel_f2v`` / ``emodel_v2f`` (/root/reference/train_ldpc.py:32-38) — same children, same state_dict keys
(``0.weight 0.bias 2.weight 2.bias``) — whose forward, for bf16 inputs on the GPU, is ONE kernel that never
writes the 64-channel hidden tensor, and whose backward recomputes it (csrc/edge_mlp.hip).  The result comes
back as a [B,net,M,k] view of edge-type-fastest storage, the layout the operator kernels read in place.
Anything outside that family (f32 inputs, other widths, CPU tensors) runs the three children as written.
"""
import torch

from . import _hip
from .mpnn.pointwise import PointwiseConv2d


class _EdgeMLPFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, w1, b1, w2, b2):
        from . import ops
        B, cin, M, k = x.shape
        net = w2.shape[0]
        E = M * k
        if not (x.stride(2) == k * x.stride(3) or M == 1 or k == 1):     # rows (m,j) must be evenly strided
            x = x.contiguous()
        x_sr = x.stride(3) if k > 1 else x.stride(2)
        y = torch.empty((B, M, k, net), device=x.device, dtype=x.dtype)
        L = _hip.lib()
        ops.timed('edge_mlp_fwd_kernel (MFMA where the layout allows)', 2 * B * E * (cin + net),
                  lambda: _hip.check(L.fgnn_edge_mlp_forward(
                      _hip._ptr(x), x.stride(0), x.stride(1), x_sr, _hip._ptr(w1), _hip._ptr(b1), _hip._ptr(w2),
                      _hip._ptr(b2), _hip._ptr(y), B, E, cin, net, _hip.stream_ptr())),
                  nflops=2 * B * E * 64 * (cin + net))
        ctx.save_for_backward(x, w1, b1, w2)
        ctx.params = (w1, b1, w2, b2)
        ctx.x_sr = x_sr
        return y.permute(0, 3, 1, 2)

    @staticmethod
    def backward(ctx, gy):
        from . import ops
        ops.backward_node_begins()
        ops.backward_tail_begins()      # (behind this node the pass is one chain on this stream: what waits for its end goes to the side stream)
        x, w1, b1, w2 = ctx.saved_tensors
        B, cin, M, k = x.shape
        net = w2.shape[0]
        E = M * k
        if gy.dtype != x.dtype:
            gy = gy.to(x.dtype)
        # gy is [B,net,M,k]; rows (m,j) must be evenly strided — true of both layouts the operator emits
        if not ((gy.stride(3) * k == gy.stride(2)) or M == 1 or k == 1):
            gy = gy.contiguous()
        gy_sr = gy.stride(3) if k > 1 else gy.stride(2)
        sinks = [ops.grad_sink(p) for p in ctx.params]
        shapes = [(64, cin, 1, 1), (64,), (net, 64, 1, 1), (net,)]
        outs = [s if s is not None else torch.zeros(shp, device=x.device, dtype=torch.float32)
                for s, shp in zip(sinks, shapes)]
        L = _hip.lib()
        ws = ops._workspace(x.device, int(L.fgnn_edge_mlp_workspace_bytes(B, E)))
        ops.timed('edge_mlp_bwd_kernel (MFMA where the layout allows)', 2 * B * E * (cin + net),
                  lambda: _hip.check(L.fgnn_edge_mlp_backward(
                      _hip._ptr(x), x.stride(0), x.stride(1), ctx.x_sr, _hip._ptr(gy), gy.stride(0), gy.stride(1),
                      gy_sr, _hip._ptr(w1), _hip._ptr(b1), _hip._ptr(w2), B, E, cin, net, _hip._ptr(outs[0]),
                      _hip._ptr(outs[1]), _hip._ptr(outs[2]), _hip._ptr(outs[3]), _hip._ptr(ws), ws.numel() * 4,
                      _hip.stream_ptr())),
                  nflops=2 * B * E * 64 * (2 * cin + 2 * net))
        grads = [None if s is not None else o.to(p.dtype) for s, o, p in zip(sinks, outs, ctx.params)]
        return (None, *grads)


class EdgeMLP(torch.nn.Sequential):
    def __init__(self, cin, hidden, cout):
        super().__init__(PointwiseConv2d(cin, hidden, 1), torch.nn.ReLU(), PointwiseConv2d(hidden, cout, 1))

    def _fusable(self, x):
        c1, c2 = self[0], self[2]
        return (x.is_cuda and x.dim() == 4 and x.dtype == torch.bfloat16 and c1.in_channels <= 8
                and c1.out_channels == 64 and c2.out_channels <= 4 and c1.bias is not None and c2.bias is not None
                and c1.weight.dtype == torch.float32 and not x.requires_grad)

    def forward(self, x):
        if not self._fusable(x):
            return super().forward(x)
        c1, c2 = self[0], self[2]
        return _EdgeMLPFn.apply(x, c1.weight, c1.bias, c2.weight, c2.bias)
