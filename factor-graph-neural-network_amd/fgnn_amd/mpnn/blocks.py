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
        return self.main[1](self.main[0](x, bn=self.main[1]))


class iid_mapping_in(torch.nn.Module):
    def __init__(self, nin, nout, bias=True):
        super().__init__()
        # InstanceNorm and the ReLU behind it run as one kernel; Identity keeps the Sequential's indices
        self.main = _conv_norm_act(nin, nout, NodeInstanceNorm(relu=True), torch.nn.Identity(), bias)

    def forward(self, x):
        if x.dim() == 4 and x.shape[2] * x.shape[3] == 1:
            # a single node (the LDPC hyper-factor's f2f map, factor_mpnn_sp.py:77,140): InstanceNorm of one value is exactly 0
            # whatever the map produced, ReLU(0) = 0, and no gradient reaches the map — nothing to launch
            return x.new_zeros((x.shape[0], self.main[0].out_channels, 1, 1))
        y = self._fused(x)
        return self.main(x) if y is None else y

    def _fused(self, x):
        """Conv -> InstanceNorm -> ReLU as ONE kernel (csrc/linear_fwd_b16.hip: linear_instnorm_fwd_kernel) for bf16 channel-fastest
        states of 96 / 48 nodes: the norm is per sample, so the wave that multiplies a sample's rows holds its whole population.
        The pre-norm tensor is written only when a backward will read it, and never read back in the forward.  The autograd graph
        is the staged one (the map's and the norm's own Functions, handed their outputs): the backward is unchanged.  None: the
        staged path runs (other widths / dtypes / layouts, autocast off, CPU)."""
        from .. import _hip, ops
        from .pointwise import _InstNormAct, _RowLinear
        if not FUSE_IID_IN or not x.is_cuda or x.dim() != 4 or x.shape[3] != 1 or x.shape[2] not in (48, 96):
            return None
        conv, norm = self.main[0], self.main[1]
        B, C, N, _ = x.shape
        cout = conv.out_channels
        dt = torch.get_autocast_dtype('cuda') if torch.is_autocast_enabled() else x.dtype
        if dt != torch.bfloat16 or C % 64 or cout % 64 or C > 256 or cout > 256 or conv.weight.dtype != torch.float32:
            return None
        grad = torch.is_grad_enabled() and (conv.weight.requires_grad or x.requires_grad)
        # with the pre-norm tensor to store (training), 256-channel inputs stay staged (tools/ibench.py, B = 4096, N = 96, fused with z /
        # staged: 213 / 199 us at 256 -> 256, 124 / 112 at 256 -> 128; 156 / 164 at 128 -> 256, 47 / 57 at 64 -> 64); without it
        # (inference) the one-kernel form is ahead at every width (187 / 199, 101 / 112, 116 / 164, 30 / 57)
        if C > (_IID_FUSE_MAX_CIN if grad else 256):
            return None
        rows = x.permute(0, 2, 3, 1)
        if not rows.is_contiguous():
            rows = rows.contiguous()
        rows = rows.view(B * N, C)
        if rows.dtype != torch.bfloat16:
            rows = rows.to(torch.bfloat16)
        weight = conv.weight.view(cout, C)
        w = weight.detach()
        b = None if conv.bias is None else conv.bias.detach().float().contiguous()
        z = torch.empty((B * N, cout), device=x.device, dtype=torch.bfloat16) if grad else None
        y = torch.empty((B, N, 1, cout), device=x.device, dtype=torch.bfloat16)
        L = _hip.lib()
        rc = []
        ops.timed('linear_instnorm_fwd_kernel', 2 * B * N * (C + cout * (2 if grad else 1)),
                  lambda: rc.append(L.fgnn_linear_instnorm_forward(_hip._ptr(rows), _hip._ptr(w), _hip._ptr(b), _hip._ptr(z), _hip._ptr(y),
                                                                   B, N, C, cout, int(norm.relu), float(norm.eps), _hip.stream_ptr())),
                  nflops=2 * B * N * C * cout)
        if rc[0] == _hip.EUNSUPPORTED:
            return None
        _hip.check(rc[0])
        if not grad:
            return y.permute(0, 3, 1, 2)
        zz = _RowLinear.apply(rows, weight, conv.bias, False, z, ops.fan_box(x))
        return _InstNormAct.apply(zz.view(B, N, 1, cout).permute(0, 3, 1, 2), norm.relu, y)


_IID_FUSE_MAX_CIN = 128
FUSE_IID_IN = True      # iid_mapping_in: map + InstanceNorm + ReLU as one kernel where the shape allows


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
    return ops.is_identity_list(nn_idx)


TAIL_WGRAD_MOMENTS = True    # training: conv2's weight gradient from the moments of the tail's reduce pass (no gz3 / a2 in memory)
FUSE_TRAIN_TAIL = True       # training: BatchNorm2 + ReLU -> conv2 -> BatchNorm3 + LeakyReLU (+ addends) without storing conv2's output


FUSE_TRAIN_HEAD = True    # training: BatchNorm1's input gradient and conv1's input gradient in one pass


# Input widths the fused head backward takes.  Measured alone (R = 393 216 rows): 73 vs 76 us staged at 64 channels in, 90 vs 88 at
# 128, 127 vs 119 at 256 (R = 196 608: 40 / 46 / 65 vs 45 / 50 / 67) — the tensor it avoids re-reading (gz1, 25-50 MB) is read back
# from the 256 MB infinity cache in the staged path, so the fusion only saves the launch and a cached pass; in the step 64 and 128
# are worth ~0.05 ms together, 256 nothing.
_HEAD_WIDTHS = (64, 128)


class _BlockHead(torch.autograd.Function):
    """conv1 -> BatchNorm -> LeakyReLU in front of the operator of a training-mode ``mp_conv_residual`` (mp_nn_residual.py:25-29,
    42-44), 64 output channels.  The forward is the staged one (the 1x1 map with the statistics epilogue, the finaliser, the apply
    pass); the BACKWARD runs BatchNorm's reduction pass and then ONE kernel (csrc/block_tail.hip::block_head_bwd_kernel) that forms
    BatchNorm's input gradient gz1 and multiplies it by conv1's weight on the way out — the staged path wrote gz1 and read it back
    in the input-gradient GEMM.  gz1 is still stored once for conv1's weight-gradient kernel, which is parked like every other."""

    @staticmethod
    def forward(ctx, rows, weight, bias, bn_w, bn_b, rm, rv, nbt, momentum, eps, slope, box=None):
        ctx.box = box                   # the ops.FanBox of the state `rows` views: the backward deposits (gz1, W1) there instead of forming gx
        from .. import _hip
        from . import pointwise
        L = _hip.lib()
        P = _hip._ptr
        spec = (bn_w, bn_b, rm, rv, nbt, momentum, eps)
        z1 = pointwise.hip_linear(rows, weight, bias, bn=spec)          # the map's last workgroup finalises BatchNorm1's statistics
        if z1 is None:
            raise _hip.FgnnHipError('fused block head: the 1x1 map is outside csrc/linear_fwd_b16.hip (checked by the caller)')
        R = rows.shape[0]
        stats = pointwise.batch_stats(z1, spec)                       # [4, 64] mean, invstd, scale, shift
        a1 = torch.empty_like(z1)
        ops.timed('bn_apply (forward)', 2 * z1.numel() * 2, lambda: _hip.check(L.fgnn_bn_apply(
            P(z1), P(a1), R, 64, _hip.BF16, P(stats[2]), P(stats[3]), slope, None, None, None, None, _hip.stream_ptr())))
        ctx.save_for_backward(rows, z1, stats, weight, bn_w, bn_b)
        ctx.slope = slope
        ctx.params = (weight, bias, bn_w, bn_b)
        return a1

    @staticmethod
    def backward(ctx, ga1):
        from .. import _hip
        ops.backward_node_begins()
        L = _hip.lib()
        P = _hip._ptr
        rows, z1, stats, weight, bn_w, bn_b = ctx.saved_tensors
        pW, pbias, pw, pb = ctx.params
        R, cin = rows.shape
        dev = rows.device
        ga1 = ga1.contiguous()
        if ga1.dtype != z1.dtype:
            ga1 = ga1.to(z1.dtype)

        def sink(param, shape):
            g = ops.grad_sink(param)
            return (g, True) if g is not None else (torch.zeros(shape, device=dev, dtype=torch.float32), False)
        gw1, s_w1 = sink(pw, (64,))
        gb1, s_b1 = sink(pb, (64,))
        Wbase = pW._base if pW._base is not None and pW._base.numel() == pW.numel() else pW
        gW, s_W = sink(Wbase, (64, cin))
        gbias, s_bias = sink(pbias, (64,)) if pbias is not None else (None, True)
        gz1 = torch.empty_like(z1)
        lazy = ctx.needs_input_grad[0] and ctx.box is not None and weight.is_contiguous() and weight.dtype == torch.float32
        gx = None if lazy else torch.empty((R, cin), device=dev, dtype=z1.dtype)
        ws = ops._workspace(dev, int(L.fgnn_bn_workspace_bytes(R, 64)))
        ops.timed('block_head_backward (reduce + finalise + grad)', 2 * R * (5 * 64 + cin), lambda: _hip.check(L.fgnn_block_head_backward(
            P(z1), P(ga1), P(stats[0]), P(stats[1]), P(bn_w.detach()), P(bn_b.detach()), ctx.slope, P(weight.detach()), P(gz1), P(gx),
            P(gw1), P(gb1), R, cin, P(ws), ws.numel() * 4, P(ops._fold_scratch(dev)), _hip.stream_ptr())), nflops=2 * R * 64 * cin)

        taken = 0
        if lazy:
            # (gz1 @ W1 joins the state's other gradients in the fan-out's backward; with conv1's parameter gradients going to sinks its
            # weight gradient joins the state's other consumers' there too — ops.FanBox: one pass over the state's rows for all)
            taken = ctx.box.deposit(gz1, weight.detach(), wgrad=(rows, gW, gbias) if (s_W and s_bias) else None)
        if taken != 2:
            record = s_W and s_bias and ops.folds_deferrable()

            def launch(rows=rows, gz1=gz1, gW=gW, gbias=gbias):
                with ops.fold_scope(record) as scope:
                    wsw = scope.slabs(dev, int(L.fgnn_linear_wgrad_workspace_bytes(R, cin, 64)))
                    ops.timed('linear_wgrad_b16_kernel', 2 * R * (cin + 64), lambda: _hip.check(L.fgnn_linear_wgrad(
                        P(rows), P(gz1), R, cin, 64, _hip.BF16, P(gW.view(64, cin)), P(gbias), P(wsw), wsw.numel() * 4, _hip.stream_ptr())),
                        nflops=2 * R * cin * 64)
            if s_W and s_bias:
                ops.defer_wgrad(launch, (rows, gz1))
            else:
                launch()
        if lazy:
            if taken:
                gx = ctx.box.placeholder(rows.shape)
            else:                                        # no slot: the product after all (gz1 is in the infinity cache)
                from . import pointwise
                gx = pointwise.hip_linear(gz1, weight.detach(), None, transposed=True)
                if gx is None:
                    gx = gz1 @ pointwise.cast_cached(pW._base if pW._base is not None else pW, gz1.dtype).view(weight.shape)
        return (gx if ctx.needs_input_grad[0] else None, None if s_W else gW.view(pW.shape).to(pW.dtype),
                None if (s_bias or gbias is None) else gbias, None if s_w1 else gw1, None if s_b1 else gb1,
                None, None, None, None, None, None, None)


LATE_JOIN = True     # the tail asks for addends of another stream behind its statistics pass
ROUTE_ADDEND_GRADS = True    # the addends' gradient leaves through its own autograd node, ahead of the tail's backward kernels


class _AddendRoute(torch.autograd.Function):
    """`out` already holds `+ addends` (the tail's apply kernel added them); this node only gives the sum its autograd
    edges.  The addends' gradient is the output's gradient unchanged, so it should not wait for the tail's backward KERNELS:
    as outputs of `_BlockTail.backward` the producers of the addends — FactorNN's side-stream branch: the variables' node-wise
    map and the hyper-factor's messages — could start their backward only after the main stream's tail backward (~0.2 ms
    per layer, which the main stream then spent waiting for that branch at the layer's gradient sum: 1.6 ms of a step,
    profiles/r03/README.md).  No kernel runs here."""

    @staticmethod
    def forward(ctx, out, periods, streams, *adds):
        """``periods[i]`` > 1: addend i is a per-sample row [R / period, C] the apply kernel broadcast over the sample's nodes; its
        gradient is the node sum of the output's (one pass, csrc/sum_n.hip).  ``streams[i]``: the stream addend i was produced on
        (or None): with FGNN_NODE_SUM_HOME=1 that pass is issued THERE — its only reader is the addend's own backward node, which
        runs on that stream — instead of in front of this stream's tail backward."""
        ctx.periods = tuple(periods)[:len(adds)]
        ctx.streams = tuple(streams)[:len(adds)]
        return out.view_as(out)

    @staticmethod
    def backward(ctx, g):
        from .pointwise import node_sum
        outs = []
        for i, q in enumerate(ctx.periods):
            if not ctx.needs_input_grad[3 + i]:
                outs.append(None)
            elif q == 1:
                outs.append(g)
            else:
                ops.backward_node_begins()
                home = ctx.streams[i] if NODE_SUM_HOME and g.is_cuda else None
                cur = torch.cuda.current_stream(g.device) if g.is_cuda else None
                if home is not None and home != cur:
                    ready = torch.cuda.Event()
                    ready.record(cur)
                    with torch.cuda.stream(home):
                        home.wait_event(ready)
                        outs.append(node_sum(g.contiguous(), q))
                    g.record_stream(home)
                else:
                    outs.append(node_sum(g.contiguous(), q))
        return (g, None, None) + tuple(outs)


NODE_SUM_HOME = True      # see _AddendRoute (round 5: -0.13 ms together with the factor states' merge on the side stream)


class _BlockTail(torch.autograd.Function):
    """Everything behind the message operator in a training-mode ``mp_conv_residual`` (mp_nn.py:165-175 BatchNorm + ReLU,
    mp_nn_residual.py:31-35,49-51 conv2 + BatchNorm + LeakyReLU, + the caller's addends) through csrc/block_tail.hip: the
    Cout-wide pre-BatchNorm tensor of conv2 is recomputed from the operator's 64-channel output wherever it is needed
    (statistics, normalisation, BatchNorm3's backward sums and input gradient) instead of being stored and re-read.  Every
    reducing launch finalises its own sums (csrc/fgnn_gridfold.h): forward = statistics + apply (+ BatchNorm2's statistics pass where the
    operator's epilogue did not bring them), backward = reduce + grad + the 64-channel BatchNorm's apply."""

    @staticmethod
    def forward(ctx, e, w2, b2, slope2, slope3, momentum2, eps2, momentum3, eps3, W2, bias2, w3, b3, rm2, rv2, nbt2, rm3, rv3,
                nbt3, population, add0=None, add1=None, add2=None, periods=(1, 1, 1)):
        """``population``: rows the batch statistics stand for in the running variances (0 = R; R * m for a per-sample vector the
        reference broadcasts over m nodes).  ``periods``: see ``_AddendRoute``."""
        import ctypes
        from .. import _hip
        from . import pointwise
        L = _hip.lib()
        P = _hip._ptr
        R, Cout = e.shape[0], W2.shape[0]
        dev = e.device
        # BatchNorm2: the operator's epilogue finalised it (pending), else one reducing launch
        st2 = pointwise.batch_stats(e, (w2, b2, rm2, rv2, nbt2, momentum2, eps2), population)         # mean, invstd, scale, shift
        ws = ops._workspace(dev, int(L.fgnn_bn_workspace_bytes(R, Cout)))
        fold = ops._fold_scratch(dev)
        # a2 is stored only for a weight-gradient KERNEL; with the moments form (TAIL_WGRAD_MOMENTS) the backward recomputes it
        a2 = None if TAIL_WGRAD_MOMENTS else torch.empty_like(e)
        W2c = W2.detach()
        bias2c = None if bias2 is None else bias2.detach()
        flops = 2 * R * 64 * Cout
        st3, fin3 = pointwise.make_final((w3, b3, rm3, rv3, nbt3, momentum3, eps3), Cout, dev, R, population)
        ops.timed('block_tail_stats_kernel', 2 * R * 64, lambda: _hip.check(L.fgnn_block_tail_stats(
            P(e), P(st2[2]), P(st2[3]), slope2, P(W2c), P(bias2c), R, Cout, P(ws), fin3, P(fold), _hip.stream_ptr())), nflops=flops)
        out = torch.empty((R, Cout), device=dev, dtype=e.dtype)
        if callable(add0):          # the addends come from another stream: asked for (and waited on) only HERE, behind the statistics
            (add0, add1, add2), periods = add0()   # pass, which does not read them (the join used to sit in front of it)
        adds = (add0, add1, add2)
        nadd = sum(a is not None for a in adds)
        ops.timed('block_tail_apply_kernel', 2 * R * (2 * 64 + (1 + nadd) * Cout), lambda: _hip.check(L.fgnn_block_tail_apply(
            P(e), P(st2[2]), P(st2[3]), slope2, P(W2c), P(bias2c), P(st3[2]), P(st3[3]), slope3, P(add0), P(add1), P(add2),
            pointwise.period_array(periods), P(out), P(a2), R, Cout, _hip.stream_ptr())), nflops=flops)
        pointwise.note_state_change()                   # running statistics / num_batches_tracked were just updated in place
        ctx.save_for_backward(e, a2, st2, st3, w2, b2, W2, w3)
        ctx.slopes = (slope2, slope3)
        ctx.has_add = tuple(a is not None and a.requires_grad for a in adds)
        ctx.periods = tuple(periods)
        ctx.has_bias2 = bias2 is not None
        ctx.params = (w2, b2, W2, bias2, w3, b3)
        return out

    @staticmethod
    def backward(ctx, gout):
        from .. import _hip
        from . import pointwise
        ops.backward_node_begins()
        L = _hip.lib()
        P = _hip._ptr
        e, a2, st2, st3, w2, b2, W2, w3 = ctx.saved_tensors
        slope2, slope3 = ctx.slopes
        pw2, pb2, pW2, pbias2, pw3, pb3 = ctx.params
        R, Cout = e.shape[0], W2.shape[0]
        dev = e.device
        gout = gout.contiguous()
        if gout.dtype != e.dtype:
            gout = gout.to(e.dtype)

        def sink(param, shape):
            g = ops.grad_sink(param)
            return (g, True) if g is not None else (torch.zeros(shape, device=dev, dtype=torch.float32), False)
        gw3, s_w3 = sink(pw3, (Cout,))
        gb3, s_b3 = sink(pb3, (Cout,))
        gw2, s_w2 = sink(pw2, (64,))
        gb2, s_b2 = sink(pb2, (64,))
        Wbase = pW2._base if pW2._base is not None and pW2._base.numel() == pW2.numel() else pW2
        gW2, s_W2 = sink(Wbase, (Cout, 64))
        gbias2, s_bias2 = sink(pbias2, (Cout,)) if ctx.has_bias2 else (None, True)
        moments = a2 is None
        gz3 = None if moments else torch.empty((R, Cout), device=dev, dtype=e.dtype)
        ga2 = torch.empty_like(e)
        nws = (2048 * Cout + 2 * Cout + 1024 * 128) * 4
        ws = ops._workspace(dev, max(nws, int(L.fgnn_linear_wgrad_workspace_bytes(R, 64, Cout))))
        dsum2 = torch.empty((2, 64), device=dev, dtype=torch.float32)       # BatchNorm2's backward sums, finalised by the grad kernel
        bias2c = None if pbias2 is None else pbias2.detach()
        # BatchNorm3 backward (sums, parameter gradients, input gradient gz3) + ga2 = gz3 W2 + BatchNorm2's backward sums and
        # parameter gradients: two launches
        if moments:
            # conv2's weight gradient is a closed form of three moments the reduce pass accumulates (csrc/block_tail.hip): neither gz3
            # nor a2 exists in memory; the buffer (slabs, A, Bc) is this call's own until the finish launch below has run
            mom = torch.empty(int(L.fgnn_block_tail_moments_bytes(R, Cout)) // 4, device=dev, dtype=torch.float32)
            ops.timed('block_tail_backward (reduce + moments + grad)', 2 * R * (2 * 64 + 2 * Cout + 64),
                      lambda: _hip.check(L.fgnn_block_tail_backward_moments(
                          P(e), P(st2[2]), P(st2[3]), slope2, P(W2.detach()), P(bias2c), P(st3[0]), P(st3[1]), P(w3.detach()), P(st3[2]),
                          P(st3[3]), slope3, P(gout), None, P(ga2), P(gw3), P(gb3), P(st2[0]), P(st2[1]), P(gw2), P(gb2), P(dsum2),
                          R, Cout, P(ws), ws.numel() * 4, P(ops._fold_scratch(dev)), P(mom), mom.numel() * 4, _hip.stream_ptr())),
                      nflops=8 * R * 64 * Cout)
        else:
            ops.timed('block_tail_backward (reduce + grad)', 2 * R * (2 * 64 + 2 * Cout + 64 + Cout),
                      lambda: _hip.check(L.fgnn_block_tail_backward(
                          P(e), P(st2[2]), P(st2[3]), slope2, P(W2.detach()), P(bias2c), P(st3[0]), P(st3[1]), P(w3.detach()), P(st3[2]),
                          P(st3[3]), slope3, P(gout), P(gz3), P(ga2), P(gw3), P(gb3), P(st2[0]), P(st2[1]), P(gw2), P(gb2), P(dsum2),
                          R, Cout, P(ws), ws.numel() * 4, P(ops._fold_scratch(dev)), _hip.stream_ptr())),
                      nflops=6 * R * 64 * Cout)
        # BatchNorm2 + activation backward on the 64-channel tensor: one element-wise pass (no reduction pass, no finaliser)
        ge = torch.empty_like(e)
        ops.timed('bn_backward (apply)', 3 * e.numel() * 2, lambda: _hip.check(L.fgnn_bn_backward_apply(
            P(e), P(ga2), P(ge), R, 64, _hip.BF16, P(st2[0]), P(st2[1]), P(w2.detach()), P(b2.detach()), slope2, P(dsum2),
            _hip.stream_ptr())))
        # conv2's weight / bias gradient: gz3^T a2 over the R rows (csrc/linear_wgrad_b16.hip); parked when it goes to the flat
        # bucket (ops.defer_wgrad: nothing in the backward reads it)
        record = s_W2 and s_bias2 and ops.folds_deferrable()

        if moments:
            scale3 = st3[2]

            def launch(mom=mom, gW2=gW2, scale3=scale3, W2d=W2.detach(), bias2c=bias2c):
                # (conv2's BIAS gradient in front of a batch-statistics BatchNorm is identically zero — sum gz3 = 0 — nothing is added to it)
                ops.timed('block_tail_wgrad_finish (fold + combine)', 4 * mom.numel(), lambda: _hip.check(L.fgnn_block_tail_wgrad_finish(
                    P(mom), mom.numel() * 4, R, Cout, P(W2d), P(bias2c), P(scale3), P(gW2.view(Cout, 64)), _hip.stream_ptr())))
            operands = (mom, scale3)
        else:
            def launch(a2=a2, gz3=gz3, gW2=gW2, gbias2=gbias2):
                with ops.fold_scope(record) as scope:
                    wsw = scope.slabs(dev, int(L.fgnn_linear_wgrad_workspace_bytes(R, 64, Cout)))
                    ops.timed('linear_wgrad_b16_kernel', 2 * R * (64 + Cout), lambda: _hip.check(L.fgnn_linear_wgrad(
                        P(a2), P(gz3), R, 64, Cout, _hip.BF16, P(gW2.view(Cout, 64)), P(gbias2), P(wsw), wsw.numel() * 4, _hip.stream_ptr())),
                        nflops=2 * R * 64 * Cout)
            operands = (a2, gz3)
        if s_W2 and s_bias2:
            ops.defer_wgrad(launch, operands)
        else:
            launch()
        ha = ctx.has_add
        gadd = [None, None, None]
        for i in range(3):
            if ha[i]:
                gadd[i] = gout if ctx.periods[i] == 1 else pointwise.node_sum(gout, ctx.periods[i])
        return (ge, None if s_w2 else gw2, None if s_b2 else gb2, None, None, None, None, None, None,
                None if s_W2 else gW2.view(pW2.shape).to(pW2.dtype), None if (s_bias2 or gbias2 is None) else gbias2,
                None if s_w3 else gw3, None if s_b3 else gb3, None, None, None, None, None, None, None,
                gadd[0], gadd[1], gadd[2], None)


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
        adds = as_addends(addend)
        if len(adds) > 3:
            return None
        # an addend that is a per-sample vector broadcast over the nodes (ops.broadcast_nodes: the hyper-factor's message to the
        # variables, formed on ONE row per codeword — below) is handed over as that [B, nout] row: the parity block's kernel adds it to
        # every destination (fgnn_mpconv_block_forward_rows); the 96 identical rows are neither written nor read
        row_mask = 0
        for i, a in enumerate(adds):
            src = getattr(a, '_fgnn_bcast_src', None)
            if src is not None and a.shape[2] > 1 and not (fanout or fanin) and src.dtype == x.dtype:
                adds[i] = src
                row_mask |= 1 << i
        for a in adds:
            if a.dtype != x.dtype or not a.permute(0, 2, 3, 1).is_contiguous():
                return None
        a0, a1, a2 = (adds + [None, None, None])[:3]
        W1, s1, t1, F, s2, t2, W2, s3, t3 = self.folded_for_inference(x.device)
        y = torch.empty((B, M, 1, nout), device=x.device, dtype=x.dtype).permute(0, 3, 1, 2)
        d = _hip.make_desc(x, nn_idx, etype, 64, mp.nedge_types, _hip.EXT_NONE, _hip.AGG_MAX, True, y)
        d.nin = 64                      # the inner operator's width; x / y strides stay the block's
        P = _hip._ptr
        if fanin:
            rc = _hip.lib().fgnn_mpconv_block_forward_fanin(ctypes.byref(d), P(x), P(etype), P(W1), P(s1), P(t1), P(F),
                                                            P(s2), P(t2), P(W2), P(s3), P(t3), float(bn1.slope), nin, nout,
                                                            P(a0), P(a1), P(a2), P(y), _hip.stream_ptr())
            if rc == _hip.EUNSUPPORTED:
                return None
            _hip.check(rc)
            return y
        if fanout and not adds and B > 1 and ops.single_source_fanout(x, nn_idx, etype):
            # every destination receives the same message and the rest of the block maps identical rows to identical rows (eval-mode
            # BatchNorms are per-channel affines): ONE row per codeword, handed on as a broadcast (round 5 did this for training)
            y1 = torch.empty((B, 1, 1, nout), device=x.device, dtype=x.dtype).permute(0, 3, 1, 2)
            et1 = etype[:, :, :1, :]
            d1 = _hip.make_desc(x, nn_idx[:, :1, :], et1, 64, 1, _hip.EXT_NONE, _hip.AGG_MAX, True, y1)
            d1.nin = 64
            rc = _hip.lib().fgnn_mpconv_block_forward_fanout(ctypes.byref(d1), P(x), P(et1), P(W1), P(s1), P(t1), P(F), P(s2), P(t2), P(W2),
                                                             P(s3), P(t3), float(bn1.slope), nin, nout, None, None, None, P(y1), _hip.stream_ptr())
            if rc != _hip.EUNSUPPORTED:
                _hip.check(rc)
                return ops.broadcast_nodes(y1, M)
        if fanout:
            rc = _hip.lib().fgnn_mpconv_block_forward_fanout(ctypes.byref(d), P(x), P(etype), P(W1), P(s1), P(t1), P(F),
                                                             P(s2), P(t2), P(W2), P(s3), P(t3), float(bn1.slope), nin, nout,
                                                             P(a0), P(a1), P(a2), P(y), _hip.stream_ptr())
            if rc == _hip.EUNSUPPORTED:
                return None
            _hip.check(rc)
            return y
        rc = _hip.lib().fgnn_mpconv_block_forward_rows(ctypes.byref(d), P(x), P(nn_idx), P(etype), P(W1), P(s1), P(t1), P(F),
                                                       P(s2), P(t2), P(W2), P(s3), P(t3), float(bn1.slope), nin, nout, P(a0), P(a1),
                                                       P(a2), row_mask, P(y), _hip.stream_ptr())
        if rc == _hip.EUNSUPPORTED:
            return None
        _hip.check(rc)
        return y

    def forward(self, node_feature, nn_idx, etype, addend=None):
        """``addend`` (optional, the caller's running sum of the same shape as the output — a tensor, a list of tensors,
        or a callable returning either, evaluated right before it is consumed) is added by conv2's fused
        BatchNorm+activation kernel instead of a separate elementwise pass."""
        nn_idx = ops.shared_graph_view(nn_idx)
        if etype.dtype != node_feature.dtype and torch.is_autocast_enabled('cuda') and etype.is_floating_point():
            # (conv1 brings the state to the autocast dtype; edge weights a script built from f32 constants follow — ops.autocast_operands)
            etype = etype.to(torch.get_autocast_dtype('cuda'))
        staged = self.training and torch.is_grad_enabled()
        if callable(addend) and not staged:
            addend = addend()                                            # the one-kernel block needs it up front
        if not callable(addend):
            addend = as_addends(addend)                                  # a tensor, a list or None -> a list (never truth-tested as a tensor)
            if not staged and len(addend) > 3:                           # ... the one-kernel block takes up to three (a fourth joins the third first)
                addend = addend[:2] + [ops.add_n(addend[2:])]
            addend = addend if addend else None
        if not staged:
            y = self._fused_eval(node_feature, nn_idx, etype, addend)
            if y is not None:
                return y
        # ONE source node feeding every destination through identical edges (the LDPC hyper-factor -> variables call): the whole
        # block's output is a per-sample vector — conv1 on the single source, the message, BatchNorm + ReLU, conv2, BatchNorm +
        # LeakyReLU all act on M identical rows — so it is computed on ONE row per sample and handed on as a broadcast
        # (ops.broadcast_nodes).  The batch statistics of M identical rows per sample are those of one row per sample (same mean and
        # biased variance); only the running variance's unbiased correction counts the rows: population_mult = M.
        M = ops.single_source_fanout(node_feature, nn_idx, etype) if (staged and not self.with_residual and addend is None
                                                                     and node_feature.shape[0] > 1) else 0
        if M:
            B, C = node_feature.shape[:2]
            one = node_feature.reshape(B, 1, 1, C).permute(0, 3, 1, 2)   # canonical channel-fastest strides for the one-node tensor
            y1 = self._forward_staged(one, nn_idx[:, :1, :], etype[:, :, :1, :], None, M)
            return ops.broadcast_nodes(y1, M)
        return self._forward_staged(node_feature, nn_idx, etype, addend, 1)

    def _forward_staged(self, node_feature, nn_idx, etype, addend, mult):
        fuse = self.training            # BatchNorm statistics ride in the 1x1 map's epilogue when training
        h = self._fused_train_head(node_feature)
        if h is None:
            h = self.conv1[1](self.conv1[0](node_feature, bn=self.conv1[1] if fuse else None))
        y = self._fused_train_tail(h, nn_idx, etype, addend, mult)
        if y is not None:
            return y + node_feature if self.with_residual else y
        h = self.mp_conv(h, nn_idx, etype, population_mult=mult)
        h = self.conv2[0](h, bn=self.conv2[1] if (fuse and mult == 1) else None)
        if callable(addend):            # produced on another stream: asked for (and waited on) only where it is consumed
            addend = addend()
        h = self.conv2[1](h, addend=addend, population_mult=mult)
        return h + node_feature if self.with_residual else h

    def _fused_train_head(self, x):
        """Training, bf16, 64 channels in the middle: conv1 + BatchNorm + LeakyReLU as ``_BlockHead`` (same forward kernels, fused
        backward).  None = not this family."""
        from .. import _hip
        conv, bn = self.conv1[0], self.conv1[1]
        if not (FUSE_TRAIN_HEAD and self.training and torch.is_grad_enabled() and x.is_cuda and isinstance(conv, PointwiseConv2d)
                and isinstance(bn, BatchNormAct2d) and bn.training and conv.out_channels == 64 and conv.in_channels in _HEAD_WIDTHS
                and bn.track_running_stats and bn.affine and bn.momentum is not None and conv.weight.dtype == torch.float32
                and bn.weight.dtype == torch.float32 and (x.requires_grad or conv.weight.requires_grad)):
            return None
        B, C, H, W = x.shape
        rows = x.permute(0, 2, 3, 1)
        if not rows.is_contiguous():
            rows = rows.contiguous()
        rows = rows.view(B * H * W, C)
        if torch.is_autocast_enabled():
            rows = rows.to(torch.get_autocast_dtype('cuda'))
        L = _hip.lib()
        R = B * H * W
        if rows.dtype != torch.bfloat16 or R < 2 or not L.fgnn_block_tail_partials(R, C) or not L.fgnn_linear_forward_partials(R, C, 64):
            return None
        a1 = _BlockHead.apply(rows, conv.weight.view(64, C), conv.bias, bn.weight, bn.bias, bn.running_mean, bn.running_var,
                              bn.num_batches_tracked, bn.momentum, bn.eps, float(bn.slope), ops.fan_box(x))
        return a1.view(B, H, W, 64).permute(0, 3, 1, 2)

    def _fused_train_tail(self, h, nn_idx, etype, addend, mult=1):
        """Training, bf16: the operator's pre-BatchNorm output goes straight into csrc/block_tail.hip (``_BlockTail``) —
        BatchNorm + ReLU, conv2, BatchNorm + LeakyReLU and the addends without conv2's Cout-wide output ever being stored.
        None = not this family (the staged path runs)."""
        from .. import _hip
        from .message_op import _EXT_CODE
        mp = self.mp_conv
        bn2, conv2, bn3 = mp.bn, self.conv2[0], self.conv2[1]
        if not (FUSE_TRAIN_TAIL and self.training and torch.is_grad_enabled() and h.is_cuda and h.dtype == torch.bfloat16
                and mp.nou == 64 and mp.extension == mp_conv_type.NO_EXTENSION      # (the operand preparation of the extension branches lives in mp_conv_v2.forward)
                and isinstance(mp.aggregtor, str) and isinstance(mp.activation_fn, torch.nn.ReLU)
                and isinstance(bn2, BatchNormAct2d) and bn2.training and isinstance(bn3, BatchNormAct2d) and bn3.training
                and isinstance(conv2, PointwiseConv2d) and conv2.in_channels == 64 and conv2.out_channels in (64, 128, 256)
                and all(b.track_running_stats and b.affine and b.momentum is not None for b in (bn2, bn3))
                and conv2.weight.dtype == torch.float32 and bn2.weight.dtype == torch.float32):
            return None
        B, M = h.shape[0], nn_idx.shape[1]
        if B * M < 2 or not _hip.lib().fgnn_block_tail_partials(B * M, conv2.out_channels):
            return None
        from . import pointwise
        z = ops.mpconv(h, nn_idx, etype, mp.filters, mp.bias, mp.nou, mp.nedge_types, _EXT_CODE[mp.extension],
                       _hip.AGG_CODES[mp.aggregtor], bn=pointwise.bn_spec(bn2) if mult == 1 else None)
        rows = z.permute(0, 2, 3, 1)
        if not rows.is_contiguous():
            pointwise.take_pending_stats(rows)                        # (drop them: they describe another buffer)
            rows = rows.contiguous()
        rows = rows.view(B * M, 64)
        Cout = conv2.out_channels
        got = {}

        def addend_rows():
            """Evaluates the caller's addend (a callable joins the stream that produced it) and returns (the three row views the
            apply kernel reads, their periods); the differentiable tensors stay in `got` for the gradient route."""
            with torch.enable_grad():
                a = addend() if callable(addend) else addend
                addends = as_addends(a)
                if len(addends) > 3:
                    addends = addends[:2] + [ops.add_n(addends[2:])]
                addends, periods = pointwise.split_broadcast(addends)
                arows = [None, None, None]
                for i, t in enumerate(addends):
                    ar = t.permute(0, 2, 3, 1)
                    if ar.dtype != rows.dtype or not ar.is_contiguous():
                        ar = ar.to(rows.dtype).contiguous()
                    arows[i] = ar.view(-1, Cout)
            got['arows'] = arows
            got['periods'] = tuple((periods + [1, 1, 1])[:3])
            got['streams'] = tuple(([getattr(t, '_fgnn_home_stream', None) for t in addends] + [None, None, None])[:3])
            return arows, got['periods']

        late = ROUTE_ADDEND_GRADS and LATE_JOIN and callable(addend) and torch.is_grad_enabled()
        arows, periods = (None, (1, 1, 1)) if late else addend_rows()
        route = late or (ROUTE_ADDEND_GRADS and torch.is_grad_enabled() and any(a is not None and a.requires_grad for a in arows))
        if late:
            def detached():
                ar, pr = addend_rows()
                return [None if a is None else a.detach() for a in ar], pr
            tail_adds = [detached, None, None]
        else:
            tail_adds = [(a.detach() if (route and a is not None) else a) for a in arows]
        y = _BlockTail.apply(rows, bn2.weight, bn2.bias, 0.0, float(bn3.slope), bn2.momentum, bn2.eps, bn3.momentum, bn3.eps,
                             conv2.weight.view(Cout, 64), conv2.bias, bn3.weight, bn3.bias, bn2.running_mean, bn2.running_var,
                             bn2.num_batches_tracked, bn3.running_mean, bn3.running_var, bn3.num_batches_tracked,
                             0 if mult == 1 else B * M * mult, *tail_adds, periods)
        if route:
            pairs = [(a, q, st) for a, q, st in zip(got['arows'], got['periods'], got['streams']) if a is not None]
            y = _AddendRoute.apply(y, tuple(q for _, q, _ in pairs), tuple(st for _, _, st in pairs), *[a for a, _, _ in pairs])
        return y.view(B, M, 1, Cout).permute(0, 3, 1, 2)
