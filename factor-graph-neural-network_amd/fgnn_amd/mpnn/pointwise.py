"""Node-wise (1x1) maps as GEMMs on channel-fastest activations.

The reference expresses every node-wise linear map as ``Conv2d(cin, cout, 1)`` on [B,C,N,1]
tensors (base_model.py:43-90, mp_nn_residual.py:25-35).  On ROCm that lowers to MIOpen
convolution solvers that are a poor fit for H*W = N*1 "images" (rocprof, profiles/r01: they
dominate the step).  ``PointwiseConv2d`` keeps Conv2d's parameters / state_dict keys
(``weight [cout,cin,1,1]``, ``bias``) but runs the map as one rocBLAS/hipBLASLt GEMM
[B*N, cin] x [cin, cout] on the channels-last view, which is also the layout the fused message
kernel reads and writes without a transpose.  Outputs are logical [B,C,N,W] with
channels-last strides; every consumer in this package is stride-agnostic.
"""
import weakref

import torch

from .. import _hip


_CAST_CACHE = {}          # id(tensor) -> [weakref(tensor), version, epoch, low-precision copy]
_CAST_EPOCH = 0
_FLAT_MIRRORS = []        # [flat f32 buffer (weak), {dtype: mirror}, version, epoch, {id(param): version}, params (weak)]


def invalidate_casts():
    """Mark every cached low-precision copy stale (call after updating parameters through an alias that does not
    bump their version counters, e.g. a flat parameter buffer; dp.FlatAdam does).  The copies are refreshed IN
    PLACE at their next use — their storage is never replaced, because a captured hipGraph may hold its address;
    graph.StepGraph calls this right before capturing, so the refresh kernels are part of the graph and every
    replay casts the current parameters."""
    global _CAST_EPOCH
    _CAST_EPOCH += 1


_STATE_EPOCH = 0          # bumped whenever this package changes parameters / buffers behind torch's version counters


def note_state_change():
    """Kernels write BatchNorm running statistics through raw pointers and flat optimizers update parameters through
    an alias: neither bumps the tensors' version counters.  Everything derived from them and cached across calls
    (folded eval-mode BatchNorm affines, the one-kernel inference block's weights) keys on this counter as well."""
    global _STATE_EPOCH
    _STATE_EPOCH += 1


def state_epoch():
    return _STATE_EPOCH


def refresh_in_place(old, new):
    """Derived constants cached across calls (folded BatchNorm affines, the one-kernel blocks' weight images) keep their
    STORAGE when they are recomputed: an inference hipGraph captured earlier holds these addresses, so a refresh must
    land in the same memory instead of freeing it.  Returns the tuple to keep."""
    new = tuple(t.contiguous() for t in new)
    if old is not None and len(old) == len(new) and all(
            o.shape == n.shape and o.dtype == n.dtype and o.device == n.device for o, n in zip(old, new)):
        for o, n in zip(old, new):
            o.copy_(n)
        return old
    return new


def register_flat_parameters(flat, params=()):
    """``flat`` (f32) backs many parameters (dp.FlatGradBucket(flatten_params=True)): their low-precision copies
    become views of ONE mirror buffer that is refreshed by a single cast kernel per step instead of one per tensor."""
    _FLAT_MIRRORS[:] = [e for e in _FLAT_MIRRORS if e[0]() is not None]      # buffers that are gone take their mirrors along
    _FLAT_MIRRORS.append([weakref.ref(flat), {}, -1, -1, {}, [weakref.ref(q) for q in params]])


def _from_flat_mirror(t, dtype):
    for ent in _FLAT_MIRRORS:
        flat = ent[0]()
        if flat is None:
            continue
        off = t.data_ptr() - flat.data_ptr()
        if 0 <= off < flat.numel() * 4 and t.dtype == flat.dtype and t.device == flat.device and t.is_contiguous() \
                and off % 4 == 0 and off // 4 + t.numel() <= flat.numel():
            mirror = ent[1].get(dtype)
            if mirror is None:
                mirror = ent[1][dtype] = torch.empty_like(flat, dtype=dtype)
                ent[2] = -1
            if ent[2] != flat._version or ent[3] != _CAST_EPOCH or ent[4].get(id(t)) != t._version:
                for mm in ent[1].values():
                    mm.copy_(flat)                               # one kernel for every parameter of the model
                ent[2], ent[3] = flat._version, _CAST_EPOCH
                # load_state_dict writes through the parameters, not through flat: remember every version as of now
                ent[4] = {id(q()): q()._version for q in ent[5] if q() is not None}
                ent[4][id(t)] = t._version
            return mirror[off // 4: off // 4 + t.numel()].view(t.shape)
    return None


def cast_cached(t, dtype):
    """``t.to(dtype)`` remembered until ``t`` is modified in place (optimizer step, load_state_dict):
    the forward re-uses the bf16 copies of the weights instead of re-casting ~170 tensors.  A stale copy is
    refreshed in place (same storage — see invalidate_casts)."""
    if t is None or t.dtype == dtype:
        return t
    if _FLAT_MIRRORS:
        v = _from_flat_mirror(t, dtype)
        if v is not None:
            return v
    key = id(t)
    hit = _CAST_CACHE.get(key)
    if hit is not None and hit[0]() is t and hit[3].dtype == dtype and hit[3].device == t.device and hit[3].shape == t.shape:
        if hit[1] != t._version or hit[2] != _CAST_EPOCH:
            hit[3].copy_(t.detach())
            hit[1], hit[2] = t._version, _CAST_EPOCH
        return hit[3]
    out = t.detach().to(dtype)
    # the source is held weakly: when the parameter dies its entry (and the copy) goes with it
    _CAST_CACHE[key] = [weakref.ref(t, lambda _, key=key: _CAST_CACHE.pop(key, None)), t._version, _CAST_EPOCH, out]
    return out


_PENDING_STATS = None     # (rows tensor, stats [4, C], running_mean) finalised batch statistics of the last statistics-producing launch
# A producer that takes ``bn=`` finalises the BatchNorm ITSELF: momentum applied to the running statistics, num_batches_tracked + 1.
# When the consumer then never sees the pending statistics (they describe another buffer: a non-contiguous copy; another map ran
# in between; the BatchNorm fell back to torch), whoever re-derives the batch statistics must NOT finalise a second time — the
# momentum would be applied twice and the counter advance by 2 for one forward.  The running_mean buffers of such dropped
# finalisations are remembered here until that BatchNorm's own statistics pass (or its torch fallback) has run without them.
_ALREADY_FINAL = set()    # data_ptr() of running_mean buffers whose BatchNorm a producer finalised for the forward in flight


def _drop_pending():
    global _PENDING_STATS
    pend, _PENDING_STATS = _PENDING_STATS, None
    if pend is not None and pend[2] is not None:
        _ALREADY_FINAL.add(pend[2].data_ptr())


def finalised_by_producer(running_mean):
    """True ONCE when a producer launch already finalised the BatchNorm that owns ``running_mean`` for this forward and its
    pending statistics were dropped: the caller forms the batch statistics again but leaves the running buffers alone."""
    if running_mean is None or not _ALREADY_FINAL:
        return False
    key = running_mean.data_ptr()
    if key in _ALREADY_FINAL:
        _ALREADY_FINAL.discard(key)
        return True
    return False


def bn_spec(bn):
    """(gamma, beta, running_mean, running_var, num_batches_tracked, momentum, eps) of a training-mode BatchNorm module the
    hand-written kernels can finalise themselves, else None (eval mode, no running statistics, cumulative momentum, ...)."""
    if bn is None or not bn.training or not bn.track_running_stats or not bn.affine or bn.momentum is None:
        return None
    if bn.weight.dtype != torch.float32 or not bn.weight.is_cuda:
        return None
    return (bn.weight, bn.bias, bn.running_mean, bn.running_var, bn.num_batches_tracked, float(bn.momentum), float(bn.eps))


def make_final(spec, C, device, count, population=0):
    """(stats [4, C] f32 — rows mean, invstd, scale, shift —, the fgnn_bn_final describing it) for a BatchNorm ``spec`` (bn_spec) whose
    statistics run over ``count`` rows standing for ``population`` rows of the reference's tensor (0 = count)."""
    gamma, beta, rm, rv, nbt, momentum, eps = spec
    stats = torch.empty((4, C), device=device, dtype=torch.float32)
    fin = _hip.bn_final(stats, gamma.detach(), beta.detach(), rm, rv, nbt, momentum, eps, count, population)
    return stats, fin


def hip_linear(rows, weight, bias, bn=None, transposed=False):
    """``rows @ weight.T + bias`` through csrc/linear_fwd_b16.hip (bf16 rows, f32 parameters, channel counts in
    multiples of 64): one pass over rows and the output at HBM rate, no cast of the weights.  With ``bn`` (a ``bn_spec``
    tuple) the launch also forms the per-channel batch statistics of the output and FINALISES that BatchNorm itself (its last
    workgroup: csrc/fgnn_gridfold.h) — running statistics, num_batches_tracked, scale / shift; the result waits for the
    BatchNorm that follows (``take_pending_stats``).  None = shape not handled."""
    global _PENDING_STATS
    _drop_pending()
    if not (rows.is_cuda and rows.dtype == torch.bfloat16 and weight.dtype == torch.float32 and rows.is_contiguous()):
        return None
    R, cin = rows.shape
    cout = weight.shape[1] if transposed else weight.shape[0]      # transposed: weight is [cin, cout] (y = rows @ weight)
    # measured on MI355X (tools/lbench.py, cold tensors, R = 393 k rows): the streaming kernel beats hipBLASLt up to 128x128 maps
    # (23.7 vs 34.4 us at 64x64, 37.7 vs 45.9 at 64x128, 52.7 vs 57.6 at 128x128: the step time does not move with that last
    # one); at 64x256 / 256x64 it loses 10 us (72 vs 61) but with the statistics epilogue saves the BatchNorm's own 25 us pass
    # over the output; wider maps stay with hipBLASLt (4 TB/s there)
    if cin * cout > (16384 if bn is not None else 8192):
        return None
    L = _hip.lib()
    npart = L.fgnn_linear_forward_partials(R, cin, cout)
    if npart == 0 or (bn is not None and R < 2):
        return None
    from .. import ops
    w = weight.detach()
    if not w.is_contiguous():
        w = w.contiguous()
    b = None if bias is None else bias.detach().float().contiguous()
    y = torch.empty((R, cout), device=rows.device, dtype=rows.dtype)
    ws = fold = fin = stats = None
    if bn is not None:
        ws = ops._workspace(rows.device, int(L.fgnn_bn_workspace_bytes(R, cout)))
        fold = ops._fold_scratch(rows.device)
        stats, fin = make_final(bn, cout, rows.device, R)
    ops.timed('linear_fwd_b16_kernel', 2 * R * (cin + cout),
              lambda: _hip.check(L.fgnn_linear_forward(_hip._ptr(rows), _hip._ptr(w), _hip._ptr(b), _hip._ptr(y), R, cin,
                                                       cout, _hip._ptr(ws), fin, _hip._ptr(fold), int(transposed), _hip.stream_ptr())),
              nflops=2 * R * cin * cout)
    if bn is not None:
        note_state_change()                 # running statistics / num_batches_tracked were just updated in place
        _ALREADY_FINAL.discard(bn[2].data_ptr())
        _PENDING_STATS = (y, stats, bn[2])
    return y


def set_pending_stats(rows, stats, running_mean=None):
    """A kernel just finalised the batch statistics ``stats`` [4, C] of ``rows`` ([R, C]) for the BatchNorm that follows (the one
    that owns ``running_mean``)."""
    global _PENDING_STATS
    _drop_pending()
    if running_mean is not None:
        _ALREADY_FINAL.discard(running_mean.data_ptr())
    _PENDING_STATS = (rows, stats, running_mean)


def take_pending_stats(rows):
    """The finalised statistics [4, C] waiting for exactly this tensor, else None."""
    global _PENDING_STATS
    pend = _PENDING_STATS
    if pend is not None and pend[0].data_ptr() == rows.data_ptr() and pend[0].shape == rows.shape:
        _PENDING_STATS = None
        return pend[1]
    _drop_pending()             # (missed: the producer's finalisation is remembered, see _ALREADY_FINAL)
    return None


def batch_stats(rows, spec, population=0):
    """stats [4, C] of a training-mode BatchNorm over ``rows`` [R, C]: the producer's (take_pending_stats) when it finalised them,
    else one reducing launch of csrc/bnact.hip (its last workgroup finalises)."""
    from .. import ops
    stats = take_pending_stats(rows)
    if stats is not None:
        return stats
    L = _hip.lib()
    R, C = rows.shape
    if finalised_by_producer(spec[2]):
        # the producing launch already applied this forward's momentum update and counted the batch, but its statistics did not
        # reach us (they described another buffer): scale / shift only, the running buffers and the counter stay as they are
        spec = (spec[0], spec[1], None, None, None) + tuple(spec[5:])
    stats, fin = make_final(spec, C, rows.device, R, population)
    ws = ops._workspace(rows.device, int(L.fgnn_bn_workspace_bytes(R, C)))
    fold = ops._fold_scratch(rows.device)
    ops.timed('bn_stats (reduce + finalise)', rows.numel() * rows.element_size(), lambda: _hip.check(L.fgnn_bn_stats(
        _hip._ptr(rows), R, C, _hip.dtype_code(rows), fin, _hip._ptr(ws), ws.numel() * 4, _hip._ptr(fold), _hip.stream_ptr())))
    note_state_change()
    return stats


def node_sum(g, M):
    """[R, C] -> [R / M, C]: sum over each sample's M consecutive rows (the gradient of a per-sample row that was broadcast over the
    sample's nodes), one pass (csrc/sum_n.hip: node_sum_kernel)."""
    from .. import ops
    R, C = g.shape
    g = g.contiguous()
    out = torch.empty((R // M, C), device=g.device, dtype=g.dtype)
    rc = []
    ops.timed('node_sum_kernel', g.numel() * g.element_size(), lambda: rc.append(_hip.lib().fgnn_node_sum(
        _hip._ptr(g), _hip._ptr(out), R // M, M, C, _hip.dtype_code(g), _hip.stream_ptr())))
    if rc[0] == _hip.EUNSUPPORTED:
        # channel counts outside the kernel's 16-byte chunks (C % 8 for bf16, % 4 for f32): the broadcast path is taken for any
        # width (ops.single_source_fanout), so its backward must exist for any width too — a device-side f32 sum
        return g.view(R // M, M, C).sum(1, dtype=torch.float32).to(g.dtype)
    _hip.check(rc[0])
    return out


class _RowLinear(torch.autograd.Function):
    """y = rows @ W^T + b over R = B*N rows.  Forward and grad-input are plain GEMMs (rocBLAS /
    hipBLASLt do those well); the weight / bias gradient is the tall-skinny product
    gW = gy^T rows (K = R ~ 4e5, M,N <= 256) that rocBLAS runs 40x off its streaming bound, so it goes
    to the hand-written split-rows kernel csrc/linear_wgrad.hip (f32 accumulation, one pass over
    rows and gy)."""

    @staticmethod
    def forward(ctx, rows, weight, bias, bn=None, precomputed=None, box=None):
        """``bn``: a ``bn_spec`` tuple — the training-mode BatchNorm behind the map, finalised by the map's own launch.
        ``precomputed``: the output, already formed by a fused kernel (blocks.iid_mapping_in): only the graph node is made.
        ``box``: the ``ops.FanBox`` of the state ``rows`` views — the backward deposits (gy, weight) there instead of forming
        ``gy @ weight`` (the fan-out's backward multiplies all its consumers' pairs in one launch)."""
        ctx.box = box
        ctx.save_for_backward(rows, weight)
        ctx.has_bias = bias is not None
        ctx.params = (weight, bias)                     # leaf tensors (ops.grad_sink)
        if precomputed is not None:
            return precomputed
        y = hip_linear(rows, weight, bias, bn)
        if y is not None:
            return y
        w = cast_cached(weight._base if weight._base is not None else weight, rows.dtype).view(weight.shape)
        b = cast_cached(bias, rows.dtype)
        return torch.nn.functional.linear(rows, w, b)

    @staticmethod
    def backward(ctx, gy):
        rows, weight = ctx.saved_tensors
        from .. import ops
        ops.backward_node_begins()
        gy = gy.contiguous()
        if gy.dtype != rows.dtype:
            gy = gy.to(rows.dtype)
        R, cin = rows.shape
        cout = weight.shape[0]
        wparam, bparam = ctx.params
        # weight may be a [cout,cin,1,1] Conv2d parameter viewed as [cout,cin]: its .grad lives on the base
        base = wparam._base if wparam._base is not None and wparam._base.numel() == wparam.numel() else wparam
        gw_sink, gb_sink = ops.grad_sink(base), ops.grad_sink(bparam)
        sinks = gw_sink is not None and (not ctx.has_bias or gb_sink is not None)
        grows = None
        taken = 0
        if ctx.needs_input_grad[0] and ctx.box is not None:
            # (gy @ weight joins the state's other gradients in the fan-out's backward — and, with the parameter gradients going to
            # sinks, this map's weight gradient joins the other consumers' there too: one pass over the state's rows for all of them)
            taken = ctx.box.deposit(gy, weight, wgrad=(rows, gw_sink, gb_sink if ctx.has_bias else None) if sinks else None)
        if taken:
            grows = ctx.box.placeholder(rows.shape)
        elif ctx.needs_input_grad[0]:
            grows = hip_linear(gy, weight, None, transposed=True)      # gy [R,cout] @ weight [cout,cin]
            if grows is None:
                grows = gy @ cast_cached(weight._base if weight._base is not None else weight, gy.dtype).view(weight.shape)
        if taken == 2:
            return grows, None, None, None, None, None
        # wide maps: bf16 multiples of 64 up to 256 and f32 multiples of 4 (csrc/linear_wgrad_f32.hip) have their own kernels;
        # what is left (odd widths) goes to the library
        if cin * cout >= 256 * 256 and not (rows.dtype == torch.bfloat16 and cin % 64 == 0 and cout % 64 == 0
                                            and cin <= 256 and cout <= 256) and not (
                rows.dtype == torch.float32 and cin % 4 == 0 and cout % 4 == 0 and cin <= 1024 and cout <= 1024 and R >= 2048):
            gw = (gy.t() @ rows).float()                # f32 square 256-wide maps: rocBLAS is ahead there
            gb = gy.float().sum(0) if ctx.has_bias else None
            return grows, gw.to(weight.dtype), (gb.to(weight.dtype) if gb is not None else None), None, None, None
        L = _hip.lib()
        gw = gw_sink if gw_sink is not None else torch.zeros((cout, cin), device=rows.device, dtype=torch.float32)
        gb = None
        if ctx.has_bias:
            gb = gb_sink if gb_sink is not None else torch.zeros((cout,), device=rows.device, dtype=torch.float32)
        record = sinks and ops.folds_deferrable()        # (decided here, inside the pass: a parked launch may go out from its end-of-pass callback)

        def launch(rows=rows, gy=gy, gw=gw, gb=gb):      # (the closure keeps rows / gy alive until the kernel is issued)
            with ops.fold_scope(record) as scope:
                ws = scope.slabs(rows.device, int(L.fgnn_linear_wgrad_workspace_bytes(R, cin, cout)))
                ops.timed('linear_wgrad_b16_kernel' if rows.dtype == torch.bfloat16 else 'linear_wgrad_kernel',
                          rows.element_size() * R * (cin + cout),
                          lambda: _hip.check(L.fgnn_linear_wgrad(_hip._ptr(rows), _hip._ptr(gy), R, cin, cout,
                                                                 _hip.dtype_code(rows), _hip._ptr(gw), _hip._ptr(gb),
                                                                 _hip._ptr(ws), ws.numel() * 4, _hip.stream_ptr())),
                          nflops=2 * R * cin * cout)
        # nothing in the backward reads a weight gradient: with both gradients going to the flat bucket the kernel is parked
        # and issued where its stream would otherwise wait for the other one (ops.defer_wgrad)
        if sinks:
            ops.defer_wgrad(launch, (rows, gy))
        else:
            launch()
        return (grows, None if gw_sink is not None else gw.to(weight.dtype),
                None if (gb is None or gb_sink is not None) else gb.to(weight.dtype), None, None, None)


class PointwiseConv2d(torch.nn.Conv2d):
    def __init__(self, in_channels, out_channels, kernel_size=1, bias=True):
        assert kernel_size in (1, (1, 1)), 'PointwiseConv2d is a 1x1 map'
        super().__init__(in_channels, out_channels, 1, bias=bias)

    def forward(self, x, bn=None):
        """``bn``: the BatchNormAct2d module the caller hands the result straight to.  In training mode the map's launch then forms
        that BatchNorm's batch statistics in its epilogue and finalises them (running statistics included): the BatchNorm neither
        re-reads the tensor for them nor launches a finaliser."""
        bn = bn_spec(bn)
        B, C, H, W = x.shape
        rows = x.permute(0, 2, 3, 1)                    # [B,H,W,C] view; free when channels-last
        if not rows.is_contiguous():
            rows = rows.contiguous()
        rows = rows.view(B * H * W, C)
        if torch.is_autocast_enabled():
            rows = rows.to(torch.get_autocast_dtype('cuda'))
        weight = self.weight.view(self.out_channels, C)
        if rows.is_cuda and rows.dtype in (torch.float32, torch.bfloat16) and torch.is_grad_enabled() and (
                weight.requires_grad or rows.requires_grad):
            from .. import ops
            y = _RowLinear.apply(rows, weight, self.bias, bn, None, ops.fan_box(x))
        else:
            needs_grad = torch.is_grad_enabled() and (weight.requires_grad or rows.requires_grad)
            y = hip_linear(rows, weight, self.bias) if rows.is_cuda and not needs_grad else None
            if y is None and needs_grad:
                # dtypes / devices outside the hand-written path (f64 gradcheck, fp16 autocast, CPU bf16): a plain
                # differentiable cast — never the detached cached copies, which would train with zero weight gradients
                b = None if self.bias is None else self.bias.to(rows.dtype)
                y = torch.nn.functional.linear(rows, weight.to(rows.dtype), b)
            elif y is None:
                w = cast_cached(self.weight, rows.dtype).view(self.out_channels, C)
                b = cast_cached(self.bias, rows.dtype)
                y = torch.nn.functional.linear(rows, w, b)
        return y.view(B, H, W, self.out_channels).permute(0, 3, 1, 2)


class _InstNormAct(torch.autograd.Function):
    """act(InstanceNorm(x)) over the node axis of a channel-fastest [B,C,N,1] tensor: one HIP kernel
    forward, one backward (csrc/instnorm.hip); only x is saved."""

    @staticmethod
    def forward(ctx, x, relu, precomputed=None):
        """``precomputed``: the output rows [B,N,1,C], already formed by a fused kernel (blocks.iid_mapping_in)."""
        B, C, N, _ = x.shape
        rows = x.permute(0, 2, 3, 1)                    # [B,N,1,C]
        if not rows.is_contiguous():
            rows = rows.contiguous()
        ctx.save_for_backward(rows)
        ctx.relu = relu
        if precomputed is not None:
            return precomputed.permute(0, 3, 1, 2)
        y = torch.empty_like(rows)
        from .. import ops
        ops.timed('instnorm_fwd_kernel', 2 * rows.numel() * rows.element_size(),
                  lambda: _hip.check(_hip.lib().fgnn_instnorm_forward(_hip._ptr(rows), _hip._ptr(y), B, N, C,
                                                                      _hip.dtype_code(rows), int(relu),
                                                                      _hip.stream_ptr())))
        return y.permute(0, 3, 1, 2)

    @staticmethod
    def backward(ctx, gy):
        (rows,) = ctx.saved_tensors
        from .. import ops
        ops.backward_node_begins()
        B, N, _, C = rows.shape
        g = gy.permute(0, 2, 3, 1)
        if not g.is_contiguous() or g.dtype != rows.dtype:
            g = g.to(rows.dtype).contiguous()
        gx = torch.empty_like(rows)
        from .. import ops
        ops.timed('instnorm_bwd_kernel', 3 * rows.numel() * rows.element_size(),
                  lambda: _hip.check(_hip.lib().fgnn_instnorm_backward(_hip._ptr(rows), _hip._ptr(g), _hip._ptr(gx), B, N,
                                                                       C, _hip.dtype_code(rows), int(ctx.relu),
                                                                       _hip.stream_ptr())))
        return gx.permute(0, 3, 1, 2), None, None


class _InstNormDot(torch.autograd.Function):
    """out[b,0,n,0] = bias + sum_c w[c] relu(InstanceNorm(x)[b,c,n,0]) — the classifier's InstanceNorm2d -> ReLU -> Conv2d(128, 1, 1)
    (factor_mpnn_sp.py:104-108) as one kernel forward and one backward (csrc/instnorm.hip: instnorm_dot_kernel); only x is saved,
    neither the normalised tensor nor its gradient exists in memory."""

    @staticmethod
    def forward(ctx, x, weight, bias):
        B, C, N, _ = x.shape
        rows = x.permute(0, 2, 3, 1)
        if not rows.is_contiguous():
            rows = rows.contiguous()
        w = weight.detach().reshape(C)
        out = torch.empty((B, 1, N, 1), device=x.device, dtype=x.dtype)
        from .. import ops
        ops.timed('instnorm_dot_kernel<fwd>', rows.numel() * rows.element_size(), lambda: _hip.check(
            _hip.lib().fgnn_instnorm_dot_forward(_hip._ptr(rows), _hip._ptr(w), _hip._ptr(None if bias is None else bias.detach()),
                                                 _hip._ptr(out), B, N, C, _hip.dtype_code(rows), _hip.stream_ptr())))
        ctx.save_for_backward(rows, weight)
        ctx.params = (weight, bias)
        return out

    @staticmethod
    def backward(ctx, gout):
        rows, weight = ctx.saved_tensors
        from .. import ops
        ops.backward_node_begins()
        B, N, _, C = rows.shape
        L = _hip.lib()
        g = gout.reshape(B, N)
        if not g.is_contiguous() or g.dtype != rows.dtype:
            g = g.to(rows.dtype).contiguous()
        wparam, bparam = ctx.params
        gw_sink, gb_sink = ops.grad_sink(wparam), ops.grad_sink(bparam)
        gw = gw_sink if gw_sink is not None else torch.zeros(wparam.shape, device=rows.device, dtype=torch.float32)
        gb = None
        if bparam is not None:
            gb = gb_sink if gb_sink is not None else torch.zeros((1,), device=rows.device, dtype=torch.float32)
        gx = torch.empty_like(rows)
        ws = ops._workspace(rows.device, int(L.fgnn_instnorm_dot_workspace_bytes(B)))
        ops.timed('instnorm_dot_kernel<bwd>', 2 * rows.numel() * rows.element_size(), lambda: _hip.check(
            L.fgnn_instnorm_dot_backward(_hip._ptr(rows), _hip._ptr(weight.detach()), _hip._ptr(g), _hip._ptr(gx), _hip._ptr(gw),
                                         _hip._ptr(gb), B, N, C, _hip.dtype_code(rows), _hip._ptr(ws), ws.numel() * 4,
                                         _hip.stream_ptr())))
        return (gx.permute(0, 3, 1, 2), None if gw_sink is not None else gw.to(wparam.dtype),
                None if (gb is None or gb_sink is not None) else gb.to(bparam.dtype))


INSTNORM_DOT = True       # (module switch: the classifier's closing pair staged when False)


def instnorm_relu_dot(x, conv):
    """``conv(relu(InstanceNorm2d(x)))`` for a one-output 1x1 ``conv`` over 128 channels (the classifier head's closing pair) as
    one kernel, or None when the shape / dtype / device is not the kernel's (the caller then runs the staged modules)."""
    if not (INSTNORM_DOT and x.is_cuda and x.dim() == 4 and x.shape[3] == 1 and x.shape[1] == 128 and 2 <= x.shape[2] <= 128
            and x.dtype in (torch.float32, torch.bfloat16) and isinstance(conv, torch.nn.Conv2d) and conv.out_channels == 1
            and conv.in_channels == 128 and conv.weight.dtype == torch.float32 and conv.weight.is_contiguous()
            and (conv.bias is None or conv.bias.dtype == torch.float32)):
        return None
    if torch.is_grad_enabled() and (x.requires_grad or conv.weight.requires_grad):
        return _InstNormDot.apply(x, conv.weight, conv.bias)
    B, C, N, _ = x.shape
    rows = x.permute(0, 2, 3, 1)
    if not rows.is_contiguous():
        rows = rows.contiguous()
    out = torch.empty((B, 1, N, 1), device=x.device, dtype=x.dtype)
    from .. import ops
    ops.timed('instnorm_dot_kernel<fwd>', rows.numel() * rows.element_size(), lambda: _hip.check(
        _hip.lib().fgnn_instnorm_dot_forward(_hip._ptr(rows), _hip._ptr(conv.weight.detach()),
                                             _hip._ptr(None if conv.bias is None else conv.bias.detach()),
                                             _hip._ptr(out), B, N, C, _hip.dtype_code(rows), _hip.stream_ptr())))
    return out


class NodeInstanceNorm(torch.nn.Module):
    """InstanceNorm2d(affine=False, no running stats) over the node axis of [B,C,N,1], any strides,
    optionally fused with the ReLU that follows it in every reference use (``relu=True``).

    A single node (the LDPC hyper-factor, factor_mpnn_sp.py:77,140) normalises to exactly 0
    — (x-mean)/sqrt(0+eps) — which is what the reference's torch-1.0 era computed and what
    newer torch refuses to compute.  No parameters, so state_dicts match
    torch.nn.InstanceNorm2d's (empty) contribution.
    """
    eps = 1e-5

    def __init__(self, relu=False):
        super().__init__()
        self.relu = relu

    def forward(self, x):
        if x.shape[2] * x.shape[3] == 1:
            return torch.zeros_like(x)
        if x.is_cuda and x.shape[3] == 1 and x.dtype in (torch.float32, torch.bfloat16):
            return _InstNormAct.apply(x, self.relu)
        xf = x.float()                                   # statistics in f32 also for bf16 activations
        var, mean = torch.var_mean(xf, dim=(2, 3), unbiased=False, keepdim=True)
        y = ((xf - mean) * torch.rsqrt(var + self.eps)).to(x.dtype)
        return torch.relu(y) if self.relu else y


def as_addends(addend):
    """``addend`` arguments are one tensor, None, or a list of tensors / Nones: the list of tensors."""
    if addend is None:
        return []
    if isinstance(addend, (list, tuple)):
        return [a for a in addend if a is not None]
    return [addend]


def add_all(y, addend):
    for a in as_addends(addend):
        y = y + a
    return y


def split_broadcast(addends):
    """``addends`` (tensors [B, C, N, 1]) -> (tensors, periods): an addend that is a per-sample vector broadcast over the N nodes
    (``ops.broadcast_nodes``: the LDPC hyper-factor's message to the variables) is replaced by its [B, C, 1, 1] source with period N —
    the apply kernels read it as one row per N output rows and its gradient is the node sum of the output's."""
    ts, periods = [], []
    for a in addends:
        src = getattr(a, '_fgnn_bcast_src', None)
        if src is not None and a.shape[2] > 1:
            ts.append(src)
            periods.append(int(a.shape[2]))
        else:
            ts.append(a)
            periods.append(1)
    return ts, periods


def period_array(periods):
    import ctypes
    periods = (list(periods) + [1, 1, 1])[:3]
    return None if all(q == 1 for q in periods) else (ctypes.c_int32 * 3)(*periods)


class _BatchNormAct(torch.autograd.Function):
    """Train-mode BatchNorm + LeakyReLU(slope) on channel-fastest rows [R, C] (csrc/bnact.hip)."""

    @staticmethod
    def forward(ctx, rows, weight, bias, running_mean, running_var, momentum, eps, slope, addend=None, nbt=None,
                addend2=None, addend3=None, periods=(1, 1, 1), population=0):
        """``periods``: an addend with period m has one row per m rows of the output (a per-sample vector broadcast over the
        sample's m nodes).  ``population``: rows the statistics stand for in the running variance's unbiased correction (0 = R)."""
        from .. import ops
        L = _hip.lib()
        R, C = rows.shape
        dt = _hip.dtype_code(rows)
        stats = batch_stats(rows, (weight, bias, running_mean, running_var, nbt, momentum, eps), population)
        y = torch.empty_like(rows)
        ctx.has_addend = tuple(a is not None for a in (addend, addend2, addend3))
        ctx.periods = tuple(periods)
        ops.timed('bn_apply (forward)', (2 + sum(ctx.has_addend)) * rows.numel() * rows.element_size(),
                  lambda: _hip.check(L.fgnn_bn_apply(_hip._ptr(rows), _hip._ptr(y), R, C, dt, _hip._ptr(stats[2]),
                                                     _hip._ptr(stats[3]), slope, _hip._ptr(addend),
                                                     _hip._ptr(addend2), _hip._ptr(addend3), period_array(periods), _hip.stream_ptr())))
        ctx.save_for_backward(rows, weight, bias, stats)
        ctx.slope = slope
        ctx.params = (weight, bias)
        return y

    @staticmethod
    def backward(ctx, gy):
        from .. import ops
        ops.backward_node_begins()
        rows, weight, bias, stats = ctx.saved_tensors
        L = _hip.lib()
        R, C = rows.shape
        gy = gy.contiguous()
        if gy.dtype != rows.dtype:
            gy = gy.to(rows.dtype)
        gx = torch.empty_like(rows)
        gw_sink, gb_sink = ops.grad_sink(ctx.params[0]), ops.grad_sink(ctx.params[1])
        gw = gw_sink if gw_sink is not None else torch.zeros(C, device=rows.device, dtype=torch.float32)
        gb = gb_sink if gb_sink is not None else torch.zeros(C, device=rows.device, dtype=torch.float32)
        ws = ops._workspace(rows.device, int(L.fgnn_bn_workspace_bytes(R, C)))
        fold = ops._fold_scratch(rows.device)
        ops.timed('bn_backward (reduce + finalise + apply)', 5 * rows.numel() * rows.element_size(),
                  lambda: _hip.check(L.fgnn_bn_backward(
                      _hip._ptr(rows), _hip._ptr(gy), _hip._ptr(gx), R, C, _hip.dtype_code(rows), _hip._ptr(stats[0]),
                      _hip._ptr(stats[1]), _hip._ptr(weight), _hip._ptr(bias), ctx.slope, _hip._ptr(gw), _hip._ptr(gb),
                      _hip._ptr(ws), ws.numel() * 4, _hip._ptr(fold), _hip.stream_ptr())))
        ga = [None, None, None]
        for i in range(3):
            if ctx.has_addend[i] and ctx.needs_input_grad[(8, 10, 11)[i]]:
                ga[i] = gy if ctx.periods[i] == 1 else node_sum(gy, ctx.periods[i])
        return (gx, None if gw_sink is not None else gw, None if gb_sink is not None else gb,
                None, None, None, None, None, ga[0], None, ga[1], ga[2], None, None)


class BatchNormAct2d(torch.nn.BatchNorm2d):
    """``BatchNorm2d`` (same parameters / buffers / state_dict keys) that also applies the activation the
    reference puts right behind it — ``slope`` 0.01 = LeakyReLU (conv1/conv2 of mp_conv_residual), 0 = ReLU
    (iid_mapping_bn, mp_conv_v2's own bn) — in one fused HIP kernel pair when training on a ROCm device;
    everything else (eval mode, CPU, odd channel counts) goes through torch's batch_norm + activation."""

    def __init__(self, num_features, slope=0.0):
        super().__init__(num_features)
        self.slope = float(slope)

    @staticmethod
    def _activate(y, slope):
        if slope == 0.0:
            return torch.relu(y)
        if slope == 1.0:
            return y
        return torch.nn.functional.leaky_relu(y, slope)

    def forward(self, x, addend=None, slope=None, population_mult=1):
        """``addend`` (a tensor of the output's shape, or a list of up to three) is added AFTER the activation — the
        ``acc + block(x) (+ residual + skip)`` that follows every block in FactorNN rides in the apply kernel
        instead of being separate passes.  ``slope`` overrides the module's activation for this call (mp_conv_v2 asks its
        plain BatchNorm for the fused ReLU this way: an argument, not a toggled attribute).  ``population_mult`` = m: every row of
        ``x`` stands for m identical rows of the reference's tensor (a per-sample vector the reference broadcasts over m nodes
        before this BatchNorm): same mean and biased variance, the running variance's unbiased correction counts m times the rows."""
        slope = self.slope if slope is None else float(slope)
        B, C, H, W = x.shape
        addends = as_addends(addend)
        if len(addends) > 3:
            from ..ops import add_n
            addends = addends[:2] + [add_n(addends[2:])]
        # momentum=None (cumulative average) and a one-value batch in training mode (torch raises) stay with torch
        ok = (x.is_cuda and x.dtype in (torch.float32, torch.bfloat16) and self.track_running_stats and
              self.affine and self.momentum is not None and (not self.training or B * H * W > 1) and
              _hip.lib().fgnn_bn_supported(B * H * W, C, _hip.dtype_code(x)))
        wants_grad = torch.is_grad_enabled() and (x.requires_grad or self.weight.requires_grad or
                                                  any(a.requires_grad for a in addends))
        if not ok or (not self.training and wants_grad):
            if population_mult != 1 and self.training:
                raise _hip.FgnnHipError('BatchNormAct2d: population_mult needs the HIP path (a ROCm tensor, running statistics)')
            _drop_pending()
            if self.training and finalised_by_producer(self.running_mean):
                # the map in front already finalised this BatchNorm (momentum applied, batch counted): batch statistics only here
                y = torch.nn.functional.batch_norm(x, None, None, self.weight, self.bias, True, 0.0, self.eps)
                return add_all(self._activate(y, slope), addends)
            return add_all(self._activate(super().forward(x), slope), addends)
        rows = x.permute(0, 2, 3, 1)
        if not rows.is_contiguous():
            rows = rows.contiguous()
        rows = rows.view(B * H * W, C)
        addends, periods = split_broadcast(addends)
        arows = [None, None, None]
        for i, a in enumerate(addends):
            ar = a.permute(0, 2, 3, 1)
            if ar.dtype != rows.dtype or not ar.is_contiguous():
                ar = ar.to(rows.dtype).contiguous()
            arows[i] = ar.view(-1, C)
        periods = tuple((periods + [1, 1, 1])[:3])
        if self.training:                               # num_batches_tracked += 1 rides in the statistics finaliser
            y = _BatchNormAct.apply(rows, self.weight, self.bias, self.running_mean, self.running_var,
                                    self.momentum, self.eps, slope, arows[0], self.num_batches_tracked,
                                    arows[1], arows[2], periods, 0 if population_mult == 1 else B * H * W * population_mult)
        else:                                           # eval: folded affine + activation in one pass
            scale, shift = self._folded()
            y = torch.empty_like(rows)
            _hip.check(_hip.lib().fgnn_bn_apply(_hip._ptr(rows), _hip._ptr(y), B * H * W, C, _hip.dtype_code(rows),
                                                _hip._ptr(scale), _hip._ptr(shift), slope, _hip._ptr(arows[0]),
                                                _hip._ptr(arows[1]), _hip._ptr(arows[2]), period_array(periods), _hip.stream_ptr()))
        return y.view(B, H, W, C).permute(0, 3, 1, 2)

    def _folded(self):
        """Eval-mode BatchNorm as (scale, shift); recomputed only when a parameter / buffer changed."""
        key = (self.weight._version, self.bias._version, self.running_mean._version, self.running_var._version,
               self.weight.device, state_epoch())
        if getattr(self, '_fold_key', None) != key:
            with torch.no_grad():
                scale = self.weight.float() * torch.rsqrt(self.running_var.float() + self.eps)
                shift = self.bias.float() - self.running_mean.float() * scale
            self._fold_key = key
            self._fold = refresh_in_place(getattr(self, '_fold', None), (scale, shift))
        return self._fold
