"""Pure data parallelism for the FGNN models: one process per GPU, one RCCL all-reduce per step.

The reference is single-process (no distributed code at all); BASELINE.json asks for
graph batches sharded over the 8 GPUs of a node with an all-reduce of the loss gradient only.
Every sample (codeword / graph) is independent, so the forward and backward need NO data-path
collective; the only exchange is the parameter-gradient sum.

Design for xGMI (7 point-to-point links x ~153 GB/s per GPU): the whole gradient (1.38 M floats
= 5.5 MB for LDPCModel) lives in ONE flat fp32 buffer whose slices back every ``param.grad``,
so a step issues exactly one ``all_reduce`` — the collective is latency-bound at this size and
bucketing it would only add launches.  BatchNorm statistics stay per-replica (what DDP does
with the reference's plain BatchNorm2d; its "SyncBatchNorm" is an alias, mp_nn.py:4).
"""
import torch
import torch.distributed as dist


def flatten_parameters(params):
    """Move float32 parameters into ONE flat buffer (every ``p.data`` becomes a view of it) and register it with the
    low-precision weight cache: the bf16 copies the library GEMMs read are then views of one mirror that a single cast
    kernel refreshes (pointwise.register_flat_parameters).  Returns the flat buffer."""
    params = list(params)
    if not params:
        raise ValueError('no parameters')
    if any(p.dtype != torch.float32 for p in params):
        raise ValueError('flatten_parameters needs float32 parameters')
    flat = torch.empty(sum(p.numel() for p in params), device=params[0].device, dtype=torch.float32)
    off = 0
    with torch.no_grad():
        for p in params:
            n = p.numel()
            view = flat[off:off + n].view_as(p)
            view.copy_(p.data)
            p.data = view
            off += n
    from .mpnn import pointwise
    pointwise.register_flat_parameters(flat, params)
    return flat


class FlatGradBucket:
    """Owns one contiguous gradient buffer; ``param.grad`` are views into it."""

    def __init__(self, params, process_group=None, flatten_params=False, reduce_single_rank=False):
        """``flatten_params=True`` also moves the parameters themselves into one flat f32 buffer
        (``flat_param``; every ``p.data`` becomes a view of it) so that the optimizer is a handful of
        kernels over 5.5 MB instead of a multi-tensor sweep over ~330 tensors — see ``FlatAdam``."""
        self.params = [p for p in params if p.requires_grad]
        self.group = process_group
        # a group of one rank skips the collective (the sum over one rank is the identity); reduce_single_rank issues it
        # anyway — how a one-GPU box exercises the RCCL call itself (tests/test_dp_two_ranks_gpu.py)
        self.reduce_single_rank = bool(reduce_single_rank)
        if not self.params:
            raise ValueError('no trainable parameters')
        dev, dt = self.params[0].device, torch.float32
        self.numel = sum(p.numel() for p in self.params)
        self.flat = torch.zeros(self.numel, device=dev, dtype=dt)
        self.flat_param = flatten_parameters(self.params) if flatten_params else None
        off = 0
        for p in self.params:
            n = p.numel()
            p.grad = self.flat[off:off + n].view_as(p)
            off += n
        # the gradient kernels may add straight into these slices (ops.grad_sink): the bucket's contract is a plain
        # loss.backward() per step, which is what makes that safe
        from . import ops
        ops.enable_grad_sink(self.params)

    def zero(self):
        self.flat.zero_()

    @property
    def world(self):
        return dist.get_world_size(self.group) if dist.is_available() and dist.is_initialized() else 1

    def all_reduce_sum(self):
        """Sum over ranks, nothing else: the division by the world size rides in ``FlatAdam.step(grad_scale=1/world)``."""
        if self.world > 1 or (self.reduce_single_rank and dist.is_initialized()):
            dist.all_reduce(self.flat, op=dist.ReduceOp.SUM, group=self.group)

    def all_reduce_mean(self, async_op=False):
        """Sum over ranks then divide by world size (mean gradient of the global batch)."""
        w = self.world
        if w == 1 and not (self.reduce_single_rank and dist.is_initialized()):
            return None
        work = dist.all_reduce(self.flat, op=dist.ReduceOp.SUM, group=self.group, async_op=async_op)
        if async_op:
            return work
        self.flat.div_(w)
        return None

    def finish(self, work):
        if work is not None:
            work.wait()
            self.flat.div_(self.world)


class FlatAdam:
    """Adam (torch.optim.Adam's update, no amsgrad) on the flat parameter / gradient buffers of a
    ``FlatGradBucket(..., flatten_params=True)``: one kernel per step (csrc/flat_adam.hip), whatever the number of
    parameter tensors.  In-place updates of the flat buffer do not bump the per-parameter version counters,
    so the step also invalidates this package's cached low-precision weight copies."""

    def __init__(self, bucket, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.0):
        if bucket.flat_param is None:
            raise ValueError('FlatAdam needs FlatGradBucket(..., flatten_params=True)')
        self.bucket = bucket
        self.lr, self.betas, self.eps, self.weight_decay = lr, betas, eps, weight_decay
        self.exp_avg = torch.zeros_like(bucket.flat_param)
        self.exp_avg_sq = torch.zeros_like(bucket.flat_param)
        self.t = 0

    @torch.no_grad()
    def step(self, grad_scale=1.0):
        """One update.  On a ROCm device: ONE kernel over the four flat buffers (csrc/flat_adam.hip); ``grad_scale`` folds a
        pending division of the gradient (1 / world after an all-reduce SUM) into it.  On the CPU (tests): the same
        rule as elementwise torch ops."""
        p, g = self.bucket.flat_param, self.bucket.flat
        b1, b2 = self.betas
        self.t += 1
        if p.is_cuda:
            from . import _hip
            P = _hip._ptr
            _hip.check(_hip.lib().fgnn_flat_adam(P(p), P(g), P(self.exp_avg), P(self.exp_avg_sq), None, p.numel(),
                                                 float(self.lr), float(b1), float(b2), float(self.eps),
                                                 float(self.weight_decay), float(grad_scale), int(self.t),
                                                 _hip.stream_ptr()))
        else:
            if grad_scale != 1.0:
                g = g * grad_scale
            if self.weight_decay:
                g = g.add(p, alpha=self.weight_decay)
            self.exp_avg.lerp_(g, 1.0 - b1)
            self.exp_avg_sq.mul_(b2).addcmul_(g, g, value=1.0 - b2)
            bc1, bc2 = 1.0 - b1 ** self.t, 1.0 - b2 ** self.t
            denom = (self.exp_avg_sq.sqrt() / (bc2 ** 0.5)).add_(self.eps)
            p.addcdiv_(self.exp_avg, denom, value=-self.lr / bc1)
        from .mpnn import pointwise
        pointwise.invalidate_casts()
        pointwise.note_state_change()

    def zero_grad(self, set_to_none=False):
        self.bucket.zero()


def broadcast_parameters(module, src=0, group=None):
    """Make every replica start from rank ``src``'s parameters and buffers (one flat broadcast)."""
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size(group) == 1:
        return
    tensors = [t for t in list(module.parameters()) + list(module.buffers()) if t.is_floating_point()]
    flat = torch.cat([t.detach().reshape(-1).float() for t in tensors])
    dist.broadcast(flat, src=src, group=group)
    off = 0
    with torch.no_grad():
        for t in tensors:
            n = t.numel()
            t.copy_(flat[off:off + n].view_as(t))
            off += n


def shard_range(total, rank, world):
    """Contiguous [begin, end) of ``total`` samples owned by ``rank`` (sizes differ by at most 1)."""
    base, rem = divmod(total, world)
    begin = rank * base + min(rank, rem)
    return begin, begin + base + (1 if rank < rem else 0)
