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


FLAT_ALIGN = 4      # elements: every parameter starts on a 16-byte boundary of the flat buffers (the kernels read filters with vector loads)


def flat_layout(params, align=FLAT_ALIGN):
    """(offsets, total): where each of ``params`` starts in a flat buffer, each start rounded up to ``align`` elements — a one-element
    bias in the middle of a model would otherwise leave every later filter on an odd 4-byte boundary (the padding holds zeros: zero
    gradients, zero moments, a zero Adam update)."""
    offsets, off = [], 0
    for p in params:
        off = (off + align - 1) // align * align
        offsets.append(off)
        off += p.numel()
    return offsets, (off + align - 1) // align * align


def flatten_parameters(params):
    """Move float32 parameters into ONE flat buffer (every ``p.data`` becomes a view of it) and register it with the
    low-precision weight cache: the bf16 copies the library GEMMs read are then views of one mirror that a single cast
    kernel refreshes (pointwise.register_flat_parameters).  Returns the flat buffer."""
    params = list(params)
    if not params:
        raise ValueError('no parameters')
    if any(p.dtype != torch.float32 for p in params):
        raise ValueError('flatten_parameters needs float32 parameters')
    offsets, total = flat_layout(params)
    flat = torch.zeros(total, device=params[0].device, dtype=torch.float32)
    with torch.no_grad():
        for p, off in zip(params, offsets):
            view = flat[off:off + p.numel()].view_as(p)
            view.copy_(p.data)
            p.data = view
    from .mpnn import pointwise
    pointwise.register_flat_parameters(flat, params)
    return flat


class DirectAllReduce:
    """The flat bucket's all-reduce through RCCL's C API (``ncclAllReduce`` on torch's CURRENT stream, ctypes on librccl.so), with a
    communicator of its own — no ``torch.distributed`` work object, no watchdog thread polling an event: the form that can be
    RECORDED INTO the step's hipGraph with the optimizer behind it.  (Captured through ``torch.distributed`` the collective works
    most of the time, but ProcessGroupNCCL's watchdog polls the work's completion event, which was recorded on a capturing
    stream: hipErrorCapturedEvent, process abort, ~1 run in 6 — profiles/r04/README.md.)  The 128-byte unique id travels from rank 0
    through the existing process group (any backend: one ``broadcast_object_list``), then every rank calls ``ncclCommInitRank``
    on its current device.  xGMI note: one 5.5 MB in-place sum per step; nothing to bucket."""

    _NCCL_FLOAT32, _NCCL_SUM = 7, 0

    def __init__(self, group=None, library=None):
        import ctypes
        import os
        if not (dist.is_available() and dist.is_initialized()):
            raise RuntimeError('DirectAllReduce needs an initialised torch.distributed group to exchange the communicator id')
        self.rank, self.world = dist.get_rank(group), dist.get_world_size(group)
        path = library or os.environ.get('FGNN_RCCL_LIB')
        if path is None:
            # the RCCL torch itself links (same ROCm runtime objects), else the ROCm installation's
            cand = [os.path.join(os.path.dirname(torch.__file__), 'lib', 'librccl.so'), '/opt/rocm/lib/librccl.so', 'librccl.so']
            path = next((c for c in cand if c == 'librccl.so' or os.path.exists(c)), 'librccl.so')
        self._lib = L = ctypes.CDLL(path)

        class UniqueId(ctypes.Structure):
            _fields_ = [('internal', ctypes.c_char * 128)]
        L.ncclGetUniqueId.restype = ctypes.c_int
        L.ncclGetUniqueId.argtypes = [ctypes.POINTER(UniqueId)]
        L.ncclCommInitRank.restype = ctypes.c_int
        L.ncclCommInitRank.argtypes = [ctypes.POINTER(ctypes.c_void_p), ctypes.c_int, UniqueId, ctypes.c_int]
        L.ncclAllReduce.restype = ctypes.c_int
        L.ncclAllReduce.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_int, ctypes.c_int, ctypes.c_void_p,
                                    ctypes.c_void_p]
        L.ncclCommDestroy.restype = ctypes.c_int
        L.ncclCommDestroy.argtypes = [ctypes.c_void_p]
        L.ncclGetErrorString.restype = ctypes.c_char_p
        L.ncclGetErrorString.argtypes = [ctypes.c_int]
        uid = UniqueId()
        if self.rank == 0:
            self._check(L.ncclGetUniqueId(ctypes.byref(uid)), 'ncclGetUniqueId')
        # (a ctypes c_char array read as .value stops at the first NUL: take the raw 128 bytes)
        box = [ctypes.string_at(ctypes.addressof(uid), 128) if self.rank == 0 else None]
        dist.broadcast_object_list(box, src=0, group=group)
        ctypes.memmove(ctypes.addressof(uid), box[0], 128)
        self._comm = ctypes.c_void_p()
        self._check(L.ncclCommInitRank(ctypes.byref(self._comm), self.world, uid, self.rank), 'ncclCommInitRank')

    def _check(self, rc, what):
        if rc != 0:
            raise RuntimeError('%s failed: %s' % (what, self._lib.ncclGetErrorString(rc).decode()))

    def all_reduce_sum_(self, flat):
        """In-place sum of the contiguous f32 tensor ``flat`` over the ranks, enqueued on torch's current stream."""
        if not (flat.is_cuda and flat.dtype == torch.float32 and flat.is_contiguous()):
            raise ValueError('DirectAllReduce takes a contiguous float32 tensor on the device')
        st = torch.cuda.current_stream(flat.device).cuda_stream
        self._check(self._lib.ncclAllReduce(flat.data_ptr(), flat.data_ptr(), flat.numel(), self._NCCL_FLOAT32, self._NCCL_SUM,
                                            self._comm, st), 'ncclAllReduce')

    def close(self):
        if getattr(self, '_comm', None):
            self._lib.ncclCommDestroy(self._comm)
            self._comm = None


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
        self.offsets, self.numel = flat_layout(self.params)          # (numel: with the alignment padding)
        self.flat = torch.zeros(self.numel, device=dev, dtype=dt)
        self.flat_param = flatten_parameters(self.params) if flatten_params else None
        for p, off in zip(self.params, self.offsets):
            p.grad = self.flat[off:off + p.numel()].view_as(p)
        # the gradient kernels may add straight into these slices (ops.grad_sink): the bucket's contract is a plain
        # loss.backward() per step, which is what makes that safe
        from . import ops
        ops.enable_grad_sink(self.params)

    def zero(self):
        # weight-gradient launches a backward pass parked and never issued (it raised half-way, ops.defer_wgrad) add into this
        # buffer: issue them BEFORE the zeroing, so that they cannot land in the next step's sums
        from . import ops
        if any(ops._DEFERRED.values()):
            ops.flush_deferred()
        # flush_deferred issues each launch on ITS stream; the zeroing below runs on the current one: join them first (the
        # engine's end-of-pass join never ran for a pass that died), or a parked kernel could still add behind the memset
        if ops._DEFER_ISSUED and self.flat.is_cuda:
            cur = torch.cuda.current_stream(self.flat.device)
            for st in ops._DEFER_ISSUED:
                if st != cur:
                    cur.wait_stream(st)
            ops._DEFER_ISSUED.clear()
        if self.flat.is_cuda:    # (a CPU / gloo bucket never recorded a fold and must not need libfgnn_hip.so at all)
            ops.flush_folds()    # folds a dead pass recorded and never flushed: in front of the zeroing, like its parked launches
        self.flat.zero_()

    @property
    def world(self):
        return dist.get_world_size(self.group) if dist.is_available() and dist.is_initialized() else 1

    def use_direct_all_reduce(self, library=None):
        """Route ``all_reduce_sum`` through RCCL's C API (``DirectAllReduce``): the collective can then be captured into the step's
        hipGraph.  Collective: every rank of the group calls this once, outside any capture."""
        self.direct = DirectAllReduce(self.group, library)
        return self.direct

    def all_reduce_sum(self):
        """Sum over ranks, nothing else: the division by the world size rides in ``FlatAdam.step(grad_scale=1/world)``."""
        if self.world > 1 or (self.reduce_single_rank and dist.is_initialized()):
            if getattr(self, 'direct', None) is not None:
                self.direct.all_reduce_sum_(self.flat)
            else:
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

    def __init__(self, bucket, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.0, capturable=False):
        """``capturable=True`` (ROCm device only): the step count and the learning rate live in device memory
        (``fgnn_flat_adam_dev``), so ``step`` has no step-dependent launch argument and can be recorded into a hipGraph —
        with the gradient all-reduce in front of it — and replayed.  Assigning ``opt.lr`` between replays still works (it
        writes the device scalar); ``opt.t`` reads the device counter back (a synchronisation)."""
        if bucket.flat_param is None:
            raise ValueError('FlatAdam needs FlatGradBucket(..., flatten_params=True)')
        self.bucket = bucket
        self.capturable = bool(capturable)
        if self.capturable:
            if not bucket.flat_param.is_cuda:
                raise ValueError('FlatAdam(capturable=True) needs parameters on a ROCm device')
            dev = bucket.flat_param.device
            self._lr_dev = torch.full((1,), float(lr), device=dev, dtype=torch.float32)
            self._step_dev = torch.zeros(1, device=dev, dtype=torch.int64)
            self._coef_dev = torch.zeros(2, device=dev, dtype=torch.float32)
        self._lr = float(lr)
        self.betas, self.eps, self.weight_decay = betas, eps, weight_decay
        self.exp_avg = torch.zeros_like(bucket.flat_param)
        self.exp_avg_sq = torch.zeros_like(bucket.flat_param)
        self._t = 0

    @property
    def lr(self):
        return self._lr

    @lr.setter
    def lr(self, value):
        value = float(value)
        if self.capturable and value != self._lr:
            self._lr_dev.fill_(value)            # a device write on the current stream: ordered before the next replay
        self._lr = value

    @property
    def t(self):
        return int(self._step_dev.item()) if self.capturable else self._t

    @t.setter
    def t(self, value):
        if self.capturable:
            self._step_dev.fill_(int(value))
        self._t = int(value)

    @torch.no_grad()
    def step(self, grad_scale=1.0):
        """One update.  On a ROCm device: ONE kernel over the four flat buffers (csrc/flat_adam.hip); ``grad_scale`` folds a
        pending division of the gradient (1 / world after an all-reduce SUM) into it.  On the CPU (tests): the same
        rule as elementwise torch ops."""
        p, g = self.bucket.flat_param, self.bucket.flat
        b1, b2 = self.betas
        if self.capturable:
            from . import _hip
            P = _hip._ptr
            _hip.check(_hip.lib().fgnn_flat_adam_dev(P(p), P(g), P(self.exp_avg), P(self.exp_avg_sq), None, p.numel(),
                                                     P(self._lr_dev), float(b1), float(b2), float(self.eps),
                                                     float(self.weight_decay), float(grad_scale), P(self._step_dev),
                                                     P(self._coef_dev), _hip.stream_ptr()))
            from .mpnn import pointwise
            pointwise.invalidate_casts()
            pointwise.note_state_change()
            return
        self._t += 1
        if p.is_cuda:
            from . import _hip
            P = _hip._ptr
            _hip.check(_hip.lib().fgnn_flat_adam(P(p), P(g), P(self.exp_avg), P(self.exp_avg_sq), None, p.numel(),
                                                 float(self.lr), float(b1), float(b2), float(self.eps),
                                                 float(self.weight_decay), float(grad_scale), int(self._t),
                                                 _hip.stream_ptr()))
        else:
            if grad_scale != 1.0:
                g = g * grad_scale
            if self.weight_decay:
                g = g.add(p, alpha=self.weight_decay)
            self.exp_avg.lerp_(g, 1.0 - b1)
            self.exp_avg_sq.mul_(b2).addcmul_(g, g, value=1.0 - b2)
            bc1, bc2 = 1.0 - b1 ** self._t, 1.0 - b2 ** self._t
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


def _cpulist(text):
    """'0-3,8,10-11' -> [0, 1, 2, 3, 8, 10, 11] (the kernel's cpulist format)."""
    out = []
    for part in text.strip().split(','):
        if not part:
            continue
        lo, _, hi = part.partition('-')
        out.extend(range(int(lo), int(hi or lo) + 1))
    return out


def device_locality(index):
    """Where device ``index`` sits: PCI address, NUMA node and that node's CPUs, read from sysfs (-1 / [] when the platform
    does not say — a VM without NUMA information)."""
    import os
    pr = torch.cuda.get_device_properties(index)
    bdf = '%04x:%02x:%02x.0' % (pr.pci_domain_id, pr.pci_bus_id, pr.pci_device_id)
    node, cpus = -1, []
    base = '/sys/bus/pci/devices/' + bdf
    try:
        node = int(open(base + '/numa_node').read())
        cpus = _cpulist(open(base + '/local_cpulist').read())
    except (OSError, ValueError):
        pass
    if node >= 0 and not cpus:
        try:
            cpus = _cpulist(open('/sys/devices/system/node/node%d/cpulist' % node).read())
        except OSError:
            cpus = []
    return {'pci': bdf, 'numa_node': node, 'local_cpus': cpus}


def plan_rank_cpus(allowed, local_cpus, slot, slots):
    """The cores rank number ``slot`` of the ``slots`` ranks SHARING one locality domain should run on: an equal contiguous
    share of the domain's allowed cores (``local_cpus`` ∩ ``allowed``; all of ``allowed`` when the platform names no domain).
    Never empty while ``allowed`` is not: with fewer cores than ranks the ranks share them round-robin."""
    allowed = sorted(allowed)
    pool = [c for c in allowed if c in set(local_cpus)] or allowed
    if not pool:
        return []
    if len(pool) < slots:
        return [pool[slot % len(pool)]]
    b, e = shard_range(len(pool), slot, slots)
    return pool[b:e]


def bind_rank_to_local_cores(device_index, local_rank, local_world, set_threads=True):
    """Pin this process (one rank = one GPU) to its share of the cores NUMA-local to its device, and size torch's intra-op
    pool to that share: eight ranks launched by ``torch.distributed.run`` otherwise all float over every core with
    ``OMP_NUM_THREADS`` unset or 1 and the host side of a replayed step (graph launch, all-reduce enqueue) migrates between
    sockets.  The ranks that share a NUMA domain are taken to be the ones whose devices report the same node; without NUMA
    information the allowed cores are split evenly by local rank.  Returns what was done (for the bench line)."""
    import os
    loc = device_locality(device_index)
    if not hasattr(os, 'sched_setaffinity'):
        return dict(loc, bound_cpus=None, note='no sched_setaffinity on this platform')
    allowed = sorted(os.sched_getaffinity(0))
    slot, slots = local_rank, max(1, local_world)
    # ranks on the same node: those local ranks whose device reports this node.  That needs the device of every local rank: known
    # only under the launchers' convention "local rank i drives device i" — with a device override (FGNN_BENCH_DEVICE) or one visible
    # device per rank the identity does not hold, and the node's cores are then split evenly by local rank instead
    identity = (device_index == local_rank and torch.cuda.device_count() >= local_world
                and os.environ.get('FGNN_BENCH_DEVICE') is None)
    if identity and loc['numa_node'] >= 0 and loc['local_cpus']:
        same = []
        for r in range(min(local_world, torch.cuda.device_count())):
            try:
                if device_locality(r)['numa_node'] == loc['numa_node']:
                    same.append(r)
            except Exception:       # noqa: BLE001
                pass
        if local_rank in same:
            slot, slots = same.index(local_rank), len(same)
    cpus = plan_rank_cpus(allowed, loc['local_cpus'], slot, slots)
    note = None
    try:
        if cpus:
            os.sched_setaffinity(0, cpus)
            if set_threads and 'OMP_NUM_THREADS' not in os.environ:
                torch.set_num_threads(max(1, len(cpus)))
    except OSError as e:
        note = 'sched_setaffinity failed: %s' % e
        cpus = None
    return dict(loc, local_cpus='%d cores' % len(loc['local_cpus']), bound_cpus=_ranges(cpus) if cpus else None,
                threads=torch.get_num_threads(), note=note)


def _ranges(cpus):
    """[0, 1, 2, 3, 8] -> '0-3,8'."""
    out, run = [], []
    for c in sorted(cpus) + [None]:
        if run and (c is None or c != run[-1] + 1):
            out.append('%d-%d' % (run[0], run[-1]) if len(run) > 1 else '%d' % run[0])
            run = []
        if c is not None:
            run.append(c)
    return ','.join(out)


def xgmi_topology():
    """What the KFD topology says about this node's GPU links (sysfs, no tool needed): per GPU node the number of xGMI
    (io_link type 11) and PCIe (type 2) links.  {} when the files are absent (container without /sys/class/kfd)."""
    import glob
    import os
    out = {}
    for nd in sorted(glob.glob('/sys/class/kfd/kfd/topology/nodes/*'), key=lambda p: int(os.path.basename(p))):
        try:
            props = dict(l.split()[:2] for l in open(nd + '/properties').read().splitlines() if len(l.split()) >= 2)
            if int(props.get('simd_count', '0')) == 0:
                continue                                   # a CPU node
            kinds = {}
            for lk in glob.glob(nd + '/io_links/*/properties'):
                lp = dict(l.split()[:2] for l in open(lk).read().splitlines() if len(l.split()) >= 2)
                kinds[lp.get('type', '?')] = kinds.get(lp.get('type', '?'), 0) + 1
            out[os.path.basename(nd)] = {'xgmi_links': kinds.get('11', 0), 'pcie_links': kinds.get('2', 0)}
        except (OSError, ValueError):
            continue
    return out


def shard_range(total, rank, world):
    """Contiguous [begin, end) of ``total`` samples owned by ``rank`` (sizes differ by at most 1)."""
    base, rem = divmod(total, world)
    begin = rank * base + min(rank, rem)
    return begin, begin + base + (1 if rank < rem else 0)
