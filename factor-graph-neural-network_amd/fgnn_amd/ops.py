"""Host-side entry points of the fused VF/FV message operator (HIP only).

``mpconv(...)`` is what ``mp_conv_v2.forward`` calls.  It computes

    z[b,o,m] = agg_j  sum_e etype[b,e,m,j] * msg[b,m,j,o,e]  + bias[o]

(optionally followed, when no gradient is needed, by a folded per-channel affine and
ReLU) in one kernel launch through the C ABI, and differentiates through a hand-written
backward kernel.  Reference semantics: /root/reference/lib/model/mpnn/mp_nn.py:115-175.
"""
import ctypes
import os

import torch

from . import _hip


class KernelTimer:
    """Optional per-launch timing with events recorded on the stream the kernels are launched on
    (torch's current stream).  bench.py installs one for a single instrumented step to obtain the
    average launch duration of each kernel symbol next to its algorithmic bytes."""

    def __init__(self):
        self.records = []

    def run(self, symbol, nbytes, nflops, launch, use_note=True):
        start, end = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        start.record()
        launch()
        end.record()
        # message-operator calls: the kernel the C dispatch chose; other entry points: the symbol given
        name = (_hip.lib().fgnn_last_kernel().decode() or symbol) if use_note else symbol
        self.records.append((name, nbytes, nflops, start, end))

    def summary(self):
        torch.cuda.synchronize()
        out = {}
        for symbol, nbytes, nflops, start, end in self.records:
            r = out.setdefault(symbol, {'launches': 0, 'ms': 0.0, 'bytes': 0, 'flops': 0})
            r['launches'] += 1
            r['ms'] += start.elapsed_time(end)
            r['bytes'] += nbytes
            r['flops'] += nflops
        return out


TIMER = None     # set to a KernelTimer to time every launch (bench.py only)

_AGG_NAME = {0: 'max', 1: 'lse', 2: 'mean'}


def _symbol(kind, d):
    net = d.net if d.net in (1, 4, 16) else 0
    return 'mpconv_%s_kernel<%s, %d, %s>' % (kind, 'bf16' if d.dtype else 'float', net,
                                             _AGG_NAME[d.agg])


def _flops(d, passes):
    """SURVEY §8d algorithmic FLOPs: node-level projection(s) + per-edge type contraction."""
    R = d.nin if d.ext == _hip.EXT_NONE else 2 * d.nin
    return passes * (2 * d.B * d.N * R * d.nou * d.net + 2 * d.B * d.M * d.k * d.nou * d.net)


def _launch(kind, d, nbytes, fn):
    if TIMER is None:
        fn()
    else:
        TIMER.run(_symbol(kind, d), nbytes, _flops(d, 3 if kind == 'bwd' else 1), fn)


def timed(symbol, nbytes, fn, nflops=0):
    """Run one C-ABI call of a non-operator kernel (node-wise maps, norms, sums); with bench.py's KernelTimer
    installed, also record its duration against its algorithmic bytes."""
    if TIMER is None:
        fn()
    else:
        TIMER.run(symbol, nbytes, nflops, fn, use_note=False)


# ---- deferred weight-gradient kernels -------------------------------------------------------------------------------------
# Nothing in a backward pass READS a weight gradient: the kernels that produce them (2.6 ms of a 18.8 ms LDPC step) only have to be
# done before the optimizer.  FactorNN's two streams spend the backward waiting for each other at every layer's joins (the
# main stream 3.5 ms, the side stream 7 ms of a step: profiles/r03/train_step_timeline.txt), so a weight-gradient kernel that sits
# in stream order IN FRONT of a kernel the other stream waits for is on the critical path for nothing.  With DEFER_WGRAD the
# launches are parked per stream and issued when the autograd engine moves on to a node of ANOTHER stream — i.e. right behind the
# stream's last critical kernel, into the slot where it would otherwise idle until the join — and at the latest when the
# backward pass ends (an engine callback, which also joins every such stream into the caller's).  Only gradients that go to
# a sink (ops.grad_sink: the flat bucket) are deferred: a gradient tensor handed back to autograd must be complete in stream order.
DEFER_WGRAD = True          # (module switch: tests compare parked against inline launches)
# (round 5 also tried the parked launches on a THIRD stream and on the side stream, each behind an event on its operands: 15.5 / 15.1
#  against 13.8 ms — every kernel that overlaps an operator launch waits for a CU; profiles/r05/README.md.  Removed.)
SIDE_ACTIVE = False         # set by the assemblies the first time a forward actually forks onto the side stream: with ONE stream in
                            # play parking buys no overlap and only keeps every layer's operands alive until the end of the backward
_DEFERRED = {}              # stream -> [(launch closure, operands)]
_DEFER_CALLBACK = [None]     # id of the backward pass (autograd graph task) whose end-of-pass callback is queued
_DEFER_ISSUED = set()       # streams that got parked launches issued during the current backward pass


def defer_wgrad(launch, operands=()):
    """Park `launch` (a closure that enqueues one weight-gradient kernel on the CURRENT stream).  `operands`: the tensors the
    kernel reads.  They are kept alive until the kernel is issued and then marked as in use by its stream: an activation the
    OTHER stream allocated is otherwise handed back to that stream's allocator the moment the last reference drops — the
    engine's join with this stream happened before the parked kernel went out, so nothing else orders the reuse behind it
    (found as three weight gradients of hyper-factor maps reading overwritten rows on hipGraph replay)."""
    if not DEFER_WGRAD or not SIDE_ACTIVE:
        launch()
        return
    st = torch.cuda.current_stream()
    task = torch._C._current_graph_task_id()           # (-1 outside a backward pass: then nothing would ever issue the launch)
    if task < 0:
        launch()
        return
    _register_backward_callback(task)
    _DEFERRED.setdefault(st, []).append((launch, tuple(operands)))


def _register_backward_callback(task):
    if _DEFER_CALLBACK[0] != task:
        # Another backward pass than the one that parked what is in the lists: a NESTED (re-entrant) pass inside it — checkpointing,
        # a custom Function calling backward() — or a pass that died half-way.  Either way the parked launches are ISSUED, never
        # dropped (they accumulate into sinked .grad buffers behind autograd's back: dropping them would lose gradients silently;
        # issuing those of a dead pass only finishes an accumulation its owner discards).  The outer pass re-registers itself
        # with its next parked launch; its own end-of-pass callback is still queued.
        if any(_DEFERRED.values()):
            flush_deferred()
        flush_folds()
        _DEFER_CALLBACK[0] = task
        torch.autograd.Variable._execution_engine.queue_callback(_flush_at_end_of_backward)


# ---- recorded parameter-gradient folds -------------------------------------------------------------------------------------------
# Every weight / filter gradient kernel leaves per-workgroup slabs that a small launch folds into the accumulator: ~100 launches of
# 5-15 us per LDPC step, a third of them in the main stream's dependent chain (csrc/fold_batch.hip).  When the gradient goes to a
# sink (ops.grad_sink: nothing in the pass reads it) the fold is RECORDED instead — the call gets a slab buffer of its own, kept
# alive here — and ONE launch at the end of the pass folds them all (a fixed summation order of its own: reproducible, equal to the
# immediate folds' sums to f32 rounding).
DEFER_FOLDS = True          # (module switch: tests compare recorded against immediate folds)
_FOLD_KEEP = []


def folds_deferrable():
    """True inside a backward pass whose end this module gets to see (the engine callback that flushes the recorded folds)."""
    if not DEFER_FOLDS:
        return False
    task = torch._C._current_graph_task_id()
    if task < 0:
        return False
    _register_backward_callback(task)
    return True


class fold_scope:
    """``with fold_scope(defer):`` — the gradient entry points called inside record their slab folds (csrc/fold_batch.hip) when
    ``defer``; ``slabs(device, nbytes)`` is the workspace to hand them: a private buffer (alive until the flush) then, else the
    stream's shared scratch."""

    def __init__(self, defer):
        self.defer = bool(defer)

    def slabs(self, device, nbytes):
        if not self.defer:
            return _workspace(device, nbytes)
        t = torch.empty((nbytes + 3) // 4, device=device, dtype=torch.float32)
        _FOLD_KEEP.append(t)
        return t

    def __enter__(self):
        if self.defer:
            _hip.lib().fgnn_fold_defer(1)
        return self

    def __exit__(self, *exc):
        if self.defer:
            _hip.lib().fgnn_fold_defer(0)
        return False


def flush_folds():
    """Fold everything recorded, on the current stream (the caller has ordered it behind every producer)."""
    L = _hip.lib()
    if L.fgnn_fold_pending():
        _hip.check(L.fgnn_fold_flush(_hip.stream_ptr()))
        cur = torch.cuda.current_stream()
        for t in _FOLD_KEEP:            # slabs another stream allocated are read by this stream's launch: not that stream's to reuse yet
            t.record_stream(cur)
    _FOLD_KEEP.clear()


def flush_deferred(except_stream=None):
    """Issue the parked launches of every stream but `except_stream`, each on its own stream."""
    for st, lst in _DEFERRED.items():
        if not lst or st == except_stream:
            continue
        with torch.cuda.stream(st):
            for fn, operands in lst:
                fn()
                for t in operands:
                    t.record_stream(st)
        lst.clear()
        _DEFER_ISSUED.add(st)


def _flush_at_end_of_backward():
    _DEFER_CALLBACK[0] = None
    flush_deferred()
    cur = torch.cuda.current_stream()
    for st in _DEFER_ISSUED:    # the engine joined its streams BEFORE this callback: what was issued since needs its own join
        if st != cur:
            cur.wait_stream(st)
    flush_folds()               # every producer is now in front of the current stream: one launch folds all their slabs
    _DEFER_ISSUED.clear()


def backward_node_begins():
    """Called at the top of every hand-written backward: the engine has moved to a node on the current stream, so the other
    streams' parked weight-gradient launches go out now (behind their last critical kernel)."""
    if STAMPS is not None:
        import sys
        f = sys._getframe(1)
        cur = torch.cuda.current_stream()
        stamp('bwd %s %s:%d' % ('side' if (SIDE_STREAM and cur == side_stream(cur.device)) else 'main',
                                f.f_code.co_filename.rsplit('/', 1)[-1], f.f_lineno))
    if _DEFER_CALLBACK[0] is not None and _DEFER_CALLBACK[0] == torch._C._current_graph_task_id():
        flush_deferred(except_stream=torch.cuda.current_stream())


TAIL_TO_SIDE = True         # see backward_tail_begins (module switch for the A/B in tools/ and the tests)


def backward_tail_begins():
    """Called by a backward node behind which the pass is ONE dependent chain on the current stream — the edge-type MLPs' backward,
    which needs the edge-weight gradients of all eight layers (/root/reference/train_ldpc.py:68-69: `emodel_*` feed every layer).
    What is parked for the end of the pass — this stream's weight-gradient launches and every recorded parameter-gradient fold —
    used to run BEHIND that chain, alone on the chip (0.3 ms of a 13 ms step: profiles/r06/README.md); it goes to the side stream
    NOW, behind an event on this stream, and runs beside the chain.  The end-of-pass callback joins the side stream as before."""
    if not (TAIL_TO_SIDE and SIDE_ACTIVE and _DEFER_CALLBACK[0] is not None
            and _DEFER_CALLBACK[0] == torch._C._current_graph_task_id()):
        return
    cur = torch.cuda.current_stream()
    side = side_stream(cur.device)
    if side == cur:
        return
    flush_deferred(except_stream=cur)            # (the other streams' parked launches go out on their own streams, as at any node)
    mine = _DEFERRED.get(cur) or []
    if not mine and not _hip.lib().fgnn_fold_pending():
        return
    ready = torch.cuda.Event()
    ready.record(cur)
    with torch.cuda.stream(side):
        side.wait_event(ready)
        for fn, operands in mine:
            fn()
            for t in operands:
                t.record_stream(side)
        mine.clear()
        flush_folds()       # every producer recorded so far is in front of `ready` on this stream or earlier on the side stream
    _DEFER_ISSUED.add(side)


def _require_device(*tensors):
    for t in tensors:
        if t is not None and not t.is_cuda:
            raise _hip.FgnnHipError(
                'fgnn_amd operators run on a ROCm device only (got a %s tensor); '
                'there is no CPU fallback — move the module and its inputs to cuda.' % t.device)


def _channels_last_like(x):
    """True when x [B,C,N,1] is laid out channel-fastest."""
    return x.shape[1] > 1 and x.stride(1) == 1 and (x.shape[2] == 1 or x.stride(2) != 1)


def _alloc_out(x, nou, M, dtype=None):
    B = x.shape[0]
    dtype = x.dtype if dtype is None else dtype
    if _channels_last_like(x):
        # channel-fastest storage as a tensor of its OWN (not a permuted view: an in-place activation on the operator's output
        # would otherwise be an in-place write to a view made inside the autograd Function, which autograd refuses)
        return torch.empty_strided((B, nou, M, 1), (M * nou, 1, nou, nou), device=x.device, dtype=dtype)
    return torch.empty((B, nou, M, 1), device=x.device, dtype=dtype)


def _check_shapes(x, nn_idx, etype, filters, nou, net, ext):
    if x.dim() != 4 or x.shape[3] != 1:
        raise ValueError('x must be [B, nin, N, 1], got %s' % (tuple(x.shape),))
    if nn_idx.dim() != 3 or nn_idx.dtype != torch.int64:
        raise ValueError('nn_idx must be int64 [B, M, k], got %s %s' % (nn_idx.dtype, tuple(nn_idx.shape)))
    B = nn_idx.shape[0]
    assert B == x.shape[0]          # same assertion as mp_nn.py:100
    if etype.shape != (B, net, nn_idx.shape[1], nn_idx.shape[2]):
        raise ValueError('etype must be [B, net, M, k] = %s, got %s' %
                         ((B, net, nn_idx.shape[1], nn_idx.shape[2]), tuple(etype.shape)))
    R = x.shape[1] if ext == _hip.EXT_NONE else 2 * x.shape[1]
    if filters.shape != (R, nou * net):
        raise ValueError('filters must be [%d, %d], got %s' % (R, nou * net, tuple(filters.shape)))
    if etype.dtype != x.dtype:
        raise ValueError('x and etype must share a dtype (%s vs %s)' % (x.dtype, etype.dtype))
    if CHECK_INDICES:
        _check_index_range(nn_idx, x.shape[2])


CHECK_INDICES = bool(int(os.environ.get('FGNN_CHECK_INDICES', '0')))


def _check_index_range(nn_idx, N):
    """Debug aid (FGNN_CHECK_INDICES=1): the kernels CLAMP neighbour ids into [0, N) so that a bad table can never
    fault; the reference's ``torch.gather`` raises instead (CPU) or device-asserts (CUDA).  With the check on, an
    out-of-range table raises IndexError here — one device reduction + host sync per distinct table, remembered on the
    tensor that owns the memory (same scheme as blocks._is_identity_list)."""
    owner = nn_idx._base if nn_idx._base is not None else nn_idx
    key = (nn_idx._version, nn_idx.storage_offset(), tuple(nn_idx.shape), tuple(nn_idx.stride()), N)
    memo = getattr(owner, '_fgnn_index_range', None)
    if memo is None or memo[0] != key:
        lo, hi = (int(v) for v in torch.aminmax(nn_idx)) if nn_idx.numel() else (0, 0)
        memo = (key, lo, hi)
        owner._fgnn_index_range = memo
    if memo[1] < 0 or memo[2] >= N:
        raise IndexError('nn_idx holds neighbour ids in [%d, %d] but x has %d nodes' % (memo[1], memo[2], N))


DEDUPE_GRAPHS = True     # recognise batch-identical neighbour tables passed as B copies (the reference's calling convention)


class Verdicts:
    """The host-side verdicts of one forward (is this table shared by the batch?  an identity list?  its in-degree?  are these
    edge weights equal over the nodes?) in call order, so that a hipGraph capture of the SAME forward can take the fast paths for
    tensors it meets for the first time — tensors the model builds inside its forward (the reference's
    ``self.hnn_idx_f2v.repeat(bsize, 1, 1)``, /root/reference/train_ldpc.py:77-84) — where no host read is possible.

    ``with Verdicts.recording() as v:`` around an EAGER run notes every device check + host read; ``with v.replaying():`` around the
    capture hands them back in the same order, each only to a tensor of the same site, shape, strides and dtype (anything else:
    the conservative answer, as before).  Sound when the capture runs the same module on the same inputs in the same state — the
    tensors checked are functions of the integer inputs and of frozen parameters only; fastpath.GraphedForward re-validates both
    before every replay."""
    current = None

    def __init__(self):
        self.fifo = {}
        self.mode = None
        self.taken = 0          # verdicts handed to a capture

    @staticmethod
    def _sig(t):
        return (tuple(t.shape), tuple(t.stride()), t.dtype)

    @classmethod
    def note(cls, site, t, value):
        v = cls.current
        if v is not None and v.mode == 'record':
            v.fifo.setdefault(site, []).append((cls._sig(t), value))
        return value

    @classmethod
    def recall(cls, site, t):
        """The recorded verdict for the next check at ``site`` (None: nothing recorded / another tensor geometry)."""
        v = cls.current
        if v is None or v.mode != 'replay':
            return None
        q = v.fifo.get(site)
        if not q or q[0][0] != cls._sig(t):
            return None
        v.taken += 1
        return q.pop(0)[1]

    class _Mode:
        def __init__(self, v, mode):
            self.v, self.mode = v, mode

        def __enter__(self):
            self.prev = Verdicts.current
            self.v.mode = self.mode
            Verdicts.current = self.v
            return self.v

        def __exit__(self, *exc):
            Verdicts.current = self.prev
            self.v.mode = None
            return False

    @classmethod
    def recording(cls):
        return cls._Mode(cls(), 'record')

    def replaying(self):
        return Verdicts._Mode(self, 'replay')


def shared_graph_view(nn_idx):
    """The reference's scripts hand the operator one neighbour table PER SAMPLE even though every sample of a batch
    shares one graph (`.repeat(B,1,1)` in train_syn_*.py:264-293; DataLoader-collated copies of the same alist tables in
    train_ldpc.py:216).  The kernels have a fast path for a table shared by the batch (batch stride 0: the incidence and
    its CSR transpose are built once per workgroup instead of once per sample).  This returns a stride-0 view of sample
    0's table when all B copies are equal — one device comparison + one host read per distinct table, remembered on the
    tensor that owns the memory (keyed by version and view geometry) — and ``nn_idx`` itself otherwise.  While a hipGraph
    is being captured no host read is possible: an unseen table is then taken as per-sample (correct, slower)."""
    B = nn_idx.shape[0]
    if not DEDUPE_GRAPHS or B <= 1 or nn_idx.stride(0) == 0 or not nn_idx.is_cuda:
        return nn_idx
    owner = nn_idx._base if nn_idx._base is not None else nn_idx
    key = (nn_idx._version, nn_idx.data_ptr(), tuple(nn_idx.shape), tuple(nn_idx.stride()))
    memo = getattr(owner, '_fgnn_shared_graph', None)
    if memo is None or memo[0] != key:
        if torch.cuda.is_current_stream_capturing():
            v = Verdicts.recall('shared_graph', nn_idx)
            if v is None:
                return nn_idx
            memo = (key, v)
        else:
            memo = (key, Verdicts.note('shared_graph', nn_idx, bool((nn_idx == nn_idx[:1]).all().item())))
        owner._fgnn_shared_graph = memo
    return nn_idx[:1].expand(B, -1, -1) if memo[1] else nn_idx


def is_identity_list(nn_idx):
    """nn_idx [B, 1, k] lists nodes 0..k-1 in order for every sample (the hyper-factor's neighbour table).  Checked on the device
    once per table; the verdict is remembered ON the tensor that owns the memory (the view's base, e.g. LDPCModel's frozen
    `hnn_idx_v2f` behind its per-call `expand`), keyed by version and view geometry, so it can never outlive or be confused with
    another table.  While a hipGraph is being captured no host read is possible: an unseen table is then taken as general."""
    owner = nn_idx._base if nn_idx._base is not None else nn_idx
    key = (nn_idx._version, nn_idx.storage_offset(), tuple(nn_idx.shape), tuple(nn_idx.stride()))
    memo = getattr(owner, '_fgnn_identity_list', None)
    if memo is None or memo[0] != key:
        if nn_idx.is_cuda and torch.cuda.is_current_stream_capturing():
            hit = Verdicts.recall('identity_list', nn_idx)
            if hit is None:
                return False
        else:
            k = nn_idx.shape[-1]
            hit = Verdicts.note('identity_list', nn_idx, bool(
                (nn_idx == torch.arange(k, device=nn_idx.device, dtype=nn_idx.dtype)).all().item()))
        memo = (key, hit)
        owner._fgnn_identity_list = memo
    return memo[1]


def max_in_degree(nn_idx, N):
    """Largest number of times one source node appears in a batch-SHARED neighbour table (the degree of the transposed
    incidence), 0 when unknown.  The second-generation backward (csrc/mpconv_bwd_sg.hip) keeps every source node's
    in-edges in registers and needs this bound up front (LDPC 96.3.963: 3 for the variables, 6 for the checks); it is
    measured once per table — one bincount + host read, remembered on the tensor that owns the memory — and travels to the
    C ABI in ``fgnn_mpconv_desc.reserved``.  Unknown (a per-sample table, or a first sight during hipGraph capture) sends
    the call to the first-generation kernels."""
    if nn_idx.shape[0] > 1 and nn_idx.stride(0) != 0:
        return 0
    owner = nn_idx._base if nn_idx._base is not None else nn_idx
    key = (nn_idx._version, nn_idx.data_ptr(), tuple(nn_idx.shape), tuple(nn_idx.stride()), N)
    memo = getattr(owner, '_fgnn_in_degree', None)
    if memo is None or memo[0] != key:
        if torch.cuda.is_current_stream_capturing():
            deg = Verdicts.recall('in_degree', nn_idx)
            if deg is None:
                return 0
            memo = (key, deg)
        else:
            flat = nn_idx[0].reshape(-1).clamp(0, N - 1)
            memo = (key, Verdicts.note('in_degree', nn_idx, int(torch.bincount(flat, minlength=N).max().item()) if flat.numel() else 0))
        owner._fgnn_in_degree = memo
    return memo[1]


# Off by default: measured on MI355X (gpurun_out/r05c, 4096 codewords) the pre-built tables change nothing — stand-alone 90.3 / 79.9 us
# with them against 87.8 / 81.5 us without (V->F / F->V 64 -> 64), the training step 15.49 against 15.46 ms: the ~17 k cycles a
# workgroup spends building its tables (profiles/r04) overlap the first samples' LDS-DMA and the staging of W, they are not on the
# launch's critical path.  ops.BACKWARD_TABLES = True turns them on (identical results).
BACKWARD_TABLES = False     # (module switch: the GPU suite runs the table-driven entry point both ways)


def backward_tables(nn_idx, d):
    """The per-graph tables of the table-driven backward (include/fgnn_hip.h: fgnn_mpconv_backward_tables) for the batch-SHARED
    neighbour table ``nn_idx`` and the descriptor ``d`` of the backward call, or None (another kernel family, a per-sample table, an
    unknown in-degree).  Built once per table — one small launch, remembered on the tensor that owns the memory next to its
    in-degree (``max_in_degree``), keyed by version, view geometry and (N, M, k) — so a training run, whose graph never changes, pays
    for the transposed incidence once instead of in every workgroup of every backward launch."""
    if not BACKWARD_TABLES or (d.reserved & 0xffff) == 0 or (nn_idx.shape[0] > 1 and nn_idx.stride(0) != 0):
        return None
    L = _hip.lib()
    nbytes = int(L.fgnn_mpconv_backward_tables_bytes(ctypes.byref(d)))
    if nbytes == 0:
        return None
    owner = nn_idx._base if nn_idx._base is not None else nn_idx
    key = (nn_idx._version, nn_idx.data_ptr(), tuple(nn_idx.shape), tuple(nn_idx.stride()), d.N, d.M, d.k, d.reserved & 0xffff)
    memo = getattr(owner, '_fgnn_bwd_tables', None)
    if memo is None or memo[0] != key:
        if torch.cuda.is_current_stream_capturing():
            return None          # (a first sight during capture: the launch would be recorded and its buffer owned by the graph's pool)
        t = torch.empty(nbytes // 4, device=nn_idx.device, dtype=torch.int32)
        rc = L.fgnn_mpconv_backward_tables(ctypes.byref(d), _hip._ptr(nn_idx), _hip._ptr(t), _hip.stream_ptr())
        if rc == _hip.EUNSUPPORTED:
            t = None             # (an in-degree beyond the table-driven kernel's slots: the call goes to another kernel anyway)
        else:
            _hip.check(rc)
            torch.cuda.current_stream(nn_idx.device).synchronize()      # other streams will read it
        memo = (key, t)
        owner._fgnn_bwd_tables = memo
    return memo[1]


def mpconv_forward_raw(x, nn_idx, etype, filters, bias, nou, net, ext, agg, *,
                       post_scale=None, post_shift=None, relu=False, want_argmax=False, bn=None, addends=None):
    """One launch of fgnn_mpconv_forward.  Returns (y, argmax-or-None).  ``bn`` (a ``pointwise.bn_spec`` tuple: the training-mode
    BatchNorm behind the operator): where the shape has a statistics epilogue, the launch also forms the batch statistics of y,
    finalises that BatchNorm in its last workgroup and announces the result to it (pointwise.set_pending_stats)."""
    _require_device(x, nn_idx, etype, filters, bias)
    _check_shapes(x, nn_idx, etype, filters, nou, net, ext)
    nn_idx = shared_graph_view(nn_idx)
    L = _hip.lib()
    M = nn_idx.shape[1]
    y = _alloc_out(x, nou, M)
    amax = None
    if want_argmax and agg == _hip.AGG_MAX:
        amax = torch.empty_like(y, dtype=torch.uint8)        # same element strides as y
    filters = filters.detach()
    if filters.dtype != torch.float32 or not filters.is_contiguous():
        filters = filters.float().contiguous()
    f32 = lambda t: None if t is None else t.detach().float().contiguous()
    bias, post_scale, post_shift = f32(bias), f32(post_scale), f32(post_shift)
    d = _hip.make_desc(x, nn_idx, etype, nou, net, ext, agg, relu, y)
    if M == 1 and net == 1 and agg == _hip.AGG_MAX and nn_idx.shape[2] == x.shape[2] and x.shape[2] > 1 and is_identity_list(nn_idx):
        d.reserved |= _hip.DESC_IDENTITY_LIST          # (the hyper-factor's table: the fan-in kernel reduces in its accumulators)
    nbytes = int(L.fgnn_mpconv_algorithmic_bytes(ctypes.byref(d))) if TIMER is not None else 0
    npart = 0
    if bn is not None and STATS_EPILOGUE and post_scale is None and not relu:
        npart = int(L.fgnn_mpconv_forward_stats_partials(ctypes.byref(d)))
    if npart > 0:
        from .mpnn import pointwise
        R = x.shape[0] * M
        ws = _workspace(x.device, int(L.fgnn_bn_workspace_bytes(R, nou)))
        stats, fin = pointwise.make_final(bn, nou, x.device, R)
        fold = _fold_scratch(x.device)
        _launch('fwd', d, nbytes, lambda: _hip.check(L.fgnn_mpconv_forward_stats(
            ctypes.byref(d), _hip._ptr(x), _hip._ptr(nn_idx), _hip._ptr(etype), _hip._ptr(filters),
            _hip._ptr(bias), _hip._ptr(y), _hip._ptr(amax), _hip._ptr(ws), fin, _hip._ptr(fold), _hip.stream_ptr())))
        pointwise.note_state_change()
        pointwise.set_pending_stats(y.permute(0, 2, 3, 1).reshape(R, nou), stats, bn[2])
        return y, amax
    if addends:
        # inference: the caller's running sums ride in the kernel's epilogue where it has one for them (``addends``: up to three
        # tensors of y's layout and dtype); otherwise they join in ONE n-input pass behind it (csrc/sum_n.hip)
        fits = (amax is None and post_scale is not None and relu and len(addends) <= 3 and
                all(a.dtype == y.dtype and a.shape == y.shape and a.stride() == y.stride() for a in addends))
        took = []
        if fits:
            ap = [_hip._ptr(a) for a in addends] + [None] * (3 - len(addends))
            _launch('fwd', d, nbytes + sum(a.numel() * a.element_size() for a in addends), lambda: took.append(L.fgnn_mpconv_forward_addends(
                ctypes.byref(d), _hip._ptr(x), _hip._ptr(nn_idx), _hip._ptr(etype), _hip._ptr(filters), _hip._ptr(bias),
                _hip._ptr(post_scale), _hip._ptr(post_shift), ap[0], ap[1], ap[2], _hip._ptr(y), _hip.stream_ptr())))
            if took[0] < 0:
                _hip.check(took[0])
            if took[0] == 1:
                return y, amax
            return add_n([y] + list(addends)), amax
    _launch('fwd', d, nbytes, lambda: _hip.check(L.fgnn_mpconv_forward(
        ctypes.byref(d), _hip._ptr(x), _hip._ptr(nn_idx), _hip._ptr(etype), _hip._ptr(filters),
        _hip._ptr(bias), _hip._ptr(post_scale), _hip._ptr(post_shift), _hip._ptr(y),
        _hip._ptr(amax), _hip.stream_ptr())))
    if addends:
        return add_n([y] + list(addends)), amax
    return y, amax


def grad_sink(param):
    """``param.grad`` when a kernel may accumulate straight into it, else None.

    Every gradient kernel of this package ACCUMULATES into its filter / bias outputs.  For a parameter that has
    been OPTED IN (``enable_grad_sink``; ``dp.FlatGradBucket`` does it for the parameters whose ``.grad`` are slices
    of its flat buffer) the kernel adds into ``.grad`` directly and the autograd Function returns None for that
    input — the value autograd's AccumulateGrad would produce, minus one zero-fill, one add and one cast per
    parameter per step.  The price of opting in: gradients reach such a parameter only through a plain
    ``loss.backward()`` — ``torch.autograd.grad``, ``backward(inputs=...)`` and tensor hooks do not see them.  Without
    the opt-in (the default for a bare module) gradients are returned to autograd like any other Function's."""
    if not ACCUMULATE_INTO_GRAD or param is None or not getattr(param, '_fgnn_grad_sink', False):
        return None
    if not param.is_leaf or not param.requires_grad:
        return None
    g = param.grad
    if (g is None or g.dtype != torch.float32 or not g.is_contiguous() or g.device != param.device
            or g.shape != param.shape or g.requires_grad):
        return None
    return g


def enable_grad_sink(params, on=True):
    """Opt parameters in to (or out of) in-place gradient accumulation by the kernels (see ``grad_sink``)."""
    for q in params:
        q._fgnn_grad_sink = bool(on)


STAMPS = None            # diagnosis (tools/stamps.py): {'buf': uint64 device tensor, 'tags': [...]} — see stamp()


def stamp(tag):
    """Diagnosis only: when ``STAMPS`` is armed, a one-thread kernel on the current stream writes the device clock into the next
    slot (fgnn_stamp).  Captured with the step, the slots tell when each point was reached in a replay with no profiler attached."""
    st = STAMPS
    if st is None:
        return
    i = len(st['tags'])
    if i >= st['buf'].numel():
        return
    st['tags'].append(tag)
    _hip.check(_hip.lib().fgnn_stamp(st['buf'].data_ptr() + 8 * i, _hip.stream_ptr()))


ACCUMULATE_INTO_GRAD = True      # master switch for the opted-in parameters (False: always return gradients)
STATS_EPILOGUE = True    # the operator's forward leaves the following BatchNorm's batch statistics
_WS = {}


def _workspace(device, nbytes):
    """One grow-only scratch buffer per (device, stream) for the per-workgroup gradient slabs and statistics
    partials (reused by every call on that stream; contents are undefined between calls).  Per stream because
    FactorNN runs the hyper-factor branch of a layer on a side stream next to the parity-check branch.  A buffer
    that was handed out while a hipGraph was being captured is never freed: the graph holds its address, so when a
    later, larger request outgrows it the old block is retired to ``_WS_CAPTURED`` (kept alive) instead of released."""
    key = (device, torch.cuda.current_stream(device).cuda_stream)
    ent = _WS.get(key)
    capturing = torch.cuda.is_current_stream_capturing()
    if ent is None or ent[0].numel() * 4 < nbytes:
        if ent is not None and ent[1]:
            _WS_CAPTURED.append(ent[0])
        ent = _WS[key] = [torch.empty((nbytes + 3) // 4, device=device, dtype=torch.float32), capturing]
    elif capturing:
        ent[1] = True
    return ent[0]


_WS_CAPTURED = []       # workspaces whose addresses live in captured graphs
_FOLD = {}


def _fold_scratch(device):
    """The ticket counters + second-level rows of the in-kernel grid folds (csrc/fgnn_gridfold.h), one zero-initialised buffer per
    (device, stream), never freed or moved (captured graphs hold its address).  Kernels of one stream never overlap, so they share
    it; every user leaves the counters at zero."""
    key = (device, torch.cuda.current_stream(device).cuda_stream)
    buf = _FOLD.get(key)
    if buf is None:
        # allocated (and zeroed) outside any capture: a capture that meets an unseen stream first would record the memset into the
        # graph (every replay would zero live counters — harmless — but the block would come from the graph's private pool)
        if torch.cuda.is_current_stream_capturing():
            raise _hip.FgnnHipError('fgnn_amd: first use of a stream inside hipGraph capture — run the step once eagerly on the capture '
                                    'stream first (graph.StepGraph does)')
        buf = _FOLD[key] = torch.zeros(_hip.FOLD_SCRATCH_BYTES // 4, device=device, dtype=torch.int32)
    return buf


_SIDE = {}
SIDE_STREAM = os.environ.get('FGNN_NO_SIDE_STREAM') is None      # FactorNN: factor types beyond the first run on a second stream (captured as parallel graph branches)


def side_stream(device):
    """The second stream FactorNN issues its hyper-factor branch on (one per device, created on first use)."""
    st = _SIDE.get(device)
    if st is None:
        st = _SIDE[device] = torch.cuda.Stream(device)
    return st


# Test hook (tests/test_parity_pins_gpu.py: routing-forced gradient pin): called with (filters parameter, argmax tensor) by every
# training-mode operator forward, so that a reference can be run along the SAME max routes.  None in production.
ROUTE_TAP = None


class _MPConv(torch.autograd.Function):
    """z = agg(messages) + bias with the hand-written HIP forward and backward."""

    @staticmethod
    def forward(ctx, x, nn_idx, etype, filters, bias, nou, net, ext, agg, bn=None):
        nn_idx = shared_graph_view(nn_idx)               # saved in this form: the backward takes the same fast path
        # edge weights shared by the batch arrive un-expanded ([1, net, M, k], see mpconv()): their gradient is the batch SUM
        ctx.shared_et = etype.shape[0] == 1 and x.shape[0] > 1
        if ctx.shared_et:
            etype = etype.expand(x.shape[0], -1, -1, -1)
        z, amax = mpconv_forward_raw(x, nn_idx, etype, filters, bias, nou, net, ext, agg, bn=bn, want_argmax=True)
        if ROUTE_TAP is not None:
            ROUTE_TAP(filters, amax)
        ctx.cfg = (nou, net, ext, agg)
        ctx.has_bias = bias is not None
        ctx.params = (filters, bias)                     # the leaf tensors themselves (for grad_sink)
        ctx.save_for_backward(x, nn_idx, etype, filters, amax)
        return z

    @staticmethod
    def backward(ctx, gz):
        backward_node_begins()
        x, nn_idx, etype, filters, amax = ctx.saved_tensors
        nou, net, ext, agg = ctx.cfg
        L = _hip.lib()
        B, M, k = nn_idx.shape
        # gz must share z's layout/dtype (the descriptor carries one set of y strides)
        zl = _alloc_out(x, nou, M)
        if gz.dtype != zl.dtype or gz.stride() != zl.stride():
            zl.copy_(gz)
            gz = zl
        dense = x.is_contiguous() or x.permute(0, 2, 3, 1).is_contiguous()
        xx = x if dense else x.contiguous()
        gx = torch.empty_like(xx)                        # preserve_format keeps xx's strides
        want_get = ctx.needs_input_grad[2]
        get = torch.empty((B, net, M, k), device=x.device, dtype=etype.dtype) if want_get else None
        fparam, bparam = ctx.params
        gw_sink, gb_sink = grad_sink(fparam), grad_sink(bparam)
        gw = gw_sink if gw_sink is not None else torch.zeros(filters.shape, device=x.device, dtype=torch.float32)
        gb = None
        if ctx.has_bias:
            gb = gb_sink if gb_sink is not None else torch.zeros((nou,), device=x.device, dtype=torch.float32)
        w = filters.detach().float().contiguous()
        d = _hip.make_desc(xx, nn_idx, etype, nou, net, ext, agg, False, gz)
        d.reserved = max_in_degree(nn_idx, x.shape[2])
        # shared edge weights: where the kernel sums their gradient over the batch itself, ask for that form
        reduced = bool(want_get and ctx.shared_et and L.fgnn_mpconv_backward_reduces_getype(ctypes.byref(d)))
        if reduced:
            d.reserved |= _hip.DESC_GETYPE_REDUCED
            get = torch.empty((1, net, M, k), device=x.device, dtype=torch.float32)
        scope = fold_scope(gw_sink is not None and (gb is None or gb_sink is not None) and folds_deferrable())
        ws = scope.slabs(x.device, int(L.fgnn_mpconv_backward_workspace_bytes(ctypes.byref(d))))
        nbytes = 0
        if TIMER is not None:
            # algorithmic bytes of the backward: x, etype, nn_idx, gz, argmax read once;
            # gx, getype written once; filters read and gfilters written once
            shared_et = etype.stride(0) == 0 and B > 1
            shared_idx = nn_idx.stride(0) == 0 and B > 1
            nbytes = (xx.element_size() * (2 * xx.numel() + gz.numel())
                      + etype.element_size() * net * M * k * (1 if shared_et else B)
                      + 8 * M * k * (1 if shared_idx else B)
                      + (B * nou * M if amax is not None else 0)
                      + (get.element_size() * get.numel() if want_get else 0) + 8 * w.numel())
        tables = backward_tables(nn_idx, d)           # the transposed incidence, built once per graph (None: the kernel builds its own)
        with scope:
            _launch('bwd', d, nbytes, lambda: _hip.check(L.fgnn_mpconv_backward_with_tables(
                ctypes.byref(d), _hip._ptr(xx), _hip._ptr(nn_idx), _hip._ptr(etype), _hip._ptr(w),
                _hip._ptr(gz), None, _hip._ptr(amax), _hip._ptr(gx), _hip._ptr(get),
                _hip._ptr(gw), _hip._ptr(gb), _hip._ptr(ws), ws.numel() * 4, _hip._ptr(tables), _hip.stream_ptr())))
        if want_get and ctx.shared_et and not reduced:
            get = get.sum(dim=0, keepdim=True)
        if want_get and get.dtype != etype.dtype:
            get = get.to(etype.dtype)
        return (gx, None, get, None if gw_sink is not None else gw.to(filters.dtype),
                None if gb_sink is not None else gb, None, None, None, None, None)


def _unexpanded(etype):
    """[1, net, M, k] tensor that ``etype`` is a batch-``expand`` of, or None.  The reference scripts build one edge-weight
    table and repeat it over the batch (train_syn_hop_factor.py:284-295); given the un-expanded tensor the operator returns
    its gradient already summed over the batch.  Sharing is taken from PROVENANCE only when a gradient flows: a stride-0
    batch axis (the tensor the caller expanded when autograd can name it, else the first row of the view — autograd pads
    that row's gradient with zeros before its own sum) or a `repeat` node over the batch axis.  B materialised rows that
    merely hold equal VALUES are collapsed only when ``etype`` needs no gradient: with a gradient, row b's upstream (a leaf,
    an edge model whose rows coincide by accident — zero-initialised last layer, saturated ReLU) must receive its own g_b,
    not the batch sum in row 0."""
    if etype.dim() != 4 or etype.shape[0] < 2:
        return None
    if etype.stride(0) != 0:
        # B materialised copies (`etype.repeat(bsize, 1, 1, 1)`, train_syn_hop_factor.py:291).  With a gradient: shared only
        # if autograd itself says so — the tensor IS the output of `repeat` over the batch axis of a one-row tensor (row 0's
        # zero-padded gradient then reaches the un-repeated tensor through RepeatBackward's own sum; no host read, works under
        # hipGraph capture).  Without a gradient: one device comparison + host read (not while capturing).
        if etype.requires_grad:
            fn = etype.grad_fn
            reps = getattr(fn, '_saved_repeats', None) if fn is not None and fn.name() == 'RepeatBackward0' else None
            src = getattr(fn, '_saved_self_sym_sizes', None) if reps is not None else None
            if (not DEDUPE_GRAPHS or reps is None or src is None or tuple(reps) != (etype.shape[0], 1, 1, 1)
                    or tuple(src) != (1,) + tuple(etype.shape[1:])):
                return None
            return etype[:1]
        if not DEDUPE_GRAPHS or not etype.is_cuda:
            return None
        if torch.cuda.is_current_stream_capturing():
            same = Verdicts.recall('equal_rows', etype)
        else:
            same = Verdicts.note('equal_rows', etype, bool((etype == etype[:1]).all().item()))
        return etype[:1] if same else None
    base = etype._base
    if (base is not None and base.dim() == 4 and base.shape[0] == 1 and base.shape[1:] == etype.shape[1:]
            and base.data_ptr() == etype.data_ptr() and base.stride()[1:] == etype.stride()[1:]
            and base.requires_grad == etype.requires_grad):
        return base
    return etype[:1]


def autocast_operands(x, etype):
    """Inside an autocast region the operator behaves like torch's matmul-class ops: floating-point operands of DIFFERENT precisions
    (bf16 activations from the layers in front, f32 edge weights a script built from its own frozen f32 tensors —
    /root/reference/train_ldpc.py:83-84 `self.hetype_f2v.repeat(bsize, 1, 1, 1)`) are brought to the autocast dtype.  Outside a
    region nothing is cast (mismatched operands raise, as torch's own ops do)."""
    if (x.dtype != etype.dtype and x.is_cuda and x.is_floating_point() and etype.is_floating_point()
            and torch.is_autocast_enabled('cuda')):
        dt = torch.get_autocast_dtype('cuda')
        x = x if x.dtype == dt else x.to(dt)
        etype = etype if etype.dtype == dt else etype.to(dt)
    return x, etype


def mpconv(x, nn_idx, etype, filters, bias, nou, net, ext, agg, bn=None):
    """Differentiable pre-BatchNorm operator output z [B, nou, M, 1].  ``bn``: see mpconv_forward_raw."""
    if ext != _hip.EXT_NONE:                             # a [1, net, M, k] etype (shared edge weights, not expanded) is taken as is
        base = _unexpanded(etype)
        if base is not None:
            etype = base
    elif etype.shape[0] == 1 and x.shape[0] > 1:
        etype = etype.expand(x.shape[0], -1, -1, -1)
    return _MPConv.apply(x, nn_idx, etype, filters, bias, nou, net, ext, agg, bn)


def single_source_fanout(x, nn_idx, etype):
    """M when the call is ONE source node feeding M > 1 destinations through identical single edges — x [B, C, 1, 1], nn_idx
    [B, M, 1], etype [B, net, M, 1] equal for all m — so that every destination receives the same message and the operator's
    output is a per-sample vector broadcast over the nodes (the LDPC hyper-factor's call, /root/reference/train_ldpc.py:40-46,82-88:
    `hnn_idx_f2v` == 0, `hetype_f2v` == 1); else 0.  Equality of the edge weights over the nodes is read from a stride-0 node axis,
    or checked on the device once per tensor WITHOUT a gradient (remembered on the tensor that owns the memory, as
    blocks._is_identity_list; never while a hipGraph is being captured: an unseen tensor is then taken as general)."""
    if not FANOUT_BROADCAST or x.dim() != 4 or x.shape[2] != 1 or x.shape[3] != 1 or nn_idx.dim() != 3:
        return 0
    B, M, k = nn_idx.shape
    if k != 1 or M <= 1 or not x.is_cuda or etype.shape[2] != M:
        return 0
    if etype.stride(2) == 0:
        return M
    if etype.requires_grad:
        return 0
    owner = etype._base if etype._base is not None else etype
    key = (etype._version, etype.data_ptr(), tuple(etype.shape), tuple(etype.stride()))
    memo = getattr(owner, '_fgnn_node_invariant', None)
    if memo is None or memo[0] != key:
        if torch.cuda.is_current_stream_capturing():
            v = Verdicts.recall('node_invariant', etype)
            if v is None:
                return 0
            memo = (key, v)
        else:
            memo = (key, Verdicts.note('node_invariant', etype, bool((etype == etype[:, :, :1, :]).all().item())))
        owner._fgnn_node_invariant = memo
    return M if memo[1] else 0


FANOUT_BROADCAST = True     # (module switch: tests compare against the materialised rows)


class _BroadcastNodes(torch.autograd.Function):
    """y1 [B, C, 1, 1] -> a stride-0 view [B, C, M, 1]; backward = one node-sum pass (csrc/sum_n.hip).  The view carries its source
    (``_fgnn_bcast_src``): consumers that can add a per-sample row themselves (the fused BatchNorm / block-tail apply kernels,
    ``addend_period``) take the [B, C] source directly — nothing of size [B, C, M] is then written or read for this tensor, forward or
    backward — and every other consumer sees an ordinary (expanded) tensor."""

    @staticmethod
    def forward(ctx, y1, M):
        ctx.M = M
        return y1.expand(-1, -1, M, -1)

    @staticmethod
    def backward(ctx, g):
        backward_node_begins()
        from .mpnn import pointwise
        B, C, M, _ = g.shape
        rows = g.permute(0, 2, 3, 1)
        if not rows.is_contiguous():
            rows = rows.contiguous()
        return pointwise.node_sum(rows.reshape(B * M, C), M).view(B, 1, 1, C).permute(0, 3, 1, 2), None


def broadcast_nodes(y1, M):
    """The per-sample vector ``y1`` [B, C, 1, 1] as the [B, C, M, 1] tensor the reference materialises (see _BroadcastNodes)."""
    if torch.is_grad_enabled() and y1.requires_grad:
        out = _BroadcastNodes.apply(y1, M)
    else:
        out = y1.expand(-1, -1, M, -1)
    out._fgnn_bcast_src = y1
    if y1.is_cuda:
        y1._fgnn_home_stream = torch.cuda.current_stream(y1.device)      # (where y1's producer ran, i.e. where its backward node will run)
    return out


def _dense_same_layout(ts):
    t0 = ts[0]
    if not (t0.is_cuda and t0.dtype in (torch.float32, torch.bfloat16) and 2 <= len(ts) <= 8):
        return False
    if (t0.numel() * t0.element_size()) % 16 or t0.numel() == 0:
        return False
    dense = t0.is_contiguous() or (t0.dim() == 4 and t0.permute(0, 2, 3, 1).is_contiguous())
    order = lambda t: tuple(st for st, n in zip(t.stride(), t.shape) if n > 1)      # strides of size-1 axes are moot
    o0 = order(t0)
    return dense and all(t.dtype == t0.dtype and t.shape == t0.shape and order(t) == o0 and
                         t.data_ptr() % 16 == 0 for t in ts)


def sum_tensors(ts):
    """Sum of same-shape tensors in ONE pass (csrc/sum_n.hip) when they share a dense layout; pairwise adds else."""
    ts = [t for t in ts if t is not None]
    if len(ts) == 1:
        return ts[0]
    if not _dense_same_layout(ts):
        out = ts[0]
        for t in ts[1:]:
            out = out + t
        return out
    out = torch.empty_like(ts[0])                        # preserve_format: same strides as the inputs
    arr = (ctypes.c_void_p * len(ts))(*[t.data_ptr() for t in ts])
    timed('sum_n_kernel', (len(ts) + 1) * ts[0].numel() * ts[0].element_size(),
          lambda: _hip.check(_hip.lib().fgnn_sum_n(arr, len(ts), ts[0].numel(), _hip.dtype_code(ts[0]),
                                                   _hip._ptr(out), _hip.stream_ptr())))
    return out


MERGE_FAN_GRADS = True      # (module switches: tests compare against the per-consumer sums / launches)
MERGE_FAN_WGRADS = True
_PLACEHOLDERS = {}


class FanBox:
    """The gradient of a state with several consumers (``fan_out``) as ONE product.  A consumer whose input gradient is
    ``gz @ W`` — a node-wise map or the 1x1 map in front of a block's operator — does not form it: its backward DEPOSITS (gz, W)
    here and returns ``placeholder`` (a zero tensor without memory traffic: one element, expanded) to autograd; the fan-out's own
    backward then runs csrc/linear_fwd_b16.hip::linear_multi_b16_kernel over the deposited pairs with the gradients that did arrive
    as tensors (the residual, a skip link) as addends — instead of one [R, C] tensor per consumer and an n-input sum over them.
    Slots: 0 and 2 take 64-channel sources, 1 takes 64 / 128 / 256; a deposit that finds no slot is refused and its consumer forms
    the gradient itself (a real tensor: an addend).

    The same consumers' WEIGHT gradients all contract their gz with the same rows — the state.  A depositor whose parameter
    gradients go to sinks hands that job over too (``wgrad = (rows, gW, gb)``): the fan-out's backward then parks ONE
    csrc/linear_wgrad_b16.hip launch over all of them (``fgnn_linear_wgrad_multi``: the state is read once instead of once per
    consumer)."""

    def __init__(self, x):
        self.shape, self.dtype, self.device = tuple(x.shape), x.dtype, x.device
        B, C, H, W = x.shape
        self.R, self.C = B * H * W, C
        self.ok = (MERGE_FAN_GRADS and x.is_cuda and x.dtype == torch.bfloat16 and C in (64, 128, 256) and self.R >= 1024
                   and (x.permute(0, 2, 3, 1).is_contiguous() or x.is_contiguous() and H * W == 1))
        self.slots = [None, None, None]
        self.wg = [None, None, None]
        self.task = None               # the backward pass (autograd graph task id) the deposits belong to
        self._ph = None

    def placeholder(self, shape):
        if self._ph is None:
            key = (self.device, self.dtype)
            ph = _PLACEHOLDERS.get(key)
            if ph is None:                     # one element per (device, dtype) for the life of the process: no fill launch per step
                ph = _PLACEHOLDERS[key] = torch.zeros(1, device=self.device, dtype=self.dtype)
            self._ph = ph
        return self._ph.expand(shape)

    def is_placeholder(self, g):
        return self._ph is not None and g.data_ptr() == self._ph.data_ptr()

    def deposit(self, gz, weight, wgrad=None):
        """gz [R, K] bf16 dense rows, weight [K, C] f32 (a map's [cout, cin] weight as it lies in memory).  Returns 0 = refused,
        1 = taken, 2 = taken together with the weight-gradient job ``wgrad`` = (rows [R, C] bf16 — the state as the depositor read
        it —, gW [K, C] f32 accumulator, gb [K] f32 accumulator or None): the depositor then launches no weight-gradient kernel."""
        if not self.ok or gz.dim() != 2 or gz.shape[0] != self.R or gz.dtype != torch.bfloat16 or not gz.is_contiguous():
            return 0
        self._enter_pass()
        K = gz.shape[1]
        if (tuple(weight.shape) != (K, self.C) or weight.dtype != torch.float32 or not weight.is_contiguous() or gz.data_ptr() % 16
                or weight.data_ptr() % 16):
            return 0
        order = (0, 2, 1) if K == 64 else ((1,) if K in (128, 256) else ())
        for i in order:
            if self.slots[i] is None:
                self.slots[i] = (gz, weight)
                if MERGE_FAN_WGRADS and wgrad is not None and self._wgrad_ok(K, *wgrad):
                    self.wg[i] = wgrad
                    return 2
                return 1
        return 0

    def _enter_pass(self):
        """Deposits belong to ONE backward pass.  A pass that cut the fan-out node off (``torch.autograd.grad(..., inputs=...)`` on
        a retained graph, a pass that raised half-way) leaves its pairs behind; a later pass over the same graph must neither be
        refused its slots by them nor have ``merge`` multiply them on top of its own gradients: they are dropped here."""
        task = torch._C._current_graph_task_id()
        if self.task != task:
            if any(self.slots):
                self.slots, self.wg = [None, None, None], [None, None, None]
            self.task = task

    def _wgrad_ok(self, K, rows, gW, gb):
        if not (rows.dtype == torch.bfloat16 and tuple(rows.shape) == (self.R, self.C) and rows.is_contiguous() and rows.data_ptr() % 16 == 0):
            return False
        if not (gW.dtype == torch.float32 and gW.numel() == K * self.C and gW.is_contiguous()):
            return False
        if gb is not None and not (gb.dtype == torch.float32 and gb.numel() == K and gb.is_contiguous()):
            return False
        first = next((w for w in self.wg if w is not None), None)
        return first is None or first[0].data_ptr() == rows.data_ptr()        # (every job of a box contracts with the SAME rows)

    def _park_wgrads(self, slots, wg):
        """One parked weight-gradient launch for the jobs handed over with the deposits (several: fgnn_linear_wgrad_multi)."""
        jobs = [(slots[i][0], wg[i]) for i in range(3) if wg[i] is not None]
        if not jobs:
            return
        if len(jobs) == 3 and self.C == 256 and jobs[1][0].shape[1] == 256:
            # 256 -> (64 | 256 | 64): 24 slices, beyond a workgroup's 16 waves — the two 64-channel maps together, the wide one alone
            self._park_wgrad_group([jobs[0], jobs[2]])
            self._park_wgrad_group([jobs[1]])
            return
        self._park_wgrad_group(jobs)

    def _park_wgrad_group(self, jobs):
        L = _hip.lib()
        P = _hip._ptr
        rows = jobs[0][1][0]
        R, C = self.R, self.C
        gzs = [j[0] for j in jobs]
        record = folds_deferrable()          # (decided inside the pass: the parked launch may go out from its end-of-pass callback)
        couts = (ctypes.c_int32 * len(jobs))(*[g.shape[1] for g in gzs])
        multi = len(jobs) > 1 and int(L.fgnn_linear_wgrad_multi_workspace_bytes(R, C, len(jobs), couts)) > 0

        def launch(rows=rows, jobs=jobs):
            with fold_scope(record) as scope:
                if multi:
                    nb = int(L.fgnn_linear_wgrad_multi_workspace_bytes(R, C, len(jobs), couts))
                    ws = scope.slabs(rows.device, nb)
                    gy = (ctypes.c_void_p * len(jobs))(*[P(j[0]) for j in jobs])
                    gW = (ctypes.c_void_p * len(jobs))(*[P(j[1][1]) for j in jobs])
                    gb = (ctypes.c_void_p * len(jobs))(*[P(j[1][2]) for j in jobs])
                    timed('linear_wgrad_b16_kernel', 2 * R * (C + sum(couts)),
                          lambda: _hip.check(L.fgnn_linear_wgrad_multi(P(rows), R, C, len(jobs), gy, couts, gW, gb, P(ws), ws.numel() * 4,
                                                                       _hip.stream_ptr())), nflops=2 * R * C * sum(couts))
                    return
                for gz, (_, gW1, gb1) in jobs:       # (outside the merged kernel's family: one launch per map, as their owners would have)
                    K = gz.shape[1]
                    ws = scope.slabs(rows.device, int(L.fgnn_linear_wgrad_workspace_bytes(R, C, K)))
                    timed('linear_wgrad_b16_kernel', 2 * R * (C + K),
                          lambda: _hip.check(L.fgnn_linear_wgrad(P(rows), P(gz), R, C, K, _hip.BF16, P(gW1), P(gb1), P(ws), ws.numel() * 4,
                                                                 _hip.stream_ptr())), nflops=2 * R * C * K)
        defer_wgrad(launch, (rows,) + tuple(gzs))

    def merge(self, grads):
        """The state's gradient from the deposits + the gradients that arrived as tensors."""
        real = [g for g in grads if g is not None and not self.is_placeholder(g)]
        self._enter_pass()
        slots, self.slots = self.slots, [None, None, None]
        wg, self.wg = self.wg, [None, None, None]
        if not any(slots):
            return sum_tensors(real) if real else None
        self._park_wgrads(slots, wg)
        if slots[0] is None and slots[2] is not None:      # (a lone 64-channel source sits in slot 0 by construction; keep the kernel's rule anyway)
            slots[0], slots[2] = slots[2], None
        B, C, H, W = self.shape
        adds = []
        for g in real:
            rows = g.permute(0, 2, 3, 1)
            if rows.dtype != self.dtype or not rows.is_contiguous():
                rows = rows.to(self.dtype).contiguous()
            adds.append(rows.view(self.R, C))
        if len(adds) > 3:
            adds = adds[:2] + [sum_tensors(adds[2:])]
        cur = torch.cuda.current_stream(self.device)
        for sl in slots:
            if sl is not None:
                sl[0].record_stream(cur)                   # (a side-stream consumer allocated it; this stream's launch reads it)
        out = torch.empty((self.R, C), device=self.device, dtype=self.dtype)
        P = _hip._ptr
        xs = (ctypes.c_void_p * 3)(*[P(sl[0]) if sl else None for sl in slots])
        ws = (ctypes.c_void_p * 3)(*[P(sl[1]) if sl else None for sl in slots])
        ks = (ctypes.c_int32 * 3)(*[sl[0].shape[1] if sl else 0 for sl in slots])
        ad = (ctypes.c_void_p * 3)(*([P(a) for a in adds] + [None] * (3 - len(adds))))
        ktot = sum(ks)
        timed('linear_multi_b16_kernel', 2 * self.R * (ktot + C * (1 + len(adds))),
              lambda: _hip.check(_hip.lib().fgnn_linear_multi_forward(xs, ks, ws, ad, P(out), self.R, C, _hip.stream_ptr())),
              nflops=2 * self.R * ktot * C)
        return out.view(B, H, W, C).permute(0, 3, 1, 2)


def fan_box(x):
    """The FanBox a consumer may deposit its input gradient into, or None (``x`` is not a fan-out handle / merging is off)."""
    box = getattr(x, '_fgnn_box', None)
    return box if (box is not None and box.ok) else None


class _FanOut(torch.autograd.Function):
    """n aliases of x whose gradients come back TOGETHER: one n-input sum instead of autograd's n-1 adds — or, where the consumers
    deposited their (gz, W) pairs in the handles' FanBox, one product over them."""

    @staticmethod
    def forward(ctx, x, n, box):
        ctx.set_materialize_grads(False)                 # an unused alias contributes nothing, not a zero tensor
        ctx.box = box
        return tuple(x.view_as(x) for _ in range(n))

    @staticmethod
    def backward(ctx, *grads):
        backward_node_begins()
        return ctx.box.merge(grads), None, None


class _SumN(torch.autograd.Function):
    @staticmethod
    def forward(ctx, *ts):
        return sum_tensors([t.detach() for t in ts])

    @staticmethod
    def backward(ctx, g):
        return tuple(g for _ in ctx.needs_input_grad)


def fan_out(x, n):
    """``n`` handles on ``x`` for ``n`` consumers (a single fused gradient sum in the backward)."""
    if n <= 1 or not (torch.is_grad_enabled() and x.requires_grad and x.is_cuda):
        return [x] * n
    box = FanBox(x) if x.dim() == 4 else None
    if box is None:
        class _Plain:       # (a non-4-D state: nothing to deposit into)
            ok = False
            merge = staticmethod(lambda grads: (lambda gs: sum_tensors(gs) if gs else None)([g for g in grads if g is not None]))
        box = _Plain()
    outs = list(_FanOut.apply(x, n, box))
    for o in outs:
        o._fgnn_box = box
    return outs


def add_n(ts):
    """Differentiable sum of tensors, one kernel when the layouts allow."""
    ts = [t for t in ts if t is not None]
    if len(ts) == 1:
        return ts[0]
    if not (ts[0].is_cuda and _dense_same_layout([t.detach() for t in ts])):
        out = ts[0]
        for t in ts[1:]:
            out = out + t
        return out
    return _SumN.apply(*ts)


def _concat2_plan(a, b, dim):
    """(samples, inner, chunk bytes a / b, sample / chunk strides in bytes of a / b) when ``torch.cat([a, b], dim)`` of two [B, C, N, 1]
    device tensors is in fgnn_concat_pair's family — channel-fastest (stride(1) == 1; slices of a larger channel-fastest activation
    qualify), dim 1 (channels) or 2 (nodes), 16-byte granularity — else None."""
    if not (a.is_cuda and b.is_cuda and a.dim() == 4 and b.dim() == 4 and a.dtype == b.dtype and a.shape[3] == 1 and b.shape[3] == 1
            and a.shape[0] == b.shape[0] and dim in (1, 2) and a.shape[3 - dim] == b.shape[3 - dim] and a.numel() and b.numel()):
        return None
    es = a.element_size()
    for t in (a, b):
        C, N = t.shape[1], t.shape[2]
        if (C > 1 and t.stride(1) != 1) or (N > 1 and t.stride(2) < C) or (t.shape[0] > 1 and t.stride(0) < C * N):
            return None
    B = a.shape[0]
    if dim == 2 and any(t.shape[2] > 1 and t.stride(2) != t.shape[1] for t in (a, b)):
        # rows strided inside a sample (a channel slice of a wider activation): the row form (fgnn_concat_rows)
        rb_ = a.shape[1] * es
        vals = (rb_, a.stride(0) * es, a.stride(2) * es, b.stride(0) * es, b.stride(2) * es)
        if any(v % 16 for v in vals) or (a.data_ptr() | b.data_ptr()) % 16:
            return None
        return ('rows', B, a.shape[2], b.shape[2]) + vals
    if dim == 2:
        inner, ca, cb = 1, a.shape[1] * a.shape[2] * es, b.shape[1] * b.shape[2] * es
        sa, sb = (a.stride(0) * es, 0), (b.stride(0) * es, 0)
    else:
        inner, ca, cb = a.shape[2], a.shape[1] * es, b.shape[1] * es
        sa, sb = (a.stride(0) * es, a.stride(2) * es), (b.stride(0) * es, b.stride(2) * es)
    vals = (ca, cb) + sa + sb
    if any(v % 16 for v in vals) or (a.data_ptr() | b.data_ptr()) % 16:
        return None
    return B, inner, ca, cb, sa[0], sa[1], sb[0], sb[1]


def _concat2_raw(a, b, dim, plan=None):
    """``torch.cat([a, b], dim)`` as one kernel (fgnn_concat_pair), channel-fastest result; None outside the family (see _concat2_plan)."""
    plan = plan or _concat2_plan(a, b, dim)
    if plan is None:
        return None
    B = a.shape[0]
    if dim == 2:
        out = torch.empty((B, a.shape[2] + b.shape[2], 1, a.shape[1]), device=a.device, dtype=a.dtype).permute(0, 3, 1, 2)
    else:
        out = torch.empty((B, a.shape[2], 1, a.shape[1] + b.shape[1]), device=a.device, dtype=a.dtype).permute(0, 3, 1, 2)
    if plan[0] == 'rows':
        _hip.check(_hip.lib().fgnn_concat_rows(_hip._ptr(a), _hip._ptr(b), _hip._ptr(out), *plan[1:], _hip.stream_ptr()))
    else:
        _hip.check(_hip.lib().fgnn_concat_pair(_hip._ptr(a), _hip._ptr(b), _hip._ptr(out), *plan, _hip.stream_ptr()))
    return out


class _Concat2(torch.autograd.Function):
    """torch.cat of two tensors as one kernel; the backward is torch.cat's own (two narrow views of the gradient)."""

    @staticmethod
    def forward(ctx, a, b, dim):
        ctx.dim, ctx.na = dim, a.shape[dim]
        return _concat2_raw(a, b, dim)

    @staticmethod
    def backward(ctx, g):
        nb = g.shape[ctx.dim] - ctx.na
        return g.narrow(ctx.dim, 0, ctx.na), g.narrow(ctx.dim, ctx.na, nb), None


def concat2(a, b, dim):
    """``torch.cat([a, b], dim)`` — one launch for two channel-fastest device activations (factor_mpnn's node-axis and channel-axis
    concatenations, /root/reference/lib/model/mpnn/factor_mpnn.py:104-107,116), torch.cat otherwise."""
    if _concat2_plan(a.detach(), b.detach(), dim) is None:
        out = torch.cat([a, b], dim=dim)
        return out.contiguous(memory_format=torch.channels_last) if (out.dim() == 4 and dim == 2) else out
    if torch.is_grad_enabled() and (a.requires_grad or b.requires_grad):
        return _Concat2.apply(a, b, dim)
    return _concat2_raw(a, b, dim)


def algorithmic_bytes(x, nn_idx, etype, nou, net, ext, agg):
    """SURVEY §8d algorithmic HBM bytes of one forward call (for bench.py's roofline)."""
    y = _alloc_out(x, nou, nn_idx.shape[1]) if x.is_cuda else None
    d = _hip.make_desc(x, nn_idx, etype, nou, net, ext, agg, False, y)
    return int(_hip.lib().fgnn_mpconv_algorithmic_bytes(ctypes.byref(d)))
