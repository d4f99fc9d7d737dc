"""One switch that gives an UNCHANGED reference script the benched path.

Through the ``lib.model.mpnn`` shim a reference script (``train_ldpc.py``) gets the hand-written kernels, but in the regime
its own lines select: f32 activations, ``Sequential(Conv2d, ReLU, Conv2d)`` edge models run by torch, ``torch.optim.Adam``
over ~330 tensors (59.3 ms per step at 4096 codewords against 17 ms for ``bench.py``'s bf16 line).  ``enable_fast_path()``
— or ``FGNN_FAST_PATH=1`` in the environment, read when ``fgnn_amd`` is imported — changes that without touching the script:

* the first time a module that OWNS a ``FactorNN`` (the script's ``LDPCModel``, ``/root/reference/train_ldpc.py:19-99``) is
  called, its edge models — children of the form ``Sequential(Conv2d(c, 64, 1), ReLU, Conv2d(64, net, 1))``
  (``train_ldpc.py:32-38``) — are replaced by ``EdgeMLP`` modules built around THE SAME parameter tensors (same state_dict
  keys, same optimizer slots), and its ``forward`` is wrapped: floating-point CUDA inputs are cast to bf16, the call runs under
  ``torch.autocast('cuda', bfloat16)``, floating-point outputs come back as f32;
* ``torch.optim.Adam`` is wrapped: constructed over CUDA f32 parameters it returns ``FastAdam`` — a ``torch.optim.Optimizer``
  (schedulers and ``state_dict`` work) whose parameters and gradients live in the flat buffers of ``dp.FlatGradBucket`` and whose
  ``step`` is the one-kernel ``dp.FlatAdam`` update (``zero_grad`` = one memset).  Anything else gets the stock class.

* round 6 — the hipGraph replay too (``GraphedForward``): once the optimised module has been called ``FGNN_FAST_GRAPH_AFTER``
  (default 3) times in training mode with inputs of one geometry and every trainable parameter's gradient lives in a ``FastAdam``
  bucket, its forward and its backward are captured as two hipGraphs (the script's own lines between them — the loss, ``backward()``,
  ``optimizer.step()`` — stay eager: ~30 short launches) and replayed from then on: new inputs are copied into the captured
  buffers, the outputs come back through an autograd node whose backward replays the second graph.  Any change the graphs
  cannot follow — another batch size, neighbour tables with other contents, eval mode, no-grad, a replaced parameter — runs that
  call eagerly as before.  ``FGNN_FAST_GRAPH_AFTER=0`` keeps every call eager.

``disable_fast_path()`` undoes the patches (modules already optimised stay so).
"""
import os
import threading
import weakref

import torch

_STATE = {'hook': None, 'adam': None}
_SEEN = weakref.WeakSet()      # modules the global hook has looked at (kept here, not as an attribute on every nn.Module of the process)


def _is_edge_model(m):
    if not isinstance(m, torch.nn.Sequential) or len(m) != 3:
        return False
    c1, act, c2 = m[0], m[1], m[2]
    ok = lambda c: isinstance(c, torch.nn.Conv2d) and c.kernel_size == (1, 1) and c.stride == (1, 1) and c.groups == 1
    return ok(c1) and isinstance(act, torch.nn.ReLU) and ok(c2) and c1.out_channels == c2.in_channels


_AUTOCAST = threading.local()      # per thread: the autocast contexts the fast modules' pre-hooks opened (closed by their post-hooks)


def _fast_pre(module, args, kwargs):
    """Forward pre-hook of an optimised module: f32 CUDA inputs -> bf16, and the call runs under bf16 autocast (closed by
    ``_fast_post``).  Hooks — module-level functions, registered on the module — instead of a replaced ``forward``: the module
    stays picklable and ``copy.deepcopy``-able (a closure installed as ``module.forward`` is neither)."""
    cast = lambda t: t.to(torch.bfloat16) if torch.is_tensor(t) and t.is_cuda and t.dtype == torch.float32 else t
    ctx = torch.autocast('cuda', dtype=torch.bfloat16)
    ctx.__enter__()
    stack = getattr(_AUTOCAST, 'stack', None)
    if stack is None:
        stack = _AUTOCAST.stack = []
    stack.append((id(module), ctx))
    return tuple(cast(a) for a in args), {k: cast(v) for k, v in kwargs.items()}


def _fast_post(module, args, kwargs, out):
    """Closes the autocast region (also when the forward raised: registered with ``always_call``); bf16 outputs -> f32."""
    stack = getattr(_AUTOCAST, 'stack', None)
    if stack and stack[-1][0] == id(module):      # (not at the call during which the hooks were registered: its pre-hook never ran)
        stack.pop()[1].__exit__(None, None, None)
    back = lambda t: t.float() if torch.is_tensor(t) and t.dtype == torch.bfloat16 else t
    if isinstance(out, (tuple, list)):
        return tuple(back(o) for o in out)
    return back(out)


GRAPH_AFTER = int(os.environ.get('FGNN_FAST_GRAPH_AFTER', '3') or 0)


def _flat_tensors(out):
    if torch.is_tensor(out):
        return [out]
    if isinstance(out, (tuple, list)):
        return [t for o in out for t in _flat_tensors(o)]
    return []


class _Replay(torch.autograd.Function):
    """The captured forward as ONE autograd node: forward = replay of the first graph, backward = replay of the second (which
    accumulates every parameter gradient into the optimizer's flat bucket, as the eager backward does)."""

    @staticmethod
    def forward(ctx, cap, anchor):
        ctx.cap = cap
        cap.fwd.replay()
        outs = tuple(o.detach().clone() for o in cap.static_out)
        ctx.mark_non_differentiable(*[o for o, g in zip(outs, cap.static_grad) if g is None])
        return outs

    @staticmethod
    def backward(ctx, *grads):
        cap = ctx.cap
        with torch.no_grad():
            for g, sg in zip(grads, cap.static_grad):
                if sg is not None:
                    if g is None:
                        sg.zero_()
                    else:
                        sg.copy_(g)
        cap.bwd.replay()
        cap.after_replay()
        return None, None


class _Captured:
    def __init__(self):
        self.fwd = self.bwd = None
        self.static_in = self.static_out = self.static_grad = None
        self.out_tree = None
        self.pointers = None
        self.frozen = None

    @staticmethod
    def after_replay():
        from .mpnn import pointwise
        pointwise.note_state_change()       # the replayed kernels moved BatchNorm buffers / gradient slices behind torch's version counters
        pointwise.invalidate_casts()


class GraphedForward:
    """Installed as ``module.forward`` of an optimised module (``fast_path``): eager calls until one input geometry has been seen
    ``GRAPH_AFTER`` times in training mode, then two hipGraphs (see the module docstring).  The reference loop it serves:
    /root/reference/train_ldpc.py:207-231."""

    def __init__(self, module):
        self.module = module      # (a reference cycle with the module's own attribute: collected with it)
        self.seen = {}
        self.cap = None
        self.key = None
        self.failed = False
        self.anchor = None
        self.replays = 0

    def eager(self, *args, **kwargs):
        """The module's own forward (the class's: this object shadows it as an instance attribute)."""
        return type(self.module).forward(self.module, *args, **kwargs)

    def __getstate__(self):        # graphs and captured buffers do not travel (pickle / deepcopy give an eager module that re-captures)
        return {'module': self.module, 'seen': {}, 'cap': None, 'key': None, 'failed': False, 'anchor': None, 'replays': 0}

    # ---- what a call must look like to be captured / replayed ----
    @staticmethod
    def _signature(args):
        sig = []
        for a in args:
            if torch.is_tensor(a):
                if not a.is_cuda:
                    return None
                sig.append((tuple(a.shape), a.dtype, a.device))
            else:
                return None                # (a non-tensor argument would be baked into the graph)
        return tuple(sig) if sig else None

    def _trainable_in_bucket(self):
        """Every trainable parameter's ``.grad`` is a persistent slice the gradient kernels add into (``FastAdam``'s flat bucket):
        only then does a replayed backward land where the optimizer reads — an autograd-owned ``.grad`` is re-created (or set to None
        by ``zero_grad``) every step, at addresses a graph cannot know."""
        any_p = False
        for q in self.module.parameters():
            if q.requires_grad:
                any_p = True
                if q.grad is None or not getattr(q, '_fgnn_grad_sink', False):
                    return False
        return any_p

    def _pointers(self):
        """Addresses the graphs hold: parameters, buffers, and the gradient slices the replayed backward adds into (a
        ``model.zero_grad()`` — set_to_none — or a re-created optimizer moves those)."""
        ps = list(self.module.parameters())
        return (tuple(t.data_ptr() for t in ps + list(self.module.buffers()))
                + tuple(0 if (q.grad is None or not q.requires_grad) else q.grad.data_ptr() for q in ps))

    def _frozen_versions(self):
        return tuple(q._version for q in self.module.parameters() if not q.requires_grad)

    def __call__(self, *args, **kwargs):
        m = self.module
        if GRAPH_AFTER <= 0 or self.failed or kwargs or not m.training or not torch.is_grad_enabled():
            return self.eager(*args, **kwargs)
        key = self._signature(args)            # (None: a CPU or non-tensor argument — nothing below touches the device then)
        if key is None or torch.cuda.is_current_stream_capturing():
            return self.eager(*args, **kwargs)
        if self.cap is not None:
            if key == self.key and self._valid(args):
                return self._replay(args)
            return self.eager(*args, **kwargs)
        n = self.seen.get(key, 0) + 1
        self.seen[key] = n
        if n <= GRAPH_AFTER or not self._trainable_in_bucket():
            return self.eager(*args, **kwargs)
        try:
            self._capture(key, args)
        except Exception as e:      # noqa: BLE001 — a model the capture cannot follow keeps the eager fast path; say so once
            import warnings
            self.failed, self.cap = True, None
            warnings.warn('fgnn_amd fast path: hipGraph capture of %s failed (%s: %s); its steps stay eager'
                          % (type(m).__name__, type(e).__name__, e))
            return self.eager(*args, **kwargs)
        return self._replay(args, fresh=True)

    def _valid(self, args):
        cap = self.cap
        if cap.pointers != self._pointers() or cap.frozen != self._frozen_versions():
            self.cap, self.seen = None, {}             # a parameter / buffer was replaced, a frozen table rewritten: capture again later
            return False
        for a, st in zip(args, cap.static_in):
            if a.dtype.is_floating_point or a.data_ptr() == st.data_ptr():
                continue
            # one device comparison + host read per table and step (the loop this serves reads its loss back every step anyway).  No
            # memo on (address, version): a collated batch is a NEW tensor every step, and the allocator hands the same address out
            # again — a table with other contents at a remembered address would be replayed with the captured one's kernels
            if not torch.equal(a, st):
                return False                           # other neighbour tables: the graphs' fast paths were chosen for the captured ones
        return True

    def _replay(self, args, fresh=False):
        cap = self.cap
        if not fresh:
            with torch.no_grad():
                for a, st in zip(args, cap.static_in):
                    if a.dtype.is_floating_point and a.data_ptr() != st.data_ptr():
                        st.copy_(a)
        outs = _Replay.apply(cap, self.anchor)
        self.replays += 1
        it = iter(outs)
        return cap.out_tree(it)

    class _FreshLeaves:
        """For the duration of the capture procedure every trainable parameter of the module tree is replaced by a NEW leaf on the
        same storage with the same ``.grad`` slice.  Why: a parameter of a plain torch module (the script's regressor head,
        train_ldpc.py:60-66) gets its gradient through an AccumulateGrad node that is cached on the parameter while ANY autograd
        graph that used it is alive — the previous iteration's, still referenced by the script's ``loss`` variable — and that node
        runs on the stream it was created for, the script's default stream: outside the capture (torch warns 'AccumulateGrad
        node's stream does not match', hipStreamEndCapture then segfaults; ``torch.autograd.grad`` routes through the same nodes).
        A fresh leaf gets a fresh node, created on the capture stream.  The graphs only hold addresses, and those are the
        originals'."""

        def __init__(self, module):
            self.module, self.swapped = module, []

        def __enter__(self):
            alias = {}
            for mod in self.module.modules():
                for name, q in list(mod._parameters.items()):
                    if q is None or not q.requires_grad:
                        continue
                    a = alias.get(id(q))
                    if a is None:
                        a = alias[id(q)] = torch.nn.Parameter(q.detach(), requires_grad=True)
                        a.grad = q.grad
                        a._fgnn_grad_sink = getattr(q, '_fgnn_grad_sink', False)
                    mod._parameters[name] = a
                    self.swapped.append((mod, name, q))
            return self

        def __exit__(self, *exc):
            for mod, name, q in self.swapped:
                mod._parameters[name] = q
            self.swapped = []
            return False

    def _capture(self, key, args):
        from . import ops
        from .mpnn import pointwise
        m = self.module
        dev = args[0].device
        with self._FreshLeaves(m):
            cap = self._capture_with_fresh_leaves(key, args, dev)
        cap.pointers, cap.frozen = self._pointers(), self._frozen_versions()
        self.anchor = torch.zeros((), device=dev, requires_grad=True)
        self.cap, self.key = cap, key

    def _capture_with_fresh_leaves(self, key, args, dev):
        from . import ops
        from .mpnn import pointwise
        m = self.module
        cap = _Captured()
        with torch.no_grad():
            cap.static_in = [a.clone() for a in args]
        saved = [(b, b.detach().clone()) for b in m.buffers()]      # the warm-up runs below move BatchNorm's running statistics: put back
        stream = torch.cuda.Stream(dev)
        stream.wait_stream(torch.cuda.current_stream(dev))
        verdicts = None
        with torch.cuda.stream(stream):
            for i in range(2):          # allocator / workspace / per-stream scratch warm-up ON the capture stream; the second run's
                rec = ops.Verdicts.recording()     # host-side verdicts are what the capture takes for tensors built inside the forward
                with rec as v, _uncached_autocast():
                    out = self.eager(*cap.static_in)
                    live = [t for t in _flat_tensors(out) if t.requires_grad]
                    # zero upstream gradients: the pass exercises every backward kernel and adds exactly +0 to the gradient slices
                    torch.autograd.backward(live, [torch.zeros_like(t) for t in live])
                verdicts = v
                del out, live
        torch.cuda.current_stream(dev).wait_stream(stream)
        torch.cuda.synchronize(dev)
        with torch.no_grad():
            for b, old in saved:
                b.copy_(old)
        pointwise.note_state_change()
        pointwise.invalidate_casts()           # the weight casts are recorded: every replay derives them from the parameters of that moment
        cap.fwd = torch.cuda.CUDAGraph()
        with verdicts.replaying(), _uncached_autocast():
            with torch.cuda.graph(cap.fwd, stream=stream):
                out = self.eager(*cap.static_in)
        flat = _flat_tensors(out)
        if not flat or not any(t.requires_grad for t in flat):
            raise RuntimeError('the forward returned nothing differentiable')
        cap.static_out = flat
        cap.static_grad = [torch.zeros_like(t) if t.requires_grad else None for t in flat]
        cap.bwd = torch.cuda.CUDAGraph()
        with verdicts.replaying():             # (the backward asks too: the in-degree of a table its forward built)
            with torch.cuda.graph(cap.bwd, pool=cap.fwd.pool(), stream=stream):
                torch.autograd.backward([t for t in flat if t.requires_grad], [g for g in cap.static_grad if g is not None])
        cap.recorded = verdicts.taken
        cap.unused = sum(len(q) for q in verdicts.fifo.values())
        # the captured passes did not EXECUTE: the first real execution is the replay the caller gets now
        cap.out_tree = _tree_builder(out)
        return cap


def _uncached_autocast():
    """The autocast region the pre-hook opened caches its weight casts until it closes.  A warm-up run would leave casts there
    that the captured forward then REUSES instead of recording their kernels (and that are freed when the region closes): every
    run of the capture procedure starts from an empty cache and keeps none."""
    torch.clear_autocast_cache()
    on = torch.is_autocast_enabled('cuda')
    return torch.autocast('cuda', dtype=torch.get_autocast_dtype('cuda') if on else None, enabled=on, cache_enabled=False)


def _tree_builder(out):
    """A function that rebuilds ``out``'s nesting (tensor | tuple / list of ...) from an iterator over replacement tensors."""
    if torch.is_tensor(out):
        return lambda it: next(it)
    if isinstance(out, (tuple, list)):
        subs = [_tree_builder(o) for o in out]
        kind = type(out)
        return lambda it: kind(b(it) for b in subs)
    return lambda it, out=out: out


def fast_path(module):
    """Optimise ONE module that owns a ``FactorNN`` (what the global hook does at its first call).  Idempotent."""
    from .edge_mlp import EdgeMLP
    if getattr(module, '_fgnn_fast', False):
        return module
    module._fgnn_fast = True
    for name, child in list(module.named_children()):
        if _is_edge_model(child) and not isinstance(child, EdgeMLP):
            c1, c2 = child[0], child[2]
            new = EdgeMLP(c1.in_channels, c1.out_channels, c2.out_channels)
            new[0].weight, new[0].bias, new[2].weight, new[2].bias = c1.weight, c1.bias, c2.weight, c2.bias      # the SAME Parameters
            new.train(child.training)
            setattr(module, name, new)
    module.register_forward_pre_hook(_fast_pre, with_kwargs=True)
    module.register_forward_hook(_fast_post, with_kwargs=True, always_call=True)
    if GRAPH_AFTER > 0 and 'forward' not in module.__dict__:
        module.forward = GraphedForward(module)      # (inside the hooks: it sees bf16 inputs under autocast)
    return module


def _pre_hook(module, args):
    from .mpnn import FactorNN
    if module in _SEEN:            # one look per module: the hook runs in front of every module call
        return None
    _SEEN.add(module)
    if not isinstance(module, FactorNN) and any(isinstance(c, FactorNN) for c in module.children()):
        fast_path(module)          # (the hooks registered here take effect from the NEXT call: this one's hook list is already fixed)
    return None


class FastAdam(torch.optim.Optimizer):
    """``torch.optim.Adam``'s update (no amsgrad) on flat parameter / gradient buffers: one kernel per step.

    Persistence (``/root/reference/train_ldpc.py:179-181,187`` saves and restores ``optimizer.state_dict()``): the moments and
    the step count live in ``self.flat`` (``dp.FlatAdam``), not in ``self.state``; ``state_dict()`` exports them in STOCK Adam's
    layout — per parameter ``step`` (a float32 scalar tensor), ``exp_avg``, ``exp_avg_sq`` (clones, shaped like the parameter) — and
    ``load_state_dict()`` accepts that layout, whether it was written by this class or by ``torch.optim.Adam`` itself, and
    copies it into the flat buffers.  A checkpoint therefore moves freely between the two.

    Differences from the stock class that remain: the update is DENSE — every parameter of the flat buffer is updated every
    step with whatever its gradient slice holds (zero for a parameter the loss did not reach), whereas ``torch.optim.Adam``
    skips parameters whose ``.grad`` is None; with ``weight_decay`` != 0 an unused parameter therefore decays here and not
    there.  One hyper-parameter set for all parameters (the first group's); ``zero_grad(set_to_none=True)`` zeroes."""

    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.0):
        from .dp import FlatAdam, FlatGradBucket
        params = list(params)
        super().__init__(params, dict(lr=lr, betas=betas, eps=eps, weight_decay=weight_decay))
        self.bucket = FlatGradBucket([p for g in self.param_groups for p in g['params']], flatten_params=True)
        self.flat = FlatAdam(self.bucket, lr=lr, betas=betas, eps=eps, weight_decay=weight_decay)

    @torch.no_grad()
    def step(self, closure=None):
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        g = self.param_groups[0]                     # (schedulers write the current rate here)
        self.flat.lr, self.flat.betas, self.flat.eps, self.flat.weight_decay = g['lr'], g['betas'], g['eps'], g['weight_decay']
        self.bucket.all_reduce_sum()                 # a no-op outside torch.distributed
        self.flat.step(grad_scale=1.0 / self.bucket.world)
        return loss

    def zero_grad(self, set_to_none=False):          # the gradients are views of one buffer: never set to None
        self.bucket.zero()

    def _slices(self):
        for q, off in zip(self.bucket.params, self.bucket.offsets):
            yield q, off, q.numel()

    @torch.no_grad()
    def state_dict(self):
        """Stock Adam's layout (see the class docstring).  Before the first step the state is empty, as the stock class's is."""
        t = self.flat.t
        self.state.clear()
        if t > 0:
            for q, off, n in self._slices():
                self.state[q] = {'step': torch.tensor(float(t), dtype=torch.float32),
                                 'exp_avg': self.flat.exp_avg[off:off + n].view_as(q).clone(),
                                 'exp_avg_sq': self.flat.exp_avg_sq[off:off + n].view_as(q).clone()}
        try:
            return super().state_dict()
        finally:
            self.state.clear()                       # the live state stays in the flat buffers only

    @torch.no_grad()
    def load_state_dict(self, state_dict):
        super().load_state_dict(state_dict)          # (validates group sizes, restores the hyper-parameters, casts to the parameters' device)
        steps = set()
        for q, off, n in self._slices():
            st = self.state.get(q)
            if not st:                               # no entry: a parameter the saved optimizer never stepped
                self.flat.exp_avg[off:off + n].zero_()
                self.flat.exp_avg_sq[off:off + n].zero_()
                continue
            if st.get('amsgrad') or 'max_exp_avg_sq' in st:
                raise ValueError('FastAdam cannot resume an amsgrad checkpoint')
            self.flat.exp_avg[off:off + n].copy_(st['exp_avg'].reshape(-1))
            self.flat.exp_avg_sq[off:off + n].copy_(st['exp_avg_sq'].reshape(-1))
            steps.add(int(float(st['step'])))
        if len(steps) > 1:
            raise ValueError('FastAdam keeps ONE step count for all parameters; the checkpoint holds %s' % sorted(steps))
        self.flat.t = steps.pop() if steps else 0
        self.state.clear()
        g = self.param_groups[0]
        self.flat.lr, self.flat.betas, self.flat.eps, self.flat.weight_decay = g['lr'], g['betas'], g['eps'], g['weight_decay']


def _wants_fast_adam(params, amsgrad, kw):
    """One group, nothing exotic asked for, every TRAINABLE parameter a CUDA f32 tensor (and at least one of them).  Frozen
    parameters may ride along — the reference hands ``model.parameters()`` over with its four frozen hyper-edge tables in it,
    two of them int64 (/root/reference/train_ldpc.py:45-58,160-161); stock Adam never touches a parameter without a gradient and
    neither does ``FastAdam`` (they stay in ``param_groups``, so ``state_dict()`` numbers the parameters as the stock class does)."""
    plain = bool(params) and all(torch.is_tensor(p) for p in params)
    if not plain or amsgrad or kw:
        return False
    live = [p for p in params if p.requires_grad]
    return bool(live) and all(p.is_cuda and p.dtype == torch.float32 for p in live)


def _adam_factory(stock):
    """A CLASS that stands in for ``torch.optim.Adam`` while the fast path is on: constructing it returns ``FastAdam`` where
    that applies (see ``_wants_fast_adam``; frozen parameters, parameter groups, amsgrad / fused / foreach requests keep the
    stock class) and a stock Adam otherwise.  It stays a type — ``class MyAdam(torch.optim.Adam)`` and
    ``isinstance(opt, torch.optim.Adam)`` keep working (a ``FastAdam`` passes the check too)."""

    class _Meta(type(stock)):
        def __instancecheck__(cls, obj):
            return isinstance(obj, FastAdam) or type.__instancecheck__(cls, obj) or isinstance(obj, stock)

    class Adam(stock, metaclass=_Meta):
        __doc__ = stock.__doc__

        def __new__(cls, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=0, amsgrad=False, **kw):
            if cls is Adam:                          # (a user's subclass is built as what it is)
                params = list(params)
                if _wants_fast_adam(params, amsgrad, kw):
                    return FastAdam(params, lr=lr, betas=betas, eps=eps, weight_decay=weight_decay)      # not a cls instance: __init__ is skipped
                obj = stock.__new__(stock)
                stock.__init__(obj, params, lr=lr, betas=betas, eps=eps, weight_decay=weight_decay, amsgrad=amsgrad, **kw)
                return obj                           # a plain stock instance (again not a cls instance)
            return stock.__new__(cls)

    Adam.__name__, Adam.__qualname__ = 'Adam', 'Adam'
    Adam.stock = stock
    return Adam


def enable_fast_path(adam=True):
    if _STATE['hook'] is None:
        _STATE['hook'] = torch.nn.modules.module.register_module_forward_pre_hook(_pre_hook)
    if adam and _STATE['adam'] is None:
        _STATE['adam'] = torch.optim.Adam
        torch.optim.Adam = _adam_factory(torch.optim.Adam)


def disable_fast_path():
    if _STATE['hook'] is not None:
        _STATE['hook'].remove()
        _STATE['hook'] = None
    if _STATE['adam'] is not None:
        torch.optim.Adam = _STATE['adam']
        _STATE['adam'] = None


if os.environ.get('FGNN_FAST_PATH', '') not in ('', '0'):
    enable_fast_path()
