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

What the switch cannot give an unchanged script is the hipGraph replay (``graph.StepGraph`` needs the step as a closure); the
eager step is launch-bound at ~25-30 ms.  ``disable_fast_path()`` undoes both patches (modules already optimised stay so).
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
    """Every parameter a trainable CUDA f32 tensor, one group, nothing exotic asked for."""
    plain = bool(params) and all(torch.is_tensor(p) for p in params)
    return (plain and not amsgrad and not kw and
            all(p.is_cuda and p.dtype == torch.float32 and p.requires_grad for p in params))


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
