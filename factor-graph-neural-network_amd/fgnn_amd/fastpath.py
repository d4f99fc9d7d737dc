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

import torch

_STATE = {'hook': None, 'adam': None}


def _is_edge_model(m):
    if not isinstance(m, torch.nn.Sequential) or len(m) != 3:
        return False
    c1, act, c2 = m[0], m[1], m[2]
    ok = lambda c: isinstance(c, torch.nn.Conv2d) and c.kernel_size == (1, 1) and c.stride == (1, 1) and c.groups == 1
    return ok(c1) and isinstance(act, torch.nn.ReLU) and ok(c2) and c1.out_channels == c2.in_channels


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
    inner = module.forward

    def fast_forward(*args, **kwargs):
        cast = lambda t: t.to(torch.bfloat16) if torch.is_tensor(t) and t.is_cuda and t.dtype == torch.float32 else t
        back = lambda t: t.float() if torch.is_tensor(t) and t.dtype == torch.bfloat16 else t
        with torch.autocast('cuda', dtype=torch.bfloat16):
            out = inner(*[cast(a) for a in args], **{k: cast(v) for k, v in kwargs.items()})
        return tuple(back(o) for o in out) if isinstance(out, (tuple, list)) else back(out)

    module.forward = fast_forward
    return module


def _pre_hook(module, args):
    from .mpnn import FactorNN
    if getattr(module, '_fgnn_fast_seen', False):      # one look per module: the hook runs in front of every module call
        return None
    module._fgnn_fast_seen = True
    if not isinstance(module, FactorNN) and any(isinstance(c, FactorNN) for c in module.children()):
        fast_path(module)          # (this call still runs the old forward — torch bound it before the hooks; the next one is fast)
    return None


class FastAdam(torch.optim.Optimizer):
    """``torch.optim.Adam``'s update (no amsgrad) on flat parameter / gradient buffers: one kernel per step."""

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


def _adam_factory(stock):
    def Adam(params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=0, amsgrad=False, **kw):
        """torch.optim.Adam, or FastAdam when every parameter is a CUDA f32 tensor (and nothing exotic is asked for)."""
        params = list(params)
        plain = bool(params) and all(torch.is_tensor(p) for p in params)
        if plain and not amsgrad and not kw and all(p.is_cuda and p.dtype == torch.float32 for p in params):
            return FastAdam(params, lr=lr, betas=betas, eps=eps, weight_decay=weight_decay)
        return stock(params, lr=lr, betas=betas, eps=eps, weight_decay=weight_decay, amsgrad=amsgrad, **kw)
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
