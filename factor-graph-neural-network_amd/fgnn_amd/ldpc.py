"""``LDPCModel`` — the 8-layer FGNN decoder for the 96.3.963 code, and its synthetic inputs.

Same constructor / forward signature / state_dict keys (651 entries) as the class in the
reference's training script (/root/reference/train_ldpc.py:19-99): ``FactorNN`` with
dims [64,64,64,128,256,256,128,64,64], 4 edge types on the 288 parity edges plus a degree-96
"hyper-factor" touching every variable, two edge-type MLPs and an SNR regressor head.
32 message-operator calls per forward = 6144 VF+FV messages per codeword.
"""

import numpy as np
import torch

from .mpnn.assemblies import FactorNN
from .mpnn.pointwise import cast_cached, bn_spec, _RowLinear, _BatchNormAct
from .edge_mlp import EdgeMLP
from .tables import LdpcGraph

MESSAGES_PER_CODEWORD = 8 * (288 + 288 + 96 + 96)
_FAST_REGRESSOR = True      # (module switch: the regressor head through torch's modules when False)


def _edge_mlp(cin, hidden, cout):
    return EdgeMLP(cin, hidden, cout)


class LDPCModel(torch.nn.Module):
    def __init__(self, nfeature_dim, hop_order, nedge_type, with_residual=True, aggregator='max'):
        super().__init__()
        self.main = FactorNN(nfeature_dim, [hop_order, 96],
                             [64, 64, 64, 128, 256, 256, 128, 64, 64], [nedge_type, 1], 2,
                             skip_link={4: 3, 5: 2, 7: 0}, ret_high=True, aggregator=aggregator)
        self.emodel_f2v = _edge_mlp(7, 64, nedge_type)
        self.emodel_v2f = _edge_mlp(7, 64, nedge_type)
        frozen = lambda t: torch.nn.Parameter(t, requires_grad=False)
        # hyper-factor: one factor listening to all 96 variables, every variable listening to it
        self.hnn_idx_v2f = frozen(torch.arange(96, dtype=torch.int64).reshape(1, 1, 96))
        self.hnn_idx_f2v = frozen(torch.zeros(1, 96, 1, dtype=torch.int64))
        self.hetype_v2f = frozen(torch.ones(1, 1, 1, 96))
        self.hetype_f2v = frozen(torch.ones(1, 1, 96, 1))
        self.with_residual = with_residual
        self.nhop_regressor = torch.nn.Sequential(
            torch.nn.Linear(64, 128), torch.nn.BatchNorm1d(128), torch.nn.ReLU(),
            torch.nn.Linear(128, 128), torch.nn.ReLU(), torch.nn.Linear(128, 1), torch.nn.ReLU())

    def _regress(self, hop):
        """``nhop_regressor`` (train_ldpc.py:48-54,93: Linear -> BatchNorm1d -> ReLU -> Linear -> ReLU -> Linear -> ReLU on the
        hyper-factor's state).  Training on bf16 activations: the same seven steps through this package's node-wise map / BatchNorm
        kernels (statistics in the first map's epilogue, running statistics in their finaliser, weight gradients straight into the
        flat bucket) — 8 + 16 launches instead of the ~30 + ~35 five-microsecond ones autocast + autograd issue for the seven torch
        modules, all of them alone on the GPU at the step's turn-around (profiles/r05/train_step_sequence.csv; round 5 also tried
        them on the side stream: slower, the joins cost more than the overlap buys)."""
        reg = self.nhop_regressor
        spec = bn_spec(reg[1]) if (_FAST_REGRESSOR and hop.is_cuda and hop.dtype == torch.bfloat16 and self.training
                                   and torch.is_grad_enabled() and hop.shape[0] > 1) else None
        if spec is None:
            return reg(hop.float())
        bn = reg[1]
        z = _RowLinear.apply(hop.contiguous(), reg[0].weight, reg[0].bias, spec)
        a = _BatchNormAct.apply(z, bn.weight, bn.bias, bn.running_mean, bn.running_var, bn.momentum, bn.eps, 0.0, None,
                                bn.num_batches_tracked, None, None, (1, 1, 1), 0)
        a = torch.relu(_RowLinear.apply(a, reg[3].weight, reg[3].bias))
        return torch.relu(_RowLinear.apply(a, reg[5].weight, reg[5].bias)).float()

    def _fanout_weights(self, dt):
        """``hetype_f2v`` [1, 1, 96, 1] in the activations' dtype.  When its 96 weights are equal (they are ones unless a checkpoint
        says otherwise; checked on the device ONCE per version of the frozen parameter, remembered on it) the tensor handed on has a
        stride-0 node axis: that is how the operator knows — also while a hipGraph is being captured — that the hyper-factor sends
        every variable the same message and computes it once per codeword (ops.single_source_fanout)."""
        p = self.hetype_f2v
        w = cast_cached(p, dt)
        if not (self.training and torch.is_grad_enabled()):
            return w             # (inference: the one-kernel layer / block paths read the 96 weights as they lie in memory)
        memo = getattr(p, '_fgnn_equal_weights', None)
        if memo is None or memo[0] != (p._version, p.data_ptr()):
            if p.is_cuda and torch.cuda.is_current_stream_capturing():
                return w
            memo = ((p._version, p.data_ptr()), bool((p == p[:, :, :1, :]).all().item()))
            p._fgnn_equal_weights = memo
        return w[:, :, :1, :].expand(-1, -1, p.shape[2], -1) if memo[1] else w

    def forward(self, node_feature, hop_feature, nn_idx_f2v, nn_idx_v2f, efeature_f2v,
                efeature_v2f):
        B = node_feature.shape[0]
        etype_f2v = self.emodel_f2v(efeature_f2v)
        etype_v2f = self.emodel_v2f(efeature_v2f)
        hyper_in = node_feature[:, 0, :, :].detach().reshape(B, 96, 1, 1)
        dt = node_feature.dtype
        # expand (batch stride 0) instead of the reference's repeat: same values, and the kernel
        # sees "one graph shared by the batch"
        res, hops = self.main(
            node_feature, [hop_feature, hyper_in],
            [nn_idx_f2v, self.hnn_idx_f2v.expand(B, -1, -1)],
            [nn_idx_v2f, self.hnn_idx_v2f.expand(B, -1, -1)],
            [etype_f2v, self._fanout_weights(dt).expand(B, -1, -1, -1)],
            [etype_v2f, cast_cached(self.hetype_v2f, dt).expand(B, -1, -1, -1)])
        if self.with_residual:
            res = res + node_feature[:, :1, :, :]
        res = res.reshape(B, 96)
        snr_pred = self._regress(hops[1].reshape(B, -1))
        return res[:, :48].contiguous(), snr_pred


class _DecodingLoss(torch.autograd.Function):
    @staticmethod
    def forward(ctx, logits, pred, label, sigma_b, mse_weight):
        from . import _hip
        B, n = logits.shape
        from . import ops
        loss = torch.empty((), device=logits.device, dtype=torch.float32)
        ws = ops._workspace(logits.device, int(_hip.lib().fgnn_ldpc_loss_workspace_bytes()))
        _hip.check(_hip.lib().fgnn_ldpc_loss_forward(_hip._ptr(logits), _hip._ptr(label), _hip._ptr(pred), _hip._ptr(sigma_b), B, n,
                                                     _hip.dtype_code(logits), mse_weight, _hip._ptr(loss), _hip._ptr(ws), ws.numel() * 4,
                                                     _hip.stream_ptr()))
        ctx.save_for_backward(logits, pred, label, sigma_b)
        ctx.mse_weight = mse_weight
        return loss

    @staticmethod
    def backward(ctx, gloss):
        from . import _hip
        logits, pred, label, sigma_b = ctx.saved_tensors
        B, n = logits.shape
        gl, gp = torch.empty_like(logits), torch.empty_like(pred)
        gloss = gloss.float().contiguous()
        _hip.check(_hip.lib().fgnn_ldpc_loss_backward(_hip._ptr(logits), _hip._ptr(label), _hip._ptr(pred), _hip._ptr(sigma_b),
                                                      _hip._ptr(gloss), B, n, _hip.dtype_code(logits), ctx.mse_weight, _hip._ptr(gl),
                                                      _hip._ptr(gp), _hip.stream_ptr()))
        return gl, gp, None, None, None


def decoding_loss(logits, snr_pred, label, sigma_b, mse_weight=0.1):
    """The training loss of /root/reference/train_ldpc.py:222-227 — BCE-with-logits on the decoded message bits + ``mse_weight`` x MSE of the
    burst-amplitude regressor against 10^(sigma_b / 20), both means — as two short launches forward and one backward on a ROCm device
    (csrc/ldpc_datapath.hip: ldpc_loss_*_kernel; ~25 short torch launches otherwise); the same torch expression elsewhere."""
    if (logits.is_cuda and logits.dim() == 2 and logits.dtype in (torch.float32, torch.bfloat16) and snr_pred.dtype == torch.float32
            and snr_pred.numel() == logits.shape[0] and label.shape == logits.shape and sigma_b.numel() == logits.shape[0]):
        return _DecodingLoss.apply(logits.contiguous(), snr_pred.contiguous(), label.float().contiguous(),
                                   sigma_b.float().contiguous(), float(mse_weight))
    bce = torch.nn.functional.binary_cross_entropy_with_logits(logits.reshape(-1).float(), label.reshape(-1).float())
    mse = torch.nn.functional.mse_loss(snr_pred.reshape(-1).float(), torch.pow(10.0, sigma_b.float() / 20).reshape(-1))
    return bce + mse_weight * mse


def synthetic_batch(B, device, seed=0, dtype=torch.float32, shared_graph=True):
    """Synthetic LDPC inputs of the exact reference shapes: received words
    y ~ N(+-1, 1) and an SNR channel in {0..4} dB; features built as ldpc_dataset.py:92-106 does.
    Returns (node_feature[B,2,96,1], hop_feature[B,6,48,1], nn_idx_f2v[B,96,3], nn_idx_v2f[B,48,6],
    efeature_f2v[B,7,96,3], efeature_v2f[B,7,48,6], label[B,48], sigma_b[B])."""
    g = LdpcGraph()
    gen = torch.Generator(device='cpu').manual_seed(seed)
    bits = torch.randint(0, 2, (B, 96), generator=gen).float()
    y = (1 - 2 * bits) + torch.randn(B, 96, generator=gen)
    snr = torch.randint(0, 5, (B,), generator=gen).float()
    sigma_b = torch.randint(0, 6, (B,), generator=gen).float()
    y, snr, bits, sigma_b = y.to(device), snr.to(device), bits.to(device), sigma_b.to(device)
    f2v = torch.from_numpy(g.var_to_factors).to(device)
    v2f = torch.from_numpy(g.factor_to_vars).to(device)
    hop = y[:, v2f]                                                # [B,48,6]
    node = torch.stack([y, snr[:, None].expand(B, 96)], 1).unsqueeze(-1)
    ef_f2v = torch.cat([hop[:, f2v], y[:, :, None, None].expand(B, 96, 3, 1)], 3)   # [B,96,3,7]
    ef_v2f = torch.cat([hop[:, :, None, :].expand(B, 48, 6, 6), hop[:, :, :, None]], 3)
    if shared_graph:
        idx_f2v, idx_v2f = f2v.unsqueeze(0).expand(B, -1, -1), v2f.unsqueeze(0).expand(B, -1, -1)
    else:
        idx_f2v, idx_v2f = f2v.unsqueeze(0).repeat(B, 1, 1), v2f.unsqueeze(0).repeat(B, 1, 1)
    return (node.to(dtype).contiguous(), hop.permute(0, 2, 1).unsqueeze(-1).to(dtype).contiguous(),
            idx_f2v, idx_v2f,
            ef_f2v.permute(0, 3, 1, 2).to(dtype).contiguous(),
            ef_v2f.permute(0, 3, 1, 2).to(dtype).contiguous(),
            bits[:, :48].contiguous(), sigma_b)
