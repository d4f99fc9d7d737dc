"""Model bodies that call the message operator: ``mp_sequential``, ``factor_mpnn``, ``FactorNN``.

Same constructor signatures, ``forward`` signatures and state_dict keys as
  mp_sequential   /root/reference/lib/model/mpnn/sequential.py:8-39
  factor_mpnn     /root/reference/lib/model/mpnn/factor_mpnn.py:8-133
  FactorNN        /root/reference/lib/model/mpnn/factor_mpnn_sp.py:25-178
so the reference's train_*.py scripts construct and call them unchanged; every VF/FV
message inside runs through the fused HIP operator.
"""
import contextlib

import torch

from .blocks import iid_mapping, iid_mapping_bn, iid_mapping_in, mp_conv_residual
from .pointwise import NodeInstanceNorm, PointwiseConv2d as _Conv, add_all, instnorm_relu_dot
from .message_op import base_mp_nn, mp_conv_type, mp_conv_v2


FUSE_EVAL_LAYERS = True      # inference: a 64 -> 64 bf16 FactorNN layer of the LDPC shape runs as ONE kernel (csrc/factor_layer_fwd.hip)


def _call(module, x, nn_idx, etype, addend=None):
    """The reference's dispatch contract: graph-aware modules get (x, nn_idx, etype).  ``addend`` is this
    build's extra: blocks that end in a fused BatchNorm+activation kernel add it there (``acc + block(x)``
    without a separate elementwise pass); for every other module it is added explicitly."""
    if isinstance(module, (mp_conv_v2, mp_conv_residual)):
        return module(x, nn_idx, etype, addend=addend)
    y = module(x, nn_idx, etype) if isinstance(module, base_mp_nn) else module(x)
    return add_all(y, addend() if callable(addend) else addend)


class mp_sequential(base_mp_nn):
    def __init__(self, *module_list):
        super().__init__()
        self.module_list = list(module_list)
        for i, m in enumerate(self.module_list):
            self.add_module(str(i), m)

    def forward(self, node_feature, *argv):
        extras = []
        for m in self.module_list:
            out = m(node_feature, *argv) if isinstance(m, base_mp_nn) else m(node_feature)
            if isinstance(out, tuple):
                extras.extend(out[1:])
                out = out[0]
            node_feature = out
        return (node_feature, extras) if extras else node_feature


class _SplitNodes(torch.autograd.Function):
    """``both[:, :, :n]``, ``both[:, :, n:]`` as two views whose gradients come back in ONE concatenation.  Autograd's own
    backward of the two slices (factor_mpnn.py:108-112) is two zero-filled tensors of ``both``'s size, two strided copies and
    an add: five activation-sized kernels per block instead of one."""

    @staticmethod
    def forward(ctx, both, n):
        ctx.n, ctx.shape, ctx.cl = n, both.shape, both.stride(1) == 1
        ctx.set_materialize_grads(False)
        return both[:, :, :n, :], both[:, :, n:, :]

    @staticmethod
    def backward(ctx, g_nodes, g_factors):
        if g_nodes is None and g_factors is None:
            return None, None
        B, C, N, W = ctx.shape
        ref = g_nodes if g_nodes is not None else g_factors
        if g_nodes is None:
            g_nodes = ref.new_zeros((B, C, ctx.n, W))
        if g_factors is None:
            g_factors = ref.new_zeros((B, C, N - ctx.n, W))
        from .. import ops as _ops
        g = _ops.concat2(g_nodes, g_factors, 2) if ctx.cl else torch.cat([g_nodes, g_factors], dim=2)
        return (g.contiguous(memory_format=torch.channels_last) if ctx.cl else g), None


class factor_mpnn(torch.nn.Module):
    """Synthetic-PGM body.  Per layer and factor type the variables and that type's factors are
    concatenated along the node axis and pushed through one mp block (rows < nnode gather from
    factors = F->V, rows >= nnode gather from variables = V->F), then split again."""

    def __init__(self, node_feature_dim, factor_feature_dim_list, dim_mapping_list, netype_list,
                 gnn_immediate_dim=64, max_mpnn_dim=64, final_filter=None, skip_link={}):
        super().__init__()
        self.node_feature_dim = node_feature_dim
        self.map_dim = dim_mapping_list[0]
        self.nfactor_types = len(factor_feature_dim_list)
        self.final_filter = final_filter
        self.skip_link = skip_link
        self.mapping_modules = []
        for i, din in enumerate([node_feature_dim] + list(factor_feature_dim_list)):
            m = iid_mapping(din, self.map_dim)
            self.add_module('mapping_modules_%d' % i, m)
            self.mapping_modules.append(m)
        self.mp_nn_modules, self.mp_merge_modules = [], []
        nlayers = len(dim_mapping_list) - 1
        for L in range(nlayers):
            nin, nout = dim_mapping_list[L], dim_mapping_list[L + 1]
            row = []
            for j in range(self.nfactor_types):
                if nin == nout:
                    m = mp_conv_residual(nin, gnn_immediate_dim, netype_list[j])
                elif nin <= max_mpnn_dim and nout <= max_mpnn_dim:
                    m = mp_conv_v2(nin, nout, netype_list[j])
                else:
                    m = torch.nn.Sequential(_Conv(nin, nout, 1), NodeInstanceNorm(relu=True),
                                            torch.nn.Identity())
                self.add_module('mp_nn_%d_%d' % (L, j), m)
                row.append(m)
            self.mp_nn_modules.append(row)
            width = nout * self.nfactor_types
            if L < nlayers - 1:
                merge = iid_mapping_bn(width, nout)
            else:
                merge = torch.nn.Sequential(
                    _Conv(width, 256, 1, bias=True), torch.nn.BatchNorm2d(256),
                    torch.nn.LeakyReLU(), _Conv(256, 256, 1, bias=True),
                    torch.nn.LeakyReLU(), _Conv(256, nout, 1, bias=True))
            self.add_module('merge_module_%d' % L, merge)
            self.mp_merge_modules.append(merge)

    def forward(self, node_features, factor_features, graph_structures):
        nnode = node_features.shape[2]
        nfeat = self.mapping_modules[0](node_features)
        ffeat = [m(f) for f, m in zip(factor_features, self.mapping_modules[1:])]
        history = []
        from ..ops import fan_out
        from .. import ops as _ops
        track = torch.is_grad_enabled() and nfeat.requires_grad
        for L, row in enumerate(self.mp_nn_modules):
            to_nodes, to_factors = [], []
            # the variables' state feeds every factor type's block: one alias per consumer, so that their gradients meet in a
            # single n-way sum (ops.fan_out) instead of autograd's pairwise adds
            nf_c = fan_out(nfeat, len(row)) if (track and len(row) > 1 and L not in self.skip_link.values()) else [nfeat] * len(row)
            for j, m in enumerate(row):
                # channel-fastest, like every activation on this path (the operator kernels read node rows of channels)
                both = _ops.concat2(nf_c[j], ffeat[j], 2)            # (one launch: torch.cat is a strided copy kernel per input)
                nn_idx, etype = graph_structures[j]
                both = _call(m, both, nn_idx, etype)
                if track:
                    nd, fc = _SplitNodes.apply(both, nnode)
                else:
                    nd, fc = both[:, :, :nnode, :], both[:, :, nnode:, :]
                to_nodes.append(nd)
                to_factors.append(fc)
            nfeat = self.mp_merge_modules[L](_ops.concat2(to_nodes[0], to_nodes[1], 1) if len(to_nodes) == 2 else torch.cat(to_nodes, dim=1))
            ffeat = to_factors
            if L in self.skip_link:
                pn, pf = history[self.skip_link[L]]
                nfeat = nfeat + pn
                ffeat = [a + b for a, b in zip(ffeat, pf)]
            history.append([nfeat, ffeat])
        if self.final_filter is not None:
            nfeat = self.final_filter(nfeat, node_features)
        return nfeat, ffeat


_V2V_MAIN = {4, 6}     # layers whose v2v map stays on the main stream.  Round 4 (18.25 ms with none): layers 0,1,7: 18.27; 2-6: 18.47; 3-5: 18.54;
# all: 18.48 — the map belonged on the side stream.  Round 6: device stamps inside the replayed step (profiles/r06/graph_step_stamps.txt) show
# the SIDE chain as the longer one of every layer since the hyper-factor's message became one row per codeword; re-measured (three
# interleaved runs each, 12.78 - 12.83 ms with none): {4}: 12.62 - 12.66; {4, 6}: 12.58 - 12.59; {4, 5, 6}: 12.57 - 12.59; {2, 4, 6}: 12.61;
# {4, 6, 7} / {1, 4, 6} / {0, 4, 6}: +0.03 - 0.09 over {4, 6}; {3}: 12.74; {5}: 12.71.


_F2F_SIDE = True       # the parity factors' f2f map on the side stream (round 5: -0.1 ms)
# (Issue order of a layer's two chains — round 6, measured: the capture stream's V->F parity block issued BEFORE the side chain makes the
# step 0.85 ms slower (13.57 against 12.74 ms): the runtime continues a graph branch with the FIRST child of its fork node, and the chain
# that continues the fork's branch must be the side chain.  profiles/r06/README.md "What the replayed graph really does".)
_FAC_MERGE_SIDE = 1    # 1 = the factor states' gradient merge on the side stream (2: the variables' instead; round 5: -0.13 ms with 1)


class FactorNN(torch.nn.Module):
    """LDPC body ("sp" variant): per layer and factor type an F->V block (factors are the
    sources, variables the destinations) and a V->F block, plus node-wise v2v / f2f maps."""

    def __init__(self, node_feature_dim, factor_feature_dim_list, dim_mapping_list, netype_list,
                 nclass=2, gnn_immediate_dim=64, max_mpnn_dim=128, final_filter=None,
                 skip_link={}, aggregator='max', ret_high=False):
        super().__init__()
        self.node_feature_dim = node_feature_dim
        self.map_dim = dim_mapping_list[0]
        self.final_filter = final_filter
        self.dim_mapping_list = dim_mapping_list
        self.nfactor_types = len(factor_feature_dim_list)
        self.skip_link = skip_link
        self.ret_high = ret_high
        self.node_mapping_module = iid_mapping(node_feature_dim, self.map_dim)
        self.factor_mapping_modules = []
        for j, din in enumerate(factor_feature_dim_list):
            m = iid_mapping_bn(din, self.map_dim)
            self.add_module('factor_mapping_modules_%d' % j, m)
            self.factor_mapping_modules.append(m)

        def mp_block(nin, nout, netype):
            kw = dict(extension=mp_conv_type.NO_EXTENSION)
            if nin == nout:
                return mp_conv_residual(nin, gnn_immediate_dim, netype, with_residual=False,
                                        aggregator=aggregator, **kw)
            if nin <= max_mpnn_dim and nout <= max_mpnn_dim:
                return mp_conv_v2(nin, nout, netype, aggregtor=aggregator, **kw)
            return mp_conv_residual(nin, gnn_immediate_dim, netype, with_residual=False,
                                    nout=nout, aggregator=aggregator, **kw)

        self.f2v_modules, self.v2f_modules, self.f2f_modules, self.v2v_modules = [], [], [], []
        for L in range(len(dim_mapping_list) - 1):
            nin, nout = dim_mapping_list[L], dim_mapping_list[L + 1]
            v2v = iid_mapping_in(nin, nout)
            self.add_module('v2v_%d' % L, v2v)
            self.v2v_modules.append(v2v)
            f2v_row, v2f_row, f2f_row = [], [], []
            for j in range(self.nfactor_types):
                f2f_row.append(iid_mapping_in(nin, nout))
                f2v_row.append(mp_block(nin, nout, netype_list[j]))
                v2f_row.append(mp_block(nin, nout, netype_list[j]))
                self.add_module('f2v_%d_%d' % (L, j), f2v_row[-1])
                self.add_module('v2f_%d_%d' % (L, j), v2f_row[-1])
                self.add_module('f2f_%d_%d' % (L, j), f2f_row[-1])
            self.f2f_modules.append(f2f_row)
            self.f2v_modules.append(f2v_row)
            self.v2f_modules.append(v2f_row)
        final_dim = nclass if nclass > 2 else 1
        self.final_classifier = torch.nn.Sequential(
            _Conv(dim_mapping_list[-1], 128, 1), NodeInstanceNorm(relu=True),
            torch.nn.Identity(), _Conv(128, final_dim, 1, bias=True))

    def _classify(self, var):
        """``final_classifier`` (factor_mpnn_sp.py:104-108); its InstanceNorm2d -> ReLU -> Identity -> Conv2d(128, 1, 1) as one kernel
        where the shape allows (two classes: one logit per variable)."""
        fc = self.final_classifier
        if len(fc) == 4 and isinstance(fc[1], NodeInstanceNorm) and fc[1].relu and isinstance(fc[2], torch.nn.Identity):
            h = fc[0](var)
            out = instnorm_relu_dot(h, fc[3])
            return out if out is not None else fc[3](fc[2](fc[1](h)))
        return fc(var)

    def _fused_layer(self, L, var, fac, skip, nn_idx_f2v, nn_idx_v2f, etype_f2v, etype_v2f):
        """Inference: layer ``L`` as one kernel (csrc/factor_layer_fwd.hip) when it is a 64 -> 64 layer of the
        LDPC shape — bf16 channel-fastest states of 96 variables / 48 degree-6 checks / one hyper-factor, neighbour tables
        shared by the batch, four foldable blocks.  Returns (new_var, new_fac) or None (the staged path then runs)."""
        import ctypes
        from .. import _hip
        from .blocks import _is_identity_list
        from .pointwise import refresh_in_place, state_epoch
        dims = self.dim_mapping_list
        if (not FUSE_EVAL_LAYERS or self.training or torch.is_grad_enabled() or self.nfactor_types != 2
                or dims[L] != 64 or dims[L + 1] != 64 or not var.is_cuda or var.dtype != torch.bfloat16):
            return None
        B = var.shape[0]
        blocks = [self.v2f_modules[L][0], self.f2v_modules[L][0], self.v2f_modules[L][1], self.f2v_modules[L][1]]
        nets = [4, 4, 1, 1]
        maps = [self.v2v_modules[L], self.f2f_modules[L][0]]
        if not all(isinstance(m, mp_conv_residual) and m.fusable_for_inference() and m.mp_conv.nedge_types == n
                   and m.conv1[0].in_channels == 64 and m.conv2[0].out_channels == 64 for m, n in zip(blocks, nets)):
            return None
        if len({float(m.conv1[1].slope) for m in blocks}) != 1:
            return None
        if not all(isinstance(m, iid_mapping_in) and isinstance(m.main[0], _Conv) and m.main[0].in_channels == 64
                   and m.main[0].out_channels == 64 and isinstance(m.main[1], NodeInstanceNorm) and m.main[1].relu for m in maps):
            return None
        cl = lambda t, n: tuple(t.shape) == (B, 64, n, 1) and t.dtype == var.dtype and t.permute(0, 2, 3, 1).is_contiguous()
        if not (cl(var, 96) and cl(fac[0], 48) and tuple(fac[1].shape) == (B, 64, 1, 1) and fac[1].dtype == var.dtype):
            return None
        if skip is not None and not (cl(skip[0], 96) and cl(skip[1][0], 48) and tuple(skip[1][1].shape) == (B, 64, 1, 1)):
            return None
        iv, if_ = nn_idx_v2f[0], nn_idx_f2v[0]
        if not (tuple(iv.shape) == (B, 48, 6) and tuple(if_.shape) == (B, 96, 3) and (B == 1 or (iv.stride(0) == 0 and if_.stride(0) == 0))
                and iv.dtype == torch.int64 and if_.dtype == torch.int64):
            return None
        hv2f, hf2v = nn_idx_v2f[1], nn_idx_f2v[1]
        if not (tuple(hv2f.shape) == (B, 1, 96) and tuple(hf2v.shape) == (B, 96, 1) and _is_identity_list(hv2f)):
            return None
        ev, ef, hev, hef = etype_v2f[0][L], etype_f2v[0][L], etype_v2f[1][L], etype_f2v[1][L]

        def et_ok(e, M, k):
            q = e.permute(0, 2, 3, 1)
            return (tuple(e.shape) == (B, 4, M, k) and e.dtype == var.dtype
                    and (q.is_contiguous() or (e.stride(0) == 0 and q[0].is_contiguous())) and e.stride(0) % 4 == 0)
        if not (et_ok(ev, 48, 6) and et_ok(ef, 96, 3)):
            return None
        if not (tuple(hev.shape) == (B, 1, 1, 96) and tuple(hef.shape) == (B, 1, 96, 1) and hev.dtype == var.dtype and hef.dtype == var.dtype
                and (B == 1 or (hev.stride(0) == 0 and hef.stride(0) == 0)) and hev.stride(3) == 1 and hef.stride(2) == 1):
            return None
        # packed parameters: maps, then per block W1, s1, t1, filters, s2, t2, W2, s3, t3 (refreshed in place when anything moved)
        folded = [m.folded_for_inference(var.device) for m in blocks]
        key = tuple(m._fuse_key for m in blocks) + tuple(m.main[0].weight._version for m in maps) + (state_epoch(),)
        cache = self.__dict__.setdefault('_layer_pack', {})
        ent = cache.get(L)
        if ent is None or ent[0] != key:
            flat = torch.cat([m.main[0].weight.detach().float().reshape(-1) for m in maps] +
                             [t.reshape(-1) for f in folded for t in f])
            assert flat.numel() == int(_hip.lib().fgnn_factor_layer_param_count()), flat.numel()
            ent = cache[L] = (key, refresh_in_place(ent[1] if ent is not None else None, (flat,)))
        params = ent[1][0]
        new_var = torch.empty((B, 96, 1, 64), device=var.device, dtype=var.dtype).permute(0, 3, 1, 2)
        new_f0 = torch.empty((B, 48, 1, 64), device=var.device, dtype=var.dtype).permute(0, 3, 1, 2)
        new_f1 = torch.empty((B, 1, 1, 64), device=var.device, dtype=var.dtype).permute(0, 3, 1, 2)
        f1 = fac[1].reshape(B, 64)
        s1 = skip[1][1].reshape(B, 64) if skip is not None else None
        if not f1.is_contiguous():
            f1 = f1.contiguous()
        if s1 is not None and not s1.is_contiguous():
            s1 = s1.contiguous()
        P = _hip._ptr
        rcs = []
        # algorithmic bytes: the state read and written once (+ the skip terms), the parity edge types, the parameters
        nstate = B * (96 + 48 + 1) * 64 * 2
        nbytes = nstate * (3 if skip is not None else 2) + 2 * 288 * 4 * 2 * (B if ev.stride(0) else 1) + params.numel() * 4
        from .. import ops as _ops
        _ops.timed('factor_layer_fwd_kernel', nbytes, lambda: rcs.append(_hip.lib().fgnn_factor_layer_forward(
            B, P(var), P(fac[0]), P(f1), P(skip[0]) if skip is not None else None, P(skip[1][0]) if skip is not None else None,
            P(s1), P(iv), iv.stride(1), iv.stride(2), P(if_), if_.stride(1), if_.stride(2), P(ev), ev.stride(0), P(ef), ef.stride(0),
            P(hev), P(hef), P(params), 1, float(blocks[0].conv1[1].slope), P(new_var), P(new_f0), P(new_f1), _hip.stream_ptr())),
            nflops=B * 2 * 64 * (64 * (96 * 5 + 48 * 3 + 3) + 256 * (96 + 48) + 64 * 96))
        rc = rcs[0]
        if rc == _hip.EUNSUPPORTED:
            return None
        _hip.check(rc)
        return new_var, [new_f0, new_f1]

    def mpnn_forward(self, mpnn, node_feature, nn_idx, efeature):
        return _call(mpnn, node_feature, nn_idx, efeature)

    def forward(self, node_feature, hop_features, nn_idx_f2v, nn_idx_v2f, etype_f2v, etype_v2f):
        var = self.node_mapping_module(node_feature)
        fac = [m(f) for f, m in zip(hop_features, self.factor_mapping_modules)]
        from ..ops import fan_out
        from .. import ops as _ops
        nft = self.nfactor_types
        nL = len(self.v2f_modules)
        # the neighbour tables serve all layers: convert once, and recognise B equal copies of one graph (what the
        # reference's DataLoader collates, train_ldpc.py:209-216) as a table shared by the batch
        nn_idx_f2v = [_ops.shared_graph_view(t.long()) for t in nn_idx_f2v]
        nn_idx_v2f = [_ops.shared_graph_view(t.long()) for t in nn_idx_v2f]
        # the edge types feed every layer: one alias per layer, so their gradients meet in one n-way sum
        etype_f2v = [fan_out(e, nL) for e in etype_f2v]
        etype_v2f = [fan_out(e, nL) for e in etype_v2f]
        skip_src = set(self.skip_link.values())
        history = {}
        for L in range(nL):
            same_width = self.dim_mapping_list[L] == self.dim_mapping_list[L + 1]
            res = 1 if same_width else 0
            keep = 1 if (L - 1) in skip_src else 0       # the incoming state is also a later layer's skip input
            if not torch.is_grad_enabled() and not self.training:
                if keep:
                    history[L - 1] = [var, list(fac)]
                fused = self._fused_layer(L, var, fac, history[self.skip_link[L]] if L in self.skip_link else None,
                                          nn_idx_f2v, nn_idx_v2f, etype_f2v, etype_v2f)
                if fused is not None:
                    var, fac = fused
                    continue
            # every state feeds several consumers (v2v / f2f map, the message blocks, the residual, a skip link):
            # hand each its own alias so that the backward sums their gradients in one kernel (ops.fan_out)
            if _FAC_MERGE_SIDE == 2 and _ops.SIDE_STREAM and nft > 1 and var.is_cuda and torch.is_grad_enabled():
                with torch.cuda.stream(_ops.side_stream(var.device)):
                    var_c = fan_out(var, 1 + nft + res + keep)
            else:
                var_c = fan_out(var, 1 + nft + res + keep)
            if _FAC_MERGE_SIDE == 1 and _ops.SIDE_STREAM and nft > 1 and var.is_cuda and torch.is_grad_enabled():
                # round 5: a fan-out's backward (ops.FanBox.merge: the state's gradient as one product, 50-270 us) runs on the stream its
                # forward was issued on.  Both states' merges used to sit back to back on the main stream at the end of a layer's
                # backward while the side stream idled: the factor states' handles are made under the side stream (no kernel in the
                # forward: views), so their merge runs there, beside the variables'.
                with torch.cuda.stream(_ops.side_stream(var.device)):
                    fac_c = [fan_out(f, 2 + res + keep) for f in fac]
            else:
                fac_c = [fan_out(f, 2 + res + keep) for f in fac]
            if keep:
                history[L - 1] = [var_c.pop(), [fc.pop() for fc in fac_c]]
            skip = history[self.skip_link[L]] if L in self.skip_link else None
            # new state = node-wise map + every block's messages (+ old state when the width is kept) (+ skip link);
            # the running sum, residual and skip terms are added by the closing BatchNorm+activation kernel of the
            # last block of each chain.  The chains of the factor types beyond the first (the hyper-factor: ~200 short
            # launches per layer) touch only their own state and the shared variables, so they are issued on a side
            # stream and overlap the parity-check chain; their messages to the variables (h) join as extra addends.  The
            # variables' own node-wise map (v2v) goes with them: that balances the two streams (21.2 -> 20.9 ms; moving the
            # parity factors' f2f map as well serialises the join and costs 2 ms).
            two = _ops.SIDE_STREAM and nft > 1 and var.is_cuda
            if two:
                _ops.SIDE_ACTIVE = True        # (weight gradients are only parked once a second stream exists: ops.defer_wgrad)
            new_fac, h = [None] * nft, []
            if two:
                main, side = torch.cuda.current_stream(var.device), _ops.side_stream(var.device)
                side.wait_stream(main)
                if _ops.STAMPS is not None:
                    _ops.stamp('L%d main0' % L)
                    with torch.cuda.stream(side):
                        _ops.stamp('L%d side0' % L)
            f2f_side = two and _F2F_SIDE
            if f2f_side:
                # round 5 (tuning knob): with the hyper-factor's message carried as a per-sample vector the side branch is the lighter
                # one (11.1 vs 15.0 ms busy, gpurun_out/r05b/timeline): the parity factors' own node-wise map can move over too — FIRST
                # in the side stream's order; the main stream asks for it (an event, not a full join) only where the V->F block's
                # tail adds it
                with torch.cuda.stream(side):
                    nf0 = self.f2f_modules[L][0](fac_c[0][0])
                    f2f_done = torch.cuda.Event()
                    f2f_done.record(side)
            for j in range(1, nft):
                with (torch.cuda.stream(side) if two else contextlib.nullcontext()):
                    nf = self.f2f_modules[L][j](fac_c[j][0])
                    new_fac[j] = _call(self.v2f_modules[L][j], var_c[1 + j], nn_idx_v2f[j], etype_v2f[j][L],
                                       addend=[nf, fac_c[j][-1] if same_width else None, skip[1][j] if skip else None])
                    h.append(_call(self.f2v_modules[L][j], fac_c[j][1], nn_idx_f2v[j], etype_f2v[j][L]))
            v2v_main = L in _V2V_MAIN and self.training and torch.is_grad_enabled()      # (inference: the capture stream's two one-kernel blocks are the longer chain)
            with (torch.cuda.stream(side) if (two and not v2v_main) else contextlib.nullcontext()):
                new_var = self.v2v_modules[L](var_c[0])        # the variables' node-wise map rides with the side branch
                if two and _ops.STAMPS is not None:
                    _ops.stamp('L%d side_end' % L)
            if f2f_side:
                def fac_addends(nf=nf0, f2f_done=f2f_done, same_width=same_width, skip=skip, fac_c=fac_c):
                    main.wait_event(f2f_done)
                    return [nf, fac_c[0][-1] if same_width else None, skip[1][0] if skip else None]
                new_fac[0] = _call(self.v2f_modules[L][0], var_c[1], nn_idx_v2f[0], etype_v2f[0][L], addend=fac_addends)
            else:
                nf = self.f2f_modules[L][0](fac_c[0][0])
                new_fac[0] = _call(self.v2f_modules[L][0], var_c[1], nn_idx_v2f[0], etype_v2f[0][L],
                                   addend=[nf, fac_c[0][-1] if same_width else None, skip[1][0] if skip else None])
            def joined(new_var=new_var, h=h, L=L, same_width=same_width, skip=skip, var_c=var_c):
                # called by the block right before its closing BatchNorm consumes the addends: only there does the main
                # stream wait for the side branch
                if two:
                    main.wait_stream(side)
                return [new_var] + h + [var_c[-1] if same_width else None, skip[0] if skip else None]
            if two and _ops.STAMPS is not None:
                _ops.stamp('L%d main_v2f_end' % L)
            new_var = _call(self.f2v_modules[L][0], fac_c[0][1], nn_idx_f2v[0], etype_f2v[0][L], addend=joined)
            if two and _ops.STAMPS is not None:
                _ops.stamp('L%d joined' % L)
            var, fac = new_var, new_fac
        out = self._classify(var)
        if self.final_filter is not None:
            out = self.final_filter(out, node_feature)
        return (out, fac) if self.ret_high else out
