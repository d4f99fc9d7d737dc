"""The remaining exports of the reference's ``lib.model.mpnn`` package: ``global_pooling`` and
``mp_ensemble`` (plus ``parallel_net`` / ``identity_module`` from the same family).

None of the four training scripts constructs them — ``train_syn_fixed_pw_hop.py:9`` only *imports*
``global_pooling`` — but the import line is part of the drop-in surface, so they exist here with the
reference's constructor / forward contracts:
  global_pooling   /root/reference/lib/model/mpnn/pooling.py:11-47
  mp_ensemble      /root/reference/lib/model/mpnn/ensemble.py:8-19
  parallel_net     /root/reference/lib/model/mpnn/parallel_net.py:14-44
  identity_module  /root/reference/lib/model/mpnn/identity.py:4-13
They are graph-level glue around whatever modules the caller passes in (the message operator inside
those modules is the HIP one); their own bodies are a pool / broadcast / concat and stay torch ops.
"""
import torch

from .message_op import base_mp_nn


def max_pool(feature, dim):
    return feature.max(dim=dim, keepdim=True)[0]


def _pool_nodes(x):
    return max_pool(x, dim=2)


class global_pooling(base_mp_nn):
    """[per-node features ; pooled graph feature broadcast to every node] along the channel axis."""

    def __init__(self, orig_mapper=None, gfeature_mapper=None, pool_func=_pool_nodes):
        super().__init__()
        self.orig_mapper = orig_mapper            # nn.Module attributes register themselves under these names
        self.gfeature_mapper = gfeature_mapper
        self.pool_func = pool_func

    def forward(self, node_feature, nn_idx, etype):
        nnodes = node_feature.shape[2]
        pooled = self.pool_func(node_feature)     # of the INPUT features, before orig_mapper (pooling.py:34)
        if self.orig_mapper is not None:
            node_feature = self.orig_mapper(node_feature, nn_idx, etype)
        if self.gfeature_mapper is not None:
            pooled = self.gfeature_mapper(pooled)
        return torch.cat([node_feature, pooled.expand(-1, -1, nnodes, -1)], dim=1)


class mp_ensemble(base_mp_nn):
    """model3(cat(model1 on graph 1, model2 on graph 2))."""

    def __init__(self, model1, model2, model3):
        super().__init__()
        self.model1, self.model2, self.model3 = model1, model2, model3

    def forward(self, node_feature, nn_idx, etype, *argv):
        a = self.model1(node_feature, nn_idx, etype)
        b = self.model2(node_feature, *argv)
        return self.model3(torch.cat((a, b), dim=1))


def add_agg(*outs):
    total = outs[0]
    for o in outs[1:]:
        total = total + o
    return total


class parallel_net(base_mp_nn):
    """n branches over the same input, combined by ``aggregator`` (default: sum)."""

    def __init__(self, *module_list, aggregator=add_agg):
        super().__init__()
        self.aggregator = aggregator
        self.module_list = list(module_list)
        for i, m in enumerate(self.module_list):
            self.add_module(str(i), m)

    def forward(self, node_feature, nn_idx=None, etype=None):
        outs = [m(node_feature, nn_idx, etype) if isinstance(m, base_mp_nn) else m(node_feature)
                for m in self.module_list]
        return self.aggregator(*outs)


class identity_module(torch.nn.Module):
    def forward(self, input):
        return input
