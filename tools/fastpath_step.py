#!/usr/bin/env python3
"""What `fgnn_amd.enable_fast_path()` buys a script that composes its model and optimizer the way the reference's train_ldpc.py
does (FactorNN from the `lib.model.mpnn` shim + Sequential(Conv2d, ReLU, Conv2d) edge models + torch.optim.Adam, eager steps,
per-sample graph tables as the DataLoader collates them): ms per training step without / with the switch.
    python tools/fastpath_step.py [batch] [steps]"""
import contextlib, io, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'factor-graph-neural-network_amd'))
import torch
import fgnn_amd
from fgnn_amd.ldpc import synthetic_batch
from lib.model.mpnn import FactorNN
dev = torch.device('cuda:0')
B = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 8


class ScriptModel(torch.nn.Module):                       # train_ldpc.py:19-99 restated: the tables its forward builds with .repeat included
    def __init__(self):
        super().__init__()
        self.main = FactorNN(2, [6, 96], [64, 64, 64, 128, 256, 256, 128, 64, 64], [4, 1], 2,
                             skip_link={4: 3, 5: 2, 7: 0}, ret_high=True, aggregator='max')
        mk = lambda: torch.nn.Sequential(torch.nn.Conv2d(7, 64, 1), torch.nn.ReLU(inplace=True), torch.nn.Conv2d(64, 4, 1))
        self.emodel_f2v, self.emodel_v2f = mk(), mk()
        frozen = lambda t: torch.nn.Parameter(t, requires_grad=False)
        self.hnn_idx_v2f = frozen(torch.arange(96).reshape(1, 1, 96))
        self.hnn_idx_f2v = frozen(torch.zeros(1, 96, 1, dtype=torch.int64))
        self.hetype_v2f, self.hetype_f2v = frozen(torch.ones(1, 1, 1, 96)), frozen(torch.ones(1, 1, 96, 1))
        self.nhop_regressor = torch.nn.Sequential(torch.nn.Linear(64, 128), torch.nn.BatchNorm1d(128), torch.nn.ReLU(),
                                                  torch.nn.Linear(128, 128), torch.nn.ReLU(), torch.nn.Linear(128, 1), torch.nn.ReLU())

    def forward(self, node_feature, hop_feature, nn_idx_f2v, nn_idx_v2f, efeature_f2v, efeature_v2f):
        etype_f2v, etype_v2f = self.emodel_f2v(efeature_f2v), self.emodel_v2f(efeature_v2f)
        with torch.no_grad():
            bsize = node_feature.shape[0]
            nhop = node_feature[:, 0, :, :].reshape(bsize, 96, 1, 1)
        res, nhops = self.main(node_feature, [hop_feature, nhop],
                               [nn_idx_f2v, self.hnn_idx_f2v.repeat(bsize, 1, 1)], [nn_idx_v2f, self.hnn_idx_v2f.repeat(bsize, 1, 1)],
                               [etype_f2v, self.hetype_f2v.repeat(bsize, 1, 1, 1)], [etype_v2f, self.hetype_v2f.repeat(bsize, 1, 1, 1)])
        res = (res + node_feature[:, :1, :, :]).squeeze()
        return res[:, :48].contiguous(), self.nhop_regressor(nhops[1].squeeze())


def run(fast, graph_after=0):
    from fgnn_amd import fastpath
    fastpath.GRAPH_AFTER = graph_after
    if fast:
        fgnn_amd.enable_fast_path()
    try:
        torch.manual_seed(3)
        with contextlib.redirect_stdout(io.StringIO()):
            m = ScriptModel().to(dev).train()
        # what the script's DataLoader + to_cuda deliver: fresh f32 tensors every iteration, per-sample copies of the tables
        pool = [synthetic_batch(B, dev, seed=5 + i, dtype=torch.float32, shared_graph=False) for i in range(4)]
        opt = torch.optim.Adam(m.parameters(), lr=1e-4, weight_decay=1e-8)
        it = [0]

        def step():                                           # train_ldpc.py:207-231, line for line
            d = pool[it[0] % len(pool)]
            it[0] += 1
            opt.zero_grad()
            pred, sb = m(*d[:6])
            loss = torch.nn.functional.binary_cross_entropy_with_logits(pred.view(-1), d[6].view(-1).float())
            sloss = torch.nn.functional.mse_loss(sb.view(-1), torch.pow(10.0, d[7].float() / 20).view(-1))
            (loss + 0.1 * sloss).backward()
            opt.step()
            return loss.item()                                # (the script reads the loss back every step)
        for _ in range(6):
            step()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            step()
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / steps * 1e3, type(opt).__name__
    finally:
        if fast:
            fgnn_amd.disable_fast_path()


for fast, ga, what in ((False, 0, 'as the script is written:'), (True, 0, 'enable_fast_path(), eager steps:'), (True, 3, 'enable_fast_path(), hipGraph replay:')):
    ms, on = run(fast, ga)
    print('%-38s %8.2f ms per training step at %d codewords (optimizer: %s)' % (what, ms, B, on), flush=True)
