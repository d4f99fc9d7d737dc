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


class ScriptModel(torch.nn.Module):                       # the composition of train_ldpc.py's LDPCModel (train_ldpc.py:19-99)
    def __init__(self):
        super().__init__()
        self.main = FactorNN(2, [6, 96], [64, 64, 64, 128, 256, 256, 128, 64, 64], [4, 1], 2,
                             skip_link={4: 3, 5: 2, 7: 0}, ret_high=True, aggregator='max')
        mk = lambda: torch.nn.Sequential(torch.nn.Conv2d(7, 64, 1), torch.nn.ReLU(inplace=True), torch.nn.Conv2d(64, 4, 1))
        self.emodel_f2v, self.emodel_v2f = mk(), mk()

    def forward(self, node_feature, hop_feature, nn_idx_f2v, nn_idx_v2f, ef_f2v, ef_v2f):
        Bn = node_feature.shape[0]
        hyper = node_feature[:, 0, :, :].detach().reshape(Bn, 96, 1, 1)
        ones = lambda *s: torch.ones(*s, device=node_feature.device, dtype=node_feature.dtype)
        res, hops = self.main(node_feature, [hop_feature, hyper],
                              [nn_idx_f2v, torch.zeros(Bn, 96, 1, dtype=torch.int64, device=node_feature.device)],
                              [nn_idx_v2f, torch.arange(96, device=node_feature.device).reshape(1, 1, 96).repeat(Bn, 1, 1)],
                              [self.emodel_f2v(ef_f2v), ones(Bn, 1, 96, 1)], [self.emodel_v2f(ef_v2f), ones(Bn, 1, 1, 96)])
        return res.reshape(Bn, 96)[:, :48]


def run(fast):
    if fast:
        fgnn_amd.enable_fast_path()
    try:
        torch.manual_seed(3)
        with contextlib.redirect_stdout(io.StringIO()):
            m = ScriptModel().to(dev).train()
        data = synthetic_batch(B, dev, seed=5, dtype=torch.float32)
        label = (torch.rand(B, 48, device=dev) > 0.5).float()
        opt = torch.optim.Adam(m.parameters(), lr=1e-4, weight_decay=1e-8)

        def step():
            opt.zero_grad()
            torch.nn.functional.binary_cross_entropy_with_logits(m(*data[:6]), label).backward()
            opt.step()
        for _ in range(3):
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


for fast in (False, True):
    ms, on = run(fast)
    print('%-28s %8.2f ms per eager training step at %d codewords (optimizer: %s)' % ('with enable_fast_path():' if fast else 'as the script is written:', ms, B, on), flush=True)
