#!/usr/bin/env python3
"""ATen ops of one factor_mpnn training step (forward + backward), by call site."""
import collections, os, sys, traceback, contextlib, io
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'factor-graph-neural-network_amd')); sys.path.insert(0, ROOT)
import torch
from torch.utils._python_dispatch import TorchDispatchMode
import fgnn_amd
import bench
dev = torch.device('cuda:0')
hop_dim, hi_nodes, pw_idx, pw_ef, hi_idx, hi_ef = bench.syn_tables('syn_hop', 9)
with contextlib.redirect_stdout(io.StringIO()):
    model = fgnn_amd.factor_mpnn(2, [4, hop_dim], bench.SYN_DIMS, [16, 16]).to(dev).train()
B = 128
nf, pws, hi = torch.rand(B, 2, 30, 1, device=dev), torch.rand(B, 4, 30, 1, device=dev), torch.rand(B, hop_dim, hi_nodes, 1, device=dev)
t = lambda a: torch.from_numpy(a).to(dev)[None]
et_pw = torch.randn(1, 16, *pw_idx.shape, device=dev, requires_grad=True)
et_hi = torch.randn(1, 16, *hi_idx.shape, device=dev, requires_grad=True)
gs = lambda: [[t(pw_idx).expand(B, -1, -1), et_pw.expand(B, -1, -1, -1)], [t(hi_idx).expand(B, -1, -1), et_hi.expand(B, -1, -1, -1)]]


class Count(TorchDispatchMode):
    def __init__(self):
        super().__init__(); self.c = collections.Counter()
    def __torch_dispatch__(self, func, types, args=(), kwargs=None):
        name = str(func)
        if not any(s in name for s in ('view', 'reshape', 'permute', 'expand', 'detach', 'alias', 'slice.Tensor', 'select', 'as_strided', 'unsqueeze', 'squeeze', 't.default', 'transpose', 'size', 'stride', 'is_contiguous', 'sym_', 'empty')):
            where = '?'
            for fr in reversed(traceback.extract_stack(limit=16)):
                if 'fgnn_amd' in fr.filename:
                    where = '%s:%d' % (os.path.basename(fr.filename), fr.lineno); break
            self.c[(name, where)] += 1
        return func(*args, **(kwargs or {}))


from fgnn_amd.dp import FlatGradBucket
bucket = FlatGradBucket(list(model.parameters()) + [et_pw, et_hi], flatten_params=False)
for _ in range(2):
    bucket.zero(); pred, _ = model(nf, [pws, hi], gs()); pred.sum().backward()
torch.cuda.synchronize()
bucket.zero()
with Count() as cnt:
    pred, _ = model(nf, [pws, hi], gs())
    pred.sum().backward()
print('ATen ops in one training step (views / empties excluded): %d' % sum(cnt.c.values()))
for (name, where), n in cnt.c.most_common(45):
    print('  %4d  %-42s %s' % (n, name, where))
