#!/usr/bin/env python3
"""Which ATen ops does one inference forward of LDPCModel still launch, and from where?  (TorchDispatchMode + the innermost
fgnn_amd stack frame.)  python tools/opcount.py [train]"""
import collections, os, sys, traceback
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'factor-graph-neural-network_amd'))
import contextlib, io
import torch
from torch.utils._python_dispatch import TorchDispatchMode
from fgnn_amd.ldpc import LDPCModel, synthetic_batch

dev = torch.device('cuda:0')
with contextlib.redirect_stdout(io.StringIO()):
    model = LDPCModel(2, 6, 4).to(dev).eval()
data = synthetic_batch(256, dev, seed=1, dtype=torch.bfloat16)
amp = torch.autocast('cuda', dtype=torch.bfloat16)


class Count(TorchDispatchMode):
    def __init__(self):
        super().__init__()
        self.c = collections.Counter()

    def __torch_dispatch__(self, func, types, args=(), kwargs=None):
        name = str(func)
        if not any(s in name for s in ('view', 'reshape', 'permute', 'expand', 'detach', 'alias', 'slice', 'select', 'as_strided', 'unsqueeze', 'squeeze', 't.default', 'transpose', 'size', 'stride', 'is_contiguous', 'sym_')):
            where = '?'
            for fr in reversed(traceback.extract_stack(limit=14)):
                if 'fgnn_amd' in fr.filename:
                    where = '%s:%d' % (os.path.basename(fr.filename), fr.lineno)
                    break
            self.c[(name, where)] += 1
        return func(*args, **(kwargs or {}))


with torch.no_grad(), amp:
    for _ in range(2):
        model(*data[:6])
    torch.cuda.synchronize()
    with Count() as cnt:
        model(*data[:6])
tot = sum(cnt.c.values())
print('ATen ops in one eval forward (views excluded): %d' % tot)
for (name, where), n in cnt.c.most_common(60):
    print('  %4d  %-45s %s' % (n, name, where))
