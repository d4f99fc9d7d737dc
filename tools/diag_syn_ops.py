#!/usr/bin/env python3
"""Which host-side ops launch the torch (non-package) kernels of one synthetic-PGM training step?  Groups the device time of one eager
step by (aten op, the package's calling line).   python tools/diag_syn_ops.py [syn_pw|syn_hop] [batch]"""
import collections, contextlib, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'factor-graph-neural-network_amd'))
import torch
import bench
import fgnn_amd
from fgnn_amd.dp import FlatGradBucket, FlatAdam

wl = sys.argv[1] if len(sys.argv) > 1 else 'syn_hop'
B = int(sys.argv[2]) if len(sys.argv) > 2 else 1024
dev = torch.device('cuda:0')
torch.backends.cudnn.enabled = False
torch.manual_seed(0)
hop_dim, hi_nodes, pw_idx, pw_ef, hi_idx, hi_ef = bench.syn_tables(wl, 9)
C = torch.nn.Conv2d
with contextlib.redirect_stdout(sys.stderr):
    model = fgnn_amd.factor_mpnn(2, [4, hop_dim], bench.SYN_DIMS, [16, 16]).to(dev)
em_pw = torch.nn.Sequential(C(3, 64, 1), torch.nn.ReLU(inplace=True), C(64, 16, 1)).to(dev)
em_hi = torch.nn.Sequential(C(hi_ef.shape[0], 64, 1), torch.nn.ReLU(inplace=True), C(64, 16, 1)).to(dev)
everything = torch.nn.ModuleList([model, em_pw, em_hi])
g = torch.Generator().manual_seed(100)
nf = torch.rand(B, 2, 30, 1, generator=g).to(dev)
pws = torch.rand(B, 4, 30, 1, generator=g).to(dev)
hi = torch.rand(B, hop_dim, hi_nodes, 1, generator=g).to(dev)
label = torch.randint(0, 2, (B, 30), generator=g).to(dev)
idx_pw, idx_hi = torch.from_numpy(pw_idx).to(dev)[None], torch.from_numpy(hi_idx).to(dev)[None]
ef_pw, ef_hi = torch.from_numpy(pw_ef).to(dev)[None], torch.from_numpy(hi_ef).to(dev)[None]
rep_i, rep_e = (lambda t: t.expand(B, -1, -1)), (lambda t: t.expand(B, -1, -1, -1))
everything.train(True)
bucket = FlatGradBucket(everything.parameters(), flatten_params=True)
opt = FlatAdam(bucket, lr=3e-3)


def compute():
    bucket.zero()
    et_pw, et_hi = em_pw(ef_pw), em_hi(ef_hi)
    pred, _ = model(nf, [pws, hi], [[rep_i(idx_pw), rep_e(et_pw)], [rep_i(idx_hi), rep_e(et_hi)]])
    loss = torch.nn.functional.cross_entropy(pred.squeeze(-1).permute(0, 2, 1).reshape(-1, 2), label.reshape(-1))
    loss.backward()


for _ in range(3):
    compute()
torch.cuda.synchronize()
from torch.profiler import profile, ProfilerActivity
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True) as prof:
    compute()
    torch.cuda.synchronize()
rows = collections.defaultdict(lambda: [0, 0.0])
for e in prof.key_averages(group_by_stack_n=12):
    dt = getattr(e, 'self_device_time_total', None)
    if dt is None:
        dt = getattr(e, 'self_cuda_time_total', 0)
    if dt <= 0:
        continue
    fr = [f for f in (e.stack or []) if 'fgnn_amd' in f or 'diag_syn_ops' in f]
    where = ' <- '.join(os.path.basename(f.split(',')[0]).strip() + ':' + f.split('(')[1].split(')')[0] if '(' in f else f for f in fr[:2])
    k = (e.key, where)
    rows[k][0] += e.count
    rows[k][1] += dt
tot = sum(v[1] for v in rows.values())
print('device time of one eager step: %.2f ms' % (tot / 1e3))
for (op, where), (n, dt) in sorted(rows.items(), key=lambda kv: -kv[1][1])[:45]:
    print('%8.1f us %5d x  %-44s %s' % (dt, n, op[:44], where[:110]))
