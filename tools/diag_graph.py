#!/usr/bin/env python3
"""Does a replayed hipGraph of the synthetic-PGM training step compute what the eager step computes?  Per parameter tensor,
after one and after two optimizer steps.   python tools/diag_graph.py [syn_pw|syn_hop] [batch]"""
import contextlib, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'factor-graph-neural-network_amd'))
import torch
import bench
import fgnn_amd
from fgnn_amd import ops
from fgnn_amd.dp import FlatGradBucket, FlatAdam
from fgnn_amd.graph import StepGraph

wl = sys.argv[1] if len(sys.argv) > 1 else 'syn_hop'
B = int(sys.argv[2]) if len(sys.argv) > 2 else 64
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
names = [n for n, _ in everything.named_parameters()]
params = [p for _, p in everything.named_parameters()]
out = {}
DROP = os.environ.get('DIAG_DROP') is not None


def compute():
    bucket.zero()
    et_pw, et_hi = em_pw(ef_pw), em_hi(ef_hi)
    if os.environ.get('DIAG_HOOK'):
        out['et_pw'], out['et_hi'] = et_pw.detach().clone(), et_hi.detach().clone()
        et_pw.register_hook(lambda g: out.__setitem__('get_pw', g.clone()))
        et_hi.register_hook(lambda g: out.__setitem__('get_hi', g.clone()))
    pred, _ = model(nf, [pws, hi], [[rep_i(idx_pw), rep_e(et_pw)], [rep_i(idx_hi), rep_e(et_hi)]])
    loss = torch.nn.functional.cross_entropy(pred.squeeze(-1).permute(0, 2, 1).reshape(-1, 2), label.reshape(-1))
    if DROP:                                                # as bench.py: nothing of the step outlives it
        out['pred'], out['loss'] = torch.zeros(1), torch.zeros(())
    else:
        out['pred'], out['loss'] = pred, loss.detach()
    loss.backward()


graphed = StepGraph(compute)
g_pred, g_loss = out['pred'], out['loss']
gout = dict(out)
for it in range(3):
    bn_state = [b.clone() for b in everything.buffers()]
    graphed.replay()
    torch.cuda.synchronize()
    gg = bucket.flat.clone()
    pg, lg = g_pred.detach().clone(), float(g_loss)
    hk = {k: v.clone() for k, v in gout.items() if k.startswith('get_') or k.startswith('et_')}
    if os.environ.get('DIAG_NO_EAGER'):                     # replays only, as bench.py steps
        print('step %d: loss graph %.6f, max|g| %.3e' % (it, lg, float(gg.abs().max())))
        gflat = bucket.flat
        norm = torch.linalg.vector_norm(gflat)
        gflat.mul_(torch.clamp(1.0 / (norm + 1e-6), max=1.0))
        opt.step(grad_scale=1.0)
        continue
    for b, s in zip(everything.buffers(), bn_state):       # the eager step starts from the same running statistics
        b.copy_(s)
    compute()
    torch.cuda.synchronize()
    ge = bucket.flat.clone()
    pe, le = out['pred'].detach().clone(), float(out['loss'])
    print('step %d: loss graph %.6f eager %.6f | pred max|diff| %.3e | grad max|diff| %.3e of max|g| %.3e (graph %.3e)' % (
        it, lg, le, float((pg - pe).abs().max()), float((gg - ge).abs().max()), float(ge.abs().max()), float(gg.abs().max())))
    for k2, v in hk.items():
        print('   %s: graph max %.3e eager max %.3e diff %.3e' % (k2, float(v.abs().max()), float(out[k2].abs().max()), float((v - out[k2]).abs().max())))
    worst = []
    for n, p in zip(names, params):
        off = (p.data_ptr() - bucket.params_flat.data_ptr()) // 4 if hasattr(bucket, 'params_flat') else None
        ga = p.grad
        if ga is None:
            continue
        o = (ga.data_ptr() - bucket.flat.data_ptr()) // 4
        a, b = gg[o:o + ga.numel()], ge[o:o + ga.numel()]
        d = float((a - b).abs().max())
        if d > 1e-3 * max(1e-6, float(b.abs().max())):
            worst.append((n, d, float(b.abs().max()), float(a.abs().max())))
    for w in worst[:12]:
        print('   %-50s diff %.3e  eager max %.3e  graph max %.3e' % w)
    print('   (%d of %d parameter tensors differ)' % (len(worst), len(names)))
    bucket.flat.copy_(ge)
    gflat = bucket.flat
    norm = torch.linalg.vector_norm(gflat)
    gflat.mul_(torch.clamp(1.0 / (norm + 1e-6), max=1.0))
    opt.step(grad_scale=1.0)
