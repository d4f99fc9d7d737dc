import sys, os, torch, contextlib, io
sys.path.insert(0, 'factor-graph-neural-network_amd'); sys.path.insert(0, 'tests'); sys.path.insert(0, 'oracle')
import fgnn_amd, helpers as H, fgnn_oracle as O
dev = torch.device('cuda:0')
tag, out = sys.argv[1], sys.argv[2]
hop_dim, pw_idx, pw_ef, hi_idx, hi_ef = H.syn_setup(tag)
B = 64
torch.manual_seed(3)
with contextlib.redirect_stdout(io.StringIO()):
    model = fgnn_amd.factor_mpnn(2, [4, hop_dim], O.SYN_DIMS, [16, 16]).to(dev).train()
g = torch.Generator().manual_seed(11)
nf = torch.rand(B, 2, 30, 1, generator=g)
label = (nf[:, 1, :, 0] > nf[:, 0, :, 0]).long().to(dev)
nf, pws = nf.to(dev), torch.rand(B, 4, 30, 1, generator=g).to(dev)
hi = torch.rand(B, hop_dim, 1 if tag == 'pw' else 30, 1, generator=g).to(dev)
t = lambda a: torch.from_numpy(a).to(dev)[None]
et_pw = torch.randn(1, 16, *pw_idx.shape, generator=g).to(dev); et_hi = torch.randn(1, 16, *hi_idx.shape, generator=g).to(dev)
pred, _ = model(nf, [pws, hi], [[t(pw_idx).expand(B, -1, -1), et_pw.expand(B, -1, -1, -1)], [t(hi_idx).expand(B, -1, -1), et_hi.expand(B, -1, -1, -1)]])
loss = torch.nn.functional.cross_entropy(pred.squeeze(-1).permute(0, 2, 1).reshape(-1, 2), label.reshape(-1))
loss.backward()
torch.save({n: p.grad.cpu() for n, p in model.named_parameters() if p.grad is not None}, out)
if len(sys.argv) > 3:
    a, b = torch.load(sys.argv[3]), torch.load(out)
    for n in a:
        d = float((a[n] - b[n]).abs().max()); r = float(a[n].abs().max())
        if d > 1e-4 * max(r, 1e-12): print('%-40s shape %-18s max|diff| %.3e of %.3e' % (n, tuple(a[n].shape), d, r))
    print('compared', len(a))
