import sys, os, torch
sys.path.insert(0, 'factor-graph-neural-network_amd'); sys.path.insert(0, 'tests'); sys.path.insert(0, 'oracle')
import fgnn_amd, helpers as H, fgnn_oracle as O
dev = torch.device('cuda:0')
tag = sys.argv[1] if len(sys.argv) > 1 else 'pw'
hop_dim, pw_idx, pw_ef, hi_idx, hi_ef = H.syn_setup(tag)
B = 96
def run():
    torch.manual_seed(77)
    model = fgnn_amd.factor_mpnn(2, [4, hop_dim], O.SYN_DIMS, [16, 16]).to(dev).train()
    C = fgnn_amd.mpnn.pointwise.PointwiseConv2d if os.environ.get("PWC") else torch.nn.Conv2d
    em_pw = torch.nn.Sequential(C(3, 64, 1), torch.nn.ReLU(), C(64, 16, 1)).to(dev)
    em_hi = torch.nn.Sequential(C(hi_ef.shape[0], 64, 1), torch.nn.ReLU(), C(64, 16, 1)).to(dev)
    ev = torch.nn.ModuleList([model, em_pw, em_hi])
    g = torch.Generator().manual_seed(5)
    nf, pws = torch.rand(B, 2, 30, 1, generator=g).to(dev), torch.rand(B, 4, 30, 1, generator=g).to(dev)
    hi = torch.rand(B, hop_dim, 1 if tag == 'pw' else 30, 1, generator=g).to(dev)
    label = torch.randint(0, 2, (B, 30), generator=g).to(dev)
    t = lambda a: torch.from_numpy(a).to(dev)[None]
    et_pw, et_hi = em_pw(t(pw_ef)), em_hi(t(hi_ef))
    pred, _ = model(nf, [pws, hi], [[t(pw_idx).expand(B, -1, -1), et_pw.expand(B, -1, -1, -1)],
                                   [t(hi_idx).expand(B, -1, -1), et_hi.expand(B, -1, -1, -1)]])
    loss = torch.nn.functional.cross_entropy(pred.squeeze(-1).permute(0, 2, 1).reshape(-1, 2), label.reshape(-1))
    loss.backward()
    torch.cuda.synchronize()
    return {n: p.grad.clone() for n, p in ev.named_parameters() if p.grad is not None}, pred.detach().clone()
import contextlib, io
with contextlib.redirect_stdout(io.StringIO()):
    runs = [run() for _ in range(4)]
a, pa = runs[0]
for i in range(1, 4):
    b, pb = runs[i]
    bad = [n for n in a if not torch.equal(a[n], b[n])]
    print('run', i, 'pred equal', torch.equal(pa, pb), 'differing grads:', len(bad), bad[:12])
