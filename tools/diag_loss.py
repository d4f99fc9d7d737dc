import sys, os, torch, contextlib, io
sys.path.insert(0, 'factor-graph-neural-network_amd'); sys.path.insert(0, 'tests'); sys.path.insert(0, 'oracle')
import fgnn_amd, helpers as H, fgnn_oracle as O
dev = torch.device('cuda:0')
tag = sys.argv[1]; lr = float(sys.argv[2]); steps = int(sys.argv[3])
hop_dim, pw_idx, pw_ef, hi_idx, hi_ef = H.syn_setup(tag)
B = 64
torch.manual_seed(3)
with contextlib.redirect_stdout(io.StringIO()):
    model = fgnn_amd.factor_mpnn(2, [4, hop_dim], O.SYN_DIMS, [16, 16]).to(dev).train()
C = torch.nn.Conv2d
em_pw = torch.nn.Sequential(C(3, 64, 1), torch.nn.ReLU(), C(64, 16, 1)).to(dev)
em_hi = torch.nn.Sequential(C(hi_ef.shape[0], 64, 1), torch.nn.ReLU(), C(64, 16, 1)).to(dev)
params = list(model.parameters()) + list(em_pw.parameters()) + list(em_hi.parameters())
opt = torch.optim.Adam(params, lr=lr)
g = torch.Generator().manual_seed(11)
nf = torch.rand(B, 2, 30, 1, generator=g)
label = (nf[:, 1, :, 0] > nf[:, 0, :, 0]).long().to(dev)
nf, pws = nf.to(dev), torch.rand(B, 4, 30, 1, generator=g).to(dev)
hi = torch.rand(B, hop_dim, 1 if tag == 'pw' else 30, 1, generator=g).to(dev)
t = lambda a: torch.from_numpy(a).to(dev)[None]
losses = []
for _ in range(steps):
    opt.zero_grad()
    et_pw, et_hi = em_pw(t(pw_ef)), em_hi(t(hi_ef))
    pred, _ = model(nf, [pws, hi], [[t(pw_idx).repeat(B, 1, 1), et_pw.repeat(B, 1, 1, 1)], [t(hi_idx).repeat(B, 1, 1), et_hi.repeat(B, 1, 1, 1)]])
    loss = torch.nn.functional.cross_entropy(pred.squeeze(-1).permute(0, 2, 1).reshape(-1, 2), label.reshape(-1))
    loss.backward()
    torch.nn.utils.clip_grad_norm_(params, 1.0)
    opt.step()
    losses.append(round(float(loss.detach()), 3))
print(tag, lr, losses)
