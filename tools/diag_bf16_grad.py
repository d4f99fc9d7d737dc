"""Where does the bf16 whole-model gradient differ from the f32 oracle's?  (GPU box.)
A = oracle f32 autograd, B = the same oracle under CPU bf16 autocast (the noise floor of ANY bf16 implementation),
C = HIP bf16 (benched path), D = HIP f32.  Whole-gradient relative errors and a per-layer breakdown."""
import os
import sys
import re
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in ('factor-graph-neural-network_amd', 'oracle', 'tests'):
    sys.path.insert(0, os.path.join(ROOT, p))
import torch
import fgnn_oracle as O
import fgnn_amd
from fgnn_amd.datapath import LdpcDataPath

dev = torch.device('cuda:0')
bn_mode = sys.argv[1] if len(sys.argv) > 1 else 'eval_stats'
B = int(sys.argv[2]) if len(sys.argv) > 2 else 128
torch.manual_seed(3)
m = fgnn_amd.LDPCModel(2, 6, 4, aggregator='max').to(dev).train()
dp = LdpcDataPath(dev)
with torch.no_grad():
    for i in range(3):
        m(*dp.sample(256, seed=50 + i)[:6])
data = dp.sample(B, seed=31, dtype=torch.bfloat16)
inputs = data[:6]
label = data[6][:, :48].float().contiguous()
train = bn_mode == 'batch_stats'
m.train(train)


def loss_of(logits, snr, label):
    return torch.nn.functional.binary_cross_entropy_with_logits(logits.float().reshape(-1), label.reshape(-1)) \
        + 0.1 * torch.nn.functional.mse_loss(snr.float().reshape(-1), torch.ones(B, device=snr.device))


names = [n for n, p in m.named_parameters() if p.requires_grad]
sd0 = {k: v.detach().cpu().clone() for k, v in m.state_dict().items()}
o_in = [t.cpu().contiguous() for t in inputs]
o_in = [t.float() if t.is_floating_point() else t for t in o_in]


def oracle(autocast):
    sd = {k: v.clone() for k, v in sd0.items()}
    for n in names:
        sd[n].requires_grad_(True)
    with torch.autocast('cpu', dtype=torch.bfloat16, enabled=autocast):
        out = O.ldpc_model(sd, *o_in, training=train)
    loss_of(*out, label.cpu()).backward()
    return {n: sd[n].grad.double() if sd[n].grad is not None else None for n in names}, [t.detach().float() for t in out]


def hip(bf16):
    for p in m.parameters():
        p.grad = None
    st = {k: v.clone() for k, v in m.state_dict().items()}
    ins = inputs if bf16 else [t.float() if t.is_floating_point() else t for t in inputs]
    with torch.autocast('cuda', dtype=torch.bfloat16, enabled=bf16):
        out = m(*ins)
    loss_of(*out, label).backward()
    g = {n: (p.grad.detach().double().cpu() if p.grad is not None else None) for n, p in m.named_parameters() if p.requires_grad}
    m.load_state_dict(st)
    return g, [t.detach().float().cpu() for t in out]


def cmp(tag, g, ref, per_layer=False):
    a = torch.cat([g[n].reshape(-1) for n in names if ref[n] is not None])
    b = torch.cat([ref[n].reshape(-1) for n in names if ref[n] is not None])
    print('%-28s whole-gradient rel err %.3e  cosine %.5f' % (tag, float((a - b).norm() / b.norm()), float(torch.dot(a, b) / a.norm() / b.norm())))
    if per_layer:
        groups = {}
        for n in names:
            if ref[n] is None:
                continue
            mm = re.match(r'main\.(\w+?)_(\d)(?:_(\d))?\.', n)
            key = ('L%s %s' % (mm.group(2), mm.group(1) + ('_' + mm.group(3) if mm.group(3) else ''))) if mm else n.split('.')[0] + '.' + n.split('.')[1]
            groups.setdefault(key, []).append(n)
        for key in sorted(groups):
            a = torch.cat([g[n].reshape(-1) for n in groups[key]])
            b = torch.cat([ref[n].reshape(-1) for n in groups[key]])
            print('    %-28s |ref| %.3e  rel err %.3e  cos %.5f' % (key, float(b.norm()), float((a - b).norm() / max(float(b.norm()), 1e-30)),
                                                                  float(torch.dot(a, b) / max(float(a.norm() * b.norm()), 1e-30))))


A, outA = oracle(False)
Bg, outB = oracle(True)
C, outC = hip(True)
D, outD = hip(False)
rng = float(outA[0].abs().max())
print('logit range %.3g; logit err / range: B %.2e  C %.2e  D %.2e' % (rng, float((outB[0] - outA[0]).abs().max()) / rng,
      float((outC[0] - outA[0]).abs().max()) / rng, float((outD[0] - outA[0]).abs().max()) / rng))
cmp('D hip f32 vs A', D, A)
cmp('B oracle cpu-bf16 vs A', Bg, A)
cmp('C hip bf16 vs A', C, A, per_layer=True)
cmp('C hip bf16 vs B', C, Bg)
cmp('C hip bf16 vs D hip f32', C, D)
if os.environ.get('FGNN_NO_SG') is None:
    print('(re-run with FGNN_NO_SG=1 for the first-generation kernels)')
