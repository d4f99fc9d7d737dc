"""Which host ops put device-to-device memcpy nodes (hipMemcpyAsync: __amd_rocclr_copyBuffer) into the training step?  (GPU box)"""
import os, sys, contextlib, collections, traceback
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'factor-graph-neural-network_amd'))
sys.path.insert(0, ROOT)
import torch
from fgnn_amd.ldpc import LDPCModel
from fgnn_amd.datapath import LdpcDataPath
from fgnn_amd.dp import FlatAdam, FlatGradBucket
import bench

dev = torch.device('cuda:0')
torch.manual_seed(0)
with contextlib.redirect_stdout(sys.stderr):
    model = LDPCModel(2, 6, 4, aggregator='max').to(dev)
data = LdpcDataPath(dev).sample(4096, seed=100, dtype=torch.bfloat16)
data = data[:6] + (data[6][:, :48].float().contiguous(), data[7])
inputs, label, sigma_b = data[:6], data[6], data[7]
model.train()
bucket = FlatGradBucket(model.parameters(), flatten_params=True)
amp = torch.autocast(device_type='cuda', dtype=torch.bfloat16)


def compute():
    bucket.zero()
    with amp:
        logits, snr = model(*inputs)
    bench.loss_fn(logits, snr, label, sigma_b).backward()


for _ in range(2):
    compute()
torch.cuda.synchronize()
hits = collections.Counter()
orig = torch.Tensor.copy_


def spy(self, src, *a, **k):
    if self.is_cuda and src.is_cuda and self.dtype == src.dtype and self.is_contiguous() and src.is_contiguous():
        fr = [f for f in traceback.extract_stack(limit=12)[:-1] if 'fgnn_amd' in f.filename or 'bench' in f.filename]
        key = ' <- '.join('%s:%d' % (os.path.basename(f.filename), f.lineno) for f in fr[-3:])
        hits[(key, tuple(self.shape))] += 1
    return orig(self, src, *a, **k)


torch.Tensor.copy_ = spy
compute()
torch.Tensor.copy_ = orig
torch.cuda.synchronize()
for (key, shape), n in hits.most_common(40):
    print(n, shape, key)
# and everything the dispatcher turns into memcpy: profile one step
from torch.profiler import profile, ProfilerActivity
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True) as prof:
    compute()
    torch.cuda.synchronize()
ev = [e for e in prof.events() if 'Memcpy' in e.name or 'copyBuffer' in e.name]
print('memcpy-like device events in one eager step:', len(ev))
cnt = collections.Counter()
for e in prof.events():
    if e.name in ('aten::copy_', 'aten::contiguous', 'aten::clone', 'aten::to', 'aten::_to_copy') and e.stack:
        st = [s for s in e.stack if 'fgnn_amd' in s or 'bench.py' in s]
        if st:
            cnt[(e.name, st[0].split('/')[-1][:90])] += 1
for k, n in cnt.most_common(40):
    print(n, k)
