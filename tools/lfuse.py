#!/usr/bin/env python3
"""One-kernel FactorNN layers (csrc/factor_layer_fwd.hip) against the per-block inference path: outputs and time of the bf16
LDPCModel inference forward.   python tools/lfuse.py [batch]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'factor-graph-neural-network_amd'))
import torch
import fgnn_amd
from fgnn_amd.mpnn import assemblies
from fgnn_amd.ldpc import synthetic_batch

B = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
dev = torch.device('cuda:0')
torch.manual_seed(0)
m = fgnn_amd.LDPCModel(2, 6, 4).to(dev)
# a few training-like updates of the BatchNorm statistics so that the folded affines are not the identity
with torch.no_grad():
    for mod in m.modules():
        if isinstance(mod, torch.nn.modules.batchnorm._BatchNorm):
            mod.running_mean.normal_(0, 0.1)
            mod.running_var.uniform_(0.5, 1.5)
            mod.weight.uniform_(0.5, 1.5)
            mod.bias.normal_(0, 0.1)
m.eval()
inp = synthetic_batch(B, dev, seed=1, dtype=torch.bfloat16)[:6]


def run():
    with torch.no_grad(), torch.autocast(device_type='cuda', dtype=torch.bfloat16):
        return m(*inp)


def timed(tag):
    for _ in range(3):
        out = run()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        run()
        with torch.cuda.graph(g, stream=s):
            out = run()
    torch.cuda.current_stream().wait_stream(s)
    g.replay()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(10):
        g.replay()
    b.record()
    torch.cuda.synchronize()
    print('%s: %.3f ms per forward' % (tag, a.elapsed_time(b) / 10))
    return [o.float().clone() for o in out]


if os.environ.get('FGNN_PROF'):            # phase timeline of the layer kernel (prof build): eager launches only
    run(); run()
    torch.cuda.synchronize()
    sys.exit(0)
assemblies.FUSE_EVAL_LAYERS = False
ref = timed('per-block kernels')
assemblies.FUSE_EVAL_LAYERS = True
got = timed('one-kernel 64-wide layers')
for r, g_, name in zip(ref, got, ['logits', 'snr']):
    print('%s: max |fused - staged| %.4e of range %.4e; mean |diff| %.4e' % (name, float((r - g_).abs().max()), float(r.abs().max()),
                                                                             float((r - g_).abs().mean())))
