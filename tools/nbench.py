#!/usr/bin/env python3
"""Micro-benchmark: fused BatchNorm+act / InstanceNorm kernels vs torch (train mode fwd+bwd)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'factor-graph-neural-network_amd'))
import torch
torch.backends.cudnn.enabled = False
from fgnn_amd.mpnn import BatchNormAct2d, NodeInstanceNorm
dev = torch.device('cuda:0')
def timeit(fn, n=10):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / n * 1e3
for dt in (torch.bfloat16, torch.float32):
    for C, N in [(64, 96), (64, 48), (256, 96)]:
        B = 4096
        x = torch.randn(B, N, 1, C, device=dev).to(dt).permute(0, 3, 1, 2).requires_grad_(True)
        gy = torch.randn(B, N, 1, C, device=dev).to(dt).permute(0, 3, 1, 2)
        mine = BatchNormAct2d(C, slope=0.01).to(dev).train()
        ref = torch.nn.BatchNorm2d(C).to(dev).train()
        def f_mine(): y = mine(x); y.backward(gy)
        def f_ref(): y = torch.nn.functional.leaky_relu(ref(x), 0.01); y.backward(gy)
        inn = NodeInstanceNorm(relu=True)
        def f_in(): y = inn(x); y.backward(gy)
        mb = x.numel() * x.element_size() / 1e6
        print('%s C=%3d N=%2d (%.0f MB): BN+act fwd+bwd mine %7.1f us  torch %7.1f us | IN+relu fwd+bwd %7.1f us' % (str(dt)[6:], C, N, mb, timeit(f_mine), timeit(f_ref), timeit(f_in)))

# reference points for the streaming kernels above: a plain device copy and an elementwise op of the same tensors
for C, N in [(64, 96), (128, 96)]:
    x = torch.randn(4096, N, 1, C, device=dev).bfloat16()
    y = torch.empty_like(x)
    mb = x.numel() * 2 / 1e6
    t_copy = timeit(lambda: y.copy_(x), 20)
    t_relu = timeit(lambda: torch.relu(x), 20)
    print('bf16 C=%3d N=%2d (%.0f MB): copy %6.1f us = %.2f TB/s | relu (new tensor) %6.1f us = %.2f TB/s' % (C, N, mb, t_copy, 2 * mb / t_copy, t_relu, 2 * mb / t_relu))
