"""Node-wise (1x1) maps as GEMMs on channel-fastest activations.

The reference expresses every node-wise linear map as ``Conv2d(cin, cout, 1)`` on [B,C,N,1]
tensors (base_model.py:43-90, mp_nn_residual.py:25-35).  On ROCm that lowers to MIOpen
convolution solvers that are a poor fit for H*W = N*1 "images" (rocprof, profiles/r01: they
dominate the step).  ``PointwiseConv2d`` keeps Conv2d's parameters / state_dict keys
(``weight [cout,cin,1,1]``, ``bias``) but runs the map as one rocBLAS/hipBLASLt GEMM
[B*N, cin] x [cin, cout] on the channels-last view, which is also the layout the fused message
kernel reads and writes without a transpose.  Outputs are logical [B,C,N,W] with
channels-last strides; every consumer in this package is stride-agnostic.
"""
import torch


class PointwiseConv2d(torch.nn.Conv2d):
    def __init__(self, in_channels, out_channels, kernel_size=1, bias=True):
        assert kernel_size in (1, (1, 1)), 'PointwiseConv2d is a 1x1 map'
        super().__init__(in_channels, out_channels, 1, bias=bias)

    def forward(self, x):
        B, C, H, W = x.shape
        rows = x.permute(0, 2, 3, 1)                    # [B,H,W,C] view; free when channels-last
        if not rows.is_contiguous():
            rows = rows.contiguous()
        y = torch.nn.functional.linear(rows.view(B * H * W, C),
                                       self.weight.view(self.out_channels, C), self.bias)
        return y.view(B, H, W, self.out_channels).permute(0, 3, 1, 2)


class NodeInstanceNorm(torch.nn.Module):
    """InstanceNorm2d(affine=False, no running stats) over the node axis of [B,C,N,1], any strides.

    A single node (the LDPC hyper-factor, factor_mpnn_sp.py:77,140) normalises to exactly 0
    — (x-mean)/sqrt(0+eps) — which is what the reference's torch-1.0 era computed and what
    newer torch refuses to compute.  No parameters, so state_dicts match
    torch.nn.InstanceNorm2d's (empty) contribution.
    """
    eps = 1e-5

    def forward(self, x):
        if x.shape[2] * x.shape[3] == 1:
            return torch.zeros_like(x)
        xf = x.float()                                   # statistics in f32 also for bf16 activations
        var, mean = torch.var_mean(xf, dim=(2, 3), unbiased=False, keepdim=True)
        return ((xf - mean) * torch.rsqrt(var + self.eps)).to(x.dtype)
