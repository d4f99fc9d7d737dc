"""fgnn_concat_pair: torch.cat of two channel-fastest activations along the node axis / the channel axis as one kernel
(factor_mpnn's concatenations, /root/reference/lib/model/mpnn/factor_mpnn.py:104-107,116) — bit-exact copies, torch.cat's backward."""
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize('dtype', [torch.float32, torch.bfloat16])
@pytest.mark.parametrize('dim', [1, 2])
@pytest.mark.parametrize('shape', [(5, 64, 30, 30), (1, 64, 1, 7), (33, 128, 30, 1), (3, 8, 48, 96), (2, 64, 30, 30)])
def test_concat_pair_matches_torch_cat(shape, dim, dtype, dev):
    from fgnn_amd import ops
    B, C, na, nb = shape
    g = torch.Generator().manual_seed(B + C + na)
    if dim == 2:
        a = torch.randn(B, na, 1, C, generator=g).to(dtype).to(dev).permute(0, 3, 1, 2)
        b = torch.randn(B, nb, 1, C, generator=g).to(dtype).to(dev).permute(0, 3, 1, 2)
    else:
        a = torch.randn(B, na, 1, C, generator=g).to(dtype).to(dev).permute(0, 3, 1, 2)
        b = torch.randn(B, na, 1, 2 * C, generator=g).to(dtype).to(dev).permute(0, 3, 1, 2)
    ref = torch.cat([a, b], dim=dim)
    out = ops._concat2_raw(a, b, dim)
    assert out is not None and out.shape == ref.shape and out.stride(1) == 1
    assert torch.equal(out, ref)
    # autograd: the gradient comes back as torch.cat's does
    a1, b1 = a.clone().requires_grad_(True), b.clone().requires_grad_(True)
    a2, b2 = a.clone().requires_grad_(True), b.clone().requires_grad_(True)
    w = torch.randn(ref.shape, generator=g).to(dtype).to(dev)
    (ops.concat2(a1, b1, dim) * w).sum().backward()
    (torch.cat([a2, b2], dim=dim) * w).sum().backward()
    assert torch.equal(a1.grad, a2.grad) and torch.equal(b1.grad, b2.grad)


@pytest.mark.parametrize('dim', [1, 2])
def test_concat_pair_takes_node_slices_of_a_larger_activation(dim, dev):
    """factor_mpnn hands over `both[:, :, :n]` / `both[:, :, n:]` of the previous block's output: channel-fastest, strided per sample."""
    from fgnn_amd import ops
    g = torch.Generator().manual_seed(3)
    big = torch.randn(7, 60, 1, 64, generator=g).to(dev).permute(0, 3, 1, 2)
    other = torch.randn(7, 30, 1, 64, generator=g).to(dev).permute(0, 3, 1, 2)
    a, b = big[:, :, :30, :], big[:, :, 30:, :]
    for x, y in ((a, b), (other, b), (a, other)):
        out = ops._concat2_raw(x, y, dim)
        assert out is not None and torch.equal(out, torch.cat([x, y], dim=dim)) and out.stride(1) == 1


def test_concat_rows_takes_channel_slices(dev):
    """torch.cat's backward hands on channel slices of a wider channel-fastest gradient (rows strided inside a sample): the node-axis
    concatenation of such slices is the row form (fgnn_concat_rows)."""
    from fgnn_amd import ops
    g = torch.Generator().manual_seed(5)
    wide = torch.randn(6, 30, 1, 128, generator=g).to(dev).permute(0, 3, 1, 2)
    other = torch.randn(6, 30, 1, 64, generator=g).to(dev).permute(0, 3, 1, 2)
    lo, hi = wide[:, :64], wide[:, 64:]
    for x, y in ((lo, hi), (lo, other), (other, hi)):
        out = ops._concat2_raw(x, y, 2)
        assert out is not None and torch.equal(out, torch.cat([x, y], dim=2)) and out.stride(1) == 1 and out.stride(2) == 64


def test_concat_pair_falls_back_outside_its_family(dev):
    from fgnn_amd import ops
    a = torch.randn(4, 6, 5, 1, device=dev)                  # NCHW-contiguous, odd sizes: torch.cat
    b = torch.randn(4, 6, 3, 1, device=dev)
    assert ops._concat2_raw(a, b, 2) is None
    assert torch.equal(ops.concat2(a, b, 2), torch.cat([a, b], dim=2))
    c = torch.randn(4, 3, 5, 1, device=dev)
    assert torch.equal(ops.concat2(a, c, 1), torch.cat([a, c], dim=1))
