"""CPU, world_size 2 over gloo: the data-parallel machinery (flat gradient bucket, one all-reduce,
parameter broadcast, contiguous sharding).  The model here is a small stock-PyTorch net — the
FGNN operator itself is GPU-only — but the DP path is exactly the one bench.py drives with RCCL."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _net():
    torch.manual_seed(1234)
    return torch.nn.Sequential(torch.nn.Linear(6, 16), torch.nn.Tanh(), torch.nn.Linear(16, 3))


def _worker(rank, world, port, out):
    import sys
    here = os.path.dirname(os.path.abspath(__file__))
    sys.path.insert(0, os.path.join(os.path.dirname(here), 'factor-graph-neural-network_amd'))
    from fgnn_amd.dp import FlatGradBucket, broadcast_parameters, shard_range
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    torch.set_num_threads(1)
    net = _net()
    if rank == 1:                       # replicas start different; broadcast must fix it
        with torch.no_grad():
            for p in net.parameters():
                p.add_(1.0)
    broadcast_parameters(net)
    g = torch.Generator().manual_seed(7)
    X, Y = torch.randn(10, 6, generator=g), torch.randn(10, 3, generator=g)
    lo, hi = shard_range(10, rank, world)
    bucket = FlatGradBucket(net.parameters())
    opt = torch.optim.SGD(bucket.params, lr=0.1)
    for _ in range(3):
        bucket.zero()
        # sum-of-squares scaled so that the MEAN over ranks of per-rank grads == full-batch grad
        loss = ((net(X[lo:hi]) - Y[lo:hi]) ** 2).sum() * (world / 10.0)
        loss.backward()
        bucket.all_reduce_mean()
        opt.step()
    flat = torch.cat([p.detach().reshape(-1) for p in net.parameters()])
    torch.save(flat, os.path.join(out, 'rank%d.pt' % rank))
    dist.destroy_process_group()


def test_two_rank_dp_equals_single_process(tmp_path):
    port = _free_port()
    mp.spawn(_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    a = torch.load(os.path.join(str(tmp_path), 'rank0.pt'))
    b = torch.load(os.path.join(str(tmp_path), 'rank1.pt'))
    assert torch.equal(a, b), 'replicas diverged'
    # single-process reference on the whole batch
    net = _net()
    g = torch.Generator().manual_seed(7)
    X, Y = torch.randn(10, 6, generator=g), torch.randn(10, 3, generator=g)
    opt = torch.optim.SGD(net.parameters(), lr=0.1)
    for _ in range(3):
        opt.zero_grad()
        (((net(X) - Y) ** 2).sum() / 10.0).backward()
        opt.step()
    ref = torch.cat([p.detach().reshape(-1) for p in net.parameters()])
    assert torch.allclose(a, ref, atol=1e-6), float((a - ref).abs().max())


def _worker_adam(rank, world, port, out):
    import sys
    here = os.path.dirname(os.path.abspath(__file__))
    sys.path.insert(0, os.path.join(os.path.dirname(here), 'factor-graph-neural-network_amd'))
    from fgnn_amd.dp import FlatAdam, FlatGradBucket, broadcast_parameters, shard_range
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    torch.set_num_threads(1)
    net = _net()
    if rank == 1:
        with torch.no_grad():
            for p in net.parameters():
                p.mul_(0.5)
    broadcast_parameters(net)
    g = torch.Generator().manual_seed(7)
    X, Y = torch.randn(10, 6, generator=g), torch.randn(10, 3, generator=g)
    lo, hi = shard_range(10, rank, world)
    bucket = FlatGradBucket(net.parameters(), flatten_params=True)      # what bench.py builds
    opt = FlatAdam(bucket, lr=1e-2, weight_decay=1e-3)
    for _ in range(4):
        bucket.zero()
        (((net(X[lo:hi]) - Y[lo:hi]) ** 2).sum() * (world / 10.0)).backward()
        bucket.all_reduce_mean()
        opt.step()
    # (the flat buffers pad every parameter to a 16-byte boundary — dp.flat_layout; the padding must still be zero after 4 steps)
    used = torch.zeros(bucket.numel, dtype=torch.bool)
    for p, off in zip(bucket.params, bucket.offsets):
        used[off:off + p.numel()] = True
    assert float(bucket.flat_param[~used].abs().sum()) == 0
    torch.save(torch.cat([p.detach().reshape(-1) for p in net.parameters()]), os.path.join(out, 'adam%d.pt' % rank))
    dist.destroy_process_group()


def test_two_rank_flat_adam_equals_single_process_adam(tmp_path):
    """The exact training plumbing of bench.py (flat parameter + gradient buffers, one all-reduce, FlatAdam)
    on 2 gloo ranks == torch.optim.Adam on the whole batch in one process."""
    port = _free_port()
    mp.spawn(_worker_adam, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    a = torch.load(os.path.join(str(tmp_path), 'adam0.pt'))
    b = torch.load(os.path.join(str(tmp_path), 'adam1.pt'))
    assert torch.equal(a, b), 'replicas diverged'
    net = _net()
    g = torch.Generator().manual_seed(7)
    X, Y = torch.randn(10, 6, generator=g), torch.randn(10, 3, generator=g)
    opt = torch.optim.Adam(net.parameters(), lr=1e-2, weight_decay=1e-3)
    for _ in range(4):
        opt.zero_grad()
        (((net(X) - Y) ** 2).sum() / 10.0).backward()
        opt.step()
    ref = torch.cat([p.detach().reshape(-1) for p in net.parameters()])
    assert torch.allclose(a, ref, atol=2e-6), float((a - ref).abs().max())


def test_shard_range_covers_everything():
    import sys
    here = os.path.dirname(os.path.abspath(__file__))
    sys.path.insert(0, os.path.join(os.path.dirname(here), 'factor-graph-neural-network_amd'))
    from fgnn_amd.dp import shard_range
    for total in (0, 1, 7, 8, 4096, 32768):
        for world in (1, 2, 3, 8):
            spans = [shard_range(total, r, world) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == total
            assert all(spans[i][1] == spans[i + 1][0] for i in range(world - 1))
            sizes = [b - a for a, b in spans]
            assert max(sizes) - min(sizes) <= 1


def test_flat_bucket_views_survive_backward():
    import sys
    here = os.path.dirname(os.path.abspath(__file__))
    sys.path.insert(0, os.path.join(os.path.dirname(here), 'factor-graph-neural-network_amd'))
    from fgnn_amd.dp import FlatGradBucket
    net = _net()
    bucket = FlatGradBucket(net.parameters())
    net(torch.ones(2, 6)).sum().backward()
    for p, off in zip(bucket.params, bucket.offsets):
        assert off % 4 == 0                                                   # every parameter on a 16-byte boundary (dp.flat_layout)
        assert p.grad.data_ptr() == bucket.flat[off:off + p.numel()].data_ptr()     # still a view
    assert float(bucket.flat.abs().sum()) > 0
    bucket.zero()
    assert all(float(p.grad.abs().sum()) == 0 for p in bucket.params)
