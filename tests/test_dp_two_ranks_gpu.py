"""GPU, world_size 2 on ONE device over gloo: the data-parallel path bench.py drives with RCCL — the real LDPCModel, the
flat gradient bucket and its single all-reduce, FlatAdam on the flat parameter buffer, and the hipGraph-captured step
— exercised end to end without an 8-GPU node (BASELINE config 4 is run by the driver; this is its readiness check).

Every rank takes half of one fixed 128-codeword batch.  With BatchNorm normalising by its running statistics (so that no
statistic depends on how the batch is split) the mean of the two ranks' gradients IS the full-batch gradient: it must equal —
to f32 summation-order rounding — the gradient of a single process on the whole batch, the first optimizer step must land on the
same parameters, and after 3 steps the two replicas must still be bit-identical to each other."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu

STEPS, PER_RANK = 3, 64


def _free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _setup_paths():
    import sys
    here = os.path.dirname(os.path.abspath(__file__))
    for p in (os.path.join(os.path.dirname(here), 'factor-graph-neural-network_amd'), here):
        if p not in sys.path:
            sys.path.insert(0, p)


def _train(rank, world, graph, reduce_single_rank=False, direct=False, steps=None):
    """3 steps on this rank's shard; returns the flat parameter buffer (a CPU tensor)."""
    STEPS = steps or globals()['STEPS']
    _setup_paths()
    import contextlib
    import io
    import fgnn_amd
    from fgnn_amd.datapath import LdpcDataPath
    from fgnn_amd.dp import FlatAdam, FlatGradBucket, broadcast_parameters, shard_range
    from fgnn_amd.graph import StepGraph
    dev = torch.device('cuda', 0)
    torch.cuda.set_device(dev)
    torch.manual_seed(11)
    with contextlib.redirect_stdout(io.StringIO()):
        model = fgnn_amd.LDPCModel(2, 6, 4, aggregator='max').to(dev)
    if rank == 1:                                   # replicas start different: the broadcast must fix it
        with torch.no_grad():
            for q in model.parameters():
                if q.requires_grad:
                    q.add_(0.5)
    broadcast_parameters(model)
    model.eval()                                    # BatchNorm by running statistics: shard-independent
    data = LdpcDataPath(dev).sample(PER_RANK * 2, seed=77, dtype=torch.bfloat16)
    lo, hi = shard_range(PER_RANK * 2, rank, world)
    inputs = tuple(t[lo:hi] for t in data[:6])
    label = data[6][lo:hi, :48].float().contiguous()
    bucket = FlatGradBucket(model.parameters(), flatten_params=True, reduce_single_rank=reduce_single_rank)
    if direct:
        bucket.use_direct_all_reduce()               # RCCL's C API on the current stream: recordable into the step graph
    full = graph == 'full'                          # the collective and the optimizer recorded into the graph as well
    opt = FlatAdam(bucket, lr=1e-3, capturable=full or graph == 'capturable')
    amp = torch.autocast('cuda', dtype=torch.bfloat16)

    def compute():
        bucket.zero()
        with amp:
            logits, _ = model(*inputs)
        torch.nn.functional.binary_cross_entropy_with_logits(logits.float().reshape(-1), label.reshape(-1)).backward()

    def whole():
        compute()
        bucket.all_reduce_sum()
        opt.step(grad_scale=1.0 / world)

    first_grad = first_param = None
    if full:
        snap = (bucket.flat_param.clone(), model.state_dict())
        snap = (snap[0], {k: v.clone() for k, v in snap[1].items()})
        step = StepGraph(whole, warmup=1)           # the warm-up call and the capture are optimizer steps too: undo them
        with torch.no_grad():
            model.load_state_dict(snap[1])
            bucket.flat_param.copy_(snap[0])
        opt.exp_avg.zero_(); opt.exp_avg_sq.zero_(); opt.t = 0
        assert opt.t == 0
    else:
        step = StepGraph(compute) if graph in (True, 'capturable') else compute
    for it in range(STEPS):
        step()
        if full:
            if it == 0:
                first_grad = (bucket.flat / world).detach().cpu().clone()
                first_param = bucket.flat_param.detach().cpu().clone()
            if it == 1:
                opt.lr = 5e-4                       # a scheduler between replays: the device scalar is what the graph reads
            continue
        bucket.all_reduce_mean()
        if it == 0:
            first_grad = bucket.flat.detach().cpu().clone()
        if it == 2 and graph in ('sched', 'capturable'):
            opt.lr = 5e-4
        opt.step()
        if it == 0:
            first_param = bucket.flat_param.detach().cpu().clone()
    torch.cuda.synchronize()
    if full:
        assert opt.t == STEPS
    return bucket.flat_param.detach().cpu().clone(), first_grad, first_param


def _worker(rank, world, port, out, graph):
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    flat, grad, p1 = _train(rank, world, graph)
    torch.save({'param': flat, 'grad': grad, 'param1': p1}, os.path.join(out, 'rank%d.pt' % rank))
    dist.destroy_process_group()


@pytest.mark.parametrize('graph', [False, True], ids=['eager', 'hipgraph'])
def test_two_ranks_on_one_gpu_equal_single_process(graph, tmp_path, dev):
    mp.spawn(_worker, args=(2, _free_port(), str(tmp_path), graph), nprocs=2, join=True)
    a = torch.load(os.path.join(str(tmp_path), 'rank0.pt'))
    b = torch.load(os.path.join(str(tmp_path), 'rank1.pt'))
    assert torch.equal(a['param'], b['param']) and torch.equal(a['grad'], b['grad']), 'replicas diverged'
    ref_param, ref_grad, ref_p1 = _train(0, 1, graph)       # one process, the whole batch
    scale = float(ref_grad.abs().max())
    assert scale > 0 and torch.isfinite(a['param']).all() and torch.isfinite(ref_param).all()
    err = float((a['grad'] - ref_grad).abs().max()) / scale
    print('mean of the 2 ranks\' gradients vs the full-batch gradient: max abs diff %.2e of the gradient range' % err)
    # same per-sample gradients, summed in a different order / grouping (f32): rounding only
    assert err <= 1e-4, err
    # one optimizer step from equal gradients gives equal parameters wherever the gradient is above its own rounding noise
    # (Adam normalises every coordinate to +-lr: a coordinate that is pure cancellation noise moves by +-lr with a random
    # sign).  Later steps are NOT compared across the two runs: a bf16 decoder with max-routing amplifies those +-lr
    # differences chaotically; what data parallelism must guarantee there is that the replicas stay identical (above).
    firm = ref_grad.abs() > 1e-3 * scale
    assert float((a['param1'] - ref_p1).abs()[firm].max()) <= 1e-5


def test_bench_self_launches_its_ranks(dev):
    """`python bench.py --gpus 2` with NO launcher around it: bench.py starts the two ranks itself (here both on device 0 over
    gloo — the hooks FGNN_BENCH_DEVICE / FGNN_DIST_BACKEND; the driver's node gives each rank its own GPU over RCCL) and rank 0
    prints one line that says n_gpus 2, with both ranks' step times and the global batch of two shards."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ('WORLD_SIZE', 'RANK', 'LOCAL_RANK', 'MASTER_PORT')}
    env.update(FGNN_BENCH_DEVICE='0', FGNN_DIST_BACKEND='gloo')
    r = subprocess.run([sys.executable, os.path.join(root, 'bench.py'), '--gpus', '2', '--batch', '256', '--steps', '2',
                        '--warmup', '1', '--no-cpu-baseline'], env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith('{')]
    assert len(lines) == 1, r.stdout
    out = json.loads(lines[0])
    assert out['n_gpus'] == 2 and out['config']['global_batch'] == 512 and out['config']['parallelism'] == 'dp2'
    d = out['config']['distributed']
    assert len(d['per_rank_ms_per_step']) == 2 and d['launched_by'] == 'bench.py self_launch' and d['backend'] == 'gloo'
    assert out['value'] > 0 and out['scaling'] == 'weak'


def _worker_rccl(rank, world, port, out, graph):
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
    torch.cuda.set_device(0)
    dist.init_process_group('nccl', rank=rank, world_size=world, device_id=torch.device('cuda', 0))
    assert dist.get_backend() == 'nccl'
    flat, grad, p1 = _train(rank, world, graph, reduce_single_rank=True)      # the all-reduce IS issued, through RCCL
    torch.save({'param': flat, 'grad': grad, 'param1': p1, 'rccl': list(torch.cuda.nccl.version())},
               os.path.join(out, 'rccl_rank%d.pt' % rank))
    dist.destroy_process_group()


@pytest.mark.parametrize('graph', [False, True], ids=['eager', 'hipgraph'])
def test_rccl_backend_runs_the_flat_bucket_all_reduce(graph, tmp_path, dev):
    """The collective of the data-parallel step through RCCL itself (backend "nccl"), as bench.py's ranks issue it — here a
    group of ONE rank, which is what a one-GPU box can form (RCCL refuses two ranks on one device): communicator set-up on the
    rank's device, the all-reduce of the 5.5 MB flat gradient next to hipGraph replays of the step, FlatAdam behind it.  With one
    rank the reduction is the identity, so three steps must end on exactly the parameters of the plain single-process run."""
    mp.spawn(_worker_rccl, args=(1, _free_port(), str(tmp_path), graph), nprocs=1, join=True)
    a = torch.load(os.path.join(str(tmp_path), 'rccl_rank0.pt'))
    assert len(a['rccl']) >= 2
    ref_param, ref_grad, ref_p1 = _train(0, 1, graph)       # no process group at all
    assert torch.equal(a['grad'], ref_grad) and torch.equal(a['param1'], ref_p1) and torch.equal(a['param'], ref_param)


def test_adam_inside_the_step_graph_equals_the_eager_host_form(dev):
    """The one-GPU step as ONE hipGraph: forward, backward and the capturable FlatAdam (step count and learning rate in device
    memory, csrc/flat_adam.hip fgnn_flat_adam_dev) — what bench.py replays at N = 1.  Three replays, the learning rate halved
    before the third, must end exactly where the eager host-argument form ends.  (With N > 1 the RCCL all-reduce sits between
    backward and Adam and stays OUTSIDE the graph: captured through torch.distributed, ProcessGroupNCCL's watchdog polls the
    work's event recorded on the capturing stream — hipErrorCapturedEvent, process abort, ~1 run in 6 on a one-rank group.)"""
    full_param, full_grad, full_p1 = _train(0, 1, 'full')
    ref_param, ref_grad, ref_p1 = _train(0, 1, 'sched')     # eager, host-side Adam, same schedule
    assert torch.equal(full_grad, ref_grad) and torch.equal(full_p1, ref_p1)
    assert torch.equal(full_param, ref_param)


def _worker_rccl_capturable(rank, world, port, out):
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
    torch.cuda.set_device(0)
    dist.init_process_group('nccl', rank=rank, world_size=world, device_id=torch.device('cuda', 0))
    flat, grad, p1 = _train(rank, world, 'capturable', reduce_single_rank=True)
    torch.save({'param': flat, 'grad': grad, 'param1': p1}, os.path.join(out, 'rccl_cap_rank%d.pt' % rank))
    dist.destroy_process_group()


def test_rccl_all_reduce_then_device_state_adam_behind_graph_replays(tmp_path, dev):
    """What an N > 1 rank does per step: replay the forward + backward graph, RCCL all-reduce of the flat gradient (eager), the
    device-state Adam (eager) — on a one-rank RCCL group; same end state as the host-argument form without a process group."""
    mp.spawn(_worker_rccl_capturable, args=(1, _free_port(), str(tmp_path)), nprocs=1, join=True)
    a = torch.load(os.path.join(str(tmp_path), 'rccl_cap_rank0.pt'))
    ref_param, ref_grad, ref_p1 = _train(0, 1, 'sched')
    assert torch.equal(a['grad'], ref_grad) and torch.equal(a['param1'], ref_p1) and torch.equal(a['param'], ref_param)


def _worker_rccl_in_graph(rank, world, port, out):
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
    torch.cuda.set_device(0)
    dist.init_process_group('gloo', rank=rank, world_size=world)      # (only carries the communicator id: the collective itself is RCCL's)
    flat, grad, p1 = _train(rank, world, 'full', reduce_single_rank=True, direct=True, steps=200)
    torch.save({'param': flat, 'grad': grad, 'param1': p1}, os.path.join(out, 'rccl_graph_rank%d.pt' % rank))
    dist.destroy_process_group()


def test_rccl_all_reduce_inside_the_step_graph(tmp_path, dev):
    """The N > 1 step as ONE hipGraph: forward, backward, the flat gradient's all-reduce and the device-state Adam — with the
    collective issued through RCCL's C API on the capture stream (dp.DirectAllReduce: a communicator of its own, no
    torch.distributed work object for a watchdog to poll).  A one-rank communicator (what a one-GPU box can form), 200 replays
    without an abort, the learning rate changed between replays; the reduction over one rank is the identity, so the end state
    must be EXACTLY the one of the same 200 steps without any process group."""
    mp.spawn(_worker_rccl_in_graph, args=(1, _free_port(), str(tmp_path)), nprocs=1, join=True)
    a = torch.load(os.path.join(str(tmp_path), 'rccl_graph_rank0.pt'))
    ref_param, ref_grad, ref_p1 = _train(0, 1, 'full', steps=200)
    assert torch.equal(a['grad'], ref_grad) and torch.equal(a['param1'], ref_p1) and torch.equal(a['param'], ref_param)
