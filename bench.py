#!/usr/bin/env python3
"""bench.py — VF+FV messages/s of the FGNN hot path on the 96.3.963 LDPC graph (BASELINE.json).

A "step" is one pass of the hot path over one batch of synthetic codewords: by default one
TRAINING step of LDPCModel (8 FGNN layers, 32 fused message-operator calls = 6144 VF+FV messages
per codeword; forward + backward + gradient all-reduce + Adam), per-GPU batch 4096, inputs
resident in HBM.  `--mode fwd` times the inference forward instead.

  python bench.py --gpus N --steps K --warmup W          (N > 1 without a launcher: bench.py itself starts N ranks,
                                                          one per GPU, RCCL — see `self_launch`)
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
         --master-port P bench.py --gpus N --steps K --warmup W          (the driver's form; RANK / WORLD_SIZE from the env)

Rank 0 prints ONE JSON line: the contract fields plus
  "roofline"     — dominant hand-written kernel: algorithmic bytes (SURVEY §8d formula) / average
                   launch duration measured live with events on the launch stream, vs 8 TB/s HBM
  "cpu_baseline" — the CPU oracle (a port of the reference's op order) timed on this host's cores
                   on a bounded sample of the same workload (rank 0, N=1 only).
"""
import argparse
import contextlib
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(ROOT, 'factor-graph-neural-network_amd'))

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

HBM_PEAK_GBS = 8000.0        # MI355X_MICROARCH.md: HBM3E 8 TB/s spec (6.3 TB/s achievable)
F32_MFMA_PEAK_TFLOPS = 157.3  # MI355X_MICROARCH.md: v_mfma_f32_*_f32 = the f32 vector rate (155 TF measured)
BF16_MFMA_PEAK_TFLOPS = 2500.0  # dense bf16 MFMA


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=10)
    ap.add_argument('--warmup', type=int, default=3)
    ap.add_argument('--workload', choices=['ldpc', 'syn_pw', 'syn_hop'], default='ldpc',
                    help='ldpc: BASELINE configs 3/4 (the metric); syn_pw / syn_hop: factor_mpnn on 30-node synthetic PGMs '
                         '(configs 2 / 5: train_syn_pw_factor.py at batch 256, train_syn_hop_factor.py at batch 1024), f32')
    ap.add_argument('--batch', type=int, default=None, help='samples per GPU (weak scaling); default 4096 / 256 / 1024 by workload')
    ap.add_argument('--mode', choices=['train', 'fwd'], default='train')
    ap.add_argument('--dtype', choices=['f32', 'bf16'], default='bf16',
                    help='bf16 (BASELINE config 3): bf16 activations/messages + bf16 matrix cores in the forward, '
                         'f32 parameters / gradients / optimizer; f32: everything f32 (the 1e-4 parity path)')
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--miopen-bn', action='store_true', help='let torch route BatchNorm to MIOpen')
    ap.add_argument('--no-graph', action='store_true', help='launch every kernel from Python instead of replaying a hipGraph')
    ap.add_argument('--inputs', choices=['channel', 'features'], default='channel',
                    help='channel: encoded codewords through the reference channel (fgnn_amd/datapath.py); '
                         'features: random bits + unit-gain AWGN built with torch ops (ldpc.synthetic_batch)')
    ap.add_argument('--tables', choices=['shared', 'per_sample'], default='shared',
                    help="neighbour tables handed to the model: 'shared' = one table expanded over the batch (batch "
                         "stride 0); 'per_sample' = B materialised copies, the reference's calling convention "
                         "(train_ldpc.py:209-216) — recognised as one graph by a content check unless --no-dedupe")
    ap.add_argument('--no-dedupe', action='store_true',
                    help='with --tables per_sample: switch the content check off, so the kernels take the general '
                         'per-sample-graph path (a different graph per codeword would run at this speed)')
    ap.add_argument('--cpu-all-cores', action='store_true', help='add the all-host-cores CPU sample (a child process, 60 s limit) to the line')
    ap.add_argument('--cpu-baseline-only', action='store_true',
                    help='time only the CPU baseline (no GPU needed) with --cpu-batch / --cpu-threads / --mode and print its JSON object')
    ap.add_argument('--cpu-batch', type=int, default=512)
    ap.add_argument('--cpu-threads', type=int, default=16)
    ap.add_argument('--hop-order', type=int, default=9, help='syn_hop: order of the high-order factors (train_syn_hop_factor.py --hop_order)')
    a = ap.parse_args()
    if a.batch is None:
        a.batch = {'ldpc': 4096, 'syn_pw': 256, 'syn_hop': 1024}[a.workload]
    return a


def loss_fn(logits, snr_pred, label, sigma_b):
    """train_ldpc.py:222-227: BCE-with-logits + 0.1 * MSE on the burst-noise regressor (on the GPU: one launch forward, one backward —
    fgnn_amd.ldpc.decoding_loss; on the CPU the same torch expression the reference writes)."""
    from fgnn_amd.ldpc import decoding_loss
    return decoding_loss(logits, snr_pred, label, sigma_b, 0.1)


def cpu_baseline(batch, mode, threads, budget=20.0, max_iters=5):
    """The oracle (reference op order, PyTorch CPU, all host cores) on a bounded sample."""
    sys.path.insert(0, os.path.join(ROOT, 'oracle'))
    import fgnn_oracle as O
    from fgnn_amd.ldpc import LDPCModel, synthetic_batch, MESSAGES_PER_CODEWORD
    # intra-op threads: the operator's tensors are small (<= 4 MB per op at this batch); past ~16
    # threads PyTorch's CPU backend only adds contention (256 threads ran 200x slower), so the
    # baseline uses `threads` cores and says so
    cores = max(1, min(threads, os.cpu_count() or 1))
    torch.set_num_threads(cores)
    with contextlib.redirect_stdout(sys.stderr):
        model = LDPCModel(2, 6, 4, aggregator='max')
    sd = {k: v.detach().clone() for k, v in model.state_dict().items()}
    data = synthetic_batch(batch, torch.device('cpu'), seed=1, shared_graph=False)
    if mode == 'train':
        for k, v in sd.items():
            if v.is_floating_point() and 'running_' not in k and not k.startswith('h'):
                v.requires_grad_(True)

    def once():
        if mode == 'train':
            logits, snr = O.ldpc_model(sd, *data[:6], training=True)
            loss_fn(logits, snr, data[6], data[7]).backward()
        else:
            with torch.no_grad():
                O.ldpc_model(sd, *data[:6], training=False)

    t0 = time.time()
    once()
    times = [time.time() - t0]          # kept only if the budget allows nothing else
    t_end = time.time() + budget
    fresh = []
    while len(fresh) < max_iters and time.time() + times[0] < t_end:
        t0 = time.time()
        once()
        fresh.append(time.time() - t0)
    times = fresh or times
    times.sort()
    med = times[len(times) // 2]
    return {'value': MESSAGES_PER_CODEWORD * batch / med, 'unit': 'messages/s', 'cores': cores,
            'kind': 'port',
            'sample': 'oracle LDPCModel %s, batch %d, median of %d iterations (%.2f s each), '
                      '%d of %d host cores, torch %s CPU' % ('train fwd+bwd' if mode == 'train' else 'eval fwd', batch,
                                        len(times), med, cores, os.cpu_count() or 1,
                                        torch.__version__)}


def self_launch(args):
    """`python bench.py --gpus N` with N > 1 and no launcher around it (no WORLD_SIZE in the environment): start the N
    ranks here — one process per GPU, rendezvous on 127.0.0.1, backend RCCL ("nccl") — wait for them and return the exit
    code.  Rank 0 inherits stdout (the ONE JSON line), every rank's stderr is passed through.  Fails loudly when the node
    has fewer than N devices (FGNN_BENCH_DEVICE, the tests' hook for several ranks on one device over gloo, lifts that)."""
    import socket
    import subprocess
    n = args.gpus
    have = torch.cuda.device_count() if torch.cuda.is_available() else 0
    if os.environ.get('FGNN_BENCH_DEVICE') is None and have < n:
        raise SystemExit('bench.py --gpus %d: this node exposes %d ROCm device(s); one rank per GPU needs %d '
                         '(no oversubscription, no silent 1-rank run)' % (n, have, n))
    def attempt():
        s = socket.socket()
        s.bind(('127.0.0.1', 0))
        port = s.getsockname()[1]
        s.close()
        t_start = time.time()
        procs = []
        for r in range(n):
            env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n),
                       MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), FGNN_BENCH_SELF_LAUNCHED='1')
            env.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')       # dmabuf IPC: what RCCL needs on this driver
            env.setdefault('OMP_NUM_THREADS', str(max(1, (os.cpu_count() or 8) // n)))
            procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env,
                                          stdout=None if r == 0 else subprocess.DEVNULL))
        rc = 0
        try:
            pending = dict(enumerate(procs))
            while pending:
                for r, pr in list(pending.items()):
                    code = pr.poll()
                    if code is None:
                        continue
                    del pending[r]
                    if code != 0 and rc == 0:
                        rc = code
                        print('bench.py: rank %d exited with code %d; stopping the other ranks' % (r, code), file=sys.stderr)
                        for q in pending.values():
                            q.terminate()
                time.sleep(0.05)
        finally:
            for pr in procs:
                if pr.poll() is None:
                    pr.kill()
        return rc, time.time() - t_start

    # the rendezvous port is found by bind / close: another process can take it before rank 0 binds it again.  A launch that dies
    # within seconds (nothing has been printed yet: the JSON line comes last) is tried once more on a fresh port.
    rc, took = attempt()
    if rc != 0 and took < 30.0:
        print('bench.py: the ranks failed %.1f s after launch; retrying once on another rendezvous port' % took, file=sys.stderr)
        rc, took = attempt()
    return rc


AFFINITY = None


def dist_setup(args):
    """(rank, world, device) of this process.  world comes from WORLD_SIZE (a launcher's, or self_launch's) and must equal
    --gpus; with N > 1 the process group is RCCL on this rank's own device.  FGNN_BENCH_DEVICE / FGNN_DIST_BACKEND are the
    tests' hooks (several ranks sharing one GPU over gloo)."""
    rank = int(os.environ.get('RANK', '0'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    world = int(os.environ.get('WORLD_SIZE', '1'))
    if world != args.gpus:
        raise SystemExit('--gpus %d but WORLD_SIZE=%d: the launcher must start exactly --gpus ranks' % (args.gpus, world))
    if not torch.cuda.is_available():
        raise SystemExit('bench.py needs a ROCm device: the FGNN hot path has no CPU fallback')
    forced = os.environ.get('FGNN_BENCH_DEVICE')
    dev_index = int(forced) if forced is not None else local_rank
    if dev_index >= torch.cuda.device_count():
        raise SystemExit('rank %d wants device %d but only %d are visible' % (rank, dev_index, torch.cuda.device_count()))
    torch.cuda.set_device(dev_index)
    dev = torch.device('cuda', dev_index)
    # one rank = one GPU = its share of the cores NUMA-local to that GPU (the launcher, torch.distributed.run or self_launch,
    # leaves every rank floating over all cores); FGNN_BIND_CPUS=0 keeps the launcher's placement, =1 binds at N = 1 too
    bind = os.environ.get('FGNN_BIND_CPUS', '')
    global AFFINITY
    AFFINITY = None
    if (world > 1 and bind != '0') or bind == '1':
        from fgnn_amd.dp import bind_rank_to_local_cores
        try:
            AFFINITY = bind_rank_to_local_cores(dev_index, local_rank, int(os.environ.get('LOCAL_WORLD_SIZE', world)))
        except Exception as e:        # noqa: BLE001 — placement is an optimisation, never a reason to fail the run
            AFFINITY = {'note': 'binding failed: %s: %s' % (type(e).__name__, e)}
    if world > 1:
        backend = os.environ.get('FGNN_DIST_BACKEND', 'nccl')
        if backend == 'nccl':
            dist.init_process_group('nccl', rank=rank, world_size=world, device_id=dev)
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)
        if dist.get_world_size() != args.gpus:
            raise SystemExit('process group has %d ranks, --gpus says %d' % (dist.get_world_size(), args.gpus))
    return rank, world, dev


def dist_report(world, dev, elapsed_local, steps):
    """What a reader needs to trust an N > 1 line: the backend, the RCCL version, every rank's own step time and device."""
    if world == 1:
        return None
    backend = dist.get_backend()
    t = torch.tensor([elapsed_local / steps * 1e3, float(dev.index)], device=dev if backend == 'nccl' else 'cpu',
                     dtype=torch.float64)
    every = [torch.zeros_like(t) for _ in range(world)]
    dist.all_gather(every, t)
    ver = None
    if backend == 'nccl':
        try:
            ver = '.'.join(str(v) for v in torch.cuda.nccl.version())
        except Exception:       # noqa: BLE001
            ver = None
    placement = [None] * world
    dist.all_gather_object(placement, AFFINITY)
    from fgnn_amd.dp import xgmi_topology
    peers = None
    try:
        n = torch.cuda.device_count()
        peers = all(torch.cuda.can_device_access_peer(i, j) for i in range(n) for j in range(n) if i != j) if n > 1 else None
    except Exception:           # noqa: BLE001
        peers = None
    return {'backend': backend + (' (RCCL)' if backend == 'nccl' else ''), 'rccl_version': ver,
            'devices_visible': torch.cuda.device_count(),
            'rank_placement': placement, 'kfd_links': xgmi_topology() or None, 'all_peers_accessible': peers,
            'rccl_env': {k: v for k, v in os.environ.items() if k.startswith(('NCCL_', 'RCCL_', 'HSA_ENABLE_IPC'))},
            'per_rank_ms_per_step': [round(float(e[0]), 4) for e in every],
            'per_rank_device': [int(e[1]) for e in every],
            'launched_by': 'bench.py self_launch' if os.environ.get('FGNN_BENCH_SELF_LAUNCHED') else 'external launcher'}


def cpu_baseline_suite(args):
    """The default line's CPU legs (rank 0, N = 1), all bounded samples of the benched workload (the oracle's LDPCModel, reference
    op order; BASELINE.md §3: eval-mode forward AND train-mode fwd+bwd, batch 256).  `cpu_baseline` = the benched mode on
    `--cpu-threads` (16) threads — the fastest CPU configuration measured on the GPU boxes' hosts (PyTorch's CPU backend loses to
    contention beyond ~16 intra-op threads on these small tensors) —, `cpu_baseline_eval` / `cpu_baseline_train` = both modes
    (one of them is the same sample as `cpu_baseline`).  §3's `torch.set_num_threads(os.cpu_count())` taken literally
    (`cpu_baseline_all_cores`, a child process under a 60 s limit that one iteration does not finish within on the 224-thread
    hosts) is behind `--cpu-all-cores`: it cost a minute of every default run to report null."""
    import subprocess
    out = {'cpu_baseline': cpu_baseline(256, args.mode, args.cpu_threads, budget=12.0, max_iters=5)}
    other = 'fwd' if args.mode == 'train' else 'train'
    leg = cpu_baseline(256, other, args.cpu_threads, budget=10.0, max_iters=5)
    out['cpu_baseline_eval'] = leg if other == 'fwd' else out['cpu_baseline']
    out['cpu_baseline_train'] = leg if other == 'train' else out['cpu_baseline']
    if not args.cpu_all_cores:
        # BASELINE.md §3 asks for os.cpu_count() threads: the all-cores sample is committed (one iteration = ~300 s on the pool's
        # 256-thread hosts, which is why the live legs use --cpu-threads) and quoted here from its file
        for rnd in ('r06', 'r05'):
            pth = os.path.join(ROOT, 'profiles', rnd, 'cpu_baseline_all_cores.json')
            try:
                rec = json.loads(open(pth).read().strip().splitlines()[-1])
                rec['from_committed_profile'] = os.path.relpath(pth, ROOT)
                out['cpu_baseline_all_cores'] = rec
                break
            except (OSError, ValueError, IndexError):
                continue
        return out
    ncores = os.cpu_count() or 1
    limit = 60
    cmd = [sys.executable, os.path.abspath(__file__), '--cpu-baseline-only', '--cpu-batch', '256', '--cpu-threads', str(ncores),
           '--mode', args.mode]
    try:
        r = subprocess.run(cmd, capture_output=True, text=True, timeout=limit, env=dict(os.environ, OMP_NUM_THREADS=str(ncores)))
        lines = [ln for ln in r.stdout.splitlines() if ln.startswith('{')]
        out['cpu_baseline_all_cores'] = json.loads(lines[-1]) if lines else {'value': None, 'cores': ncores, 'sample': 'failed: ' + r.stderr[-200:]}
    except subprocess.TimeoutExpired:
        out['cpu_baseline_all_cores'] = {'value': None, 'unit': 'messages/s', 'cores': ncores, 'kind': 'port',
                                         'sample': 'oracle LDPCModel, batch 256, %d threads: one iteration did not finish within '
                                                   '%d s (intra-op oversubscription; the %d-thread samples above are the baseline)'
                                                   % (ncores, limit, args.cpu_threads)}
    return out


# ----------------------------------------------------------------------------------------------------------------------
# synthetic-PGM workloads (BASELINE configs 2 / 5): factor_mpnn as train_syn_pw_factor.py / train_syn_hop_factor.py build it
# ----------------------------------------------------------------------------------------------------------------------
SYN_DIMS = [64, 64, 128, 128, 256, 256, 128, 128, 64, 64, 2]


def syn_tables(workload, hop_order):
    """(hop feature dim, nodes of the high-order factor axis, pw table, pw edge features, high table, high edge features)."""
    from fgnn_amd import tables
    pw_idx, pw_ef = tables.pw_factor_table(30)
    if workload == 'syn_pw':                              # train_syn_pw_factor.py: one chain factor per graph besides the pairwise ones
        hi_idx, hi_ef, _ = tables.chain_high_table(30, 9)
        return 1, 1, pw_idx, pw_ef, hi_idx, hi_ef
    hi_idx, hi_ef = tables.ring_hop_table(30, hop_order)  # train_syn_hop_factor.py: 30 order-k factors on a ring
    return hop_order, 30, pw_idx, pw_ef, hi_idx, hi_ef


def syn_messages_per_graph(model, tabs):
    """VF+FV messages of one forward: edges (M x k) of every message-operator call."""
    from fgnn_amd.mpnn import mp_conv_residual, mp_conv_v2
    edges = [t.shape[0] * t.shape[1] for t in tabs]
    total = 0
    for row in model.mp_nn_modules:
        for j, m in enumerate(row):
            if isinstance(m, (mp_conv_v2, mp_conv_residual)):
                total += edges[j]
    return total


def syn_cpu_baseline(workload, hop_order, batch, mode, threads, budget=20.0, max_iters=5):
    """The oracle's factor_mpnn (reference op order, PyTorch CPU) on a bounded sample of the same workload."""
    sys.path.insert(0, os.path.join(ROOT, 'oracle'))
    import fgnn_oracle as O
    import fgnn_amd
    cores = max(1, min(threads, os.cpu_count() or 1))
    torch.set_num_threads(cores)
    hop_dim, hi_nodes, pw_idx, pw_ef, hi_idx, hi_ef = syn_tables(workload, hop_order)
    with contextlib.redirect_stdout(sys.stderr):
        model = fgnn_amd.factor_mpnn(2, [4, hop_dim], SYN_DIMS, [16, 16])
    sd = {k: v.detach().clone() for k, v in model.state_dict().items()}
    g = torch.Generator().manual_seed(1)
    nf, pws = torch.rand(batch, 2, 30, 1, generator=g), torch.rand(batch, 4, 30, 1, generator=g)
    hi = torch.rand(batch, hop_dim, hi_nodes, 1, generator=g)
    label = torch.randint(0, 2, (batch, 30), generator=g)
    et_pw, et_hi = torch.randn(1, 16, *pw_idx.shape, generator=g), torch.randn(1, 16, *hi_idx.shape, generator=g)
    gs = [[torch.from_numpy(pw_idx)[None].repeat(batch, 1, 1), et_pw.repeat(batch, 1, 1, 1)],
          [torch.from_numpy(hi_idx)[None].repeat(batch, 1, 1), et_hi.repeat(batch, 1, 1, 1)]]
    if mode == 'train':
        for k, v in sd.items():
            if v.is_floating_point() and 'running_' not in k:
                v.requires_grad_(True)

    def once():
        if mode == 'train':
            pred, _ = O.factor_mpnn(sd, '', nf, [pws, hi], gs, dims=SYN_DIMS, netypes=[16, 16], training=True)
            torch.nn.functional.cross_entropy(pred.squeeze(-1).permute(0, 2, 1).reshape(-1, 2), label.reshape(-1)).backward()
        else:
            with torch.no_grad():
                O.factor_mpnn(sd, '', nf, [pws, hi], gs, dims=SYN_DIMS, netypes=[16, 16], training=False)

    t0 = time.time()
    once()
    times = [time.time() - t0]
    t_end = time.time() + budget
    fresh = []
    while len(fresh) < max_iters and time.time() + times[0] < t_end:
        t0 = time.time()
        once()
        fresh.append(time.time() - t0)
    times = sorted(fresh or times)
    med = times[len(times) // 2]
    msgs = syn_messages_per_graph(model, [pw_idx, hi_idx])
    return {'value': msgs * batch / med, 'unit': 'messages/s', 'cores': cores, 'kind': 'port',
            'sample': 'oracle factor_mpnn (%s) %s, batch %d graphs, median of %d iterations (%.2f s each), %d of %d host cores, '
                      'torch %s CPU' % (workload, 'train fwd+bwd' if mode == 'train' else 'eval fwd', batch, len(times), med,
                                        cores, os.cpu_count() or 1, torch.__version__)}


def main_syn(args):
    """factor_mpnn training (or inference) step on synthetic 30-node PGMs, f32: edge models -> factor_mpnn (12 fused message
    operator calls through csrc/mpconv_fwd_ext.hip / mpconv_bwd_ext.hip) -> cross entropy -> backward -> gradient-norm clip
    -> Adam (train_syn_hop_factor.py:283-303).  Same JSON contract as the LDPC workload; `value` counts VF+FV messages."""
    rank, world, dev = dist_setup(args)
    torch.backends.cudnn.enabled = bool(args.miopen_bn)
    import fgnn_amd
    from fgnn_amd import ops
    from fgnn_amd.dp import FlatAdam, FlatGradBucket, broadcast_parameters

    torch.manual_seed(0)
    hop_dim, hi_nodes, pw_idx, pw_ef, hi_idx, hi_ef = syn_tables(args.workload, args.hop_order)
    C = torch.nn.Conv2d
    with contextlib.redirect_stdout(sys.stderr):
        model = fgnn_amd.factor_mpnn(2, [4, hop_dim], SYN_DIMS, [16, 16]).to(dev)
    em_pw = torch.nn.Sequential(C(3, 64, 1), torch.nn.ReLU(inplace=True), C(64, 16, 1)).to(dev)
    em_hi = torch.nn.Sequential(C(hi_ef.shape[0], 64, 1), torch.nn.ReLU(inplace=True), C(64, 16, 1)).to(dev)
    everything = torch.nn.ModuleList([model, em_pw, em_hi])
    broadcast_parameters(everything)
    B = args.batch
    g = torch.Generator().manual_seed(100 + rank)
    nf = torch.rand(B, 2, 30, 1, generator=g).to(dev)                  # log-potentials ~ U(0, 1) (lib/data/random_pgm.py)
    pws = torch.rand(B, 4, 30, 1, generator=g).to(dev)
    hi = torch.rand(B, hop_dim, hi_nodes, 1, generator=g).to(dev)
    label = torch.randint(0, 2, (B, 30), generator=g).to(dev)
    idx_pw, idx_hi = torch.from_numpy(pw_idx).to(dev)[None], torch.from_numpy(hi_idx).to(dev)[None]
    ef_pw, ef_hi = torch.from_numpy(pw_ef).to(dev)[None], torch.from_numpy(hi_ef).to(dev)[None]
    if args.tables == 'per_sample':                                    # B materialised copies, as the reference's .repeat passes them
        rep_i, rep_e = (lambda t: t.repeat(B, 1, 1)), (lambda t: t.repeat(B, 1, 1, 1))
    else:
        rep_i, rep_e = (lambda t: t.expand(B, -1, -1)), (lambda t: t.expand(B, -1, -1, -1))
    if args.no_dedupe:
        ops.DEDUPE_GRAPHS = False
    train = args.mode == 'train'
    everything.train(train)
    if train:
        bucket = FlatGradBucket(everything.parameters(), flatten_params=True)
        opt = FlatAdam(bucket, lr=3e-3)

    def forward():
        et_pw, et_hi = em_pw(ef_pw), em_hi(ef_hi)
        pred, _ = model(nf, [pws, hi], [[rep_i(idx_pw), rep_e(et_pw)], [rep_i(idx_hi), rep_e(et_hi)]])
        return pred

    def compute():
        if train:
            bucket.zero()
            pred = forward()
            torch.nn.functional.cross_entropy(pred.squeeze(-1).permute(0, 2, 1).reshape(-1, 2), label.reshape(-1)).backward()
        else:
            with torch.no_grad():
                forward()

    graphed = None
    if not args.no_graph:
        try:
            from fgnn_amd.graph import StepGraph
            graphed = StepGraph(compute)
        except Exception as e:           # noqa: BLE001 — report and fall back to eager launches
            print('bench.py: hipGraph capture failed (%s: %s); running eagerly' % (type(e).__name__, e), file=sys.stderr)
            graphed = None

    def step():
        if graphed is not None:
            graphed.replay()
        else:
            compute()
        if train:
            if os.environ.get('FGNN_BENCH_STEP_TIMES') is not None:
                torch.cuda.synchronize()
                ta = time.perf_counter()
                bucket.all_reduce_sum()
                torch.cuda.synchronize()
                gf = bucket.flat
                print('rank %d all-reduce %.3f ms; finite %s, |g| %.3e, denormal %d' % (
                    rank, (time.perf_counter() - ta) * 1e3, bool(torch.isfinite(gf).all()), float(gf.abs().max()),
                    int(((gf != 0) & (gf.abs() < 1.2e-38)).sum())), file=sys.stderr)
            else:
                bucket.all_reduce_sum()
            # torch.nn.utils.clip_grad_norm(parameters, 1.0) on the flat gradient, without a host round trip
            gflat = bucket.flat
            norm = torch.linalg.vector_norm(gflat) / world
            gflat.mul_(torch.clamp(1.0 / (norm + 1e-6), max=1.0))
            opt.step(grad_scale=1.0 / world)

    def fence():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
            torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    fence()
    trace_steps = os.environ.get('FGNN_BENCH_STEP_TIMES') is not None   # diagnosis: per-step wall times (adds a device sync per step)
    t0 = time.perf_counter()
    for _ in range(args.steps):
        t1 = time.perf_counter()
        step()
        if trace_steps:
            torch.cuda.synchronize()
            print('rank %d step %.3f ms' % (rank, (time.perf_counter() - t1) * 1e3), file=sys.stderr)
    fence()
    elapsed = time.perf_counter() - t0
    dist_info = dist_report(world, dev, elapsed, args.steps)
    if world > 1:
        t = torch.tensor([elapsed], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    roofline, kernels = None, {}
    if rank == 0:
        ops.SIDE_STREAM = False
        compute()
        torch.cuda.synchronize()
        ops.TIMER = ops.KernelTimer()
        compute()
        ops.SIDE_STREAM = True
        kernels = ops.TIMER.summary()
        ops.TIMER = None
        mp = [(k, v) for k, v in kernels.items() if k.startswith('mpconv_')]
        if mp:
            sym, r = max(mp, key=lambda kv: kv[1]['ms'])
            avg_ms = r['ms'] / r['launches']
            tfs = (r['flops'] / r['launches']) / (avg_ms * 1e-3) / 1e12
            gbs = (r['bytes'] / r['launches']) / (avg_ms * 1e-3) / 1e9
            try:       # HBM-side bytes per launch from the committed PMC passes (order-9 shape, batch 1024: profiles/r02/README.md)
                traffic = json.load(open(os.path.join(ROOT, 'profiles', 'r02', 'pmc_traffic_syn.json')))['kernels'].get(sym, {}).get(
                    'traffic_bytes_per_launch') if (args.workload == 'syn_hop' and B == 1024) else None
            except (OSError, ValueError, KeyError):
                traffic = None
            roofline = {'bound': 'mfma', 'achieved': round(tfs, 2), 'peak': F32_MFMA_PEAK_TFLOPS, 'unit': 'TFLOP/s',
                        'frac': round(tfs / F32_MFMA_PEAK_TFLOPS, 4), 'traffic': traffic, 'kernel': sym,
                        'launches_per_step': r['launches'], 'avg_launch_us': round(avg_ms * 1e3, 2),
                        'algorithmic_bytes_per_launch': r['bytes'] // r['launches'],
                        'algorithmic_flops_per_launch': r['flops'] // r['launches'],
                        'hbm': {'bound': 'hbm', 'achieved': round(gbs, 1), 'peak': HBM_PEAK_GBS, 'unit': 'GB/s',
                                'frac': round(gbs / HBM_PEAK_GBS, 4), 'traffic': None}}
            if 'fwd_extp_kernel' in sym and 'true>' in sym:
                # this kernel runs its projections as a three-term bf16 split on the bf16 matrix cores (6 MFMAs per f32 product
                # term set, f32-level accuracy: csrc/mpconv_fwd_ext.hip): `achieved` counts the ALGORITHMIC f32 FLOPs against the
                # f32 matrix-core peak, the executed bf16 FLOPs (6x the projection) against the bf16 peak are given beside it
                roofline['arithmetic'] = 'f32 result from a 3-term bf16 operand split, 6 bf16 MFMAs per product, f32 accumulation'
                roofline['executed_bf16'] = {'achieved': round(6.0 * tfs, 1), 'peak': BF16_MFMA_PEAK_TFLOPS, 'unit': 'TFLOP/s',
                                             'frac': round(6.0 * tfs / BF16_MFMA_PEAK_TFLOPS, 4)}
            if 'bwd_extq_kernel' in sym:
                # round 6: the backward's three GEMMs run as bf16 pieces on the bf16 matrix cores (csrc/mpconv_bwd_ext.hip): NP = 2 pieces
                # = 3 MFMAs per fragment product (gradients within 5e-6 of the exact-f32 kernel's), NP = 3 = 6 (4e-7).  `achieved` / `frac`
                # stay the ALGORITHMIC f32 FLOPs against the f32 matrix-core peak (what the exact kernel is bounded by: it reached 0.43);
                # the bf16 FLOPs the kernel really issues, against the bf16 peak, ride beside them
                terms = 3.0 if 'kernel<2' in sym else 6.0
                roofline['arithmetic'] = ('f32 operands as %d bf16 pieces, %d bf16 MFMAs per fragment product, f32 accumulation'
                                          % (2 if terms == 3.0 else 3, int(terms)))
                roofline['executed_bf16'] = {'achieved': round(terms * tfs, 1), 'peak': BF16_MFMA_PEAK_TFLOPS, 'unit': 'TFLOP/s',
                                             'frac': round(terms * tfs / BF16_MFMA_PEAK_TFLOPS, 4)}
    fence()
    if rank == 0:
        msgs = syn_messages_per_graph(model, [pw_idx, hi_idx])
        out = {
            'metric': 'VF+FV messages/sec on 30-node synthetic PGMs (%s)' % args.workload,
            'value': msgs * B * world * args.steps / elapsed, 'unit': 'messages/s', 'n_gpus': world,
            'steps': args.steps, 'warmup': args.warmup, 'ms_per_step': elapsed / args.steps * 1e3,
            'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None, 'dtype': 'f32', 'data': 'synthetic',
            'config': {'workload': '%s: factor_mpnn(2, [4, %d], %s, [16, 16]) + two edge models, %s step, batch %d graphs per GPU '
                                   '(30 variables, 30 pairwise factors k=2, %s), graph tables %s'
                                   % (args.workload, hop_dim, SYN_DIMS,
                                      'training (fwd + bwd + grad all-reduce + norm clip + Adam)' if train else 'inference forward', B,
                                      'one chain factor' if args.workload == 'syn_pw' else '30 order-%d factors' % args.hop_order,
                                      'batch-shared (expand)' if args.tables == 'shared' else 'per-sample copies (repeat)'),
                       'messages_per_graph': msgs, 'graphs_per_s': B * world * args.steps / elapsed,
                       'global_batch': B * world, 'parallelism': 'dp%d' % world, 'mode': args.mode,
                       'hip_graph': graphed is not None, 'distributed': dist_info,
                       'peak_hbm_GB': round(torch.cuda.max_memory_allocated(dev) / 1e9, 2)},
            'roofline': roofline,
            'kernels': {k: {'launches': v['launches'], 'avg_us': round(v['ms'] / v['launches'] * 1e3, 2),
                            'avg_us_in_step': round(v.get('ms2', v['ms']) / v['launches'] * 1e3, 2),
                            'total_ms': round(v['ms'], 3),
                            'algorithmic_GBs': round(v['bytes'] / max(v['ms'], 1e-9) / 1e6, 1)}
                        for k, v in sorted(kernels.items(), key=lambda kv: -kv[1]['ms'])},
        }
        if world == 1 and not args.no_cpu_baseline:
            out['cpu_baseline'] = syn_cpu_baseline(args.workload, args.hop_order, min(args.cpu_batch, 64), args.mode, args.cpu_threads)
        else:
            out['cpu_baseline'] = None
        print(json.dumps(out))
    if world > 1:
        dist.destroy_process_group()


def _trace(msg):
    if os.environ.get('FGNN_BENCH_TRACE'):
        print('[bench rank %s] %s' % (os.environ.get('RANK', '0'), msg), file=sys.stderr, flush=True)


def main():
    args = parse()
    if args.cpu_baseline_only:                       # the child leg of cpu_baseline_suite (or a manual CPU timing): no GPU touched
        print(json.dumps(cpu_baseline(args.cpu_batch, args.mode, args.cpu_threads, budget=40.0, max_iters=2)))
        return
    if args.gpus > 1 and 'WORLD_SIZE' not in os.environ:
        sys.exit(self_launch(args))                  # N ranks of this same command, one per GPU
    if args.workload != 'ldpc':
        return main_syn(args)
    rank, world, dev = dist_setup(args)

    # torch's own channels-last BatchNorm kernels beat MIOpen's spatial BN on these [B,C,N,1]
    # activations (profiles/r01), so MIOpen is bypassed for the plumbing ops by default
    torch.backends.cudnn.enabled = bool(args.miopen_bn)
    from fgnn_amd import ops
    from fgnn_amd.dp import FlatAdam, FlatGradBucket, broadcast_parameters
    from fgnn_amd.ldpc import LDPCModel, MESSAGES_PER_CODEWORD, synthetic_batch

    _trace('process group up')
    torch.manual_seed(0)
    dtype = torch.float32 if args.dtype == 'f32' else torch.bfloat16
    with contextlib.redirect_stdout(sys.stderr):         # the constructors announce their aggregator (as the reference does)
        model = LDPCModel(2, 6, 4, aggregator='max').to(dev)
    broadcast_parameters(model)
    _trace('parameters broadcast')
    if args.inputs == 'channel':
        # the GPU data path (fgnn_amd/datapath.py): random messages, the reference's G encode, its AWGN + burst
        # channel at 0..4 dB, features gathered on the device — generated once, resident before the timed region
        from fgnn_amd.datapath import LdpcDataPath
        data = LdpcDataPath(dev).sample(args.batch, seed=100 + rank, dtype=dtype)
        data = data[:6] + (data[6][:, :48].float().contiguous(), data[7])
    else:
        data = synthetic_batch(args.batch, dev, seed=100 + rank, dtype=dtype)
    if args.tables == 'per_sample':
        data = data[:2] + (data[2].contiguous(), data[3].contiguous()) + data[4:]      # B materialised copies
        assert data[2].stride(0) != 0
    if args.no_dedupe:
        ops.DEDUPE_GRAPHS = False
    inputs, label, sigma_b = data[:6], data[6], data[7]
    train = args.mode == 'train'
    model.train(train)
    # N = 1: the optimizer is recorded into the step graph too (FlatAdam(capturable=True): step count and learning rate in device
    # memory).  N > 1: the RCCL all-reduce sits between the backward and Adam, and it must NOT be captured through
    # torch.distributed: ProcessGroupNCCL's watchdog thread polls the work's completion event, which was recorded on a capturing
    # stream — hipErrorCapturedEvent, the process aborts (seen in ~1 of 6 runs of a one-rank group on the GPU box).  So with N > 1
    # the graph ends with the backward and the collective + Adam are two eager launches behind every replay.
    # Round 5: with N > 1 the all-reduce goes through RCCL's C API on the capture stream (dp.DirectAllReduce: a communicator of its
    # own, no torch.distributed work object for the watchdog to poll), so the WHOLE step — forward, backward, collective, Adam — is one
    # graph at every N.  FGNN_NO_RCCL_DIRECT=1 (or a non-RCCL backend) keeps the round-4 form: graph = forward + backward, then two
    # eager launches.
    direct = train and world > 1 and dist.get_backend() == 'nccl' and os.environ.get('FGNN_NO_RCCL_DIRECT') is None
    whole_in_graph = train and not args.no_graph and (world == 1 or direct)
    if train:
        # parameters and gradients live in two flat f32 buffers: one all-reduce, and Adam (the reference's
        # lr / weight_decay, train_ldpc.py) is eight elementwise kernels instead of a 330-tensor sweep
        bucket = FlatGradBucket(model.parameters(), flatten_params=True)
        if direct:
            try:
                bucket.use_direct_all_reduce()
            except Exception as e:      # noqa: BLE001 — report, keep the eager collective
                print('bench.py: RCCL C-API communicator failed (%s: %s); the all-reduce stays outside the graph' % (type(e).__name__, e), file=sys.stderr)
                direct = False
                whole_in_graph = train and not args.no_graph and world == 1
        opt = FlatAdam(bucket, lr=1e-4, weight_decay=1e-8, capturable=whole_in_graph)

    if not train:
        from fgnn_amd.dp import flatten_parameters
        flatten_parameters(p for p in model.parameters() if p.dtype == torch.float32)     # one weight-cast kernel per forward

    # bf16: activations / messages are stored in bf16 (the message kernels then use bf16 matrix cores for
    # the forward), parameters, gradients and optimizer state stay f32 (autocast for the node-wise GEMMs)
    amp = torch.autocast(device_type='cuda', dtype=torch.bfloat16, enabled=(args.dtype == 'bf16'))

    def compute():                      # everything but the collective and the optimizer: graph-capturable
        if train:
            bucket.zero()
            with amp:
                logits, snr = model(*inputs)
            loss_fn(logits, snr, label, sigma_b).backward()
        else:
            with torch.no_grad(), amp:
                model(*inputs)

    def update():
        bucket.all_reduce_sum()                     # one RCCL all-reduce of the flat gradient; the mean is folded into Adam
        opt.step(grad_scale=1.0 / world)

    def whole():
        compute()
        update()

    # The step is launch-bound at this batch (~1500 launches): record it once into a hipGraph and replay it.
    graphed = None
    if not args.no_graph:
        try:
            from fgnn_amd.graph import StepGraph
            # inference: the model is frozen — its folded BatchNorm affines and bf16 weight copies are built once (by the warm-up
            # runs), not re-derived inside every replay; training re-derives them every step (the parameters move)
            graphed = StepGraph(whole if whole_in_graph else compute, static_params=not train)
        except Exception as e:           # noqa: BLE001 — report and fall back
            graphed = None
            if whole_in_graph and world > 1:      # the collective would not capture: the round-4 form (graph = forward + backward)
                print('bench.py: capturing the step WITH the RCCL all-reduce failed (%s: %s); capturing forward + backward only'
                      % (type(e).__name__, e), file=sys.stderr)
                whole_in_graph = False
                try:
                    graphed = StepGraph(compute, static_params=not train)
                except Exception as e2:  # noqa: BLE001
                    print('bench.py: hipGraph capture failed (%s: %s); running eagerly' % (type(e2).__name__, e2), file=sys.stderr)
            else:
                print('bench.py: hipGraph capture failed (%s: %s); running eagerly' % (type(e).__name__, e), file=sys.stderr)

    _trace('graph captured: %s' % (graphed is not None))

    def step(eager=False):
        if graphed is not None and not eager:
            graphed.replay()
            if train and not whole_in_graph:
                update()
        else:
            compute()
            if train:
                update()

    def fence():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
            torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    _trace('warmup enqueued')
    fence()
    _trace('warmup done')
    trace_steps = os.environ.get('FGNN_BENCH_STEP_TIMES') is not None   # diagnosis: per-step wall times (adds a device sync per step)
    host_times = os.environ.get('FGNN_BENCH_HOST_TIMES') is not None    # diagnosis: how far ahead of the device does the host run?
    marks, host_done = [], []
    if host_times:
        e0 = torch.cuda.Event(enable_timing=True)
        e0.record()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        t1 = time.perf_counter()
        step()
        if trace_steps:
            torch.cuda.synchronize()
            print('rank %d step %.3f ms' % (rank, (time.perf_counter() - t1) * 1e3), file=sys.stderr)
        if host_times:
            e = torch.cuda.Event(enable_timing=True)
            e.record()
            marks.append(e)
            host_done.append((time.perf_counter() - t0) * 1e3)
    host_enqueue = time.perf_counter() - t0      # the host's share: when it is close to `elapsed` the replay is launch-bound
    fence()
    elapsed = time.perf_counter() - t0
    _trace('timed steps done')
    if host_times:
        print('rank %d: host enqueue %.3f ms per step, wall %.3f ms per step' % (rank, host_enqueue / args.steps * 1e3, elapsed / args.steps * 1e3),
              file=sys.stderr)
        print('   step: host returned at / device finished at (ms from the loop start): ' +
              ' '.join('%.1f/%.1f' % (h, e0.elapsed_time(m)) for h, m in zip(host_done, marks)), file=sys.stderr)
    if graphed is not None and getattr(graphed, 'stamps', None):
        graphed.stamp_report(sys.stderr)
    dist_info = dist_report(world, dev, elapsed, args.steps)
    if world > 1:
        t = torch.tensor([elapsed], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    # Instrumented steps: per-kernel-symbol launch durations from events on the launch stream.  TWO passes: (1) one stream —
    # each kernel alone on the chip ("isolated"); (2) the two streams of the timed step — the side branch's kernels run beside
    # the operator's, as inside the replayed graph ("in_step").  The roofline fractions quoted are the IN-STEP ones; the
    # committed rocprofv3 kernel trace of the replayed graph (profiles/r04) carries the in-graph averages they should agree with.
    roofline = None
    kernels = {}
    if rank == 0:
        ops.SIDE_STREAM = False
        compute()        # untimed eager pass: first eager launches of a kernel variant pay its code-object load (10 ms once seen)
        torch.cuda.synchronize()
        ops.TIMER = ops.KernelTimer()
        compute()        # eager, WITHOUT the collective / optimizer: the other ranks are not taking part
        kernels = ops.TIMER.summary()
        ops.SIDE_STREAM = True
        compute()
        torch.cuda.synchronize()
        ops.TIMER = ops.KernelTimer()
        compute()
        kernels2 = ops.TIMER.summary()
        ops.TIMER = None
        if kernels:
            try:
                pmc = {}
                for rnd in ('r01', 'r02', 'r03', 'r04', 'r05', 'r06'):     # later rounds' passes override (new kernel families; one pass PER INSTANCE)
                    pth = os.path.join(ROOT, 'profiles', rnd, 'pmc_traffic.json')
                    if os.path.exists(pth):
                        pmc.update(json.load(open(pth))['kernels'])
            except (OSError, ValueError, KeyError):
                pmc = {}
            # Figures of an EARLIER run — the committed rocprofv3 kernel trace of the replayed graph — ride along under their own key
            # (`from_committed_profile`: file, and the library kernels of the step that `kernels` below cannot see because they are not
            # launched through ops.timed); everything else in this line was measured by THIS run.
            in_graph, in_graph_file, library = {}, None, []
            try:
                import csv
                for rnd in ('r06', 'r05', 'r04'):
                    pth = os.path.join(ROOT, 'profiles', rnd, 'final_bf16_%s_kernel_stats_top40.csv' % ('train' if train else 'fwd'))
                    if os.path.exists(pth):
                        in_graph_file = os.path.relpath(pth, ROOT)
                        for row in csv.DictReader(open(pth)):
                            in_graph[row['Name']] = float(row['AverageNs'])
                            if row['Name'].startswith(('Cijk_', 'void at::native', 'at::native', 'void rocblas', 'rocblas')):
                                library.append({'kernel': row['Name'][:60], 'calls_in_trace': int(row['Calls']),
                                                'avg_us': round(float(row['AverageNs']) / 1e3, 2), 'pct_of_trace': float(row['Percentage'])})
                        break
            except (OSError, ValueError, KeyError):
                pass
            # ... and the same trace of the step on ONE stream: every kernel alone on the chip inside the replayed graph (the two-stream
            # averages carry the contention of whatever runs beside a launch on the other queue: an operator launch holds every CU)
            in_graph1, in_graph1_file = {}, None
            try:
                for rnd in ('r06', 'r05'):
                    pth = os.path.join(ROOT, 'profiles', rnd, 'final_bf16_train1_kernel_stats_top40.csv')
                    if train and os.path.exists(pth):
                        in_graph1_file = os.path.relpath(pth, ROOT)
                        for row in csv.DictReader(open(pth)):
                            in_graph1[row['Name']] = float(row['AverageNs'])
                        break
            except (OSError, ValueError, KeyError):
                pass

            def trace_row(sym, table):
                """Average duration (ns) of `sym` in a kernel-stats table: the symbol as the dispatcher names it, or with the row-stride /
                node-loop template arguments the backward kernel grew in round 5 (<KC, DEG> = <KC, DEG, 64[, NL]>)."""
                s0 = sym.replace(' ', '')
                for cand in (s0, s0[:-1] + ',64>' if s0.endswith('>') else None, s0[:-1] + ',64,' if s0.endswith('>') else None):
                    if cand:
                        for name, ns in table.items():
                            if cand in name.replace(' ', ''):
                                return ns
                return None

            def describe(sym):
                r, r2 = kernels[sym], kernels2.get(sym)
                n = r['launches']
                iso_ms = r['ms'] / n
                step_ms = (r2['ms'] / r2['launches']) if r2 else iso_ms
                nb, nf = r['bytes'] / n, r['flops'] / n
                gbs = nb / (step_ms * 1e-3) / 1e9
                # bf16 storage + bf16 MFMA (every operator kernel of a --dtype bf16 run, the hyper-factor fan-in / fan-out ones included:
                # their symbols carry no 'b16'): HBM-bound (AI ~80 << ridge ~312); the f32 kernels price against the f32 matrix rate
                bf16 = args.dtype == 'bf16' or any(t in sym for t in ('b16', '_sg_', '_ws_'))
                # (this round's passes name the backward kernel with its row-stride and node-loop arguments: <KC, DEG> = <KC, DEG, 64[, NL]>)
                pm = next((v for k, v in pmc.items() if k.startswith(sym[:-1] + ', 64')), None) or pmc.get(sym) or {}
                d = {'kernel': sym, 'launches_per_step': n, 'bound': 'hbm' if bf16 else 'mfma',
                     'algorithmic_bytes_per_launch': int(nb), 'algorithmic_flops_per_launch': int(nf),
                     'avg_launch_us_isolated': round(iso_ms * 1e3, 2), 'avg_launch_us_in_step': round(step_ms * 1e3, 2),
                     'achieved': round(gbs, 1), 'peak': HBM_PEAK_GBS, 'unit': 'GB/s', 'frac': round(gbs / HBM_PEAK_GBS, 4),
                     'frac_isolated': round(nb / (iso_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
                     'traffic': pm.get('traffic_bytes_per_launch'), 'mfma_busy': pm.get('mfma_busy')}
                # (two-launch calls — ' x2' — and launches with addends have no one-to-one row in a per-symbol average: no figure)
                if in_graph_file and not sym.endswith((' x2', ' k2')):
                    ns = trace_row(sym, in_graph)
                    if ns is not None:
                        us = round(ns / 1e3, 2)
                        d['from_committed_profile'] = {'file': in_graph_file, 'avg_launch_us_in_graph_rocprof': us,
                                                       'frac_in_graph_rocprof': round(nb / (us * 1e-6) / 1e9 / HBM_PEAK_GBS, 4),
                                                       'note': 'average over ALL launches of the symbol in an earlier, committed trace of the replayed graph'}
                        ns1 = trace_row(sym, in_graph1)
                        if ns1 is not None:
                            us1 = round(ns1 / 1e3, 2)
                            d['from_committed_profile'].update({'one_stream_file': in_graph1_file, 'avg_launch_us_in_graph_one_stream': us1,
                                                                'frac_in_graph_one_stream': round(nb / (us1 * 1e-6) / 1e9 / HBM_PEAK_GBS, 4)})
                if not bf16:                              # the f32-MFMA kernels (exact v_mfma_f32_16x16x4_f32) sit above the f32 ridge: matrix-core-bound
                    tfs = nf / (step_ms * 1e-3) / 1e12
                    d.update({'achieved': round(tfs, 2), 'peak': F32_MFMA_PEAK_TFLOPS, 'unit': 'TFLOP/s',
                              'frac': round(tfs / F32_MFMA_PEAK_TFLOPS, 4), 'hbm_frac': round(gbs / HBM_PEAK_GBS, 4)})
                return d

            def family(prefix):
                syms = [k for k in kernels if k.startswith(prefix)]
                if not syms:
                    return None, None
                top = max(syms, key=lambda k: kernels[k]['ms'])
                t2 = sum((kernels2.get(k) or kernels[k])['ms'] for k in syms)
                nb = sum(kernels[k]['bytes'] for k in syms)
                return describe(top), round(nb / (t2 * 1e-3) / 1e9 / HBM_PEAK_GBS, 4)

            fwd, fwd_frac = family('mpconv_fwd')
            bwd, bwd_frac = family('mpconv_bwd')
            # the top-level object = the operator kernel with the largest total time (the contract's "dominant kernel"); both
            # directions ride along, and operator_*_frac = sum of the direction's algorithmic bytes / sum of its in-step time
            # over ALL its calls
            dom = max([d for d in (fwd, bwd) if d], key=lambda d: kernels[d['kernel']]['ms'])

            def timed_configuration(d):
                """`frac` / `achieved` / `avg_launch_us` of the configuration that is TIMED: the kernel's average launch inside the
                replayed two-stream graph.  HIP events cannot bracket a launch inside a replayed graph, and an eager step — launch-bound —
                shows every kernel alone on the chip, so the in-graph duration is the one of the committed rocprofv3 kernel trace of
                this same command (profiles/rNN, named in `frac_source`); the figures this run measured live stay beside it as
                `frac_live_eager` (two streams, eager) and `frac_isolated` (one stream).  Without a committed trace for the symbol the
                live figure is all there is, and `frac_source` says so."""
                if not d:
                    return d
                d = dict(d)
                d['frac_live_eager'], d['achieved_live_eager'] = d['frac'], d['achieved']
                cp = d.get('from_committed_profile')
                if cp and d['bound'] == 'hbm':
                    us = cp['avg_launch_us_in_graph_rocprof']
                    d['avg_launch_us'] = us
                    d['achieved'] = round(d['algorithmic_bytes_per_launch'] / (us * 1e-6) / 1e9, 1)
                    d['frac'] = cp['frac_in_graph_rocprof']
                    d['frac_source'] = 'in-graph average of %s (rocprofv3 --kernel-trace --stats of this command, committed)' % cp['file']
                else:
                    d['avg_launch_us'] = d['avg_launch_us_in_step']
                    d['frac_source'] = 'live: events around each launch of an eager two-stream step (no committed in-graph trace names this kernel)'
                return d
            fwd, bwd, dom = timed_configuration(fwd), timed_configuration(bwd), timed_configuration(dom)
            roofline = dict(dom)
            roofline.update({'forward': fwd, 'backward': bwd,
                             'operator_fwd_frac': fwd_frac, 'operator_bwd_frac': bwd_frac,
                             'durations': 'frac / achieved / avg_launch_us = the kernel inside the replayed two-stream graph (frac_source); '
                                          'frac_live_eager / avg_launch_us_in_step = events around each launch of an eager two-stream step of THIS '
                                          'run; _isolated = the same on one stream; operator_*_frac = all calls of a direction, live eager',
                             'library_kernels_from_committed_profile': {'file': in_graph_file, 'kernels': library}})
            for k, v in kernels.items():
                v['ms2'] = (kernels2.get(k) or v)['ms']
    fence()

    if rank == 0:
        tables_note = {'shared': 'graph tables batch-shared (one table, batch stride 0)',
                       'per_sample': 'graph tables per-sample (B copies, as the reference passes them)%s'
                                     % (', content check off' if args.no_dedupe else
                                        ', recognised as one graph by a content check')}[args.tables]
        total_msgs = MESSAGES_PER_CODEWORD * args.batch * world * args.steps
        out = {
            'metric': 'VF+FV messages/sec on 96.3.963 LDPC graph',
            'value': total_msgs / elapsed, 'unit': 'messages/s', 'n_gpus': world,
            'steps': args.steps, 'warmup': args.warmup,
            'ms_per_step': elapsed / args.steps * 1e3, 'higher_is_better': True,
            'scaling': 'weak', 'vs_baseline': None, 'dtype': args.dtype, 'data': 'synthetic',
            'config': {'workload': 'LDPC 96.3.963 LDPCModel (8 FGNN layers, 32 fused VF/FV operator '
                                   'calls, 6144 messages/codeword), %s step, batch %d codewords per GPU, %s'
                                   % ('training (fwd+bwd+grad all-reduce+Adam)' if train else
                                      'inference forward', args.batch, tables_note),
                       'tables': args.tables + ('' if args.tables == 'shared' else
                                                (' (content check off)' if args.no_dedupe else ' (deduplicated by content check)')),
                       # the 6144 count includes the degree-96 hyper-factor's 2 x 96 edges per layer; parity-check edges
                       # alone are 8 x (288 + 288) = 4608 per codeword (SURVEY §8d asks for both)
                       'messages_per_codeword': MESSAGES_PER_CODEWORD, 'parity_messages_per_codeword': 4608,
                       'parity_only_value': 4608 * args.batch * world * args.steps / elapsed,
                       'global_batch': args.batch * world, 'parallelism': 'dp%d' % world,
                       'inputs': ('random messages -> reference G encode -> AWGN+burst channel, on the GPU'
                                  if args.inputs == 'channel' else 'random bits + AWGN (torch ops)'),
                       'peak_hbm_GB': round(torch.cuda.max_memory_allocated(dev) / 1e9, 2),
                       'mode': args.mode, 'hip_graph': graphed is not None,
                       'parameters': 'updated every step' if train else 'frozen: BatchNorm folding and bf16 weight copies made once, outside the replayed graph',
                       'graph_scope': (None if graphed is None else ('forward + backward + RCCL all-reduce (C API) + Adam' if world > 1 else 'forward + backward + Adam') if whole_in_graph else
                                       'forward + backward (all-reduce and Adam eager behind each replay)' if train else 'forward'),
                       'distributed': dist_info},
            'roofline': roofline,
            'kernels': {k: {'launches': v['launches'], 'avg_us': round(v['ms'] / v['launches'] * 1e3, 2),
                            'avg_us_in_step': round(v.get('ms2', v['ms']) / v['launches'] * 1e3, 2),
                            'total_ms': round(v['ms'], 3),
                            'algorithmic_GBs': round(v['bytes'] / max(v['ms'], 1e-9) / 1e6, 1)}
                        for k, v in sorted(kernels.items(), key=lambda kv: -kv[1]['ms'])},
        }
        if world == 1 and not args.no_cpu_baseline:
            out.update(cpu_baseline_suite(args))
        else:
            out['cpu_baseline'] = None
        print(json.dumps(out))
    if world > 1:
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
