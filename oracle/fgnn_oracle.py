"""CPU oracle for the FGNN VF/FV message-passing hot path.

TEST INFRASTRUCTURE ONLY.  Nothing under ``oracle/`` is part of the product:
only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline``
leg may import this file, and only as the checker / the timed CPU baseline.
The product path (``factor-graph-neural-network_amd/``) never imports it and
fails loudly when the HIP library is missing.

What this is: a from-scratch, *functional* PyTorch-CPU restatement of the
reference's algorithm, written as an interpreter over a reference-style
``state_dict`` (flat ``{name: tensor}``) instead of as ``nn.Module`` classes.
It keeps the reference's **op order** (materialised-index gather -> matmul ->
bmm with the edge-type weights -> aggregate -> +bias -> BatchNorm -> ReLU) so
that (i) it is bit-comparable with the reference on CPU and (ii) timing it on
the GPU box's host cores stands in for "the reference CPU path" (the
reference's Python cannot travel to the GPU box).

Parity pinning: ``oracle/make_golden.py`` imports the real reference from
``/root/reference`` (in the build container only), checks this restatement
against it (<= 1e-6) and writes the golden vectors under ``tests/golden/``;
``tests/test_oracle_golden.py`` re-checks the oracle against those vectors
without the reference present.

Reference citations (all relative to /root/reference):
  mp_conv            lib/model/mpnn/mp_nn.py:115-175 (+ to_edge_feature :92-113)
  aggregate          lib/model/mpnn/mp_nn.py:68-90
  residual_block     lib/model/mpnn/mp_nn_residual.py:25-56
  iid maps           lib/model/mpnn/base_model.py:43-90
  factor_nn          lib/model/mpnn/factor_mpnn_sp.py:25-178
  factor_mpnn        lib/model/mpnn/factor_mpnn.py:8-133
  ldpc_model         train_ldpc.py:19-99
  mp_sequential cfg1 train_syn_fixed_pw_hop.py:121-134, lib/model/mpnn/sequential.py:21-39
"""
import torch
import torch.nn.functional as F

NO_EXTENSION = 0
ORIG_WITH_NEIGHBOR = 1
ORIG_WITH_DIFF = 2

BN_EPS = 1e-5
BN_MOMENTUM = 0.1
LSE_GAMMA = 3.0


# --------------------------------------------------------------------------
# the operator
# --------------------------------------------------------------------------
def gather_rows(feat, nn_idx):
    """feat [B,N,C], nn_idx [B,M,k] int64 -> [B,M,k,C].

    Same dataflow as the reference's to_edge_feature (mp_nn.py:92-113): the
    index is materialised to the size of the output and fed to ``gather``.
    """
    B, M, k = nn_idx.shape
    C = feat.shape[2]
    assert feat.shape[0] == B
    wide = nn_idx.reshape(B, M * k, 1).expand(B, M * k, C).contiguous()
    return torch.gather(feat, 1, wide).reshape(B, M, k, C)


def aggregate(e, how):
    """e [B,nou,M,k] -> [B,nou,M,1]   (mp_nn.py:68-90)."""
    if how is None:
        return e
    if how == 'max':
        return e.max(dim=3, keepdim=True)[0]
    if how == 'softmax':
        return (1.0 / LSE_GAMMA) * torch.logsumexp(LSE_GAMMA * e, dim=3, keepdim=True)
    if how == 'mean':
        return e.mean(dim=3, keepdim=True)
    if callable(how):
        return how(e)
    raise ValueError('unknown aggregator %r' % (how,))


def batch_norm(x, sd, prefix, training):
    """BatchNorm2d/1d as torch does it (train: batch stats + running update)."""
    return F.batch_norm(x, sd[prefix + 'running_mean'], sd[prefix + 'running_var'],
                        sd[prefix + 'weight'], sd[prefix + 'bias'],
                        training, BN_MOMENTUM, BN_EPS)


def mp_conv(sd, prefix, x, nn_idx, etype, *, nou, net, extension, aggregator,
            training=False, relu=True):
    """The VF/FV message operator, reference op order (mp_nn.py:115-175).

    x [B,nin,N,1], nn_idx [B,M,k] int64, etype [B,net,M,k] -> [B,nou,M,1].
    Parameters read from ``sd``: prefix+'filters' [R, nou*net] (column index
    o*net+e), optional prefix+'bias', optional prefix+'bn.*'.
    """
    filters = sd[prefix + 'filters']
    B, nin, N = x.shape[0], x.shape[1], x.shape[2]
    M, k = nn_idx.shape[1], nn_idx.shape[2]
    w_edge = etype.permute(0, 2, 3, 1).contiguous().reshape(B * M * k, net, 1)
    rows = x.permute(0, 2, 3, 1).contiguous().reshape(B, N, nin)
    if extension == NO_EXTENSION:
        proj = rows.reshape(B * N, nin).matmul(filters).reshape(B, N, nou * net)
        per_edge = gather_rows(proj, nn_idx).reshape(B * M * k, nou, net)
    else:
        assert N == M, 'extension branches need one row per destination'
        nb = gather_rows(rows, nn_idx)                       # [B,M,k,nin]
        own = rows.reshape(B, N, 1, nin).expand(B, N, k, nin)
        if extension == ORIG_WITH_DIFF:
            nb = own - nb
        cat = torch.cat([own, nb], dim=3).reshape(B * M * k, 2 * nin)
        per_edge = cat.matmul(filters).reshape(B * M * k, nou, net)
    msg = per_edge.bmm(w_edge).reshape(B, M, k, nou)
    msg = msg.permute(0, 3, 1, 2).contiguous()               # [B,nou,M,k]
    y = aggregate(msg, aggregator)
    if (prefix + 'bias') in sd:
        y = y + sd[prefix + 'bias'].reshape(1, nou, 1, 1)
    if (prefix + 'bn.weight') in sd:
        y = batch_norm(y, sd, prefix + 'bn.', training)
    if relu:
        y = torch.relu(y)
    return y


# --------------------------------------------------------------------------
# blocks around the operator
# --------------------------------------------------------------------------
def conv1x1(sd, prefix, x):
    w = sd[prefix + 'weight']
    b = sd.get(prefix + 'bias')
    return F.conv2d(x, w, b)


def instance_norm_nodes(x):
    """InstanceNorm2d(affine=False) over the node axis of [B,C,N,1].

    With a single node (LDPC hyper-factor, factor_mpnn_sp.py:77,140) the
    reference era (torch 1.0) returned exactly 0; torch >= 1.9 raises unless
    F._verify_spatial_size is patched.  The value is defined here as 0.
    """
    if x.shape[2] * x.shape[3] == 1:
        return torch.zeros_like(x)
    return F.instance_norm(x, eps=BN_EPS)


def conv_bn_act(sd, prefix, x, training, slope):
    """Sequential(Conv2d 1x1, BatchNorm2d, (Leaky)ReLU) — indices 0,1 of a Sequential."""
    y = conv1x1(sd, prefix + '0.', x)
    y = batch_norm(y, sd, prefix + '1.', training)
    return F.leaky_relu(y, slope) if slope else torch.relu(y)


def residual_block(sd, prefix, x, nn_idx, etype, *, net, extension, aggregator,
                   with_residual, training=False):
    """mp_conv_residual (mp_nn_residual.py:39-56): conv1 -> mp_conv -> conv2 (+x)."""
    nmed = sd[prefix + 'conv1.0.weight'].shape[0]
    h = conv_bn_act(sd, prefix + 'conv1.', x, training, 0.01)
    h = mp_conv(sd, prefix + 'mp_conv.', h, nn_idx, etype, nou=nmed, net=net,
                extension=extension, aggregator=aggregator, training=training)
    h = conv_bn_act(sd, prefix + 'conv2.', h, training, 0.01)
    return h + x if with_residual else h


def iid_mapping(sd, prefix, x):          # Conv + LeakyReLU      base_model.py:43-59
    return F.leaky_relu(conv1x1(sd, prefix + 'main.0.', x), 0.01)


def iid_mapping_bn(sd, prefix, x, training):   # Conv + BN + ReLU  base_model.py:62-79
    return conv_bn_act(sd, prefix + 'main.', x, training, 0.0)


def iid_mapping_in(sd, prefix, x):       # Conv + InstanceNorm + ReLU  base_model.py:82-90
    return torch.relu(instance_norm_nodes(conv1x1(sd, prefix + 'main.0.', x)))


def _is_residual(sd, prefix):
    return (prefix + 'conv1.0.weight') in sd


def _mp_module(sd, prefix, x, nn_idx, etype, *, net, extension, aggregator,
               with_residual, training):
    """Dispatch on what lives at ``prefix``: residual block or bare operator."""
    if _is_residual(sd, prefix):
        return residual_block(sd, prefix, x, nn_idx, etype, net=net,
                              extension=extension, aggregator=aggregator,
                              with_residual=with_residual, training=training)
    nou = sd[prefix + 'filters'].shape[1] // net
    return mp_conv(sd, prefix, x, nn_idx, etype, nou=nou, net=net,
                   extension=extension, aggregator=aggregator, training=training)


# --------------------------------------------------------------------------
# FactorNN (LDPC body) — factor_mpnn_sp.py:121-178
# --------------------------------------------------------------------------
def factor_nn(sd, prefix, node_feature, hop_features, nn_idx_f2v, nn_idx_v2f,
              etype_f2v, etype_v2f, *, dims, netypes, skip_link, aggregator='max',
              training=False):
    """Returns (final_res, nhop_feature list)."""
    nn_f = iid_mapping(sd, prefix + 'node_mapping_module.', node_feature)
    hop_f = [iid_mapping_bn(sd, prefix + 'factor_mapping_modules_%d.' % j, f, training)
             for j, f in enumerate(hop_features)]
    history = []
    for L in range(len(dims) - 1):
        nin, nout = dims[L], dims[L + 1]
        nfe = iid_mapping_in(sd, prefix + 'v2v_%d.' % L, nn_f)
        ffe = [iid_mapping_in(sd, prefix + 'f2f_%d_%d.' % (L, j), f)
               for j, f in enumerate(hop_f)]
        for j in range(len(hop_f)):
            kw = dict(net=netypes[j], extension=NO_EXTENSION, aggregator=aggregator,
                      with_residual=False, training=training)
            nv = _mp_module(sd, prefix + 'f2v_%d_%d.' % (L, j), hop_f[j],
                            nn_idx_f2v[j].long(), etype_f2v[j], **kw)
            nfe = nfe + nv
            nf = _mp_module(sd, prefix + 'v2f_%d_%d.' % (L, j), nn_f,
                            nn_idx_v2f[j].long(), etype_v2f[j], **kw)
            ffe[j] = ffe[j] + nf
        if nin == nout:
            nn_f = nn_f + nfe
            hop_f = [a + b for a, b in zip(ffe, hop_f)]
        else:
            nn_f, hop_f = nfe, ffe
        if L in skip_link:
            on, of = history[skip_link[L]]
            nn_f = nn_f + on
            hop_f = [a + b for a, b in zip(of, hop_f)]
        history.append((nn_f, hop_f))
    p = prefix + 'final_classifier.'
    y = conv1x1(sd, p + '0.', nn_f)
    y = torch.relu(instance_norm_nodes(y))
    y = conv1x1(sd, p + '3.', y)
    return y, hop_f


LDPC_DIMS = [64, 64, 64, 128, 256, 256, 128, 64, 64]
LDPC_SKIP = {4: 3, 5: 2, 7: 0}


def ldpc_model(sd, node_feature, hop_feature, nn_idx_f2v, nn_idx_v2f,
               efeature_f2v, efeature_v2f, *, dims=None, nedge_type=4,
               aggregator='max', training=False, with_residual=True):
    """LDPCModel.forward (train_ldpc.py:67-99).  Returns (logits[B,48], snr[B,1])."""
    dims = LDPC_DIMS if dims is None else dims
    B, nvar = node_feature.shape[0], node_feature.shape[2]

    def edge_mlp(p, ef):
        return conv1x1(sd, p + '2.', torch.relu(conv1x1(sd, p + '0.', ef)))

    et_f2v = edge_mlp('emodel_f2v.', efeature_f2v)
    et_v2f = edge_mlp('emodel_v2f.', efeature_v2f)
    hyper_feature = node_feature[:, 0, :, :].reshape(B, nvar, 1, 1)
    res, hops = factor_nn(
        sd, 'main.', node_feature, [hop_feature, hyper_feature],
        [nn_idx_f2v, sd['hnn_idx_f2v'].repeat(B, 1, 1)],
        [nn_idx_v2f, sd['hnn_idx_v2f'].repeat(B, 1, 1)],
        [et_f2v, sd['hetype_f2v'].repeat(B, 1, 1, 1)],
        [et_v2f, sd['hetype_v2f'].repeat(B, 1, 1, 1)],
        dims=dims, netypes=[nedge_type, 1], skip_link=LDPC_SKIP,
        aggregator=aggregator, training=training)
    if with_residual:
        res = res + node_feature[:, :1, :, :]
    res = res.reshape(B, nvar)
    hh = hops[1].reshape(B, -1)
    p = 'nhop_regressor.'
    h = F.linear(hh, sd[p + '0.weight'], sd[p + '0.bias'])
    h = torch.relu(batch_norm(h, sd, p + '1.', training))
    h = torch.relu(F.linear(h, sd[p + '3.weight'], sd[p + '3.bias']))
    h = torch.relu(F.linear(h, sd[p + '5.weight'], sd[p + '5.bias']))
    return res[:, :nvar // 2].contiguous(), h


# --------------------------------------------------------------------------
# factor_mpnn (synthetic PGM body) — factor_mpnn.py:88-133
# --------------------------------------------------------------------------
def factor_mpnn(sd, prefix, node_features, factor_features, graph_structures, *,
                dims, netypes, training=False):
    """Returns (nfeatures, ffeatures list).  All mp blocks are ORIG_WITH_DIFF;
    residual blocks aggregate with 'max', bare operators with 'softmax'
    (the constructor defaults at factor_mpnn.py:57-61)."""
    nnode = node_features.shape[2]
    nf = iid_mapping(sd, prefix + 'mapping_modules_0.', node_features)
    ff = [iid_mapping(sd, prefix + 'mapping_modules_%d.' % (j + 1), f)
          for j, f in enumerate(factor_features)]
    ntypes = len(ff)
    for L in range(len(dims) - 1):
        cn, cf = [], []
        for j in range(ntypes):
            cat = torch.cat([nf, ff[j]], dim=2).contiguous()
            nn_idx, etype = graph_structures[j]
            p = prefix + 'mp_nn_%d_%d.' % (L, j)
            if _is_residual(sd, p):
                out = residual_block(sd, p, cat, nn_idx, etype, net=netypes[j],
                                     extension=ORIG_WITH_DIFF, aggregator='max',
                                     with_residual=True, training=training)
            elif (p + 'filters') in sd:
                out = mp_conv(sd, p, cat, nn_idx, etype,
                              nou=dims[L + 1], net=netypes[j],
                              extension=ORIG_WITH_DIFF, aggregator='softmax',
                              training=training)
            else:   # Sequential(Conv2d, InstanceNorm2d, ReLU): no message passing
                out = torch.relu(instance_norm_nodes(conv1x1(sd, p + '0.', cat)))
            cn.append(out[:, :, :nnode, :])
            cf.append(out[:, :, nnode:, :])
        cn = torch.cat(cn, dim=1)
        p = prefix + 'merge_module_%d.' % L
        if L < len(dims) - 2:
            nf = iid_mapping_bn(sd, p, cn, training)
        else:
            h = conv1x1(sd, p + '0.', cn)
            h = F.leaky_relu(batch_norm(h, sd, p + '1.', training), 0.01)
            h = F.leaky_relu(conv1x1(sd, p + '3.', h), 0.01)
            nf = conv1x1(sd, p + '5.', h)
        ff = cf
    return nf, ff


SYN_DIMS = [64, 64, 128, 128, 256, 256, 128, 128, 64, 64, 2]


# --------------------------------------------------------------------------
# config 1: mp_sequential of train_syn_fixed_pw_hop.py:121-134
# --------------------------------------------------------------------------
def fixed_pw_hop_net(sd, x, nn_idx, etype, *, net=16, training=False):
    """Children are numbered like nn.Sequential ('0.' ... '17.')."""
    h = mp_conv(sd, '0.', x, nn_idx, etype, nou=64, net=net,
                extension=ORIG_WITH_NEIGHBOR, aggregator='softmax', training=training)
    i = 1
    while True:
        h = residual_block(sd, '%d.' % i, h, nn_idx, etype, net=net,
                           extension=ORIG_WITH_DIFF, aggregator='max',
                           with_residual=True, training=training)
        h = conv1x1(sd, '%d.' % (i + 1), h)
        if ('%d.weight' % (i + 2)) not in sd:
            return h
        h = torch.relu(batch_norm(h, sd, '%d.' % (i + 2), training))
        i += 4


# --------------------------------------------------------------------------
# §8f rank 4: the LDPC data path in front of the model (numpy)
# --------------------------------------------------------------------------
def ldpc_encode(G, s):
    """Systematic encode of K-bit messages s [..., K] with the (N x K) GF(2) matrix G: [s | G s mod 2] — what
    `s2t(s, 48, 48, Gfile, smn=True)` returns per message (/root/reference/lib/data/MNC/MNC_py.cpp:22-83,
    `mod2mat_multiply`, radford/mod2mat.cpp:353).  Pinned against the reference's own compiled code by
    oracle/make_ldpc_datapath_golden.py (tests/golden/ldpc_datapath.npz)."""
    import numpy as np
    G = np.asarray(G, np.int64)
    s = np.asarray(s, np.int64)
    t = (s @ G.T) % 2
    return np.concatenate([s, t], axis=-1).astype(np.uint8)


def ldpc_channel(t, snr_db, sigma_b, rho, z1, u, z2):
    """`t2y` (/root/reference/lib/data/MNC/MNC_py.cpp:86-102) with the random draws made explicit:
    gcx = 10^(snr_db/20);  y = 2 gcx (t - 0.5) + z1;  where sigma_b >= 1e-20 and u < rho: y += gcx sigma_b z2.
    t [B,N] bits, snr_db / sigma_b [B], z1 / z2 ~ N(0,1), u ~ U[0,1) of t's shape.  float64 like the reference
    (`T` = double when called from Python).  The draws themselves, as the reference's generator makes them, are
    `ldpc_channel_stream` / `XtensorStream` below (pinned against the reference's compiled module)."""
    import numpy as np
    t = np.asarray(t, np.float64)
    gcx = np.power(10.0, np.asarray(snr_db, np.float64) / 20.0)[:, None]
    sb = np.asarray(sigma_b, np.float64)[:, None]
    y = 2.0 * gcx * (t - 0.5) + z1
    burst = (sb >= 1e-20) & (np.asarray(u) < rho)
    return y + np.where(burst, gcx * sb * np.asarray(z2), 0.0)


def philox4x32(counter, key, rounds=10):
    """Philox4x32-10 (Salmon, Moraes, Dror, Shaw: "Parallel random numbers: as easy as 1, 2, 3", SC'11) on arrays:
    counter [..., 4] uint32, key [2] uint32 -> [..., 4] uint32.  The generator csrc/ldpc_datapath.hip runs per codeword bit."""
    import numpy as np
    c = np.array(counter, dtype=np.uint64)
    k0, k1 = np.uint64(key[0]), np.uint64(key[1])
    M0, M1, mask = np.uint64(0xD2511F53), np.uint64(0xCD9E8D57), np.uint64(0xFFFFFFFF)
    for _ in range(rounds):
        p0, p1 = M0 * c[..., 0], M1 * c[..., 2]
        n = np.stack([(p1 >> np.uint64(32)) ^ c[..., 1] ^ k0, p1 & mask, (p0 >> np.uint64(32)) ^ c[..., 3] ^ k1, p0 & mask], axis=-1)
        c = n & mask
        k0 = (k0 + np.uint64(0x9E3779B9)) & mask
        k1 = (k1 + np.uint64(0xBB67AE85)) & mask
    return c.astype(np.uint32)


def philox_channel_draws(nbits, seed, offset):
    """(z1, u, z2) float32 [nbits] as `fgnn_ldpc_channel_features_rng` draws them for the codeword bits 0 .. nbits-1 of a batch:
    block A = philox(counter (i lo, i hi, offset lo, offset hi), key seed), block B = the same with the counter's top bit flipped;
    z1 = Box-Muller(A0, A1), u = (A2 >> 8) 2^-24, z2 = Box-Muller(B0, B1), Box-Muller on the top 24 bits of each word in float32:
    sqrt(-2 ln((a >> 8) + 1) 2^-24) cos(2 pi (b >> 8) 2^-24)."""
    import numpy as np
    i = np.arange(nbits, dtype=np.uint64)
    lo, hi = (i & np.uint64(0xFFFFFFFF)), (i >> np.uint64(32))
    off_lo, off_hi = np.uint64(offset & 0xFFFFFFFF), np.uint64((offset >> 32) & 0xFFFFFFFF)
    key = (seed & 0xFFFFFFFF, (seed >> 32) & 0xFFFFFFFF)
    ca = np.stack([lo, hi, np.full_like(lo, off_lo), np.full_like(lo, off_hi)], axis=-1)
    cb = ca.copy()
    cb[..., 3] ^= np.uint64(0x80000000)
    A, Bk = philox4x32(ca, key), philox4x32(cb, key)
    s = np.float32(5.9604644775390625e-08)

    def normal(a, b):
        u1 = ((a >> np.uint32(8)).astype(np.float32) + np.float32(1.0)) * s
        u2 = (b >> np.uint32(8)).astype(np.float32) * s
        return (np.sqrt(np.float32(-2.0) * np.log(u1)) * np.cos(np.float32(6.283185307179586) * u2)).astype(np.float32)

    return normal(A[:, 0], A[:, 1]), (A[:, 2] >> np.uint32(8)).astype(np.float32) * s, normal(Bk[:, 0], Bk[:, 1])


def ldpc_sum_product(nlist, nchk, bias, loops=100, tinydiv=1e-40, clip=0.9999999999):
    """MacKay's probability-domain sum-product decoder as the reference runs it for its classical baseline
    (`zb2x(z, 48, 48, A2, 1, 100)`, /root/reference/lib/data/ldpc.py:20-24 -> `bndecode`,
    lib/data/MNC/bnd/bnd.cpp:150-173): priors `bias[n]` = P(bit n = 1), target syndrome 0, defaults doclip = 1 /
    clip = 0.9999999999 / tinydiv = 1e-40 / dofudge = 0 (`bnd_defaults`).  One codeword, pure-Python float64 with the
    reference's operation order (horizontal_pass :217-291, vertical_pass :294-371, bnd_score_state :196-214), so the
    pseudo-posteriors are bit-identical to the compiled reference (pinned by oracle/make_ldpc_datapath_golden.py).
    ``nlist[n]`` = the checks of variable n in the alist file's order (-1 = padding).
    Returns (x [N] hard decisions, q1 [N], violated checks, iterations run)."""
    N = len(nlist)
    cols = [[int(m) for m in row if m >= 0] for row in nlist]
    rows = [[] for _ in range(nchk)]
    for n in range(N):                                    # bnd.cpp:239-251: a row's entries are met in increasing n
        for u, m in enumerate(cols[n]):
            rows[m].append((n, u))
    bias = [float(b) for b in bias]
    dqc = [[1.0 - 2.0 * bias[n]] * len(cols[n]) for n in range(N)]          # bnd_load_dqc :139-147
    pc0 = [[0.0] * len(c) for c in cols]
    pc1 = [[0.0] * len(c) for c in cols]
    q1 = [0.0] * N
    x = [0] * N
    viol, it = nchk, 0
    for it in range(1, loops + 1):
        for m in range(nchk):                             # horizontal pass
            ent = rows[m]
            L = len(ent)
            dpf = [1.0] * (L + 2)
            dpr = [1.0] * (L + 2)
            for l in range(1, L + 1):
                n, u = ent[l - 1]
                dpf[l] = dqc[n][u] * dpf[l - 1]
            for l in range(L, 0, -1):
                n, u = ent[l - 1]
                dpr[l] = dqc[n][u] * dpr[l + 1]
                dpc = dpf[l - 1] * dpr[l + 1] * 0.5
                pc0[n][u] = 0.5 + dpc
                pc1[n][u] = 0.5 - dpc
        for n in range(N):                                # vertical pass
            U = len(cols[n])
            qt0 = [1.0 - bias[n]] + [0.0] * U
            qt1 = [bias[n]] + [0.0] * U
            for u in range(1, U + 1):
                qt0[u] = qt0[u - 1] * pc0[n][u - 1]
                qt1[u] = qt1[u - 1] * pc1[n][u - 1]
            s = qt0[U] + qt1[U]
            if s > tinydiv:
                q1[n] = qt1[U] / s
            qb0, qb1 = 1.0, 1.0
            for u in range(U, 0, -1):
                nb0, nb1 = qb0 * pc0[n][u - 1], qb1 * pc1[n][u - 1]
                qc0, qc1 = qt0[u - 1] * qb0, qt1[u - 1] * qb1
                qb0, qb1 = nb0, nb1
                s, d = qc0 + qc1, qc0 - qc1
                if s > tinydiv:
                    v = d / s
                    v = clip if v > clip else (-clip if v < -clip else v)
                else:
                    v = 0.0
                dqc[n][u - 1] = v
        x = [1 if q >= 0.5 else 0 for q in q1]            # score
        viol = sum(1 for m in range(nchk) if sum(x[n] for n, _ in rows[m]) % 2)
        if viol == 0:
            break
    import numpy as np
    return np.array(x, np.uint8), np.array(q1, np.float64), viol, it


def ldpc_bit_prior(y, snr_db):
    """`y2b` (/root/reference/lib/data/MNC/MNC_py.cpp:104-108): P(bit = 1 | y) = 1 / (1 + exp(-2 gcx y))."""
    import numpy as np
    gcx = np.power(10.0, np.asarray(snr_db, np.float64) / 20.0)
    return 1.0 / (1.0 + np.exp(-2.0 * gcx[..., None] * np.asarray(y, np.float64)))


class XtensorStream:
    """The random stream `t2y` draws from, restated: xtensor's default engine is ONE process-wide `std::mt19937`
    (3rdparty/xtensor/include/xtensor/xrandom.hpp:300-313; `init_seed(s)` = `engine.seed(s)`, MNC_py.cpp:185-187), and each
    `xt::random::rand` / `randn` call builds a FRESH `std::uniform_real_distribution<double>` / `std::normal_distribution<double>`
    over it (xrandom.hpp:329-372).  With libstdc++ (the reference's toolchain):
      * a double in [0,1) = `generate_canonical<double,53>` = two 32-bit engine outputs, (lo + hi 2^32) 2^-64;
      * a normal = Marsaglia's polar method on x, y = 2 U - 1: the call returns y m and keeps x m for the NEXT call on the
        same distribution object (m = sqrt(-2 ln r2 / r2)); the kept value dies with the object, so `randn({1})` costs one
        accepted pair per call and `randn({n})` n/2 pairs.
    Pinned bit-for-bit against the reference's compiled MNC module by oracle/make_t2y_golden.py
    (tests/golden/ldpc_t2y_stream.npz)."""

    def __init__(self, seed):
        import numpy as np
        self._bg = np.random.MT19937()
        self._bg._legacy_seeding(int(seed) & 0xFFFFFFFF)       # init_genrand: what std::mt19937::seed(value) runs
        self._buf, self._pos = [], 0

    def _u32(self):
        if self._pos == len(self._buf):
            self._buf, self._pos = self._bg.random_raw(4096).tolist(), 0
        self._pos += 1
        return self._buf[self._pos - 1]

    def canonical(self):
        lo, hi = self._u32(), self._u32()
        r = (float(lo) + float(hi) * 4294967296.0) / 18446744073709551616.0
        return r if r < 1.0 else 1.0 - 2.0 ** -53              # libstdc++ clamps the round-to-1.0 case to nextafter(1, 0)

    def rand1(self):
        """`xt::random::rand<double>({1})[0]`."""
        return self.canonical()

    def randn(self, n, std_dev=1.0):
        """`xt::random::randn<double>({n}, 0, std_dev)` evaluated in storage order."""
        import math
        out, saved = [], None
        for _ in range(n):
            if saved is not None:
                v, saved = saved, None
            else:
                while True:
                    x = 2.0 * self.canonical() - 1.0
                    y = 2.0 * self.canonical() - 1.0
                    r2 = x * x + y * y
                    if not (r2 > 1.0 or r2 == 0.0):
                        break
                m = math.sqrt(-2.0 * math.log(r2) / r2)
                saved, v = x * m, y * m
            out.append(v * std_dev + 0.0)
        return out


def ldpc_channel_stream(t, snr_db, sigma_b, rho, stream, return_draws=False):
    """One `t2y(t, snr_db, sigma_b, rho)` call (/root/reference/lib/data/MNC/MNC_py.cpp:86-102) INCLUDING its draws from
    `stream` (an XtensorStream, advanced in place): N normals for the AWGN term first, then — only if sigma_b >= 1e-20 — per
    bit one uniform and, where it is below rho, one N(0, gcx sigma_b) from a fresh distribution object.  t [N] bits -> y [N]
    float64, bit-for-bit what the reference returns after `init_seed`.  With return_draws also (z1, u, z2) as `ldpc_channel`
    takes them: the unit normals and uniforms this call consumed (u = 1 and z2 = 0 where the reference drew none)."""
    import math
    import numpy as np
    t = np.asarray(t, np.float64)
    n = t.shape[0]
    gcx = math.pow(10.0, float(snr_db) / 20.0)
    z1 = np.asarray(stream.randn(n), np.float64)
    y = 2 * gcx * (t - 0.5) + z1
    sigma = gcx * float(sigma_b)
    u, z2 = np.ones(n), np.zeros(n)
    if sigma_b >= 1e-20:
        for i in range(n):
            u[i] = stream.rand1()
            if u[i] < rho:
                z2[i] = stream.randn(1)[0]
                y[i] += z2[i] * sigma + 0.0                    # normal_distribution(0, sigma): unit draw * sigma + mean
    return (y, z1, u, z2) if return_draws else y
