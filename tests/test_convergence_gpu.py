"""GPU: a convergence pin for the benched (bf16) training path.

Parity of single steps is pinned elsewhere (forward 2^-6, whole-model bf16 gradients 19 % / 40 % from the f32 oracle's autograd at
batch 4096: DESIGN §2).  What those numbers do not say is whether 300 such steps TRAIN the model like 300 f32 steps do.  Here:

  * `test_f32_training_steps_follow_the_oracle`: the f32 HIP path (the <= 1e-4 parity regime) and the CPU oracle
    (oracle/fgnn_oracle.py + torch.optim.Adam) take the same few optimizer steps from the same closed-form parameters on the same
    batches: the loss sequences agree — the f32 HIP run is a faithful stand-in for "the oracle trained on the GPU" (the oracle
    itself needs ~20 s per step at 4096 codewords);
  * `test_bf16_training_converges_like_f32`: 300 steps at 4096 codewords per step (25 distinct batches from the reference's
    encoder + channel, same order) of the bf16 path (autocast, bf16 activations, f32 parameters / Adam state — what bench.py times)
    beside the f32 path: the smoothed loss curves stay within a stated band, both go DOWN, and the bit error rate of the decoded
    message bits on a held-out batch (eval mode, running statistics) agrees within 10 % relative.

Reference loop: /root/reference/train_ldpc.py:207-231 (Adam, BCE-with-logits + 0.1 MSE, `pred > 0` decisions)."""
import contextlib
import io

import pytest
import torch

import fgnn_oracle as O
import helpers as H

pytestmark = pytest.mark.gpu

LR = 1e-3            # (the reference's 1e-4 moves the loss too little in 300 steps to tell a working path from a frozen one)


def _model(dev):
    import fgnn_amd
    with contextlib.redirect_stdout(io.StringIO()):
        m = fgnn_amd.LDPCModel(2, 6, 4, aggregator='max')
    m.load_state_dict(H.fill_state_dict(m.state_dict(), gain=2.0))
    return m.to(dev).train()


def _batches(dev, n, B, first_seed):
    from fgnn_amd.datapath import LdpcDataPath
    path = LdpcDataPath(dev)
    out = []
    for i in range(n):
        d = path.sample(B, seed=first_seed + i, dtype=torch.float32)
        out.append(d[:6] + (d[6][:, :48].float().contiguous(), d[7]))
    return out


def _as(batch, dtype):
    return tuple(t.to(dtype) if t.is_floating_point() and i < 6 else t for i, t in enumerate(batch))


def _train(dev, batches, steps, dtype, lr=LR):
    from fgnn_amd.dp import FlatAdam, FlatGradBucket
    from fgnn_amd.ldpc import decoding_loss
    model = _model(dev)
    bucket = FlatGradBucket(model.parameters(), flatten_params=True)
    opt = FlatAdam(bucket, lr=lr, weight_decay=0.0)
    amp = torch.autocast('cuda', dtype=torch.bfloat16, enabled=dtype == torch.bfloat16)
    data = [_as(b, dtype) for b in batches]
    losses = torch.zeros(steps, device=dev)
    for it in range(steps):
        b = data[it % len(data)]
        bucket.zero()
        with amp:
            logits, snr = model(*b[:6])
        loss = decoding_loss(logits, snr, b[6], b[7], 0.1)
        loss.backward()
        opt.step()
        losses[it] = loss.detach()
    return model, losses.cpu()


def _ber(model, batch, dtype):
    model.eval()
    b = _as(batch, dtype)
    with torch.no_grad(), torch.autocast('cuda', dtype=torch.bfloat16, enabled=dtype == torch.bfloat16):
        logits, _ = model(*b[:6])
    return float(((logits.float() > 0).float() != b[6]).float().mean())


def test_f32_training_steps_follow_the_oracle(dev):
    from fgnn_amd.ldpc import decoding_loss
    B, steps = 96, 4
    batches = _batches(dev, 2, B, 500)
    _, hip = _train(dev, batches, steps, torch.float32)
    # the oracle: the same parameters as leaf tensors, torch's own Adam
    with contextlib.redirect_stdout(io.StringIO()):
        import fgnn_amd
        ref = fgnn_amd.LDPCModel(2, 6, 4, aggregator='max')
    sd = H.fill_state_dict(ref.state_dict(), gain=2.0)
    leaves = []
    for k, v in sd.items():
        if v.is_floating_point() and 'running_' not in k and not k.startswith('h'):
            v.requires_grad_(True)
            leaves.append(v)
    opt = torch.optim.Adam(leaves, lr=LR)
    cpu = [tuple(t.cpu().contiguous() for t in b) for b in batches]
    ora = []
    for it in range(steps):
        b = cpu[it % len(cpu)]
        opt.zero_grad()
        logits, snr = O.ldpc_model(sd, *b[:6], training=True)
        loss = decoding_loss(logits, snr, b[6], b[7], 0.1)
        loss.backward()
        opt.step()
        ora.append(float(loss.detach()))
    ora = torch.tensor(ora)
    print('f32 HIP losses', hip.tolist(), 'oracle losses', ora.tolist())
    # step 0 is pure forward parity; later steps carry Adam's sign-like first updates of gradients that batch-statistics BatchNorm
    # conditions badly at 96 codewords: a looser band
    assert abs(float(hip[0]) - float(ora[0])) <= 1e-4 * max(1.0, abs(float(ora[0])))
    assert float((hip - ora).abs().max()) <= 2e-3 * float(ora.abs().max())         # (measured on MI355X: 3e-4 after four steps)


def test_bf16_training_converges_like_f32(dev):
    B, steps, nb = 4096, 300, 25
    batches = _batches(dev, nb, B, 1000)
    held_out = _batches(dev, 1, B, 999)[0]
    m32, l32 = _train(dev, batches, steps, torch.float32)
    m16, l16 = _train(dev, batches, steps, torch.bfloat16)
    win = 25
    s32, s16 = l32.view(-1, win).mean(1), l16.view(-1, win).mean(1)
    ber32, ber16 = _ber(m32, held_out, torch.float32), _ber(m16, held_out, torch.bfloat16)
    print('f32 loss per %d steps' % win, [round(float(v), 4) for v in s32])
    print('bf16 loss per %d steps' % win, [round(float(v), 4) for v in s16])
    print('held-out BER f32 %.5f bf16 %.5f' % (ber32, ber16))
    assert bool(torch.isfinite(l16).all()) and bool(torch.isfinite(l32).all())
    # both paths train: the last window sits well below the first
    assert float(s32[-1]) < 0.9 * float(s32[0]) and float(s16[-1]) < 0.9 * float(s16[0])
    # the bf16 curve follows the f32 curve (window means): band 1 % of the f32 value (measured on MI355X: 0.15 %, both curves
    # 0.334 -> 0.2285 over the 300 steps; held-out BER 0.1006 / 0.0996)
    assert float(((s16 - s32).abs() / s32.abs()).max()) <= 0.01, ((s16 - s32) / s32).tolist()
    # decoded bits on a held-out batch: 10 % relative (+ 2e-4 absolute: ~80 bit decisions of 196 608)
    assert abs(ber16 - ber32) <= 0.10 * ber32 + 2e-4, (ber16, ber32)
