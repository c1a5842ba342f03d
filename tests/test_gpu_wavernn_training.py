"""WaveRNN training step on the GPU (wavernn_training.WaveRNNTrainer, every kernel through the C ABI) against the float64
autograd oracle oracle/wavernn_train_torch.py, which tests/test_oracle_wavernn.py pins to a training step of the reference
module itself, and directly against those reference goldens."""
import os

import numpy as np
import pytest
import torch

import estts  # noqa: F401
from estts import weights
from oracle import wavernn_train_torch as OT
from wavernn_common import small_hp

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(__file__), 'golden')


def rel(a, b):
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    return np.abs(a - b).max() / max(np.abs(b).max(), 1e-12)


def check_grads(mine, ref, tol_all=0.35, tol_tight=3e-3):
    """ReLU gates (fc1, fc2, MelResNet) are kinks: a gate that sits within the device's ~1e-5 arithmetic noise of zero can
    take the other branch and move everything upstream by a percent or so (see tests/test_gpu_training.py); the tensors
    not reached by such a flip must agree to 3e-3 of their largest entry, the whole gradient to a cosine of 0.999."""
    errs, dot, n1, n2 = [], 0.0, 0.0, 0.0
    for k, r in ref.items():
        a = mine[k].numpy().astype(np.float64).reshape(-1)
        r = np.asarray(r, np.float64).reshape(-1)
        dot, n1, n2 = dot + float(a @ r), n1 + float(a @ a), n2 + float(r @ r)
        if np.abs(r).max() < 1e-9:
            continue
        errs.append((rel(a, r), k))
    errs.sort(reverse=True)
    cosine = dot / np.sqrt(n1 * n2)
    tight = sum(e < tol_tight for e, _ in errs) / len(errs)
    print(f'gradient check: cosine {cosine:.7f}, worst {errs[:3]}, within {tol_tight}: {tight:.0%}')
    assert errs[0][0] < tol_all, errs[:5]
    assert tight >= 0.5, errs[:8]
    assert cosine > 0.999


@pytest.mark.parametrize('mode', ['RAW', 'MOL'])
@pytest.mark.parametrize('use_graph,chunk,fused', [(False, None, True), (True, None, True), (False, 7, True), (True, 7, True),
                                                   (False, 7, False), (True, None, False), (True, 7, 'two chains')])
def test_train_step_against_reference_golden_and_oracle(mode, use_graph, chunk, fused):
    """chunk: steps between hand-overs of the two overlapped GRU recurrences; 7 gives the window several (uneven) chunks, so
    the second stream's batched products and the carried hidden-state gradients cross chunk borders.  fused: a recurrent step
    as one launch (product + gates; both layers in one chain of launches, or on two streams) or as two."""
    from estts.wavernn_training import WaveRNNTrainer
    g = np.load(os.path.join(GOLD, f'wavernn_train_{mode.lower()}_small.npz'))
    hp = small_hp(mode)
    w = weights.init_wavernn(hp, seed=int(g['seed']))
    model = WaveRNNTrainer.from_hparams(hp, use_graph=use_graph)
    model.CHUNK = chunk
    model.fused_gru = bool(fused)
    model.one_chain = fused is True                      # 'two chains': one-launch steps, the layers on two streams
    n_chunks = len(model._chunks(g['x'].shape[1]))       # a quarter window by default, a sixteenth for the one-chain recurrences
    assert n_chunks > 4 if (chunk is not None or fused) else n_chunks == 4
    model.load_state_dict({k: torch.from_numpy(v) for k, v in w.items()})
    sd0 = model.state_dict()
    for k, v in w.items():
        assert np.array_equal(sd0[k].numpy(), np.asarray(v)), k            # layout round trip
    ref = OT.train_step(w, hp, g['x'], g['y'], g['mels'])
    steps = 2 if use_graph else 1                                          # the second step replays the captured loops
    for it in range(steps):
        if it == 1:                                                        # same weights again: reload, keep the graphs
            model.store.flat.copy_(flat0)
            model.store.state.copy_(state0)
            model.store.m.zero_(); model.store.v.zero_(); model.iterations = 0
            for lin in model._linears:
                lin.repack()
            model._refresh_fir()
        else:
            flat0, state0 = model.store.flat.clone(), model.store.state.clone()
        loss = float(model.train_step(g['x'], g['y'], g['mels'], lr=float(g['lr'])))
        assert abs(loss - float(g['loss'])) < 3e-5 * max(1.0, abs(float(g['loss']))), (it, loss)
        assert abs(loss - ref['loss']) < 3e-5 * max(1.0, abs(ref['loss']))
        logits = model._ws['logits'][:, :model.n_classes].cpu().numpy().reshape(g['logits'].shape)
        assert np.abs(logits - ref['logits']).max() < 2e-4
        grads = model.gradients()
        assert set(ref['grads']) <= set(grads)
        check_grads(grads, ref['grads'])
        after = model.state_dict()
        for k, v in ref['new_stats'].items():
            assert np.abs(after[k].numpy() - v).max() < 2e-5, k
        for k in ref['grads']:                                             # first Adam step: lr * sign(g) where g is not tiny
            gg = g['grad/' + k]
            big = np.abs(gg) > 0.25 * np.abs(gg).max()
            assert np.abs(after[k].numpy() - g['after/' + k])[big].max() < 3e-6, k
    assert model.get_step() == steps


def test_training_reduces_loss_and_checkpoint_loads_into_generation_model():
    """(a) several steps on one batch reduce the loss, (b) the trained state_dict loads into the generation class."""
    from estts.wavernn import WaveRNN
    from estts.wavernn_training import WaveRNNTrainer
    hp = small_hp('RAW')
    w = weights.init_wavernn(hp, seed=3)
    model = WaveRNNTrainer.from_hparams(hp)
    model.load_state_dict({k: torch.from_numpy(v) for k, v in w.items()})
    rng = np.random.default_rng(0)
    B, hops = 4, 5
    L = hops * hp['hop_length']
    labels = rng.integers(0, 2 ** hp['bits'], size=(B, L + 1))
    x = (2 * labels[:, :L] / (2 ** hp['bits'] - 1.) - 1.).astype(np.float32)
    mels = rng.uniform(0, 1, size=(B, hp['num_mels'], hops + 2 * hp['voc_pad'])).astype(np.float32)
    losses = [float(model.train_step(x, labels[:, 1:], mels, lr=2e-3)) for _ in range(30)]
    assert losses[-1] < 0.85 * losses[0], losses
    gen = WaveRNN.from_hparams(hp)
    gen.load_state_dict(model.state_dict())
    long_mel = rng.uniform(0, 1, size=(1, hp['num_mels'], 30)).astype(np.float32)
    out = gen.generate(torch.from_numpy(long_mel), True, hp['voc_target'], hp['voc_overlap'], True, seed=1)
    assert np.isfinite(out).all() and len(out) > 0


def test_recurrent_layers_wider_than_one_launch_take_the_two_launch_steps():
    """The one-launch recurrent step holds K = 3 * rnn_dims <= 1536 in flight (rnn_dims <= 512); a wider layer must choose the
    two-launch steps by itself and still match the float64 oracle."""
    from estts.wavernn_training import WaveRNNTrainer
    hp = dict(small_hp('RAW'), voc_rnn_dims=528, voc_fc_dims=528)
    w = weights.init_wavernn(hp, seed=11)
    model = WaveRNNTrainer.from_hparams(hp, use_graph=False)
    assert model.fused_gru is False and model.one_chain is False
    assert WaveRNNTrainer.from_hparams(dict(hp, voc_rnn_dims=512, voc_fc_dims=512)).fused_gru is True
    model.load_state_dict({k: torch.from_numpy(np.asarray(v)) for k, v in w.items()})
    rng = np.random.default_rng(2)
    B, hops = 2, 3
    L = hops * hp['hop_length']
    labels = rng.integers(0, 2 ** hp['bits'], size=(B, L + 1))
    x = (2 * labels[:, :L] / (2 ** hp['bits'] - 1.) - 1.).astype(np.float32)
    mels = rng.uniform(0, 1, size=(B, hp['num_mels'], hops + 2 * hp['voc_pad'])).astype(np.float32)
    ref = OT.train_step(w, hp, x, labels[:, 1:], mels)
    loss = float(model.train_step(x, labels[:, 1:], mels, lr=1e-4))
    assert abs(loss - ref['loss']) < 3e-5 * max(1.0, abs(ref['loss']))
    check_grads(model.gradients(), ref['grads'])
