"""MINE / CLUB heads on the GPU (estts.mine, every kernel through the C ABI) against the float64 autograd oracle
(oracle/mine_torch.py) on the same variables, embeddings and random draws; then train_models_step over a small model.

Tolerances: estimates 2e-5 relative, gradients 1e-3 of each tensor's largest entry (float32 accumulation, split-bf16 products)."""
import numpy as np
import pytest
import torch

import estts  # noqa: F401
from oracle import mine_torch as OM

pytestmark = pytest.mark.gpu


def embeddings(B, N, de, sd, ds, seed):
    rng = np.random.default_rng(seed)
    return (rng.normal(0, 1, size=(B, N, de)).astype(np.float32), rng.normal(0, 1, size=(B, 1, sd)).astype(np.float32),
            rng.normal(0, 1, size=(B, 1, ds)).astype(np.float32))


def dev(a):
    return torch.from_numpy(a).cuda()


def compare_grads(mine, ref, tol=1e-3):
    worst = 0.0
    assert set(mine) == set(ref)
    for k, r in ref.items():
        a = mine[k].astype(np.float64)
        scale = np.abs(r).max()
        if scale < 1e-9:        # the estimate does not change when every score moves by a constant: the last bias
            assert np.abs(a).max() < 1e-6, k
            continue
        worst = max(worst, float(np.abs(a - r).max() / scale))
        assert np.abs(a - r).max() < tol * scale, (k, np.abs(a - r).max() / scale)
    return worst


@pytest.mark.parametrize('div_type,pair_type,smoothing', [('KL', 'style_text', 1.0), ('KL', 'style_text_speaker', 0.6),
                                                          ('reyni', 'style_speaker', 1.0), ('reyni', 'text_speaker', 0.7)])
def test_mine_estimate_and_gradient(div_type, pair_type, smoothing):
    from estts.mine import MINE
    B, N = 12, 9
    text, style, spk = embeddings(B, N, 64, 32, 16, seed=1)
    head = MINE(dense_hidden_units=[128, 64], beta_values=[0, 0.5, 1], divergence_type=div_type, pair_type=pair_type, seed=4)
    prev = np.array([[0.3, 0.8], [0.5, 0.4], [0.9, 0.2]], np.float32)
    holder = {'smoothing_factor': smoothing, 'exp_terms': prev}
    mi0, terms0 = head.call(dev(text), dev(style), dev(spk), holder)           # builds the variables; no update
    w = head.weights()
    draws = head.last_draws
    mi, terms = head.call(dev(text), dev(style), dev(spk), holder, draws=draws, train=True)
    assert abs(float(mi) - float(mi0)) < 1e-6
    ref = OM.mine_call(w, 2, pair_type, div_type, [0, 0.5, 1], smoothing, prev, text, style, spk, draws)
    assert abs(float(mi) - ref['mi']) < 2e-5 * max(1.0, abs(ref['mi'])), (float(mi), ref['mi'])
    assert np.abs(terms.cpu().numpy() - np.asarray(ref['exp_terms'])).max() < 2e-5
    worst = compare_grads(head.gradients(), ref['grads'])
    print(f'{div_type} {pair_type}: mi {float(mi):.5f}, worst gradient error {worst:.1e}')
    assert head.step == 1


def test_mine_ascent_raises_the_estimate():
    from estts.mine import MINE
    B, N = 16, 5
    text, style, spk = embeddings(B, N, 64, 32, 16, seed=2)
    style[:, 0, :8] = text[:, 2, :8]                   # make style and the text of position 2 dependent
    head = MINE(dense_hidden_units=[64, 32], divergence_type='KL', pair_type='style_text', seed=1)
    head._compile({'learning_rate': 1e-3, 'beta_1': 0.9, 'beta_2': 0.98, 'epsilon': 1e-9})
    holder = {'smoothing_factor': 1.0, 'exp_terms': np.zeros((3, 2), np.float32)}
    draws = {'char': 2, 'text_perm': np.roll(np.arange(B), 1), 'speaker_perm': np.arange(B)}
    first = float(head.call(dev(text), dev(style), dev(spk), holder, draws=draws)[0])
    for _ in range(40):
        head.call(dev(text), dev(style), dev(spk), holder, draws=draws, train=True)
    last = float(head.call(dev(text), dev(style), dev(spk), holder, draws=draws)[0])
    assert np.isfinite(last) and last > first + 0.05, (first, last)
    back = MINE(dense_hidden_units=[64, 32], divergence_type='KL', pair_type='style_text', seed=9)
    back.load_weights(head.weights())
    assert abs(float(back.call(dev(text), dev(style), dev(spk), holder, draws=draws)[0]) - last) < 1e-5


@pytest.mark.parametrize('pair_type', ['style_text', 'text_speaker'])
def test_club_terms_and_gradient(pair_type):
    from estts.mine import CLUB
    B, N = 10, 7
    text, style, spk = embeddings(B, N, 256, 64, 256, seed=3)
    head = CLUB(dense_hidden_units=[128, 64], pair_type=pair_type, seed=2)
    head.call(dev(text), dev(style), dev(spk))
    w, draws = head.weights(), head.last_draws
    lld, bound = head.call(dev(text), dev(style), dev(spk), draws=draws, train=True)
    ref = OM.club_call(w, 2, pair_type, text, style, spk, draws)
    assert abs(float(lld) - ref['lld']) < 2e-5 * max(1.0, abs(ref['lld'])), (float(lld), ref['lld'])
    # (the bound is the difference of two sums of the size of lld)
    assert abs(float(bound) - ref['bound']) < 2e-5 * max(1.0, abs(ref['lld']))
    compare_grads(head.gradients(), ref['grads'])


@pytest.mark.parametrize('sep_call', [False, True])
def test_train_models_step(sep_call):
    """traning_steps.py:16-95 end to end on a small speaker_style_text model with two MINE heads."""
    import sys, os
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    from test_gpu_autoregressive_training import batch, small_cfg
    from estts import weights
    from estts.autoregressive_training import AutoregressiveTransformerTrainer
    from estts.mine import MINE
    from estts.training_steps import train_models_step
    cfg = small_cfg('speaker_style_text', 0.1)
    w = weights.init_autoregressive_transformer(cfg, seed=5)
    model = AutoregressiveTransformerTrainer(w, cfg, seed=1)
    model._compile(stop_scaling=8.0, optimizer={'learning_rate': 1e-3})
    model.set_constants(reduction_factor=2, decoder_prenet_dropout=0.1, drop_n_heads=0)
    heads = [MINE(dense_hidden_units=[64, 32], divergence_type='KL', pair_type=p, seed=i) for i, p in enumerate(('style_text', 'style_speaker'))]
    for h in heads:
        h._compile({'learning_rate': 1e-3, 'beta_1': 0.9, 'beta_2': 0.98, 'epsilon': 1e-9})
    holder = {'use_mine': True, 'mine_sep_call': sep_call, 'mi_loss': 0.0, 'exp_terms': np.zeros((3, 2), np.float32),
              'weight_factor': 0.1, 'smoothing_factor': 1.0}
    tokens, mel, stop, spk = batch(cfg, 3, 13, 20, seed=5)
    tokens2, mel2, stop2, spk2 = batch(cfg, 4, 11, 16, seed=6)
    losses = []
    for _ in range(3):
        out = train_models_step(mel, tokens, stop, spk, mel2, tokens2, stop2, spk2, model, heads, holder, True, True, True, False)
        losses.append(float(out['loss']))
        assert set(out['mi_loss']) == {'MINE:style_text', 'MINE:style_speaker'}
        assert np.isfinite(float(holder['mi_loss'])) and tuple(holder['exp_terms'].shape) == (1, 2)
        for k in ('final_output', 'stop_prob', 'mel_linear', 'decoder_attention', 'text_encoder_attention', 'gst_output', 'text_enc_output',
                  'tts_loss', 'style_loss', 'losses', 'reduced_target'):
            assert k in out, k
    assert model.step == 3 and all(h.step == 3 for h in heads)
    assert all(np.isfinite(losses))
    # the total differs from the TTS loss by weight_factor * max(0, previous mi)
    assert abs(float(out['loss']) - float(out['tts_loss'])) < 0.1 * abs(float(holder['mi_loss'])) + 1.0
    styled = train_models_step(mel, tokens, stop, spk, mel2, tokens2, stop2, spk2, model, heads, holder, True, True, True, True)
    assert float(styled['style_loss']) > 0 and float(styled['tts_loss']) > float(styled['losses']['output'])
    frozen = train_models_step(mel, tokens, stop, spk, mel2, tokens2, stop2, spk2, model, heads, holder, False, True, True, False)
    assert np.isfinite(float(frozen['loss']))        # the text encoder in inference mode inside the training step
