"""WaveRNN on the GPU (HIP kernels through the C ABI) against golden vectors written by the reference module
itself (tests/golden/wavernn_*.npz) and against the NumPy oracle at full size.

Tolerances: float32 activations with split-bf16 (bf16x3) matrix products:
  upsampled conditioning 1e-4, teacher-forced logits 2e-3 (they grow through two recurrences over 48 steps),
  RAW trajectories: identical classes for the fixture seed (the draw is an inverse-CDF threshold test),
  MOL trajectories: 2e-3 on sample values.
"""
import os

import numpy as np
import pytest
import torch

from wavernn_common import small_hp
from estts import configs, weights
from estts.wavernn import WaveRNN
from oracle import wavernn_numpy as W

pytestmark = pytest.mark.gpu


def _model(hp, seed, **kw):
    m = WaveRNN.from_hparams(hp, **kw)
    w = weights.init_wavernn(hp, seed=seed)
    m.load_state_dict(w)
    return m, w


def _golden(golden_dir, mode):
    g = np.load(os.path.join(golden_dir, f'wavernn_{mode.lower()}_small.npz'))
    hp = small_hp(mode)
    m, w = _model(hp, int(g['seed']))
    return g, hp, m, w


@pytest.mark.parametrize('mode', ['RAW', 'MOL'])
def test_upsample_and_forward_vs_reference_goldens(golden_dir, mode):
    g, hp, m, w = _golden(golden_dir, mode)
    mp = np.pad(g['mel'], ((0, 0), (0, 0), (hp['voc_pad'], hp['voc_pad'])))
    up, aux = m.upsample(mp)
    assert np.abs(up.cpu().numpy() - g['upsampled_mel']).max() < 1e-4
    assert np.abs(aux.cpu().numpy() - g['aux']).max() < 1e-4
    logits = m.forward(g['fwd_x'], g['fwd_mels'])
    assert tuple(logits.shape) == g['fwd_logits'].shape
    assert np.abs(logits.cpu().numpy() - g['fwd_logits']).max() < 2e-3


@pytest.mark.parametrize('use_graph', [False, True])
def test_generate_raw_vs_reference_golden(golden_dir, use_graph):
    g, hp, m, w = _golden(golden_dir, 'RAW')
    m.steps_per_graph = 20
    wav, raw = m.generate(torch.from_numpy(g['mel']), True, hp['voc_target'], hp['voc_overlap'], hp['mu_law'],
                          uniforms=g['u'], use_graph=use_graph, return_trace=True)
    _, _, raw_ref = W.generate(g['mel'], w, hp, g['u'], return_trace=True)
    assert raw.shape == raw_ref.shape
    n = 2 ** hp['bits']
    k, k_ref = np.rint((raw + 1) * (n - 1) / 2).astype(int), np.rint((raw_ref + 1) * (n - 1) / 2).astype(int)
    assert np.array_equal(k, k_ref)                          # same classes at every step of every fold
    assert np.abs(wav[0] - g['wav']).max() < 1e-6            # hence the reference's waveform


def test_generate_mol_vs_reference_golden(golden_dir):
    g, hp, m, w = _golden(golden_dir, 'MOL')
    uni = {'mix': g['u_mix'], 'logistic': g['u_logistic']}
    _, _, raw_ref = W.generate(g['mel'], w, hp, uni, return_trace=True)
    # replay the reference trajectory (forced feedback) so that one rounding difference cannot snowball
    wav, raw = m.generate(torch.from_numpy(g['mel']), True, hp['voc_target'], hp['voc_overlap'], hp['mu_law'],
                          uniforms=uni, forced=raw_ref.T, return_trace=True)
    assert np.abs(raw - raw_ref).max() < 2e-3
    assert np.abs(wav[0] - g['wav']).max() < 2e-3


def test_full_size_raw_step_trajectory_vs_oracle():
    """Shipped dimensions (rnn 512, 9-bit RAW, hop 200): 260 steps of two folds against the float64 oracle."""
    hp = configs.wavernn_hparams(voc_mode='RAW')
    m, w = _model(hp, 77)
    rng = np.random.default_rng(3)
    T = 22
    mel = rng.uniform(0, 1, size=(1, hp['num_mels'], T)).astype(np.float32)
    target, overlap = 200, 30
    n_seq = len(__import__('estts.wavernn', fromlist=['fold_starts']).fold_starts(T * hp['hop_length'], target, overlap))
    L = target + 2 * overlap
    u = rng.uniform(0, 1, size=(L, n_seq))
    wav_ref, logits_ref, raw_ref = W.generate(mel, w, hp, u, target=target, overlap=overlap, return_trace=True)
    wav, raw = m.generate(torch.from_numpy(mel), True, target, overlap, True, uniforms=u, forced=raw_ref.T,
                          return_trace=True)
    # classes are integers k with value 2k/511 - 1: count disagreements (a logit rounding can move a draw that sits
    # within ~1e-6 of a CDF step)
    k = np.rint((raw + 1) * 511 / 2).astype(int)
    k_ref = np.rint((raw_ref + 1) * 511 / 2).astype(int)
    assert np.mean(k != k_ref) < 0.01
    assert np.abs(k - k_ref).max() <= 1


def test_batch_of_utterances_matches_single_runs():
    """Several utterances of different lengths in one call: folds are independent sequences, so each waveform must
    equal the one produced alone (same seed-independent injected uniforms)."""
    hp = small_hp('RAW')
    m, w = _model(hp, 9)
    rng = np.random.default_rng(4)
    mels = [rng.uniform(0, 1, size=(hp['num_mels'], T)).astype(np.float32) for T in (23, 31, 26)]
    from estts.wavernn import fold_starts
    folds = [len(fold_starts(x.shape[1] * hp['hop_length'], hp['voc_target'], hp['voc_overlap'])) for x in mels]
    L = hp['voc_target'] + 2 * hp['voc_overlap']
    u_all = rng.uniform(0, 1, size=(L, sum(folds)))
    outs = m.generate_batch(mels, True, hp['voc_target'], hp['voc_overlap'], True, uniforms=u_all)
    o = 0
    for x, f, got in zip(mels, folds, outs):
        alone = m.generate_batch([x], True, hp['voc_target'], hp['voc_overlap'], True, uniforms=u_all[:, o:o + f])[0]
        o += f
        assert got.shape == ((x.shape[1] - 1) * hp['hop_length'],)
        assert np.array_equal(got, alone)


def test_builtin_generator_is_deterministic_and_uniform():
    hp = small_hp('RAW')
    m, w = _model(hp, 9)
    mel = np.random.default_rng(5).uniform(0, 1, size=(1, hp['num_mels'], 40)).astype(np.float32)
    a = m.generate(torch.from_numpy(mel), True, hp['voc_target'], hp['voc_overlap'], True, seed=123)
    b = m.generate(torch.from_numpy(mel), True, hp['voc_target'], hp['voc_overlap'], True, seed=123)
    c = m.generate(torch.from_numpy(mel), True, hp['voc_target'], hp['voc_overlap'], True, seed=124)
    assert np.array_equal(a, b) and not np.array_equal(a, c)
    assert a.shape == (39 * hp['hop_length'],) and np.isfinite(a).all() and np.abs(a).max() <= 1.0


def test_checkpoint_round_trip(tmp_path):
    hp = small_hp('MOL')
    m, w = _model(hp, 11)
    path = str(tmp_path / 'latest_weights.pyt')
    m.step = 4321
    m.save(path)
    m2 = WaveRNN.from_hparams(hp)
    m2.restore(path)
    assert m2.get_step() == 4321
    sd = torch.load(path, weights_only=True)
    assert set(k for k in sd if k != 'step') == set(w)
