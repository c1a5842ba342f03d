"""The NumPy WaveRNN oracle against golden vectors produced by the reference module itself
(tests/golden/wavernn_*.npz, see tests/golden/make_wavernn_goldens.py)."""
import os

import numpy as np
import pytest

from wavernn_common import small_hp
from estts import weights
from oracle import wavernn_numpy as W


def _load(golden_dir, mode):
    g = np.load(os.path.join(golden_dir, f'wavernn_{mode.lower()}_small.npz'))
    hp = small_hp(mode)
    return g, hp, weights.init_wavernn(hp, seed=int(g['seed']))


@pytest.mark.parametrize('mode', ['RAW', 'MOL'])
def test_upsample_and_forward_match_reference(golden_dir, mode):
    g, hp, w = _load(golden_dir, mode)
    w64 = {k: v.astype(np.float64) for k, v in w.items()}
    mp = np.pad(g['mel'].astype(np.float64), ((0, 0), (0, 0), (hp['voc_pad'], hp['voc_pad'])))
    up, aux = W.upsample_network(mp, w64, hp['voc_upsample_factors'], hp['voc_res_blocks'], hp['voc_pad'])
    np.testing.assert_allclose(up, g['upsampled_mel'], atol=2e-6)
    np.testing.assert_allclose(aux, g['aux'], atol=2e-5)
    logits = W.forward(g['fwd_x'], g['fwd_mels'], w, hp)
    np.testing.assert_allclose(logits, g['fwd_logits'], atol=3e-5)


def test_generate_raw_trajectory_matches_reference(golden_dir):
    g, hp, w = _load(golden_dir, 'RAW')
    wav, logits, raw = W.generate(g['mel'], w, hp, g['u'], return_trace=True)
    # free-running: same uniforms -> same classes -> same waveform (float32 vs float64 logits never straddle a
    # CDF boundary for this seed)
    np.testing.assert_allclose(logits, g['gen_logits'], atol=5e-5)
    np.testing.assert_allclose(wav, g['wav'], atol=1e-6)


def test_hop256_fixture_matches_reference(golden_dir):
    """The vocoder shape of the benchmark (hop 256 = upsample factors (4, 8, 8)) at small widths, written by the reference."""
    from wavernn_common import small_hp_hop256
    g = np.load(os.path.join(golden_dir, 'wavernn_raw_hop256.npz'))
    hp = small_hp_hop256()
    w = weights.init_wavernn(hp, seed=int(g['seed']))
    w64 = {k: v.astype(np.float64) for k, v in w.items()}
    frames = g['upsampled_mel'].shape[1] // hp['hop_length']
    mp = np.pad(g['mel'][:, :, :frames].astype(np.float64), ((0, 0), (0, 0), (hp['voc_pad'], hp['voc_pad'])))
    up, aux = W.upsample_network(mp, w64, hp['voc_upsample_factors'], hp['voc_res_blocks'], hp['voc_pad'])
    np.testing.assert_allclose(up, g['upsampled_mel'], atol=2e-6)
    np.testing.assert_allclose(aux, g['aux'], atol=2e-5)
    np.testing.assert_allclose(W.forward(g['fwd_x'], g['fwd_mels'], w, hp), g['fwd_logits'], atol=3e-5)
    wav, logits, raw = W.generate(g['mel'], w, hp, g['u'], return_trace=True)
    np.testing.assert_allclose(logits[::8], g['gen_logits'], atol=5e-5)
    np.testing.assert_allclose(wav, g['wav'], atol=1e-6)


def test_generate_mol_trajectory_matches_reference(golden_dir):
    g, hp, w = _load(golden_dir, 'MOL')
    wav, logits, raw = W.generate(g['mel'], w, hp, {'mix': g['u_mix'], 'logistic': g['u_logistic']}, return_trace=True)
    np.testing.assert_allclose(logits, g['gen_logits'], atol=2e-4)
    np.testing.assert_allclose(wav, g['wav'], atol=2e-4)


def test_fold_unfold_shapes_and_mu_law():
    x = np.arange(10, dtype=np.float64).reshape(1, 10, 1) + 1
    f = W.fold_with_overlap(x, 2, 1)          # docstring example of fatchord_version.py:291-295
    assert f.shape == (3, 4, 1)
    assert f[:, :, 0].tolist() == [[1, 2, 3, 4], [4, 5, 6, 7], [7, 8, 9, 10]]
    # ragged tail is zero padded to one more fold
    f2 = W.fold_with_overlap(np.ones((1, 12, 1)), 2, 1)
    assert f2.shape == (4, 4, 1) and f2[3, :, 0].tolist() == [1, 1, 1, 0]
    y = W.xfade_and_unfold(np.ones((3, 8)), 4, 2)
    assert y.shape == (3 * 6 + 2,)
    labels = W.encode_mu_law(np.array([-1.0, 0.0, 1.0]), 512)
    assert labels.tolist() == [0.0, 256.0, 511.0]
    back = W.decode_mu_law(labels, 512)
    assert abs(back[0] + 1) < 1e-9 and abs(back[2] - 1) < 1e-9 and abs(back[1]) < 1e-4


@pytest.mark.parametrize('mode', ['RAW', 'MOL'])
def test_training_oracle_matches_reference_goldens(mode):
    """oracle/wavernn_train_torch.py against one training step of the reference module itself (loss, every gradient,
    BatchNorm running statistics, parameters after torch.optim.Adam)."""
    from estts import weights
    from oracle import wavernn_train_torch as OT
    from wavernn_common import small_hp
    g = np.load(os.path.join(os.path.dirname(__file__), 'golden', f'wavernn_train_{mode.lower()}_small.npz'))
    hp = small_hp(mode)
    w = weights.init_wavernn(hp, seed=int(g['seed']))
    ref = OT.train_step(w, hp, g['x'], g['y'], g['mels'])
    assert abs(ref['loss'] - float(g['loss'])) < 2e-6 * max(1.0, abs(float(g['loss'])))
    assert np.abs(ref['logits'] - g['logits']).max() < 2e-5
    names = [k[5:] for k in g.files if k.startswith('grad/')]
    assert set(names) == set(ref['grads'])
    for k in names:
        scale = max(np.abs(g['grad/' + k]).max(), 1e-6)
        assert np.abs(ref['grads'][k] - g['grad/' + k]).max() / scale < 2e-3, k       # reference runs in float32
    for k, v in ref['new_stats'].items():
        assert np.abs(v - g['after/' + k]).max() < 1e-5, k
    after = OT.adam_update(w, ref['grads'], {}, float(g['lr']), 1)
    for k in names:
        big = np.abs(g['grad/' + k]) > 0.05 * np.abs(g['grad/' + k]).max()           # the first Adam step is lr * sign(g)
        assert np.abs(after[k] - g['after/' + k])[big].max() < 2e-6, k
