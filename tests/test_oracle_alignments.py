"""The duration-extraction oracle against hand-worked cases and the invariants the reference asserts itself
(TransformerTTS/utils/alignments.py:81-82,159).  The reference holds no test or fixture for that file: unpinned otherwise."""
import numpy as np
import pytest

from oracle import alignments_numpy as al


def hard_case(durations, heads=2, pad_frames=3, pad_tokens=2, channels=4, sharp=12.0, seed=0):
    """One utterance whose attention follows `durations`: returns (maps (H, T + 1 + pad, N + 2 + pad), mels, tokens)."""
    rng = np.random.default_rng(seed)
    durations = np.asarray(durations)
    N, T = len(durations), int(durations.sum())
    centre = np.repeat(np.arange(N), durations)                               # token of every inner frame
    Lq, Lk = T + 2 + pad_frames - 1, N + 2 + pad_tokens
    maps = np.zeros((heads, Lq, Lk), np.float32)
    for h in range(heads):
        logits = rng.normal(0, 0.3 + 1.5 * h, (Lq, Lk))
        logits[1:T + 1, 1:N + 1] += sharp * (np.arange(N)[None, :] == centre[:, None])
        e = np.exp(logits - logits.max(axis=1, keepdims=True))
        maps[h] = e / e.sum(axis=1, keepdims=True)
    mels = np.zeros((T + 2 + pad_frames, channels), np.float32)
    mels[:T + 2] = rng.normal(0, 1, (T + 2, channels))
    tokens = np.zeros(Lk, np.int32)
    tokens[:N + 2] = rng.integers(1, 50, N + 2)
    return maps, mels, tokens


def test_alignment_matrix():
    m = al.duration_to_alignment_matrix([2, 0, 1, 3])
    want = np.array([[1, 1, 0, 0, 0, 0], [0, 0, 0, 0, 0, 0], [0, 0, 1, 0, 0, 0], [0, 0, 0, 1, 1, 1]], float)
    assert np.array_equal(m, want)


def test_weight_mask_is_distance_from_the_diagonal():
    w = al.weight_mask(np.zeros((4, 2)))
    assert w.shape == (4, 2)
    assert np.allclose(w, np.abs(np.arange(2)[None, :] / 2 - np.arange(4)[:, None] / 4))


@pytest.mark.parametrize('kw', [dict(binary=True), dict(binary=False), dict(binary=True, fix_jumps=True),
                                dict(binary=True, weighted=True), dict(binary=False, weighted=True)])
def test_sharp_maps_give_back_the_durations(kw):
    truth = [3, 1, 4, 2, 5, 1, 2]
    maps, mels, tokens = hard_case(truth)
    d, um, up, fa = al.get_durations_from_alignment(maps[None], mels[None], tokens[None], **kw)
    assert np.array_equal(d[0], truth)
    assert um[0].shape == (sum(truth), mels.shape[1]) and np.array_equal(um[0], mels[1:1 + sum(truth)])
    assert np.array_equal(up[0], tokens[1:1 + len(truth)])
    assert fa[0].shape == (len(truth), sum(truth))
    assert np.all(fa[0].max(axis=0) > 1.0)                                    # soft map + hard alignment overlap everywhere


def test_jump_repair_removes_a_stray_peak():
    truth = [4, 4, 4, 4, 4, 4]
    maps, mels, tokens = hard_case(truth, heads=1)
    maps[0, 1 + 5, :] = 0
    maps[0, 1 + 5, 1 + 5] = 1.0                                               # frame 5 (token 1) jumps to token 5
    plain, _, _, _ = al.get_durations_from_alignment(maps[None], mels[None], tokens[None], binary=True)
    fixed, _, _, _ = al.get_durations_from_alignment(maps[None], mels[None], tokens[None], binary=True, fix_jumps=True)
    assert np.array_equal(plain[0], [4, 3, 4, 4, 4, 5])
    assert np.array_equal(fixed[0], truth)


def test_fill_zeros():
    assert np.array_equal(al.fill_zeros(np.array([0., 3, 1]), 'next'), [1, 2, 1])
    assert np.array_equal(al.fill_zeros(np.array([0., 1, 1]), 'next'), [0, 1, 1])          # nothing above 1 to take from
    assert np.array_equal(al.fill_zeros(np.array([0., 1, 4, 0]), 'max'), [1, 1, 3, 0])      # the last entry is never filled
    assert np.array_equal(al.fill_zeros(np.array([2., 0, 0, 5]), 'next'), [2, 1, 1, 3])
    assert np.array_equal(al.fill_zeros(np.array([0., 0, 0]), 'max'), [0, 0, 0])


def test_soft_maps_sum_to_the_frame_count():
    rng = np.random.default_rng(3)
    for trial in range(5):
        truth = rng.integers(1, 6, 9)
        maps, mels, tokens = hard_case(truth, heads=3, sharp=2.0, seed=trial)               # diffuse attention
        for kw in (dict(), dict(weighted=True), dict(fill_gaps=True, fill_mode='next'), dict(fill_gaps=True, fill_mode='max')):
            d, _, _, _ = al.get_durations_from_alignment(maps[None], mels[None], tokens[None], **kw)
            assert d[0].sum() == truth.sum() and np.all(d[0] >= 0) and np.all(d[0] == np.round(d[0]))


def test_double_peak_is_rejected():
    maps, mels, tokens = hard_case([2, 2, 2], heads=1)
    maps[0, 2, 1] = maps[0, 2, 2] = 0.4
    maps[0, 2, 3:] = 0.0
    with pytest.raises(AssertionError):
        al.get_durations_from_alignment(maps[None], mels[None], tokens[None], binary=True)
