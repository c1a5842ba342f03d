"""Duration extraction on the GPU (estts.alignments: three reduction launches + host repairs on index vectors) against the
matrix-form oracle (oracle/alignments_numpy.py).  Integer work: durations, unpadded mels / tokens must be identical; the
float64 reductions agree to 1e-12 relative (summation order); the diagnostic alignment image to 1e-6."""
import itertools

import numpy as np
import pytest
import torch

import estts  # noqa: F401
from oracle import alignments_numpy as O

pytestmark = pytest.mark.gpu


def ragged_batch(B, H, max_tokens, max_frames, channels, sharp, seed, jumps=0):
    """Padded batch of attention maps that follow random durations, with start / end entries and zero padding."""
    rng = np.random.default_rng(seed)
    Lk, Tm = max_tokens + 2, max_frames + 2
    Lq = Tm - 1
    maps = np.zeros((B, H, Lq, Lk), np.float32)
    mels = np.zeros((B, Tm, channels), np.float32)
    tokens = np.zeros((B, Lk), np.int32)
    truth = []
    for b in range(B):
        N = int(rng.integers(max(3, max_tokens // 2), max_tokens + 1)) if b else max_tokens
        d = rng.integers(1, 7, N)
        while d.sum() > max_frames:
            d[np.argmax(d)] -= 1
        if b == 0:
            d[0] += max_frames - d.sum()                                     # one utterance fills the padded size exactly
        T = int(d.sum())
        centre = np.repeat(np.arange(N), d)
        for h in range(H):
            logits = rng.normal(0, 0.3 + 0.6 * h, (Lq, Lk))
            logits[1:T + 1, 1:N + 1] += sharp * (np.arange(N)[None, :] == centre[:, None])
            for _ in range(jumps):
                t = int(rng.integers(1, T + 1))
                logits[t, 1 + int(rng.integers(0, N))] += 3 * sharp
            e = np.exp(logits - logits.max(axis=1, keepdims=True))
            maps[b, h] = e / e.sum(axis=1, keepdims=True)
        mels[b, :T + 2] = rng.normal(0, 1, (T + 2, channels))
        tokens[b, :N + 2] = rng.integers(1, 100, N + 2)
        truth.append(d)
    return maps, mels, tokens, truth


OPTIONS = [dict(weighted=w, binary=bn, fix_jumps=fj, fill_gaps=fg, fill_mode=fm)
           for w, bn, fj, (fg, fm) in itertools.product((False, True), (False, True), (False, True),
                                                        ((False, 'max'), (True, 'max'), (True, 'next')))
           if bn or not fj]


@pytest.mark.parametrize('sharp,jumps', [(8.0, 0), (4.0, 3), (1.5, 0)])
def test_matches_oracle(sharp, jumps):
    from estts.alignments import get_durations_from_alignment
    maps, mels, tokens, _ = ragged_batch(B=5, H=4, max_tokens=37, max_frames=150, channels=80, sharp=sharp, seed=int(sharp * 10),
                                         jumps=jumps)
    dev = torch.device('cuda:0')
    A, M, P = (torch.from_numpy(x).to(dev) for x in (maps, mels, tokens))
    for kw in OPTIONS:
        want = O.get_durations_from_alignment(maps, mels, tokens, **kw)
        got = get_durations_from_alignment(A, M, P, **kw)
        for b in range(len(maps)):
            assert np.array_equal(got[0][b], want[0][b]), (kw, b, got[0][b], want[0][b])
            assert got[0][b].dtype == want[0][b].dtype
            assert np.array_equal(got[1][b], want[1][b]) and np.array_equal(got[2][b], want[2][b])
            assert got[3][b].shape == want[3][b].shape
            assert np.abs(got[3][b] - want[3][b]).max() < 1e-6
    got = get_durations_from_alignment(maps, mels, tokens, binary=True, return_alignment=False)      # host arrays are uploaded
    want = O.get_durations_from_alignment(maps, mels, tokens, binary=True)
    assert all(np.array_equal(g, w) for g, w in zip(got[0], want[0])) and got[3][0] is None


def test_reductions_match_numpy():
    from estts.alignments import attention_reductions
    maps, mels, tokens, _ = ragged_batch(B=4, H=3, max_tokens=70, max_frames=200, channels=16, sharp=3.0, seed=5)
    dev = torch.device('cuda:0')
    for weighted in (False, True):
        r = attention_reductions(torch.from_numpy(maps).to(dev), torch.from_numpy(mels).to(dev), torch.from_numpy(tokens).to(dev),
                                 weighted=weighted)
        for b in range(len(maps)):
            ml = int((np.abs(mels[b]).sum(-1) != 0).sum())
            pl = int((tokens[b] != 0).sum())
            assert (r['mel_len'][b], r['phon_len'][b]) == (ml, pl)
            W = maps[b, :, 1:ml - 1, 1:pl - 1].astype(np.float64)
            wm = O.weight_mask(W[0])
            score = (W * wm).sum(axis=(1, 2))
            assert np.allclose(r['score'][b], score, rtol=1e-12, atol=0)
            assert np.allclose(r['colsum'][b, :, :pl - 2], W.sum(axis=1), rtol=1e-12, atol=0)
            assert np.all(r['colsum'][b, :, pl - 2:] == 0)
            assert r['best_head'][b] == int(np.argmin(score))
            ref = (W / score[:, None, None]).sum(axis=0) if weighted else W[int(np.argmin(score))]
            assert np.array_equal(r['peak_idx'][b, :ml - 2], ref.argmax(axis=1))
            assert np.allclose(r['peak_val'][b, :ml - 2], ref.max(axis=1), rtol=1e-12, atol=0)
            assert np.all(r['peak_ties'][b, :ml - 2] == 1) and np.all(r['peak_idx'][b, ml - 2:] == -1)


def test_ties_and_bad_lengths_are_reported():
    from estts.alignments import get_durations_from_alignment
    maps, mels, tokens, _ = ragged_batch(B=2, H=1, max_tokens=10, max_frames=30, channels=8, sharp=6.0, seed=1)
    tied = maps.copy()
    tied[1, 0, 4, :] = 0
    tied[1, 0, 4, 2] = tied[1, 0, 4, 5] = 0.5
    with pytest.raises(AssertionError):
        get_durations_from_alignment(tied, mels, tokens, binary=True)
    with pytest.raises(AssertionError):
        O.get_durations_from_alignment(tied, mels, tokens, binary=True)
    short = np.zeros_like(mels)
    short[:, :2] = 1.0                                                       # start and end vector only
    with pytest.raises(ValueError):
        get_durations_from_alignment(maps, short, tokens, binary=True)
    with pytest.raises(ValueError):
        get_durations_from_alignment(maps[:, :, :20], mels, tokens, binary=True)                  # map shorter than the mel


def test_full_size_properties():
    """BASELINE config 3 shape (32 x 4 heads x 600 x 100): durations are non-negative integers that sum to the frame count, and
    the peaks of sharp maps give back the durations they were drawn from."""
    from estts.alignments import get_durations_from_alignment
    maps, mels, tokens, truth = ragged_batch(B=32, H=4, max_tokens=100, max_frames=599, channels=80, sharp=9.0, seed=11)
    dev = torch.device('cuda:0')
    A, M, P = (torch.from_numpy(x).to(dev) for x in (maps, mels, tokens))
    for kw in (dict(binary=True, fix_jumps=True, fill_gaps=True, fill_mode='max'), dict(weighted=True)):
        d, um, up, fa = get_durations_from_alignment(A, M, P, return_alignment=False, **kw)
        for b in range(32):
            assert d[b].sum() == truth[b].sum() == len(um[b]) and len(d[b]) == len(truth[b]) == len(up[b])
            assert np.all(d[b] >= 0) and np.all(d[b] == np.round(d[b]))
            if kw.get('binary'):                                             # soft sums also count the softmax tails
                assert np.array_equal(d[b], truth[b])
