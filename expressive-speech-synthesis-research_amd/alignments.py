"""Duration extraction: decoder cross-attention maps -> integer token durations (the targets `train_forward` learns).

Mirrors `get_durations_from_alignment` of TransformerTTS/utils/alignments.py:87-165 (same arguments, same four return values)
as called by TransformerTTS/extract_durations.py:177,212 on the last decoder block's attention.  The reference copies every
(heads, frames, tokens) map to the host; here the maps stay on the GPU and three launches reduce them to what the sequential
part needs (estts_alignment_lengths / _stats / _peaks: padding lengths, per-head diagonal scores and column sums, per-frame
peak position / value / multiplicity).  The sequential repairs (alignments.py:19-28,39-75,141-157) then run on the host over
index vectors instead of one-hot matrices: a cleaned map is its vector of peak positions p[t], its score
sum_t |p[t] / N' - t / T'|, its durations bincount(p).

`return_alignment=False` skips the one remaining large copy (the best head's map, needed only for the diagnostic image the
reference logs) and returns None in its place.
"""
import numpy as np
import torch

from . import _lib
from ._lib import check
from .ops import _stream


def duration_to_alignment_matrix(durations):
    """alignments.py:10-16: row i is 1 over the frames token i spans."""
    durations = np.asarray(durations).astype(np.int64)
    ends = np.cumsum(durations)
    t = np.arange(int(ends[-1]) if len(ends) else 0)[None, :]
    return ((t >= (ends - durations)[:, None]) & (t < ends[:, None])).astype(np.float64)


def _clean_positions(peaks, jump_threshold):
    """clean_attention (alignments.py:19-28) on peak positions: a jump of more than the threshold keeps the previous position."""
    out = np.empty_like(peaks)
    cur = 0
    for t, p in enumerate(peaks.tolist()):
        if abs(p - cur) <= jump_threshold:
            cur = p
        out[t] = cur
    return out


def _diagonal_score(pos, n_tokens):
    """sum over frames of weight_mask (alignments.py:31-36) at the map's single 1 per frame."""
    T = len(pos)
    return float(np.sum(np.abs(pos / n_tokens - np.arange(T) / T)))


def _fix_jumps(peaks, n_tokens, binary_score):
    """fix_attention_jumps (alignments.py:56-75)."""
    cleaned = [_clean_positions(peaks, j) for j in (2, 3, 4, 5)]
    scores = [_diagonal_score(c, n_tokens) for c in cleaned]
    best = int(np.argmin(scores))
    pos, score, jump = cleaned[best], scores[best], 5
    while score - binary_score > 2.0 and jump < 20:
        jump += 1
        pos = _clean_positions(peaks, jump)
        score = _diagonal_score(pos, n_tokens)
    return pos


def fill_zeros(duration, take_from='next'):
    """alignments.py:39-53: a zero duration (except the last) takes one frame from the next duration above 1 ('next') or from
    the largest remaining one ('max'); left alone when there is none."""
    for i in range(len(duration) - 1):
        if duration[i] != 0:
            continue
        rest = duration[i:]
        if take_from == 'next':
            above = np.flatnonzero(rest > 1)
            k = int(above[0]) if len(above) else 0
        elif take_from == 'max':
            k = int(np.argmax(rest))
        else:
            raise ValueError(f"fill_mode must be 'next' or 'max', got {take_from!r}")
        if k:
            duration[i] = 1
            duration[i + k] -= 1
    return duration


def _round_to_total(col, total):
    """alignments.py:140-154: scale the column sums to `total` frames, round half to even, then move single frames at the
    largest rounding errors until the sum is exact."""
    normalized = col * (total / np.sum(col))
    integer = np.round(normalized)
    diff = np.sum(integer) - total
    while diff != 0:
        err = integer - normalized
        if diff > 0:
            integer[np.argmax(err)] -= 1
        else:
            integer[np.argmin(err)] += 1
        diff = np.sum(integer) - total
    return integer


def attention_reductions(batch_alignments, mels, phonemes, weighted=False):
    """The device part.  batch_alignments (B, H, Lq, Lk) float32, mels (B, T, C) float32, phonemes (B, N) int32, all on the GPU.
    Returns host arrays: mel_len, phon_len (B,), score (B, H) f64, colsum (B, H, Lk) f64, peak_idx / peak_ties (B, Lq) i32,
    peak_val (B, Lq) f64, best_head (B,)."""
    A, mels, phonemes = batch_alignments, mels, phonemes
    if A.dtype != torch.float32 or mels.dtype != torch.float32 or not A.is_cuda or not mels.is_cuda or not phonemes.is_cuda:
        raise TypeError('alignments and mels must be float32 CUDA tensors, phonemes a CUDA tensor')
    A, mels = A.contiguous(), mels.contiguous()
    phonemes = phonemes.to(torch.int32).contiguous()
    B, H, Lq, Lk = A.shape
    if mels.shape[0] != B or phonemes.shape[0] != B or phonemes.shape[1] != Lk:
        raise ValueError(f'shape mismatch: alignments {tuple(A.shape)}, mels {tuple(mels.shape)}, phonemes {tuple(phonemes.shape)}')
    dev = A.device
    lens = torch.empty(2, B, dtype=torch.int32, device=dev)
    score = torch.empty(B, H, dtype=torch.float64, device=dev)
    colsum = torch.empty(B, H, Lk, dtype=torch.float64, device=dev)
    pidx = torch.empty(2, B, Lq, dtype=torch.int32, device=dev)
    pval = torch.empty(B, Lq, dtype=torch.float64, device=dev)
    best = torch.empty(B, dtype=torch.int32, device=dev)
    lib, s = _lib.load(), _stream()
    check(lib.estts_alignment_lengths(mels.data_ptr(), mels.stride(1), mels.shape[1], mels.shape[2], phonemes.data_ptr(), Lk, B,
                                      lens[0].data_ptr(), lens[1].data_ptr(), s), 'estts_alignment_lengths')
    check(lib.estts_alignment_stats(A.data_ptr(), B, H, Lq, Lk, lens[0].data_ptr(), lens[1].data_ptr(), score.data_ptr(),
                                    colsum.data_ptr(), s), 'estts_alignment_stats')
    check(lib.estts_alignment_peaks(A.data_ptr(), B, H, Lq, Lk, lens[0].data_ptr(), lens[1].data_ptr(), score.data_ptr(),
                                    int(bool(weighted)), pidx[0].data_ptr(), pval.data_ptr(), pidx[1].data_ptr(), best.data_ptr(), s),
          'estts_alignment_peaks')
    lens_h, pidx_h = lens.cpu().numpy(), pidx.cpu().numpy()
    return dict(mel_len=lens_h[0], phon_len=lens_h[1], score=score.cpu().numpy(), colsum=colsum.cpu().numpy(), peak_idx=pidx_h[0],
                peak_ties=pidx_h[1], peak_val=pval.cpu().numpy(), best_head=best.cpu().numpy())


def get_durations_from_alignment(batch_alignments, mels, phonemes, weighted=False, binary=False, fill_gaps=False, fix_jumps=False,
                                 fill_mode='max', return_alignment=True):
    """Same contract as the reference function.  Inputs may be CUDA tensors (they stay on the device) or host arrays (uploaded).
    Returns (durations, unpad_mels, unpad_phonemes, final_alignment): per utterance a vector of N' integer durations summing to T'
    (dtype as in the reference: int64 from plain binary maps, float otherwise), the (T', C) mel and (N',) tokens without
    start / end entries, and best-head map^T + the new hard alignment."""
    assert (binary is True) or (fix_jumps is False), 'Cannot fix jumps in non-binary attention.'
    if not torch.cuda.is_available():
        raise _lib.EsttsError('duration extraction needs the GPU: there is no CPU fallback')
    dev = torch.device('cuda', torch.cuda.current_device())
    A = batch_alignments if torch.is_tensor(batch_alignments) else torch.from_numpy(np.ascontiguousarray(batch_alignments, np.float32))
    M = mels if torch.is_tensor(mels) else torch.from_numpy(np.ascontiguousarray(mels, np.float32))
    P = phonemes if torch.is_tensor(phonemes) else torch.from_numpy(np.ascontiguousarray(phonemes).astype(np.int32))
    A, M, P = A.to(dev), M.to(dev), P.to(dev)
    r = attention_reductions(A, M, P, weighted=weighted)
    B, H, Lq, Lk = A.shape
    mels_h = M.cpu().numpy()
    phon_h = P.cpu().numpy()
    durations, unpad_mels, unpad_phonemes, final_alignment = [], [], [], []
    for b in range(B):
        mel_len, phon_len = int(r['mel_len'][b]), int(r['phon_len'][b])
        T, N = mel_len - 2, phon_len - 2
        if T < 1 or N < 1:
            raise ValueError(f'utterance {b}: {mel_len} frames / {phon_len} tokens leave nothing between start and end')
        if mel_len - 1 > Lq or phon_len > Lk:
            raise ValueError(f'utterance {b}: {mel_len} frames / {phon_len} tokens exceed the {Lq} x {Lk} attention map')
        unpad_mels.append(mels_h[b, 1:mel_len - 1, :])
        unpad_phonemes.append(phon_h[b, 1:phon_len - 1])
        best = int(r['best_head'][b])
        if binary:
            if np.any(r['peak_ties'][b, :T] != 1):
                raise AssertionError(f'utterance {b}: more than one attention peak in a frame')       # alignments.py:81-82
            pos = r['peak_idx'][b, :T].astype(np.int64)
            if fix_jumps:
                pos = _fix_jumps(pos, N, float(np.sum(r['peak_val'][b, :T])))
            integer = np.bincount(pos, minlength=N)                                  # int64, like the one-hot column sums ...
            if fix_jumps:
                integer = integer.astype(np.float64)                                 # ... the repaired map is a float64 one
        else:
            # the reference's best-head branch stays in the map's float32, its weighted branch is float64 (scores are)
            col = ((r['colsum'][b, :, :N] / r['score'][b][:, None]).sum(axis=0) if weighted
                   else r['colsum'][b, best, :N].astype(np.float32))
            integer = _round_to_total(col, T)
        if fill_gaps:
            integer = fill_zeros(integer, take_from=fill_mode)
        assert np.sum(integer) == T, f'{np.sum(integer)} vs {T}'
        durations.append(integer)
        if return_alignment:
            best_map = A[b, best, 1:mel_len - 1, 1:phon_len - 1].cpu().numpy()
            final_alignment.append(best_map.T + duration_to_alignment_matrix(integer))
        else:
            final_alignment.append(None)
    return durations, unpad_mels, unpad_phonemes, final_alignment
