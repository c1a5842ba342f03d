"""CPU oracle for duration extraction from attention maps (NumPy, the reference's own arithmetic types).

TEST INFRASTRUCTURE ONLY (tests/, __graft_entry__.smoke(), bench.py's cpu_baseline leg).

Restates TransformerTTS/utils/alignments.py:10-165 on whole (frames, tokens) matrices, one-hot rows and all, the way the
reference does it, so that the device path (which works on peak positions and reductions) is checked against an independent
formulation.  TensorFlow enters the reference here only through the two padding masks (model/transformer_utils.py:61-69:
token == 0, sum of |mel| over channels == 0), restated with NumPy.
PARITY STATUS: the reference holds no test or fixture for this file and cannot be imported (TensorFlow absent), so this
restatement is unpinned; tests/test_oracle_alignments.py checks the invariants the reference asserts itself
(durations sum to the frame count, one peak per frame) and hand-worked cases.

Two places where the reference depends on the NumPy version are restated by their behaviour under the NumPy (< 1.20s) it was
written for: `if next_avail:` on an empty array is False (alignments.py:50), on a one-element array it is the element's truth.
"""
import numpy as np


def duration_to_alignment_matrix(durations):                                        # alignments.py:10-16
    durations = np.asarray(durations).astype(int)
    starts = np.cumsum(np.append([0], durations[:-1])).astype(int)
    tot = int(np.sum(durations))
    out = np.zeros((len(durations), tot))
    for i, (s, d) in enumerate(zip(starts, durations)):
        out[i, s:s + d] = 1.0
    return out


def clean_attention(binary_attention, jump_threshold):                              # alignments.py:19-28
    phon_idx = 0
    clean = np.zeros(binary_attention.shape)
    for i, av in enumerate(binary_attention):
        nxt = int(np.argmax(av))
        if abs(nxt - phon_idx) > jump_threshold:
            nxt = phon_idx
        phon_idx = nxt
        clean[i, min(phon_idx, clean.shape[1] - 1)] = 1
    return clean


def weight_mask(attention_weights):                                                 # alignments.py:31-36
    max_m, max_n = attention_weights.shape
    I = np.tile(np.arange(max_n), (max_m, 1)) / max_n
    J = np.swapaxes(np.tile(np.arange(max_m), (max_n, 1)), 0, 1) / max_m
    return np.sqrt(np.square(I - J))


def fill_zeros(duration, take_from='next'):                                         # alignments.py:39-53
    next_avail = None
    for i in range(len(duration)):
        if i < len(duration) - 1 and duration[i] == 0:
            if take_from == 'next':
                cand = np.where(duration[i:] > 1)[0]
                next_avail = int(cand[0]) if len(cand) >= 1 else 0      # empty -> falsy; one element -> its own truth
            elif take_from == 'max':
                next_avail = int(np.argmax(duration[i:]))
            if next_avail:
                duration[i] = 1
                duration[i + next_avail] -= 1
    return duration


def fix_attention_jumps(binary_attn, alignments_weights, binary_score):             # alignments.py:56-75
    scores, attns = [], []
    for jumpth in [2, 3, 4, 5]:
        cl = clean_attention(binary_attn, jumpth)
        attns.append(cl)
        scores.append(np.sum(alignments_weights * cl))
    best = int(np.argmin(scores))
    best_score, best_attn = scores[best], attns[best]
    while (best_score - binary_score) > 2. and jumpth < 20:
        jumpth += 1
        best_attn = clean_attention(binary_attn, jumpth)
        best_score = np.sum(alignments_weights * best_attn)
    return best_attn


def binary_attention(attention_weights):                                            # alignments.py:78-84
    peak = attention_weights.max(axis=1)
    hits = (attention_weights.T == peak)
    assert np.sum(np.sum(hits, axis=0) != 1) == 0, 'more than one attention peak in a frame'
    binary = hits.astype(int).T
    return binary, np.sum(attention_weights * binary)


def get_durations_from_alignment(batch_alignments, mels, phonemes, weighted=False, binary=False, fill_gaps=False,
                                 fix_jumps=False, fill_mode='max'):                 # alignments.py:87-165
    assert (binary is True) or (fix_jumps is False), 'Cannot fix jumps in non-binary attention.'
    mels = np.asarray(mels)
    phonemes = np.asarray(phonemes)
    mel_pad = (np.sum(np.abs(mels), axis=-1) == 0)
    phon_pad = (phonemes == 0)
    durations, unpad_mels, unpad_phonemes, final_alignment = [], [], [], []
    for i, al in enumerate(batch_alignments):
        al = np.asarray(al)
        mel_len = int(mels.shape[1] - np.sum(mel_pad[i]))
        phon_len = int(phonemes.shape[1] - np.sum(phon_pad[i]))
        unpad = al[:, 1:mel_len - 1, 1:phon_len - 1]
        unpad_mels.append(mels[i, 1:mel_len - 1, :])
        unpad_phonemes.append(phonemes[i, 1:phon_len - 1])
        wmask = weight_mask(unpad[0])
        head_scores, scored = [], []
        for a in unpad:
            score = np.sum(wmask * a)
            scored.append(a / score)
            head_scores.append(score)
        ref = np.sum(scored, axis=0) if weighted else unpad[int(np.argmin(head_scores))]
        if binary:
            battn, bscore = binary_attention(ref)
            if fix_jumps:
                battn = fix_attention_jumps(battn, wmask, bscore)
            integer = battn.sum(axis=0)
        else:
            att_dur = np.sum(ref, axis=0)
            normalized = att_dur * ((mel_len - 2) / np.sum(att_dur))
            integer = np.round(normalized)
            diff = np.sum(integer) - (mel_len - 2)
            while diff != 0:
                rounding = integer - normalized
                if diff > 0:
                    integer[np.argmax(rounding)] -= 1
                else:
                    integer[np.argmin(rounding)] += 1
                diff = np.sum(integer) - (mel_len - 2)
        if fill_gaps:
            integer = fill_zeros(integer, take_from=fill_mode)
        assert np.sum(integer) == mel_len - 2, f'{np.sum(integer)} vs {mel_len - 2}'
        new_alignment = duration_to_alignment_matrix(integer.astype(int))
        best_attention = unpad[int(np.argmin(head_scores))]
        final_alignment.append(best_attention.T + new_alignment)
        durations.append(integer)
    return durations, unpad_mels, unpad_phonemes, final_alignment
