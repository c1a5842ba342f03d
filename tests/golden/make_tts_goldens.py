"""Writes tests/golden/forward_small.npz and tests/golden/ar_gst_small.npz from the float64 CPU oracle (oracle/tts_numpy.py).

TensorFlow cannot be run here, so these vectors pin the build against its own oracle (itself
cross-checked against a second restatement and the reference's known-answer tests), not against
TensorFlow: parity vs the original framework stays "unpinned" for the model arithmetic.
Run from the repository root:  python tests/golden/make_tts_goldens.py
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests'))

import estts  # noqa: E402
from estts import weights  # noqa: E402
from oracle import tts_numpy as O  # noqa: E402
from test_gpu_forward_transformer import _inputs, small_cfg  # noqa: E402

def ar_gst_inputs():
    """Small GST + speaker + text model and a ragged teacher-forced batch (shared with the tests that read the fixture)."""
    from test_gpu_autoregressive import small_cfg as ar_small_cfg
    cfg = ar_small_cfg('speaker_style_text', dense=1)
    rng = np.random.default_rng(31)
    B, N, Tq = 3, 13, 17
    tokens = rng.integers(1, cfg['vocab_size'], size=(B, N)).astype(np.int32)
    targets = np.clip(rng.normal(0, 1.5, size=(B, Tq, 80)), -4, 4).astype(np.float32)
    for b, (n, t) in enumerate(((13, 17), (10, 12), (7, 9))):
        tokens[b, n:] = 0
        targets[b, t:] = 0
    spk = rng.normal(size=(B, 1, cfg['speaker_embed_dim'])).astype(np.float32)
    return cfg, 41, tokens, targets, spk, 2


def make_ar_gst():
    cfg, seed, tokens, targets, spk, r = ar_gst_inputs()
    w = weights.init_autoregressive_transformer(cfg, seed=seed)
    ref = O.autoregressive_call(w, cfg, tokens, targets, spk, r=r)
    f32 = lambda a: np.asarray(a).astype(np.float32)
    np.savez_compressed(os.path.join(ROOT, 'tests', 'golden', 'ar_gst_small.npz'), weight_seed=seed, r=r, tokens=tokens,
                        targets=targets, spk=spk, final_output=f32(ref['final_output']), mel_linear=f32(ref['mel_linear']),
                        stop_prob=f32(ref['stop_prob']), gst_output=f32(ref['gst_output']),
                        gst_encoder_attention=f32(ref['gst_encoder_attention']),
                        text_enc_output=f32(ref['text_enc_output']),
                        last_cross_attention=f32(ref['decoder_attention'][-1]))
    print('wrote ar_gst_small.npz', ref['final_output'].shape)


if __name__ == '__main__':
    make_ar_gst()
    cfg = small_cfg()
    seed = 21
    w = weights.init_forward_transformer(cfg, seed=seed)
    tokens, durs = _inputs(cfg, 2, 19, seed=9)
    ref = O.forward_transformer_call(w, cfg, tokens, durs)
    np.savez_compressed(os.path.join(ROOT, 'tests', 'golden', 'forward_small.npz'), weight_seed=seed, tokens=tokens,
                        durations=durs, mel=ref['mel'].astype(np.float32), duration=ref['duration'].astype(np.float32))
    print('wrote forward_small.npz', ref['mel'].shape)
