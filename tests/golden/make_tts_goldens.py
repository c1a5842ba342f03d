"""Writes tests/golden/forward_small.npz from the float64 CPU oracle (oracle/tts_numpy.py).

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

if __name__ == '__main__':
    cfg = small_cfg()
    seed = 21
    w = weights.init_forward_transformer(cfg, seed=seed)
    tokens, durs = _inputs(cfg, 2, 19, seed=9)
    ref = O.forward_transformer_call(w, cfg, tokens, durs)
    np.savez_compressed(os.path.join(ROOT, 'tests', 'golden', 'forward_small.npz'), weight_seed=seed, tokens=tokens,
                        durations=durs, mel=ref['mel'].astype(np.float32), duration=ref['duration'].astype(np.float32))
    print('wrote forward_small.npz', ref['mel'].shape)
