#!/usr/bin/env python3
"""How much of a ForwardTransformer training step is host launch time (GPU box only)."""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import estts  # noqa
from estts import configs, weights
from estts.forward_training import ForwardTransformerTrainer

cfg = configs.forward_config()
w = weights.init_forward_transformer(cfg, seed=1234)
B, N, T = 32, 100, 600
rng = np.random.default_rng(0)
tokens = rng.integers(1, cfg['vocab_size'], size=(B, N)).astype(np.int32)
d = rng.integers(1, 12, size=(B, N)).astype(np.float64)
d = np.floor(d * T / d.sum(axis=1, keepdims=True)); d[:, 0] += T - d.sum(axis=1)
mel = np.clip(rng.normal(0, 1.5, size=(B, T, 80)), -4, 4).astype(np.float32)
model = ForwardTransformerTrainer(w, cfg)
model._compile()
tok_d, mel_d, dur_d = torch.as_tensor(tokens).cuda(), torch.as_tensor(mel).cuda(), torch.as_tensor(d.astype(np.int32)).cuda()
for _ in range(3):
    model.train_step(tok_d, mel_d, dur_d)
torch.cuda.synchronize()
for _ in range(3):
    t0 = time.perf_counter()
    out = model.train_step(tok_d, mel_d, dur_d)
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print(f'host issue {1e3 * (t1 - t0):.2f} ms, then waited {1e3 * (t2 - t1):.2f} ms for the GPU')
