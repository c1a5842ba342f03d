"""Lists the estts_gemm_wgrad calls of one ForwardTransformer training step (B = 32) with their operand layout, and times each
distinct convolution call again on its own operands right after the step (wall clock, 200 repetitions).  GPU box."""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import bench  # noqa: E402
import estts  # noqa: E402,F401
from estts import configs, train_ops as T, weights  # noqa: E402
from estts.forward_training import ForwardTransformerTrainer  # noqa: E402

B, N, Tm = 32, 100, 600
cfg = configs.forward_config()
w = weights.init_forward_transformer(cfg, seed=1234)
tokens, durs = bench.synthetic_utterance(cfg, B, N, Tm, seed=1234)
mel = np.clip(np.random.default_rng(99).normal(0, 1.5, size=(B, Tm, cfg['mel_channels'])), -4, 4).astype(np.float32)
model = ForwardTransformerTrainer(w, cfg, precision='bf16x3', seed=1234)
model.set_constants(learning_rate=1e-4, decoder_prenet_dropout=0.0, drop_n_heads=0)
tok_d, dur_d, mel_d = torch.from_numpy(tokens).cuda(), torch.from_numpy(durs).cuda(), torch.from_numpy(mel).cuda()
calls = []
orig = T.wgrad


def spy(x, dy, dw, **kw):
    calls.append((x, dy, dw, kw))
    return orig(x, dy, dw, **kw)


for _ in range(2):
    model.train_step(tok_d, mel_d, dur_d)
T.wgrad = spy
import estts.forward_training as ft  # noqa: E402
ft.T.wgrad = spy
model.train_step(tok_d, mel_d, dur_d)
ft.T.wgrad = orig
torch.cuda.synchronize()
seen = set()
for x, dy, dw, kw in calls:
    if kw.get('taps', 1) == 1:
        continue
    key = (tuple(x.shape), x.stride(0), tuple(dy.shape), dy.stride(0), tuple(dw.shape), dw.stride(0), kw.get('taps'), kw.get('lpad'), kw.get('M'),
           kw.get('seg_len'), x.data_ptr() % 256, dy.data_ptr() % 256, dw.data_ptr() % 256)
    if key in seen:
        continue
    seen.add(key)
    kw2 = dict(kw)
    kw2['db'] = None
    scratch = torch.zeros_like(dw)
    for target, name in ((dw, 'own dW'), (scratch, 'fresh dW')):
        for _ in range(5):
            orig(x, dy, target, **kw2)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(200):
            orig(x, dy, target, **kw2)
        torch.cuda.synchronize()
        print(key, name, f'{(time.perf_counter() - t0) / 200 * 1e6:.1f} us', flush=True)
