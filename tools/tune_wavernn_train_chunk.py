"""Step time of the WaveRNN training step (32 windows x 1000 samples, RAW 9-bit) against the hand-over chunk of the two
overlapped GRU recurrences (WaveRNNTrainer.CHUNK)."""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import estts  # noqa: E402,F401
from estts import configs, weights  # noqa: E402
from estts.wavernn_training import WaveRNNTrainer  # noqa: E402

hp = configs.wavernn_hparams(voc_mode='RAW')
B, hops = 32, 5
L = hops * hp['hop_length']
w = weights.init_wavernn(hp, seed=1234)
rng = np.random.default_rng(7)
n_cls = 2 ** hp['bits']
labels = rng.integers(0, n_cls, size=(B, L + 1))
xd = torch.from_numpy((2 * labels[:, :L] / (n_cls - 1.) - 1.).astype(np.float32)).cuda()
yd = torch.from_numpy(labels[:, 1:].astype(np.int64)).cuda()
md = torch.from_numpy(rng.uniform(0, 1, size=(B, hp['num_mels'], hops + 2 * hp['voc_pad'])).astype(np.float32)).cuda()
for chunk in [int(a) for a in sys.argv[1:]] or [1000, 500, 250, 125, 50, 25]:
    model = WaveRNNTrainer.from_hparams(hp)
    model.CHUNK = chunk
    model.load_state_dict({k: torch.from_numpy(np.asarray(v)) for k, v in w.items()})
    for _ in range(3):
        loss = model.train_step(xd, yd, md, lr=1e-4)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(5):
        loss = model.train_step(xd, yd, md, lr=1e-4)
    torch.cuda.synchronize()
    print(f'chunk {chunk:5d}: {(time.perf_counter() - t0) / 5 * 1e3:7.2f} ms per step, loss {float(loss):.5f}', flush=True)
    del model
