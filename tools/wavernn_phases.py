#!/usr/bin/env python3
"""Where the wall time of the config-4 WaveRNN pass goes: conditioning, step loop, device-to-host copy, host post-processing."""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import estts  # noqa
from estts import configs, weights
from estts import wavernn as W

hp = configs.wavernn_hparams_22k()
m = W.WaveRNN.from_hparams(hp, precision='bf16x3')
m.load_state_dict(weights.init_wavernn(hp, seed=1234))
rng = np.random.default_rng(1234)
mels = [rng.uniform(0, 1, size=(hp['num_mels'], 560)).astype(np.float32) for _ in range(64)]
m.generate_batch(mels[:1], True, 1000, 100, True, seed=0)
torch.cuda.synchronize()

stamps = {}
orig_prepare, orig_run = m._prepare, m._run_steps


def timed(name, fn):
    def wrapper(*a, **k):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        r = fn(*a, **k)
        torch.cuda.synchronize()
        stamps[name] = stamps.get(name, 0.0) + time.perf_counter() - t0
        return r
    return wrapper


m._prepare = timed('conditioning network (_prepare)', orig_prepare)
m._run_steps = timed('step loop (_run_steps)', orig_run)
W.xfade_and_unfold = timed('host cross-fade', W.xfade_and_unfold)
W.decode_mu_law = timed('host mu-law expansion', W.decode_mu_law)
t0 = time.perf_counter()
outs = m.generate_batch(mels, True, hp['voc_target'], hp['voc_overlap'], True, seed=1)
torch.cuda.synchronize()
total = time.perf_counter() - t0
for k, v in stamps.items():
    print(f'{k:40s} {v * 1e3:8.1f} ms')
print(f'{"everything else (args, copy, fade-out)":40s} {(total - sum(stamps.values())) * 1e3:8.1f} ms')
print(f'{"total":40s} {total * 1e3:8.1f} ms')
