#!/usr/bin/env python3
"""WaveRNN batched vocoding throughput (BASELINE config 4): U utterances x T frames, RAW 9-bit mu-law."""
import argparse
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import estts  # noqa
from estts import configs, weights
from estts.wavernn import WaveRNN

if os.environ.get('WL_LIB'):          # A/B runs of two builds on one box (devices differ by several per cent)
    from estts import _lib
    _lib.LIB_PATH = os.path.abspath(os.environ['WL_LIB'])

ap = argparse.ArgumentParser()
ap.add_argument('--utts', type=int, default=64)
ap.add_argument('--frames', type=int, default=560)
ap.add_argument('--hop', type=int, default=256)
ap.add_argument('--precision', default='bf16x3')
ap.add_argument('--target', type=int, default=11000)
ap.add_argument('--overlap', type=int, default=550)
ap.add_argument('--steps-per-graph', type=int, default=50)
ap.add_argument('--fc-tile', default=None, help='dense layer tile, sequences x units per workgroup: 128x16, 128x32, 128x64 or 64x32')
ap.add_argument('--atomic-step', action='store_true', help='one step counter advanced by the last workgroup of the sampler')
ap.add_argument('--unfused', action='store_true', help='separate products and gate launches for the GRU layers')
args = ap.parse_args()
hp = configs.wavernn_hparams_22k() if args.hop == 256 else configs.wavernn_hparams(voc_mode='RAW')
m = WaveRNN.from_hparams(hp, precision=args.precision)
m.load_state_dict(weights.init_wavernn(hp, seed=1234))
m.steps_per_graph = args.steps_per_graph
if args.fc_tile:
    seqs, units = (int(v) for v in args.fc_tile.split('x'))
    m.fc_unit_blocks, m.fc_seq_tile = units // 16, seqs
    m._pack_fc(m._sd)
if args.atomic_step:
    m.alternate_step_counters = False
if args.unfused:
    m.fused = False
rng = np.random.default_rng(1234)
mels = [rng.uniform(0, 1, size=(hp['num_mels'], args.frames)).astype(np.float32) for _ in range(args.utts)]
torch.cuda.synchronize()
t0 = time.perf_counter()
outs = m.generate_batch(mels, True, args.target, args.overlap, True, seed=1)
torch.cuda.synchronize()
dt = time.perf_counter() - t0
samples = sum(len(o) for o in outs)
audio_s = samples / hp['sample_rate']
L = args.target + 2 * args.overlap
print(f'utts {args.utts} frames {args.frames} hop {hp["hop_length"]} precision {args.precision}: {dt:.3f} s wall for {audio_s:.1f} s audio '
      f'-> RTF {dt / audio_s:.5f}, {samples / dt / 1e3:.1f} k samples/s, {dt / L * 1e6:.1f} us per step ({L} steps)')
