#!/usr/bin/env python3
"""train_wavernn.py of the reference (WaveRNN/train_wavernn.py) on MI355X, one process per GPU:

  python -m torch.distributed.run --nproc-per-node 8 --master-addr 127.0.0.1 scripts/train_wavernn.py \
      [--weights latest_weights.pyt] [--data <dir of .npz with x, y, mels windows>] [--synthetic] [--steps N]

Windows follow the reference's collate (utility/dataset.py): voc_seq_len samples, mels with voc_pad frames of context."""
import argparse
import glob
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import estts  # noqa: E402,F401
from estts import configs, parallel, weights  # noqa: E402
from estts.wavernn_training import WaveRNNTrainer  # noqa: E402


def synthetic_batches(hp, B, rank):
    rng = np.random.default_rng(rank)
    hops = hp['voc_seq_len'] // hp['hop_length']
    L, n_cls = hp['voc_seq_len'], 2 ** hp['bits']
    while True:
        labels = rng.integers(0, n_cls, size=(B, L + 1))
        wave = (2 * labels / (n_cls - 1.) - 1.).astype(np.float32)
        mels = rng.uniform(0, 1, size=(B, hp['num_mels'], hops + 2 * hp['voc_pad'])).astype(np.float32)
        yield wave[:, :L], (labels[:, 1:] if hp['voc_mode'] == 'RAW' else wave[:, 1:]), mels


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--weights')
    ap.add_argument('--data')
    ap.add_argument('--synthetic', action='store_true')
    ap.add_argument('--lr', type=float)
    ap.add_argument('--batch_size', type=int)
    ap.add_argument('--steps', type=int, default=1000)
    ap.add_argument('--mode', default=None, choices=['RAW', 'MOL'])
    args = ap.parse_args()
    rank, world, _ = parallel.init_distributed()
    hp = configs.wavernn_hparams(**({'voc_mode': args.mode} if args.mode else {}))
    hp.setdefault('voc_seq_len', hp['hop_length'] * 5)
    model = WaveRNNTrainer.from_hparams(hp)
    if args.weights and os.path.exists(args.weights):
        model.load(args.weights)
    else:
        model.load_state_dict({k: torch.from_numpy(np.asarray(v)) for k, v in weights.init_wavernn(hp, seed=1234).items()})
    B = args.batch_size or hp.get('voc_batch_size', 32)
    lr = args.lr or hp.get('voc_lr', 1e-4)
    if args.data and not args.synthetic:
        files = sorted(glob.glob(os.path.join(args.data, '*.npz')))[rank::world]

        def batches():
            while True:
                for f in files:
                    z = np.load(f)
                    yield z['x'], z['y'], z['mels']
        it = batches()
    else:
        it = synthetic_batches(hp, B, rank)
    start, running = time.time(), 0.0
    for i in range(1, args.steps + 1):
        x, y, m = next(it)
        running += float(model.train_step(x, y, m, lr=lr))
        if rank == 0 and i % 10 == 0:
            print(f'| Step {model.get_step()} | Loss: {running / i:.4f} | {i / (time.time() - start):.1f} steps/s |', flush=True)
    if rank == 0 and args.weights:
        model.save(args.weights)
    parallel.barrier()


if __name__ == '__main__':
    main()
