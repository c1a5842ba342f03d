#!/usr/bin/env python3
"""train_forward.py of the reference (TransformerTTS/train_forward.py:60-140) on MI355X, one process per GPU:

  python -m torch.distributed.run --nproc-per-node 8 --master-addr 127.0.0.1 scripts/train_forward.py --config <dir> \
      --data <dir with mel / phoneme / duration .npy triples> | --synthetic

Each rank takes every world-th batch; gradients are summed over RCCL inside train_step.  Schedules (learning rate,
decoder-prenet dropout, heads to drop) come from the yaml config as in the reference's loop."""
import argparse
import glob
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import estts  # noqa: E402,F401
from estts import parallel  # noqa: E402
from estts.config_manager import ConfigManager  # noqa: E402
from estts.scheduling import piecewise_linear_schedule, reduction_schedule  # noqa: E402


def synthetic_batches(cfg, batch_size, rank, world):
    rng = np.random.default_rng(1234 + rank)
    while True:
        N = int(rng.integers(60, 140))
        tokens = rng.integers(1, cfg['vocab_size'], size=(batch_size, N)).astype(np.int32)
        durs = rng.integers(1, 9, size=(batch_size, N)).astype(np.int32)
        for b in range(batch_size):
            n = int(rng.integers(N // 2, N + 1))
            tokens[b, n:] = 0
            durs[b, n:] = 0
        T = int(durs.sum(1).max())
        mel = np.clip(rng.normal(0, 1.5, size=(batch_size, T, cfg['mel_channels'])), -4, 4).astype(np.float32)
        for b in range(batch_size):
            mel[b, durs[b].sum():] = 0
        yield mel, tokens, durs


def file_batches(data_dir, batch_size, rank, world):
    """Samples stored as <name>.npz with arrays mel (T, C), phonemes (N,), durations (N,) (sum = T)."""
    files = sorted(glob.glob(os.path.join(data_dir, '*.npz')))[rank::world]
    if not files:
        raise FileNotFoundError(f'no .npz samples under {data_dir}')
    rng = np.random.default_rng(rank)
    while True:
        rng.shuffle(files)
        for i in range(0, len(files), batch_size):                     # the last batch of a pass may be smaller
            items = [np.load(f) for f in files[i:i + batch_size]]
            batch_size_i = len(items)
            N, T = max(len(it['phonemes']) for it in items), max(it['mel'].shape[0] for it in items)
            C = items[0]['mel'].shape[1]
            mel, tok, dur = (np.zeros((batch_size_i, T, C), np.float32), np.zeros((batch_size_i, N), np.int32),
                             np.zeros((batch_size_i, N), np.int32))
            for b, it in enumerate(items):
                mel[b, :it['mel'].shape[0]] = it['mel']
                tok[b, :len(it['phonemes'])] = it['phonemes']
                dur[b, :len(it['durations'])] = it['durations']
            yield mel, tok, dur


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--config', required=True)
    ap.add_argument('--data')
    ap.add_argument('--synthetic', action='store_true')
    ap.add_argument('--max-steps', type=int)
    ap.add_argument('--session_name', default=None)
    args = ap.parse_args()
    rank, world, _ = parallel.init_distributed()
    cm = ConfigManager(args.config, 'forward', session_name=args.session_name)
    cfg = cm.config
    if rank == 0:
        cm.create_remove_dirs()
        cm.dump_config()
    model = cm.load_model(training=True, verbose=rank == 0)
    batches = (synthetic_batches(cm.model_config_dict(), cfg['batch_size'], rank, world) if args.synthetic or not args.data
               else file_batches(args.data, cfg['batch_size'], rank, world))
    max_steps = args.max_steps or cfg['max_steps']
    losses = []
    while model.step < max_steps:
        mel, phonemes, durations = next(batches)
        model.set_constants(decoder_prenet_dropout=float(piecewise_linear_schedule(model.step, cfg['decoder_prenet_dropout_schedule'])),
                            learning_rate=float(piecewise_linear_schedule(model.step, cfg['learning_rate_schedule'])),
                            drop_n_heads=reduction_schedule(model.step, cfg['head_drop_schedule']))
        out = model.train_step(input_sequence=phonemes, target_sequence=mel, target_durations=durations)
        losses.append(float(out['loss']))
        if rank == 0 and model.step % 10 == 0:
            print(f'step {model.step} loss {losses[-1]:.4f} (avg of last 100: {np.mean(losses[-100:]):.4f})', flush=True)
        if rank == 0 and model.step % cfg['weights_save_frequency'] == 0:
            cm.save_weights(str(cm.weights_dir / f'ckpt-{model.step}.npz'), model.weights(), model.step)
    if rank == 0:
        cm.save_weights(str(cm.weights_dir / f'ckpt-{model.step}.npz'), model.weights(), model.step)
    parallel.barrier()


if __name__ == '__main__':
    main()
