#!/usr/bin/env python3
"""train_forward.py of the reference (TransformerTTS/train_forward.py:60-140) on MI355X, one process per GPU:

  python -m torch.distributed.run --nproc-per-node 8 --master-addr 127.0.0.1 scripts/train_forward.py --config <dir> \
      --data <dir with mel / phoneme / duration .npy triples> | --synthetic

Every rank builds the same global batch (world x batch_size utterances), shards it by mel length so that the ranks' frame
counts balance (parallel.shard_by_cost), and passes the global batch's padded lengths to train_step, which makes the N-rank
step equal to the one-GPU step on the whole global batch; gradients are summed over RCCL inside train_step.  Schedules
(learning rate, decoder-prenet dropout, heads to drop) come from the yaml config as in the reference's loop; checkpoints hold
the variables, Adam's moments and rank 0's BatchNorm moving statistics (broadcast to all ranks before they are written)."""
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


def collate(items, mel_channels):
    """items: list of (mel (T, C), phonemes (N,), durations (N,)) -> zero-padded (mel, tokens, durations)."""
    B = len(items)
    N, T = max(len(it[1]) for it in items), max(it[0].shape[0] for it in items)
    mel, tok, dur = np.zeros((B, T, mel_channels), np.float32), np.zeros((B, N), np.int32), np.zeros((B, N), np.int32)
    for b, (m, p, d) in enumerate(items):
        mel[b, :m.shape[0]] = m
        tok[b, :len(p)] = p
        dur[b, :len(d)] = d
    return mel, tok, dur


def shard_global_batch(items, rank, world):
    """This rank's share of one global batch, balanced by frames, and the global batch's padded (T, N)."""
    lengths = [it[0].shape[0] for it in items]
    mine = parallel.shard_by_cost(lengths, world)[rank]
    return [items[i] for i in mine], (max(lengths), max(len(it[1]) for it in items))


def synthetic_batches(cfg, batch_size, rank, world):
    """LJSpeech-shaped random utterances; every rank draws the same global batch from the same stream."""
    rng = np.random.default_rng(1234)
    while True:
        items = []
        for _ in range(batch_size * world):
            n = int(np.clip(rng.normal(100, 30), 20, 180))
            d = rng.integers(1, 9, size=n).astype(np.int32)
            mel = np.clip(rng.normal(0, 1.5, size=(int(d.sum()), cfg['mel_channels'])), -4, 4).astype(np.float32)
            items.append((mel, rng.integers(1, cfg['vocab_size'], size=n).astype(np.int32), d))
        mine, glob_len = shard_global_batch(items, rank, world)
        yield collate(mine, cfg['mel_channels']) + (glob_len,)


def file_batches(data_dir, batch_size, rank, world):
    """Samples stored as <name>.npz with arrays mel (T, C), phonemes (N,), durations (N,) (sum = T).  All ranks walk the
    same shuffled file list in global batches of world x batch_size and keep their frame-balanced share."""
    files = sorted(glob.glob(os.path.join(data_dir, '*.npz')))
    if not files:
        raise FileNotFoundError(f'no .npz samples under {data_dir}')
    rng = np.random.default_rng(0)
    while True:
        rng.shuffle(files)
        for i in range(0, len(files), batch_size * world):             # the last batch of a pass may be smaller
            group = files[i:i + batch_size * world]
            if len(group) < world:
                continue                                                 # not enough utterances for every rank
            items = []
            for f in group:
                with np.load(f) as it:
                    items.append((it['mel'], it['phonemes'], it['durations']))
            mine, glob_len = shard_global_batch(items, rank, world)
            yield collate(mine, items[0][0].shape[1]) + (glob_len,)


def checkpoint(cm, model, rank):
    model.broadcast_moving_statistics(src=0)             # per-replica BatchNorm statistics: every rank continues from rank 0's
    if rank == 0:
        cm.save_weights(str(cm.weights_dir / f'ckpt-{model.step}.npz'), model.weights(), model.step, model.optimizer_state())


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--config', required=True)
    ap.add_argument('--data')
    ap.add_argument('--synthetic', action='store_true')
    ap.add_argument('--max-steps', type=int)
    ap.add_argument('--session_name', default=None)
    # the reference's directory resets (train_forward.py:66-75)
    ap.add_argument('--reset_dir', dest='clear_dir', action='store_true', help='deletes everything under this config\'s folder.')
    ap.add_argument('--reset_logs', dest='clear_logs', action='store_true', help='deletes logs under this config\'s folder.')
    ap.add_argument('--reset_weights', dest='clear_weights', action='store_true', help='deletes weights under this config\'s folder.')
    args = ap.parse_args()
    rank, world, _ = parallel.init_distributed()
    cm = ConfigManager(args.config, 'forward', session_name=args.session_name)
    cfg = cm.config
    if rank == 0:
        cm.create_remove_dirs(clear_dir=args.clear_dir, clear_logs=args.clear_logs, clear_weights=args.clear_weights)
        cm.dump_config()
    parallel.barrier()
    model = cm.load_model(training=True, verbose=rank == 0)
    batches = (synthetic_batches(cm.model_config_dict(), cfg['batch_size'], rank, world) if args.synthetic or not args.data
               else file_batches(args.data, cfg['batch_size'], rank, world))
    max_steps = args.max_steps or cfg['max_steps']
    losses = []
    while model.step < max_steps:
        mel, phonemes, durations, global_lengths = next(batches)
        model.set_constants(decoder_prenet_dropout=float(piecewise_linear_schedule(model.step, cfg['decoder_prenet_dropout_schedule'])),
                            learning_rate=float(piecewise_linear_schedule(model.step, cfg['learning_rate_schedule'])),
                            drop_n_heads=reduction_schedule(model.step, cfg['head_drop_schedule']))
        out = model.train_step(input_sequence=phonemes, target_sequence=mel, target_durations=durations,
                               global_lengths=global_lengths if world > 1 else None)
        losses.append(float(out['loss']))
        if rank == 0 and model.step % 10 == 0:
            print(f'step {model.step} loss {losses[-1]:.4f} (avg of last 100: {np.mean(losses[-100:]):.4f})', flush=True)
        if model.step % cfg['weights_save_frequency'] == 0:
            checkpoint(cm, model, rank)
    checkpoint(cm, model, rank)
    parallel.barrier()


if __name__ == '__main__':
    main()
