#!/usr/bin/env python3
"""Token durations from the autoregressive model's attention: the data `train_forward` learns from.

The flow of the reference's extract_durations.py: `val_step` (teacher-forced pass) over every batch, the last decoder block's
cross-attention into `get_durations_from_alignment`, one file per utterance.  The attention maps never leave the GPU; only
per-frame peak positions and per-token sums do.

  python scripts/extract_durations.py --config config/wavernn [--weights ckpt.npz] (--data <dir of .npz> | --synthetic N) --out <dir> \
      [--best] [--binary] [--fix_jumps] [--fill_mode_max | --fill_mode_next]

Input samples: <name>.npz with mel (T, C) including the start / end vectors and phonemes (N,) including the start / end tokens.
Output samples: <name>.npz with mel (T - 2, C), phonemes (N - 2,), durations (N - 2,), the format scripts/train_forward.py reads.
"""
import argparse
import glob
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import estts  # noqa: E402,F401
from estts.alignments import get_durations_from_alignment  # noqa: E402
from estts.config_manager import ConfigManager  # noqa: E402


def synthetic_samples(n, cfg, seed=0):
    rng = np.random.default_rng(seed)
    for i in range(n):
        N, T = int(rng.integers(20, 60)), int(rng.integers(120, 400))
        mel = np.clip(rng.normal(0, 1.5, (T, cfg['mel_channels'])), -4, 4).astype(np.float32)
        mel[0], mel[-1] = cfg['mel_start_value'], cfg['mel_end_value']
        yield f'synthetic_{i}', mel, rng.integers(1, cfg['vocab_size'], N).astype(np.int32)


def stored_samples(data_dir):
    for f in sorted(glob.glob(os.path.join(data_dir, '*.npz'))):
        with np.load(f, allow_pickle=False) as z:
            yield os.path.splitext(os.path.basename(f))[0], z['mel'].astype(np.float32), z['phonemes'].astype(np.int32)


def batches(samples, batch_size):
    group = []
    for s in samples:
        group.append(s)
        if len(group) == batch_size:
            yield group
            group = []
    if group:
        yield group


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--config', required=True)
    ap.add_argument('--weights')
    ap.add_argument('--data')
    ap.add_argument('--synthetic', type=int, default=0)
    ap.add_argument('--out', required=True)
    ap.add_argument('--best', action='store_true', help='use the best head instead of the score-weighted sum of heads')
    ap.add_argument('--binary', action='store_true', help='count attention peaks instead of summing attention')
    ap.add_argument('--fix_jumps', action='store_true', help='repair jumps of the attention peak (with --binary)')
    ap.add_argument('--fill_mode_max', action='store_true')
    ap.add_argument('--fill_mode_next', action='store_true')
    args = ap.parse_args()
    assert not (args.fill_mode_max and args.fill_mode_next), 'Choose one gap filling mode.'
    cm = ConfigManager(args.config, 'autoregressive')
    model = cm.load_model(args.weights, verbose=True)
    model.set_constants(reduction_factor=1, decoder_prenet_dropout=0.0)      # extract_durations.py:110: r must be 1
    cfg = cm.model_config_dict()
    n_layers, n_dense = len(cm.config['decoder_num_heads']), int(cm.config['decoder_dense_blocks'])
    last = f'Decoder_ConvBlock{n_layers - n_dense}_CrossAttention' if n_layers > n_dense else f'Decoder_DenseBlock{n_dense}_CrossAttention'
    print(f'Extracting attention from layer {last}')
    samples = stored_samples(args.data) if args.data else synthetic_samples(args.synthetic or 8, cfg)
    os.makedirs(args.out, exist_ok=True)
    spk_dim = cfg['speaker_embed_dim']
    all_durations, count = [], 0
    for group in batches(samples, 5 * cm.config.get('batch_size', cm.config.get('tts_batch_size', 8))):   # extract_durations.py:99; the shipped yaml names it tts_batch_size
        B, N, T = len(group), max(len(p) for _, _, p in group), max(len(m) for _, m, _ in group)
        mel = np.zeros((B, T, cfg['mel_channels']), np.float32)
        phon = np.zeros((B, N), np.int32)
        stop = np.zeros((B, T), np.int32)
        for b, (_, m, p) in enumerate(group):
            mel[b, :len(m)], phon[b, :len(p)] = m, p
            stop[b, :len(m)], stop[b, len(m) - 1] = 1, 2
        mel_d, phon_d = torch.from_numpy(mel).cuda(), torch.from_numpy(phon).cuda()
        spk = torch.zeros(B, 1, spk_dim, device='cuda') if 'speaker' in cfg['system_type'] else None
        out = model.val_step(phon_d, mel_d, stop, spk)
        durations, unpad_mels, unpad_phonemes, _ = get_durations_from_alignment(
            out['decoder_attention'][last], mel_d, phon_d, weighted=not args.best, binary=args.binary,
            fill_gaps=args.fill_mode_max or args.fill_mode_next, fix_jumps=args.fix_jumps,
            fill_mode='max' if args.fill_mode_max else 'next', return_alignment=False)
        for (name, _, _), d, m, p in zip(group, durations, unpad_mels, unpad_phonemes):
            np.savez(os.path.join(args.out, name + '.npz'), mel=m, phonemes=p, durations=d.astype(np.int32))
            all_durations.append(d)
        count += B
    d = np.concatenate(all_durations)
    print(f'{count} utterances, {len(d)} tokens: mean duration {d.mean():.2f}, zeros {int((d == 0).sum())}, max {int(d.max())} -> {args.out}')


if __name__ == '__main__':
    main()
