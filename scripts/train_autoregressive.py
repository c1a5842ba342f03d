#!/usr/bin/env python3
"""train_autoregressive.py of the reference (TransformerTTS/train_autoregressive.py:60-290) on MI355X:

  python scripts/train_autoregressive.py --config <dir> (--data <dir of .npz samples> | --synthetic) [--max-steps N]

Every step sets the scheduled constants (decoder-prenet dropout, learning rate, reduction factor, heads to drop) and runs
train_models_step: one update of the autoregressive model and, with use_mine, one ascent step of every mutual-information
head (MINE / CLUB per mine_type and mine_pair_types).  Samples are .npz files with arrays mel (T, C), phonemes (N,) and,
for the speaker systems, speaker (speaker_embed_dim,); the start / end vectors and the stop labels are added here as the
reference's preprocessor does (TransformerTTS/preprocessing/data_handling.py).  One process per GPU under
torch.distributed.run: every rank takes its own batches, gradients are summed over RCCL inside train_step (beside the
backward pass) and the losses are divided by the global batch's padded size.  The mutual-information heads are per rank.
Checkpoints: ckpt-<step>.npz (variables, Adam moments) and mine<i>-<step>.npz per head (variables, Adam moments, step
count, the running terms of mi_holder); a restart restores both."""
import argparse
import glob
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import estts  # noqa: E402,F401
from estts import parallel  # noqa: E402
from estts.config_manager import ConfigManager  # noqa: E402
from estts.mine import CLUB, MINE  # noqa: E402
from estts.scheduling import piecewise_linear_schedule, reduction_schedule  # noqa: E402
from estts.training_steps import train_models_step  # noqa: E402


def collate(items, cfg, with_speaker):
    """items: (mel (T, C), phonemes (N,), speaker or None) -> mel [B, T + 2, C] framed by the start / end vectors,
    phonemes [B, N], stop [B, T + 2] (1 = frame, 2 = last frame, 0 = padding), speaker [B, speaker_embed_dim] or None."""
    B, C = len(items), cfg['mel_channels']
    T, N = max(it[0].shape[0] for it in items) + 2, max(len(it[1]) for it in items)
    mel, tok, stop = np.zeros((B, T, C), np.float32), np.zeros((B, N), np.int32), np.zeros((B, T), np.int32)
    for b, (m, p, _) in enumerate(items):
        n = m.shape[0] + 2
        mel[b, 0], mel[b, 1:n - 1], mel[b, n - 1] = cfg['mel_start_value'], m, cfg['mel_end_value']
        tok[b, :len(p)] = p
        stop[b, :n - 1], stop[b, n - 1] = 1, 2
    spk = np.stack([it[2] for it in items]).astype(np.float32) if with_speaker else None
    return mel, tok, stop, spk


def synthetic_batches(cfg, batch_size, with_speaker, seed):
    rng = np.random.default_rng(seed)
    while True:
        items = []
        for _ in range(batch_size):
            n = int(np.clip(rng.normal(60, 20), 10, 120))
            t = int(n * rng.uniform(4, 7))
            mel = np.clip(rng.normal(0, 1.5, size=(t, cfg['mel_channels'])), -3.9, 3.9).astype(np.float32)
            items.append((mel, rng.integers(1, cfg['vocab_size'], size=n).astype(np.int32),
                          rng.normal(0, 1, size=cfg['speaker_embed_dim']).astype(np.float32)))
        yield collate(items, cfg, with_speaker)


def file_batches(data_dir, cfg, batch_size, with_speaker, rank, world):
    files = sorted(glob.glob(os.path.join(data_dir, '*.npz')))[rank::world]
    if not files:
        raise FileNotFoundError(f'no .npz samples under {data_dir} for rank {rank}')
    rng = np.random.default_rng(rank)
    while True:
        rng.shuffle(files)
        for i in range(0, len(files), batch_size):
            items = []
            for f in files[i:i + batch_size]:
                with np.load(f) as it:
                    items.append((it['mel'], it['phonemes'], it['speaker'] if with_speaker else None))
            yield collate(items, cfg, with_speaker)


def build_heads(cfg):
    """train_autoregressive.py:69-101."""
    heads = []
    if not cfg.get('use_mine'):
        return heads
    for i, pair in enumerate(cfg['mine_pair_types']):
        kinds = {'CLUB': ['CLUB'], 'MINE': ['MINE'], 'MINE_CLUB': ['CLUB', 'MINE']}.get(cfg['mine_type'])
        if kinds is None:
            raise ValueError('mine_type is not right')
        for kind in kinds:
            if kind == 'CLUB':
                heads.append(CLUB(dense_hidden_units=cfg['mine_dense_hidden_units'], pair_type=pair, seed=100 + len(heads)))
            else:
                heads.append(MINE(conv_filters=cfg['mine_conv_filters'], conv_kernel=cfg['mine_conv_kernel'],
                                  dense_hidden_units=cfg['mine_dense_hidden_units'], beta_values=cfg['mine_beta_values'],
                                  divergence_type=cfg['divergence_type'], pair_type=pair, seed=100 + len(heads)))
    lr = float(np.array(cfg['learning_rate_mine_schedule'])[0, 1])
    for h in heads:
        h._compile({'learning_rate': lr, 'beta_1': 0.9, 'beta_2': 0.98, 'epsilon': 1e-9})
    return heads


def checkpoint(cm, model, heads, rank, mi_holder):
    """ckpt-<step>.npz for the model; mine<i>-<model step>.npz for head i: variables, Adam moments and step count, the state of
    its pairing generator, and the running terms of mi_holder (the smoothed exponentials and the last mutual-information
    value, which the next step's reported loss contains).  The heads are per rank (their gradients are not exchanged);
    rank 0's are the ones saved and every rank resumes from them."""
    model.broadcast_moving_statistics(src=0)
    if rank == 0:
        cm.save_weights(str(cm.weights_dir / f'ckpt-{model.step}.npz'), model.weights(), model.step, model.optimizer_state())
        for i, h in enumerate(heads):
            if h.store is not None:
                state = h.optimizer_state()
                state['__exp_terms__'] = np.asarray(torch.as_tensor(mi_holder['exp_terms']).detach().cpu().numpy(), np.float32)
                state['__mi_loss__'] = np.asarray(float(mi_holder['mi_loss']), np.float64)
                cm.save_weights(str(cm.weights_dir / f'mine{i}-{model.step}.npz'), h.weights(), h.step, state)


def restore_heads(cm, heads, mi_holder, model_step, verbose):
    """train_autoregressive.py:119-131: every head comes back from its own newest checkpoint (not newer than the model's),
    with its optimizer; mi_holder's running terms come back with it."""
    for i, h in enumerate(heads):
        files = [(int(p.rsplit('-', 1)[1][:-4]), p) for p in glob.glob(str(cm.weights_dir / f'mine{i}-*.npz'))]
        files = sorted(f for f in files if f[0] <= model_step)
        if not files:
            continue
        w, _, opt = cm.read_weights(files[-1][1], with_optimizer=True)
        h.load_weights(w)
        if '__adam_m__' in opt:
            h.load_optimizer_state(opt)
        if '__exp_terms__' in opt:
            mi_holder['exp_terms'] = np.asarray(opt['__exp_terms__'], np.float32)
            mi_holder['mi_loss'] = float(opt['__mi_loss__'])
        if verbose:
            print(f'restored head {i} ({h.name}:{h.pair_type}) from {files[-1][1]} at head step {h.step}')


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--config', required=True)
    ap.add_argument('--data')
    ap.add_argument('--synthetic', action='store_true')
    ap.add_argument('--max-steps', type=int)
    ap.add_argument('--session_name', default=None)
    ap.add_argument('--reset_dir', dest='clear_dir', action='store_true', help='deletes everything under this config\'s folder.')
    ap.add_argument('--reset_logs', dest='clear_logs', action='store_true', help='deletes logs under this config\'s folder.')
    ap.add_argument('--reset_weights', dest='clear_weights', action='store_true', help='deletes weights under this config\'s folder.')
    args = ap.parse_args()
    rank, world, _ = parallel.init_distributed()
    cm = ConfigManager(args.config, 'autoregressive', session_name=args.session_name)
    cfg = cm.config
    if rank == 0:
        cm.create_remove_dirs(clear_dir=args.clear_dir, clear_logs=args.clear_logs, clear_weights=args.clear_weights)
        cm.dump_config()
    parallel.barrier()
    model = cm.load_model(training=True, verbose=rank == 0)
    mcfg = cm.model_config_dict()
    with_speaker = 'speaker' in cfg['system_type']
    bs = int(cfg.get('tts_batch_size', cfg.get('batch_size', 8)))
    batches = (synthetic_batches(mcfg, bs, with_speaker, 1234 + rank) if args.synthetic or not args.data
               else file_batches(args.data, mcfg, bs, with_speaker, rank, world))
    heads = build_heads(cfg)
    sep = bool(cfg.get('use_mine') and cfg.get('mine_sep_call'))
    mine_batches = None
    if sep:                                                      # train_autoregressive.py:195-216,229-236
        mbs = int(np.array(cfg['mine_batch_size_schedule'])[0, 1])
        mine_batches = (synthetic_batches(mcfg, mbs, with_speaker, 4321 + rank) if args.synthetic or not args.data
                        else file_batches(args.data, mcfg, mbs, with_speaker, rank, world))
    n_beta = len(cfg.get('mine_beta_values', [0]))
    mi_holder = {'use_mine': bool(cfg.get('use_mine')), 'mine_sep_call': sep, 'mi_loss': 0.0,
                 'exp_terms': np.zeros((n_beta, 2), np.float32), 'weight_factor': float(cfg.get('mine_weight_factor', 0.0)),
                 'smoothing_factor': float(cfg.get('mine_smoothing_factor', 1.0))}
    restore_heads(cm, heads, mi_holder, model.step, rank == 0)
    max_steps = args.max_steps or cfg['max_steps']
    # ragged ranks: every rank divides its losses by the global batch's padded frame count (train_step(global_frames=...))
    losses = []
    while model.step < max_steps:
        mel, phonemes, stop, spk = next(batches)
        m_mel = m_ph = m_stop = m_spk = 0
        if sep:
            m_mel, m_ph, m_stop, m_spk = next(mine_batches)
        model.set_constants(decoder_prenet_dropout=float(piecewise_linear_schedule(model.step, cfg['decoder_prenet_dropout_schedule'])),
                            learning_rate=float(piecewise_linear_schedule(model.step, cfg['learning_rate_tts_schedule'])),
                            reduction_factor=reduction_schedule(model.step, cfg['reduction_factor_schedule']),
                            drop_n_heads=reduction_schedule(model.step, cfg['head_drop_schedule']))
        global_frames = None
        if world > 1:
            global_frames = int(parallel.max_over_ranks(mel.shape[1] - 1))
        out = train_models_step(mel, phonemes, stop, spk, m_mel, m_ph, m_stop, m_spk, model, heads, mi_holder,
                                cfg.get('train_text_encoder', True), cfg.get('train_style_encoder', True),
                                cfg.get('train_decoder', True), cfg.get('use_style_loss', False), global_frames=global_frames)
        losses.append(float(out['loss']))
        if rank == 0 and model.step % 10 == 0:
            print(f'step {model.step} r {model.r} loss {losses[-1]:.4f} (avg of last 100: {np.mean(losses[-100:]):.4f})', flush=True)
        if model.step % cfg.get('weights_save_frequency', 1000) == 0:
            checkpoint(cm, model, heads, rank, mi_holder)
    checkpoint(cm, model, heads, rank, mi_holder)
    parallel.barrier()


if __name__ == '__main__':
    main()
