#!/usr/bin/env python3
"""train_wavernn.py of the reference (WaveRNN/train_wavernn.py:15-124) on MI355X, one process per GPU:

  python -m torch.distributed.run --standalone --nproc-per-node 8 scripts/train_wavernn.py \
      [--data <dir of .npz with x, y, mels windows>] [--synthetic] [--lr L] [--batch_size B] [--force_train]

The loop keeps the reference's cadence: the session is restored from `checkpoints/<voc_model_id>.wavernn/latest_weights.pyt`
(a new session writes its initial weights there), training runs in epochs of `len(train_set)` iterations until
`voc_total_steps` (10 M with --force_train), every `voc_checkpoint_every` steps rank 0 vocodes `voc_gen_at_checkpoint` test
mels into `model_outputs/<voc_model_id>.wavernn/` and writes `checkpoint_<k>k_steps.pyt`, and every epoch ends with
`save(latest)` and a line in `log.txt`.  Differences: the test mels of a checkpoint are vocoded in ONE batched generation
(every fold of every utterance is one sequence of the step kernels) where the reference generates them one after the
other; gradients are summed over the ranks (RCCL) inside train_step.

Windows follow the reference's collate (utility/dataset.py:65-91): voc_seq_len samples, mels with voc_pad frames of
context.  --data: .npz files holding one batch each; --test_mels: directory of (num_mels, T) .npy mels for the
checkpoint generations (seeded random mels otherwise)."""
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
from estts.wavernn import WaveRNN  # noqa: E402
from estts.wavernn_training import WaveRNNTrainer  # noqa: E402

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from gen_wavernn import save_wav  # noqa: E402


def synthetic_batches(hp, B, rank, epoch):
    rng = np.random.default_rng(1000 * epoch + rank)
    hops = hp['voc_seq_len'] // hp['hop_length']
    L, n_cls = hp['voc_seq_len'], 2 ** hp['bits']
    while True:
        labels = rng.integers(0, n_cls, size=(B, L + 1))
        wave = (2 * labels / (n_cls - 1.) - 1.).astype(np.float32)
        mels = rng.uniform(0, 1, size=(B, hp['num_mels'], hops + 2 * hp['voc_pad'])).astype(np.float32)
        yield wave[:, :L], (labels[:, 1:] if hp['voc_mode'] == 'RAW' else wave[:, 1:]), mels


def gen_testset(trainer, hp, test_mels, samples, batched, target, overlap, save_path):
    """gen_wavernn.py:10-34 over the first `samples` test mels, all of them in one batched generation; the generator is an
    inference instance on the trainer's current variables.  Returns the files written."""
    mels = test_mels[:samples]
    if not mels:
        return []
    gen = WaveRNN.from_hparams(hp)
    gen.load_state_dict(trainer.state_dict())
    k = trainer.get_step() // 1000
    batch_str = f'gen_batched_target{target}_overlap{overlap}' if batched else 'gen_NOT_BATCHED'
    outs = gen.generate_batch(mels, batched, target, overlap, hp['mu_law'])
    files = []
    for i, y in enumerate(outs, 1):
        print('\n| Generating: %i/%i' % (i, len(mels)))
        files.append(os.path.join(save_path, f'{k}k_steps_{i}_{batch_str}.wav'))
        save_wav(y, files[-1], hp['sample_rate'])
    return files


def voc_train_loop(model, hp, train_set, total_iters, test_mels, lr, total_steps, paths, rank):
    """train_wavernn.py:15-64.  train_set(epoch) yields (x, y, mels) batches, total_iters of them make an epoch; as in the
    reference the number of whole epochs is fixed at the start ((total_steps - step) // total_iters + 1)."""
    epochs = (total_steps - model.get_step()) // total_iters + 1
    msg = ''
    for e in range(1, epochs + 1):
        start, running_loss = time.time(), 0.
        it = train_set(e)
        for i in range(1, total_iters + 1):
            x, y, m = next(it)
            running_loss += float(model.train_step(x, y, m, lr=lr))
            speed, avg_loss = i / (time.time() - start), running_loss / i
            step = model.get_step()
            if step % hp['voc_checkpoint_every'] == 0:
                model.broadcast_running_statistics(src=0)
                if rank == 0:
                    gen_testset(model, hp, test_mels, hp['voc_gen_at_checkpoint'], hp['voc_gen_batched'], hp['voc_target'],
                                hp['voc_overlap'], paths['output'])
                    model.checkpoint(paths['checkpoints'])
                parallel.barrier()
            msg = f'| Epoch: {e}/{epochs} ({i}/{total_iters}) | Loss: {avg_loss:.4f} | {speed:.1f} steps/s | Step: {step // 1000}k | '
            if rank == 0 and (i % 10 == 0 or i == total_iters):
                print(msg, flush=True)
        model.broadcast_running_statistics(src=0)
        if rank == 0:
            model.save(paths['latest'])
            model.log(paths['log'], msg)
        parallel.barrier()


def main():
    ap = argparse.ArgumentParser(description='Train WaveRNN Vocoder')
    ap.add_argument('--lr', '-l', type=float, help='[float] override the hparams learning rate')
    ap.add_argument('--batch_size', '-b', type=int, help='[int] override the hparams batch size')
    ap.add_argument('--force_train', '-f', action='store_true', help='Forces the model to train past total steps')
    ap.add_argument('--data', help='directory of .npz batches (x, y, mels)')
    ap.add_argument('--synthetic', action='store_true', help='seeded random windows instead of --data')
    ap.add_argument('--test_mels', help='directory of (num_mels, T) .npy mels vocoded at every checkpoint')
    ap.add_argument('--mode', default=None, choices=['RAW', 'MOL'])
    ap.add_argument('--checkpoints', help='checkpoint directory (default checkpoints/<voc_model_id>.wavernn/)')
    ap.add_argument('--outputs', help='directory of the checkpoint generations (default model_outputs/<voc_model_id>.wavernn/)')
    ap.add_argument('--total_steps', type=int, help='override voc_total_steps')
    ap.add_argument('--checkpoint_every', type=int, help='override voc_checkpoint_every')
    ap.add_argument('--gen_at_checkpoint', type=int, help='override voc_gen_at_checkpoint')
    ap.add_argument('--epoch_iters', type=int, default=1000, help='iterations per epoch of the synthetic train set')
    ap.add_argument('--gen_target', type=int, help='override voc_target for the checkpoint generations')
    ap.add_argument('--gen_overlap', type=int, help='override voc_overlap for the checkpoint generations')
    args = ap.parse_args()
    rank, world, _ = parallel.init_distributed()
    over = {k: v for k, v in (('voc_mode', args.mode), ('voc_checkpoint_every', args.checkpoint_every),
                              ('voc_gen_at_checkpoint', args.gen_at_checkpoint), ('voc_total_steps', args.total_steps),
                              ('voc_target', args.gen_target), ('voc_overlap', args.gen_overlap)) if v is not None}
    hp = configs.wavernn_hparams(**over)
    assert int(np.cumprod(hp['voc_upsample_factors'])[-1]) == hp['hop_length']         # train_wavernn.py:101
    ckpt_dir = args.checkpoints or os.path.join('checkpoints', f"{hp['voc_model_id']}.wavernn")
    out_dir = args.outputs or os.path.join('model_outputs', f"{hp['voc_model_id']}.wavernn")
    paths = {'checkpoints': ckpt_dir, 'latest': os.path.join(ckpt_dir, 'latest_weights.pyt'), 'output': out_dir,
             'log': os.path.join(ckpt_dir, 'log.txt')}
    if rank == 0:
        os.makedirs(ckpt_dir, exist_ok=True)
        os.makedirs(out_dir, exist_ok=True)
    parallel.barrier()

    if rank == 0:
        print('\nInitialising Model...\n')
    model = WaveRNNTrainer.from_hparams(hp)
    model.load_state_dict({k: torch.from_numpy(np.asarray(v)) for k, v in weights.init_wavernn(hp, seed=1234).items()})
    if rank == 0:
        model.restore(paths['latest'])
    parallel.barrier()
    if rank != 0:
        model.load(paths['latest'])

    B = args.batch_size or hp['voc_batch_size']
    lr = args.lr or hp['voc_lr']
    if args.data and not args.synthetic:
        files = sorted(glob.glob(os.path.join(args.data, '*.npz')))[rank::world]
        if not files:
            raise SystemExit(f'no .npz batches under {args.data} for rank {rank}')
        total_iters = len(files)

        def train_set(epoch):
            order = np.random.default_rng(epoch).permutation(len(files))
            for j in order:
                with np.load(files[j]) as z:
                    yield z['x'], z['y'], z['mels']
    else:
        total_iters = args.epoch_iters

        def train_set(epoch):
            return synthetic_batches(hp, B, rank, epoch)
    if args.test_mels:
        test_mels = [np.load(f, allow_pickle=False).astype(np.float32) for f in sorted(glob.glob(os.path.join(args.test_mels, '*.npy')))]
    else:
        rng = np.random.default_rng(4321)
        test_mels = [rng.uniform(0, 1, size=(hp['num_mels'], int(t))).astype(np.float32) for t in rng.integers(40, 90, size=hp['voc_gen_at_checkpoint'])]

    total_steps = 10_000_000 if args.force_train else hp['voc_total_steps']
    if rank == 0:
        print(f"| Remaining: {(total_steps - model.get_step()) // 1000}k Steps | Batch Size: {B} x {world} | LR: {lr} | "
              f"Sequence Len: {hp['voc_seq_len']} | Mode: {hp['voc_mode']} |", flush=True)
    voc_train_loop(model, hp, train_set, total_iters, test_mels, lr, total_steps, paths, rank)
    if rank == 0:
        print('Training Complete.')
        print('To continue training increase voc_total_steps or use --force_train')
    parallel.barrier()


if __name__ == '__main__':
    main()
