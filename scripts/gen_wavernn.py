#!/usr/bin/env python3
"""Vocode with WaveRNN on one MI355X: the reference's WaveRNN/gen_wavernn.py (`gen_from_file`, and its test-set loop as a
directory of mel files).

  python scripts/gen_wavernn.py --file speech.wav [--weights latest_weights.pyt] [-t 11000 -o 550] [--unbatched] [--out dir]
  python scripts/gen_wavernn.py --mels dir_with_npy_mels [...]

--file: 16-bit mono PCM at the hparams' sample rate; its mel is computed on the GPU (dsp.melspectrogram) and vocoded, the
target and the generated waveform are written next to each other as in the reference.  --mels: every <name>.npy there
((num_mels, T) in [0, 1]) is vocoded, all utterances in ONE batched generation (every fold of every utterance is one sequence
of the step kernels) - the reference generates them one after the other.  Without --weights the model has seeded random
weights (noise out; exercises the pipeline).
"""
import argparse
import glob
import os
import sys
import time
import wave

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import estts  # noqa: E402,F401
from estts import configs, weights  # noqa: E402
from estts.audio import wavernn_melspectrogram  # noqa: E402
from estts.wavernn import WaveRNN  # noqa: E402


def save_wav(x, path, sr):
    """dsp.save_wav: float in [-1, 1] -> 16-bit PCM."""
    pcm = (np.clip(np.asarray(x, np.float64), -1, 1) * (2 ** 15 - 1)).astype('<i2')
    with wave.open(path, 'wb') as f:
        f.setnchannels(1)
        f.setsampwidth(2)
        f.setframerate(sr)
        f.writeframes(pcm.tobytes())


def load_wav(path, sr):
    with wave.open(path, 'rb') as f:
        if f.getsampwidth() != 2 or f.getnchannels() != 1 or f.getframerate() != sr:
            raise SystemExit(f'{path}: expected 16-bit mono at {sr} Hz')
        return np.frombuffer(f.readframes(f.getnframes()), dtype='<i2').astype(np.float32) / 32768.0


def main():
    hp = configs.wavernn_hparams()
    ap = argparse.ArgumentParser(description='Generate WaveRNN samples')
    ap.add_argument('--batched', '-b', dest='batched', action='store_true', help='fast batched generation (folds)')
    ap.add_argument('--unbatched', '-u', dest='batched', action='store_false', help='one sequence per utterance')
    ap.add_argument('--target', '-t', type=int, default=hp['voc_target'], help='samples per fold')
    ap.add_argument('--overlap', '-o', type=int, default=hp['voc_overlap'], help='cross-fade samples between folds')
    ap.add_argument('--file', '-f', help='a wav to analyse and re-synthesise')
    ap.add_argument('--mels', help='directory of (num_mels, T) .npy mels')
    ap.add_argument('--weights', '-w', help='checkpoint (.pyt state dict) to load')
    ap.add_argument('--out', default='.', help='output directory')
    ap.set_defaults(batched=hp['voc_gen_batched'])
    args = ap.parse_args()
    if not args.file and not args.mels:
        ap.error('give --file or --mels')
    model = WaveRNN.from_hparams(hp)
    if args.weights:
        model.load(args.weights)
    else:
        model.load_state_dict({k: torch.from_numpy(np.asarray(v)) for k, v in weights.init_wavernn(hp, seed=1).items()})
    os.makedirs(args.out, exist_ok=True)
    k = model.get_step() // 1000
    sr = hp['sample_rate']
    tag = f'gen_batched_target{args.target}_overlap{args.overlap}' if args.batched else 'gen_NOT_BATCHED'
    if args.file:
        wav = load_wav(args.file, sr)
        name = os.path.basename(args.file)
        save_wav(wav, os.path.join(args.out, f'__{name}__{k}k_steps_target.wav'), sr)
        names, mels = [f'__{name}__{k}k_steps_{tag}.wav'], [wavernn_melspectrogram(wav, hp)]
    else:
        files = sorted(glob.glob(os.path.join(args.mels, '*.npy')))
        if not files:
            raise SystemExit(f'no .npy mels under {args.mels}')
        names = [f'{k}k_steps_{os.path.splitext(os.path.basename(f))[0]}_{tag}.wav' for f in files]
        mels = [np.load(f, allow_pickle=False).astype(np.float32) for f in files]
    t0 = time.perf_counter()
    outs = model.generate_batch(mels, args.batched, args.target, args.overlap, hp['mu_law'])
    dt = time.perf_counter() - t0
    for n, y in zip(names, outs):
        save_wav(y, os.path.join(args.out, n), sr)
    audio = sum(len(y) for y in outs) / sr
    print(f'{len(outs)} utterance(s), {audio:.2f} s of audio in {dt:.2f} s (RTF {dt / max(audio, 1e-9):.4f}) -> {args.out}')


if __name__ == '__main__':
    main()
