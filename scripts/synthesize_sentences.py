#!/usr/bin/env python3
"""End-to-end synthesis on one MI355X: phonemes -> TransformerTTS mel -> WaveRNN waveform, the path of the reference's
synthesize_sentences_speaker.py (model.predict, the (mel.T + 4) / 8 hand-off, Synthesize.generate) without leaving the GPU
between the two models.

  python scripts/synthesize_sentences.py --config /path/to/config/wavernn --model forward --text "h@loU w3:ld" \
      [--tts-weights ckpt-5.npz] [--vocoder-weights latest_weights.pyt] [--ref-mel ref.npy | --ref-wav ref.wav] --out out.npy

Text must already be phonemes (the phonemizer is an external program).  Without weight files both models run on seeded
random weights, which exercises the pipeline but produces noise.
"""
import argparse
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import estts  # noqa: E402,F401
from estts import configs, weights  # noqa: E402
from estts.config_manager import ConfigManager  # noqa: E402
from estts.wavernn import WaveRNN  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--config', required=True)
    ap.add_argument('--model', default='forward', choices=['forward', 'autoregressive'])
    ap.add_argument('--text', required=True)
    ap.add_argument('--tts-weights')
    ap.add_argument('--vocoder-weights')
    ap.add_argument('--ref-mel', help='.npy reference mel (T, 80) for the style encoder (autoregressive model)')
    ap.add_argument('--ref-wav', help='16-bit PCM mono .wav reference at the data config\'s sampling rate: its mel is computed on the GPU '
                                      '(estts.audio.Audio, the reference\'s Audio.mel_spectrogram)')
    ap.add_argument('--out', default='out.npy')
    args = ap.parse_args()
    cm = ConfigManager(args.config, args.model)
    model = cm.load_model(args.tts_weights, verbose=True)
    t0 = time.perf_counter()
    if args.model == 'forward':
        out = model.predict(args.text)
    else:
        ref = np.load(args.ref_mel) if args.ref_mel else None
        if args.ref_wav:
            import wave
            from estts.audio import Audio
            with wave.open(args.ref_wav, 'rb') as f:
                if f.getsampwidth() != 2 or f.getnchannels() != 1 or f.getframerate() != cm.config['sampling_rate']:
                    raise SystemExit(f'--ref-wav must be 16-bit mono at {cm.config["sampling_rate"]} Hz')
                pcm = np.frombuffer(f.readframes(f.getnframes()), dtype='<i2').astype(np.float32) / 32768.0
            ref = Audio(cm.config).mel_spectrogram(pcm).t().contiguous()           # (T, 80), stays on the device
        out = model.predict(args.text, ref, None, verbose=False)
    mel = out['mel']
    hp = configs.wavernn_hparams()
    voc = WaveRNN.from_hparams(hp)
    if args.vocoder_weights:
        voc.load(args.vocoder_weights)
    else:
        voc.load_state_dict({k: torch.from_numpy(np.asarray(v)) for k, v in weights.init_wavernn(hp, seed=1).items()})
    mel_voc = ((mel.t() + 4.) / 8.)[None]                      # synthesize_sentences_speaker.py:69-70, stays on the device
    wav = voc.generate(mel_voc, True, hp['voc_target'], hp['voc_overlap'], hp['mu_law'])
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    np.save(args.out, wav)
    print(f'{mel.shape[0]} frames, {len(wav)} samples ({len(wav) / hp["sample_rate"]:.2f} s) in {dt:.2f} s -> {args.out}')


if __name__ == '__main__':
    main()
