#!/usr/bin/env python3
"""End-to-end synthesis on one MI355X: phonemes -> TransformerTTS mel -> WaveRNN waveform, the path of the reference's
synthesize_sentences_speaker.py (model.predict, the (mel.T + 4) / 8 hand-off, Synthesize.generate) without leaving the GPU
between the two models.

  python scripts/synthesize_sentences.py --config /path/to/config/wavernn --model forward --text "h@loU w3:ld" \
      [--tts-weights ckpt-5.npz] [--vocoder-weights latest_weights.pyt] [--ref-mel ref.npy | --ref-wav ref.wav]
      [--spk-embed speaker.npy] [--system-type style_text] --out out.npy

The autoregressive model takes what its system_type asks for, as in the reference (synthesize_sentences_speaker.py:60-67): a
reference mel or wav for the style encoder ('style' systems) and a 256-d speaker embedding ('speaker' systems).

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
from estts.synthesizer_wavernn import Synthesize  # noqa: E402


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
    ap.add_argument('--spk-embed', help='.npy speaker embedding (speaker_embed_dim values) for the speaker systems of the '
                                        'autoregressive model (synthesize_sentences_speaker.py:65-67)')
    ap.add_argument('--system-type', choices=['text', 'style_text', 'speaker_text', 'speaker_style_text'],
                    help='override the yaml config\'s system_type (autoregressive model)')
    ap.add_argument('--max-length', type=int, default=1000, help='most mel frames the autoregressive model may decode (predict\'s max_length)')
    ap.add_argument('--out', default='out.npy')
    args = ap.parse_args()
    cm = ConfigManager(args.config, args.model)
    if args.system_type:
        cm.config['system_type'] = args.system_type
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
        system = cm.config['system_type']
        if 'style' in system and ref is None:
            raise SystemExit(f"system_type '{system}' needs --ref-mel or --ref-wav")
        spk = None
        if 'speaker' in system:
            if not args.spk_embed:
                raise SystemExit(f"system_type '{system}' needs --spk-embed")
            spk = np.load(args.spk_embed).astype(np.float32).reshape(1, -1)
            n_spk = cm.model_config_dict()['speaker_embed_dim']
            if spk.shape[1] != n_spk:
                raise SystemExit(f'--spk-embed has {spk.shape[1]} values, the model expects {n_spk}')
        out = model.predict(args.text, ref if 'style' in system else None, spk, max_length=args.max_length, verbose=False)
    mel = out['mel']
    hp = configs.wavernn_hparams()
    vocoder = Synthesize(args.vocoder_weights, hp)             # synthesize_sentences_speaker.py:49
    mel_voc = ((mel.t() + 4.) / 8.)[None]                      # synthesize_sentences_speaker.py:69-70, stays on the device
    wav = vocoder.generate(mel_voc)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    np.save(args.out, wav)
    print(f'{mel.shape[0]} frames, {len(wav)} samples ({len(wav) / hp["sample_rate"]:.2f} s) in {dt:.2f} s -> {args.out}')


if __name__ == '__main__':
    main()
