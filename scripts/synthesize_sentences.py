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


def synthesize_file(args, model):
    """The reference's loop over sentences (synthesize_sentences_speaker.py:56-75) with the sentences overlapped: the forward
    model keeps --in-flight of them on the GPU at once, and WaveRNN takes every fold of every utterance as one batch."""
    if args.model != 'forward':
        raise SystemExit('--text-file runs the forward model (the autoregressive model synthesises one sentence per call)')
    with open(args.text_file) as f:
        texts = [line.strip() for line in f if line.strip()]
    hp = configs.wavernn_hparams()
    vocoder = Synthesize(args.vocoder_weights, hp)
    model.predict_many(texts[:args.in_flight], in_flight=args.in_flight)          # library load, workspaces, first-visit shapes
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    outs = model.predict_many(texts, in_flight=args.in_flight)
    torch.cuda.synchronize()
    t_tts = time.perf_counter() - t0
    frames = sum(int(o['mel'].shape[0]) for o in outs)
    keep = [i for i, o in enumerate(outs) if o['mel'].shape[0] >= 21]            # (WaveRNN's closing fade covers 20 hops)
    t0 = time.perf_counter()
    wavs = vocoder.generate_many([((outs[i]['mel'].t() + 4.) / 8.) for i in keep]) if keep else []
    torch.cuda.synchronize()
    t_voc = time.perf_counter() - t0
    stem = args.out[:-4] if args.out.endswith('.npy') else args.out
    for i, wav in zip(keep, wavs):
        np.save(f'{stem}_{i}.npy', wav)
    audio = sum(len(w) for w in wavs) / hp['sample_rate']
    print(f'{len(texts)} sentences, {args.in_flight} in flight: {frames} mel frames in {t_tts * 1e3:.1f} ms '
          f'({frames / t_tts:.0f} mel-frames/s, {t_tts / len(texts) * 1e3:.3f} ms per sentence); '
          f'{len(wavs)} utterances vocoded in one batch: {audio:.2f} s of audio in {t_voc:.2f} s -> {stem}_<i>.npy')


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--config', required=True)
    ap.add_argument('--model', default='forward', choices=['forward', 'autoregressive'])
    ap.add_argument('--text')
    ap.add_argument('--text-file', help='one sentence (phonemes) per line: all of them are synthesised, --in-flight at a time by the '
                                        'forward model, and vocoded in one batched WaveRNN generation; outputs <out stem>_<i>.npy')
    ap.add_argument('--in-flight', type=int, default=4, help='sentences the forward model keeps on the GPU at once with --text-file '
                                                             '(ForwardTransformer.predict_many; 1 = one at a time)')
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
    if not args.text and not args.text_file:
        ap.error('give --text or --text-file')
    cm = ConfigManager(args.config, args.model)
    if args.system_type:
        cm.config['system_type'] = args.system_type
    model = cm.load_model(args.tts_weights, verbose=True)
    if args.text_file:
        return synthesize_file(args, model)
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
