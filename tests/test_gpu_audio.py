"""Mel front-end on the GPU (estts.audio, float32 matrix-core STFT + mel) against the float64 oracle (oracle/audio_numpy.py)
on a seeded speech-like signal.  Tolerance: 2e-3 on the [-4, 4] normalised mel (0.025 dB), 1e-4 on the [0, 1] vocoder mel;
the oracle itself is unpinned against librosa (see its header)."""
import numpy as np
import pytest
import torch

import estts  # noqa: F401
from oracle import audio_numpy as A

pytestmark = pytest.mark.gpu


def speechlike(n, sr, seed):
    rng = np.random.default_rng(seed)
    t = np.arange(n) / sr
    f0 = 120 + 30 * np.sin(2 * np.pi * 0.7 * t)
    y = sum(a * np.sin(2 * np.pi * k * np.cumsum(f0) / sr) for k, a in ((1, 0.3), (2, 0.2), (3, 0.1), (7, 0.05), (15, 0.02)))
    y = y * (0.5 + 0.5 * np.sin(2 * np.pi * 3 * t)) + 0.003 * rng.normal(size=n)
    y[: n // 10] *= 0.01                      # a near-silent lead-in: 60 dB of dynamic range inside one signal
    return y.astype(np.float32)


@pytest.mark.parametrize('sr,n_fft,hop,win,fmin,normalizer', [(16000, 2048, 200, 800, 40, 'WaveRNN'), (22050, 1024, 256, 1024, 0, 'MelGAN'),
                                                              (16000, 2048, 200, 800, 40, 'MelGAN')])
def test_tts_mel_spectrogram(sr, n_fft, hop, win, fmin, normalizer):
    from estts.audio import Audio
    cfg = dict(sampling_rate=sr, n_fft=n_fft, hop_length=hop, win_length=win, mel_channels=80, f_min=fmin, f_max=None,
               normalizer=normalizer)
    y = speechlike(sr + 137, sr, seed=hop)
    ref = A.tts_mel_spectrogram(y.astype(np.float64), cfg)
    out = Audio(cfg).mel_spectrogram(y)
    assert tuple(out.shape) == ref.shape == (80, 1 + len(y) // hop)
    err = np.abs(out.cpu().numpy() - ref)
    assert err.max() < (2e-3 if normalizer == 'WaveRNN' else 1e-3), err.max()
    assert err.mean() < 1e-4


def test_wavernn_melspectrogram_and_vocoder_handoff():
    from estts import configs
    from estts.audio import wavernn_melspectrogram
    hp = configs.wavernn_hparams()
    hp2 = dict(sample_rate=hp['sample_rate'], n_fft=2048, hop_length=hp['hop_length'], win_length=800, num_mels=80, fmin=40,
               min_level_db=-100)
    y = speechlike(12345, 16000, seed=2)
    ref = A.wavernn_melspectrogram(y.astype(np.float64), hp2)
    out = wavernn_melspectrogram(y, hp2)
    assert tuple(out.shape) == ref.shape
    assert np.abs(out.cpu().numpy() - ref).max() < 2e-4
    assert out.min().item() >= 0.0 and out.max().item() <= 1.0
    assert (ref == 1).any()                      # the clip of the normaliser is exercised


def test_gemm_f32_plain_product_and_argument_checks():
    from estts import _lib
    lib = _lib.load()
    rng = np.random.default_rng(0)
    M, N, K = 70, 130, 48
    a, w = rng.normal(size=(M, K)).astype(np.float32), rng.normal(size=(N, K)).astype(np.float32)
    ad, wd = torch.from_numpy(a).cuda(), torch.from_numpy(w).cuda()
    c = torch.full((M, N + 3), float('nan'), device='cuda')
    st = torch.cuda.current_stream().cuda_stream
    assert lib.estts_gemm_f32(ad.data_ptr(), K, wd.data_ptr(), K, c.data_ptr(), N + 3, M, N, K, st) == 0
    ref = a.astype(np.float64) @ w.astype(np.float64).T
    assert np.abs(c[:, :N].cpu().numpy() - ref).max() < 2e-5
    assert torch.isnan(c[:, N:]).all()
    assert lib.estts_gemm_f32(ad.data_ptr(), K, wd.data_ptr(), K, c.data_ptr(), N + 3, M, N, 40, st) != 0      # K % 16


def test_reference_wav_to_style_conditioned_synthesis(tmp_path, monkeypatch):
    """scripts/synthesize_sentences.py --model autoregressive --ref-wav: wav file -> mel front-end on the GPU -> GST reference
    encoder -> incremental AR decoding -> WaveRNN, on seeded random weights (plumbing: shapes, finiteness)."""
    import importlib.util
    import os
    import sys
    import wave
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    y = speechlike(16000, 16000, seed=9)
    path = tmp_path / 'ref.wav'
    with wave.open(str(path), 'wb') as f:
        f.setnchannels(1); f.setsampwidth(2); f.setframerate(16000)
        f.writeframes((np.clip(y, -1, 1) * 32767).astype('<i2').tobytes())
    spec = importlib.util.spec_from_file_location('synth_script', os.path.join(root, 'scripts', 'synthesize_sentences.py'))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    out = tmp_path / 'wav.npy'
    np.save(tmp_path / 'spk.npy', np.random.default_rng(3).normal(size=256).astype(np.float32))
    monkeypatch.setattr(sys, 'argv', ['synthesize_sentences.py', '--config', os.path.join(root, 'config', 'wavernn'), '--model',
                                      'autoregressive', '--text', 'D@ kwIk braUn fQks', '--ref-wav', str(path), '--system-type', 'speaker_style_text',
                                      '--spk-embed', str(tmp_path / 'spk.npy'), '--out', str(out)])
    # seeded random weights may emit the stop token after a few frames, fewer than the vocoder's 20-frame fade-out needs:
    # decode a fixed 40 frames instead
    from estts.autoregressive_transformer import AutoregressiveTransformer
    predict = AutoregressiveTransformer.predict

    def predict_without_stop(self, *a, **kw):
        self.stop_prob_index = 99
        return predict(self, *a, **kw)
    monkeypatch.setattr(AutoregressiveTransformer, 'predict', predict_without_stop)
    sys.argv += ['--max-length', '40']
    mod.main()
    wav = np.load(out)
    assert wav.ndim == 1 and len(wav) > 1000 and np.isfinite(wav).all() and np.abs(wav).max() <= 1.0 + 1e-6
