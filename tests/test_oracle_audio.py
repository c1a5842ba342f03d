"""The mel front-end oracle (oracle/audio_numpy.py) against closed-form properties.  librosa, whose algorithms it restates,
is not installable here, so these pins are analytic: the DFT definition, where a pure tone lands, the area of the Slaney
filters, the scale's fixed points, and that Griffin-Lim inversion reproduces the magnitudes it was given."""
import numpy as np

from oracle import audio_numpy as A

TTS_CFG = dict(sampling_rate=16000, n_fft=2048, hop_length=200, win_length=800, mel_channels=80, f_min=40, f_max=None,
               normalizer='WaveRNN')     # TransformerTTS/config/wavernn/data_config.yaml:37-44


def test_stft_is_the_windowed_dft_of_reflect_padded_frames():
    rng = np.random.default_rng(0)
    y = rng.normal(size=1000)
    n_fft, hop, win = 64, 16, 40
    D = A.stft(y, n_fft, hop, win)
    assert D.shape == (33, 1 + 1000 // hop)
    yp = np.concatenate([y[n_fft // 2:0:-1], y, y[-2:-n_fft // 2 - 2:-1]])
    w = np.zeros(n_fft)
    w[12:52] = 0.5 - 0.5 * np.cos(2 * np.pi * np.arange(40) / 40)
    for t in (0, 5, D.shape[1] - 1):
        seg = yp[t * hop:t * hop + n_fft] * w
        for f in (0, 3, 32):
            want = np.sum(seg * np.exp(-2j * np.pi * f * np.arange(n_fft) / n_fft))
            assert abs(D[f, t] - want) < 1e-9


def test_istft_inverts_stft():
    rng = np.random.default_rng(1)
    y = rng.normal(size=3000)
    back = A.istft(A.stft(y, 256, 64, 200), 64, 200)
    assert len(back) == 3000 - 3000 % 64 or len(back) == (3000 // 64) * 64
    n = len(back)
    assert np.abs(back[300:n - 300] - y[300:n - 300]).max() < 1e-9


def test_mel_scale_and_filterbank():
    assert abs(A.hz_to_mel(1000.0) - 15.0) < 1e-12 and abs(A.mel_to_hz(15.0) - 1000.0) < 1e-9
    assert abs(A.hz_to_mel(6400.0) - 42.0) < 1e-9                      # 27 log-steps above 1 kHz
    fb = A.mel_filterbank(16000, 2048, 80, fmin=40)
    assert fb.shape == (80, 1025) and (fb >= 0).all()
    # Slaney normalisation: every triangle has unit area in Hz (sampled at 7.8125 Hz per bin)
    area = fb.sum(axis=1) * (8000.0 / 1024)
    assert np.abs(area[5:] - 1.0).max() < 0.02
    peaks = fb.argmax(axis=1)
    assert (np.diff(peaks) > 0).all()
    assert fb[:, :int(40 / 7.8125)].sum() == 0.0


def test_tone_lands_in_the_right_mel_band_and_normalisers_map_ranges():
    sr = 16000
    t = np.arange(sr) / sr
    y = 0.5 * np.sin(2 * np.pi * 1000.0 * t)
    mel = A.tts_mel_spectrogram(y, TTS_CFG)
    assert mel.shape == (80, 1 + sr // 200)
    fb = A.mel_filterbank(sr, 2048, 80, 40)
    want = fb[:, round(1000 / (sr / 2048))].argmax()
    assert abs(int(mel[:, 40].argmax()) - int(want)) <= 1
    assert mel.min() >= -4.0 and mel.max() <= 4.0
    mg = A.tts_mel_spectrogram(y, dict(TTS_CFG, normalizer='MelGAN'))
    assert mg.min() >= np.log(1e-5) - 1e-12
    hp = dict(sample_rate=16000, n_fft=2048, hop_length=200, win_length=800, num_mels=80, fmin=40, min_level_db=-100)
    wr = A.wavernn_melspectrogram(y, hp)
    assert wr.min() >= 0.0 and wr.max() <= 1.0
    assert np.abs((wr * 8 - 4) - mel).max() < 1e-9          # the TTS-side WaveRNN normaliser is the vocoder's, rescaled


def test_gst_tacotron_spectrogram_inversion_round_trip():
    rng = np.random.default_rng(2)
    sr = 16000
    t = np.arange(sr // 2) / sr
    y = 0.3 * np.sin(2 * np.pi * 220 * t) + 0.2 * np.sin(2 * np.pi * 1330 * t) * np.hanning(len(t)) + 0.001 * rng.normal(size=len(t))
    hp = dict(A.GST_HPARAMS, griffin_lim_iters=30, power=1.0)
    spec = A.gst_spectrogram(y, hp)
    assert spec.shape[0] == 1025 and spec.min() >= 0 and spec.max() <= 1
    mel = A.gst_melspectrogram(y, hp)
    assert mel.shape == (80, spec.shape[1])
    wav = A.gst_inv_spectrogram(spec, hp, rng=np.random.default_rng(3))
    again = A.gst_spectrogram(wav, hp)
    n = min(spec.shape[1], again.shape[1])
    loud = spec[:, 2:n - 2] > 0.5
    assert np.abs(again[:, 2:n - 2] - spec[:, 2:n - 2])[loud].mean() < 0.03      # Griffin-Lim reproduces the magnitudes
    x = np.arange(10, dtype=float)
    assert np.allclose(A.inv_preemphasis(A.preemphasis(x, 0.97), 0.97), x)
    quiet = np.concatenate([0.5 * np.ones(8000), np.zeros(40000)])
    assert A.find_endpoint(quiet) < 8000 + 2 * 3200 + 1
