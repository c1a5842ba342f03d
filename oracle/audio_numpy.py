"""CPU oracle for the mel front-ends (float64 NumPy).

TEST INFRASTRUCTURE ONLY (tests/, __graft_entry__.smoke(), bench.py's cpu_baseline leg).

Restates
  * TransformerTTS/utils/audio.py:9-40,87-119   Audio.mel_spectrogram with the WaveRNN and MelGAN normalisers
  * WaveRNN/utility/dsp.py:40-80                 melspectrogram / spectrogram of the vocoder's data preparation
  * gst_tacotron/util/audio.py:19-148            pre-emphasis, spectrogram, melspectrogram, Griffin-Lim inversion,
                                                 find_endpoint (BASELINE config 1: CPU plumbing, timed only)
All three call librosa (third-party dependency, not importable here: `librosa>=0.7` in TransformerTTS/requirements.txt,
`librosa==0.6.x` era for the other two).  The published algorithms restated below:
  * librosa.stft(y, n_fft, hop_length, win_length): center=True with reflect padding of n_fft // 2, periodic Hann window of
    win_length samples zero-padded on both sides to n_fft, frame t starts at t * hop_length of the padded signal,
    1 + n_fft // 2 bins, shape (bins, frames)
  * librosa.istft: windowed overlap-add divided by the summed squared window, n_fft // 2 trimmed from both ends
  * librosa.filters.mel(sr, n_fft, n_mels, fmin, fmax=sr/2, htk=False, norm=1 / 'slaney'): triangular filters equally
    spaced on the Slaney mel scale (linear below 1 kHz, 200/3 Hz per mel; logarithmic above, step ln(6.4)/27), each
    divided by half its bandwidth in Hz
  * librosa.feature.melspectrogram(S=...) applies the filterbank to S as given (no squaring when S is passed)
PARITY STATUS: unpinned against librosa itself (not installed, no fixtures of it in the reference); pinned only by
closed-form properties in tests/test_oracle_audio.py (DFT definition, tone location, filter areas, inversion round trip).
"""
import numpy as np


# ------------------------------------------------------------------------------------------------ librosa pieces
def hann_periodic(n):
    """scipy.signal.get_window('hann', n, fftbins=True)."""
    return 0.5 - 0.5 * np.cos(2.0 * np.pi * np.arange(n) / n)


def padded_window(n_fft, win_length):
    """librosa.util.pad_center(window, n_fft)."""
    w = np.zeros(n_fft)
    lpad = (n_fft - win_length) // 2
    w[lpad:lpad + win_length] = hann_periodic(win_length)
    return w


def stft(y, n_fft, hop_length, win_length):
    """librosa.stft with its defaults (center, reflect padding, Hann).  Returns complex (1 + n_fft // 2, frames)."""
    y = np.asarray(y, np.float64)
    yp = np.pad(y, n_fft // 2, mode='reflect')
    n_frames = 1 + (len(yp) - n_fft) // hop_length
    idx = np.arange(n_fft)[None, :] + hop_length * np.arange(n_frames)[:, None]
    frames = yp[idx] * padded_window(n_fft, win_length)[None, :]
    return np.fft.rfft(frames, axis=1).T


def istft(D, hop_length, win_length):
    """librosa.istft with its defaults (center): overlap-add of windowed inverse FFTs over the summed squared window."""
    n_fft = 2 * (D.shape[0] - 1)
    n_frames = D.shape[1]
    w = padded_window(n_fft, win_length)
    frames = np.fft.irfft(D.T, n=n_fft, axis=1) * w[None, :]
    L = n_fft + hop_length * (n_frames - 1)
    y = np.zeros(L)
    wss = np.zeros(L)
    for t in range(n_frames):
        y[t * hop_length:t * hop_length + n_fft] += frames[t]
        wss[t * hop_length:t * hop_length + n_fft] += w * w
    nz = wss > np.finfo(np.float32).tiny
    y[nz] /= wss[nz]
    return y[n_fft // 2:L - n_fft // 2]


def hz_to_mel(f):
    f = np.asarray(f, np.float64)
    f_sp = 200.0 / 3
    mels = f / f_sp
    min_log_hz = 1000.0
    min_log_mel = min_log_hz / f_sp
    logstep = np.log(6.4) / 27.0
    return np.where(f >= min_log_hz, min_log_mel + np.log(np.maximum(f, 1e-10) / min_log_hz) / logstep, mels)


def mel_to_hz(m):
    m = np.asarray(m, np.float64)
    f_sp = 200.0 / 3
    min_log_hz = 1000.0
    min_log_mel = min_log_hz / f_sp
    logstep = np.log(6.4) / 27.0
    return np.where(m >= min_log_mel, min_log_hz * np.exp(logstep * (m - min_log_mel)), f_sp * m)


def mel_filterbank(sr, n_fft, n_mels=128, fmin=0.0, fmax=None):
    """librosa.filters.mel (Slaney scale, Slaney area normalisation).  Returns (n_mels, 1 + n_fft // 2)."""
    fmax = sr / 2.0 if fmax is None else fmax
    fft_f = np.linspace(0, sr / 2.0, 1 + n_fft // 2)
    mel_f = mel_to_hz(np.linspace(hz_to_mel(fmin), hz_to_mel(fmax), n_mels + 2))
    fdiff = np.diff(mel_f)
    ramps = mel_f[:, None] - fft_f[None, :]
    lower = -ramps[:-2] / fdiff[:-1, None]
    upper = ramps[2:] / fdiff[1:, None]
    weights = np.maximum(0, np.minimum(lower, upper))
    return weights * (2.0 / (mel_f[2:n_mels + 2] - mel_f[:n_mels]))[:, None]


def amp_to_db(x):
    return 20.0 * np.log10(np.maximum(1e-5, x))


def db_to_amp(x):
    return np.power(10.0, x * 0.05)


# ------------------------------------------------------------------------------------------------ TransformerTTS
def tts_mel_spectrogram(wav, cfg):
    """Audio.mel_spectrogram (TransformerTTS/utils/audio.py:36-40) -> (mel_channels, frames); cfg: the data_config keys
    sampling_rate, n_fft, hop_length, win_length, mel_channels, f_min, f_max, normalizer ('WaveRNN' | 'MelGAN')."""
    S = np.abs(stft(wav, cfg['n_fft'], cfg['hop_length'], cfg['win_length']))
    M = mel_filterbank(cfg['sampling_rate'], cfg['n_fft'], cfg['mel_channels'], cfg['f_min'], cfg.get('f_max')) @ S
    if cfg['normalizer'] == 'MelGAN':                       # audio.py:87-97
        return np.log(np.clip(M, 1e-5, None))
    if cfg['normalizer'] == 'WaveRNN':                      # audio.py:99-119: min_level_db -100, max_norm 4
        return np.clip((amp_to_db(M) + 100.0) / 100.0, 0, 1) * 8.0 - 4.0
    raise ValueError(cfg['normalizer'])


# ------------------------------------------------------------------------------------------------ WaveRNN
def wavernn_melspectrogram(y, hp):
    """dsp.melspectrogram (WaveRNN/utility/dsp.py:71-74) -> (num_mels, frames) in [0, 1]."""
    S = np.abs(stft(y, hp['n_fft'], hp['hop_length'], hp['win_length']))
    M = mel_filterbank(hp['sample_rate'], hp['n_fft'], hp['num_mels'], hp['fmin']) @ S
    return np.clip((amp_to_db(M) - hp['min_level_db']) / -hp['min_level_db'], 0, 1)


# ------------------------------------------------------------------------------------------------ gst_tacotron
GST_HPARAMS = dict(num_mels=80, num_freq=1025, sample_rate=16000, frame_length_ms=50, frame_shift_ms=12.5, preemphasis=0.97,
                   min_level_db=-100, ref_level_db=20, griffin_lim_iters=60, power=1.5)   # gst_tacotron/hparams.py:11-42


def _gst_stft_parameters(hp):
    n_fft = (hp['num_freq'] - 1) * 2
    return n_fft, int(hp['frame_shift_ms'] / 1000 * hp['sample_rate']), int(hp['frame_length_ms'] / 1000 * hp['sample_rate'])


def preemphasis(x, k):
    """scipy.signal.lfilter([1, -k], [1], x)."""
    x = np.asarray(x, np.float64)
    y = x.copy()
    y[1:] -= k * x[:-1]
    return y


def inv_preemphasis(x, k):
    """scipy.signal.lfilter([1], [1, -k], x)."""
    y = np.empty(len(x))
    acc = 0.0
    for i, v in enumerate(x):
        acc = v + k * acc
        y[i] = acc
    return y


def gst_spectrogram(y, hp=GST_HPARAMS):
    """util/audio.py:27-30 -> (num_freq, frames) in [0, 1]."""
    n_fft, hop, win = _gst_stft_parameters(hp)
    S = amp_to_db(np.abs(stft(preemphasis(y, hp['preemphasis']), n_fft, hop, win))) - hp['ref_level_db']
    return np.clip((S - hp['min_level_db']) / -hp['min_level_db'], 0, 1)


def gst_melspectrogram(y, hp=GST_HPARAMS):
    """util/audio.py:50-53 -> (num_mels, frames) in [0, 1]."""
    n_fft, hop, win = _gst_stft_parameters(hp)
    D = np.abs(stft(preemphasis(y, hp['preemphasis']), n_fft, hop, win))
    S = amp_to_db(mel_filterbank(hp['sample_rate'], n_fft, hp['num_mels']) @ D) - hp['ref_level_db']
    return np.clip((S - hp['min_level_db']) / -hp['min_level_db'], 0, 1)


def griffin_lim(S, hp=GST_HPARAMS, rng=None):
    """util/audio.py:65-75: random initial phases, then griffin_lim_iters rounds of STFT phase re-estimation."""
    n_fft, hop, win = _gst_stft_parameters(hp)
    rng = np.random.default_rng(0) if rng is None else rng
    angles = np.exp(2j * np.pi * rng.random(S.shape))
    Sc = np.abs(S).astype(np.complex128)
    y = istft(Sc * angles, hop, win)
    for _ in range(hp['griffin_lim_iters']):
        angles = np.exp(1j * np.angle(stft(y, n_fft, hop, win)))
        y = istft(Sc * angles, hop, win)
    return y


def gst_inv_spectrogram(spec, hp=GST_HPARAMS, rng=None):
    """util/audio.py:33-36: denormalise, dB -> amplitude, ** power, Griffin-Lim, inverse pre-emphasis."""
    S = db_to_amp(np.clip(spec, 0, 1) * -hp['min_level_db'] + hp['min_level_db'] + hp['ref_level_db'])
    return inv_preemphasis(griffin_lim(S ** hp['power'], hp, rng), hp['preemphasis'])


def find_endpoint(wav, sample_rate=16000, threshold_db=-40, min_silence_sec=0.8):
    """util/audio.py:56-63."""
    window_length = int(sample_rate * min_silence_sec)
    hop_length = int(window_length / 4)
    threshold = db_to_amp(threshold_db)
    for x in range(hop_length, len(wav) - window_length, hop_length):
        if np.max(wav[x:x + window_length]) < threshold:
            return x + hop_length
    return len(wav)
