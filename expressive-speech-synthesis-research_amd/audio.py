"""Mel front-end on the GPU: reference wav -> the mel spectrogram the style encoder, the aligner and the vocoder read.

Mirrors `Audio` of TransformerTTS/utils/audio.py:9-40,87-119 (`mel_spectrogram`, normalisers 'WaveRNN' and 'MelGAN') and
`melspectrogram` of WaveRNN/utility/dsp.py:71-74.  librosa's STFT (centred, reflect padding, periodic Hann window padded to
n_fft) is one float32 matrix product of the overlapping frames with a windowed DFT matrix (estts_gemm_f32, frame t = padded
signal + t * hop_length, so the frames are never copied out), followed by magnitudes, the Slaney mel filterbank as a second
product and the normaliser.  torch is used for device memory and the reflect padding (a copy) only.
"""
import math

import numpy as np
import torch

from . import _lib
from ._lib import check
from .ops import _stream


def _hz_to_mel(f):
    f = np.asarray(f, np.float64)
    return np.where(f >= 1000.0, 15.0 + np.log(np.maximum(f, 1e-10) / 1000.0) / (math.log(6.4) / 27.0), f / (200.0 / 3))


def _mel_to_hz(m):
    m = np.asarray(m, np.float64)
    return np.where(m >= 15.0, 1000.0 * np.exp((math.log(6.4) / 27.0) * (m - 15.0)), (200.0 / 3) * m)


def mel_filterbank(sr, n_fft, n_mels, fmin=0.0, fmax=None):
    """librosa.filters.mel with its defaults (Slaney scale and area normalisation), float64 [n_mels, 1 + n_fft // 2]."""
    fmax = sr / 2.0 if fmax is None else fmax
    fft_f = np.linspace(0, sr / 2.0, 1 + n_fft // 2)
    mel_f = _mel_to_hz(np.linspace(_hz_to_mel(fmin), _hz_to_mel(fmax), n_mels + 2))
    fdiff = np.diff(mel_f)
    ramps = mel_f[:, None] - fft_f[None, :]
    w = np.maximum(0, np.minimum(-ramps[:-2] / fdiff[:-1, None], ramps[2:] / fdiff[1:, None]))
    return w * (2.0 / (mel_f[2:n_mels + 2] - mel_f[:n_mels]))[:, None]


class MelFrontEnd:
    """STFT magnitudes -> mel -> normaliser for one (sample rate, n_fft, hop, window, mel) setting."""

    def __init__(self, sample_rate, n_fft, hop_length, win_length, n_mels, fmin=0.0, fmax=None, device='cuda'):
        if not torch.cuda.is_available():
            raise RuntimeError('the mel front-end runs on the GPU only')
        _lib.load()
        if hop_length % 4 or win_length > n_fft:
            raise ValueError('hop_length must be a multiple of 4 and win_length <= n_fft')
        self.device = torch.device(device)
        self.n_fft, self.hop, self.win, self.n_mels = n_fft, hop_length, win_length, n_mels
        self.nb = 1 + n_fft // 2
        self.nbp = (self.nb + 15) // 16 * 16                      # bins padded to the product's K granularity
        # only the window's support contributes: K = win_length rounded up to 16, frames start lpad samples in
        lpad = (n_fft - win_length) // 2
        self.k0 = lpad // 4 * 4
        self.K = (lpad + win_length - self.k0 + 15) // 16 * 16
        n = np.arange(self.k0, self.k0 + self.K)
        w = np.zeros(self.K)
        inside = (n >= lpad) & (n < lpad + win_length)
        w[inside] = 0.5 - 0.5 * np.cos(2.0 * np.pi * (n[inside] - lpad) / win_length)
        ang = 2.0 * np.pi * np.outer(np.arange(self.nb), n) / n_fft
        dft = np.zeros((2 * self.nbp, self.K), np.float32)
        dft[:self.nb] = (np.cos(ang) * w).astype(np.float32)
        dft[self.nbp:self.nbp + self.nb] = (-np.sin(ang) * w).astype(np.float32)
        self.dft = torch.from_numpy(dft).to(self.device)
        fb = np.zeros((n_mels, self.nbp), np.float32)
        fb[:, :self.nb] = mel_filterbank(sample_rate, n_fft, n_mels, fmin, fmax).astype(np.float32)
        self.fb = torch.from_numpy(fb).to(self.device)

    def mel(self, wav, mode=0, ref_level_db=0.0, min_level_db=-100.0, scale=1.0, shift=0.0):
        """wav: 1-D float array / tensor.  Returns the normalised mel [frames, n_mels] (float32, on the device)."""
        y = torch.as_tensor(wav, dtype=torch.float32).to(self.device).reshape(-1)
        half = self.n_fft // 2
        if y.numel() <= half:
            raise ValueError('signal shorter than n_fft / 2 (reflect padding)')
        frames = 1 + y.numel() // self.hop
        # reflect padding (a copy), plus slack so that the last frame's K window stays inside the buffer
        yp = torch.zeros(2 * half + y.numel() + self.K, dtype=torch.float32, device=self.device)
        yp[half:half + y.numel()] = y
        yp[:half] = y[1:half + 1].flip(0)
        yp[half + y.numel():2 * half + y.numel()] = y[-half - 1:-1].flip(0)
        lib, st = _lib.load(), _stream()
        spec = torch.empty(frames, 2 * self.nbp, dtype=torch.float32, device=self.device)
        check(lib.estts_gemm_f32(yp.data_ptr() + 4 * self.k0, self.hop, self.dft.data_ptr(), self.K, spec.data_ptr(), 2 * self.nbp,
                                 frames, 2 * self.nbp, self.K, st), 'estts_gemm_f32')
        mag = torch.empty(frames, self.nbp, dtype=torch.float32, device=self.device)
        check(lib.estts_spec_magnitude(spec.data_ptr(), 2 * self.nbp, self.nbp, mag.data_ptr(), self.nbp, frames, st),
              'estts_spec_magnitude')
        melp = torch.empty(frames, self.n_mels, dtype=torch.float32, device=self.device)
        check(lib.estts_gemm_f32(mag.data_ptr(), self.nbp, self.fb.data_ptr(), self.nbp, melp.data_ptr(), self.n_mels, frames,
                                 self.n_mels, self.nbp, st), 'estts_gemm_f32')
        out = torch.empty_like(melp)
        check(lib.estts_mel_normalize(melp.data_ptr(), self.n_mels, out.data_ptr(), self.n_mels, frames, self.n_mels, mode,
                                      ref_level_db, min_level_db, scale, shift, st), 'estts_mel_normalize')
        self._keep = (yp, spec, mag, melp)
        return out


class Audio:
    """TransformerTTS/utils/audio.py:9-40: `Audio(config).mel_spectrogram(wav)` -> (mel_channels, frames), normalised by
    config['normalizer'] ('WaveRNN': [-4, 4] from a -100 dB floor; 'MelGAN': log of the clipped mel)."""

    def __init__(self, config, device='cuda'):
        self.config = config
        if config['normalizer'] not in ('WaveRNN', 'MelGAN'):
            raise ValueError(f"unknown normalizer {config['normalizer']}")
        self.front = MelFrontEnd(config['sampling_rate'], config['n_fft'], config['hop_length'], config['win_length'],
                                 config['mel_channels'], config['f_min'], config.get('f_max'), device)

    def mel_spectrogram(self, wav):
        if self.config['normalizer'] == 'MelGAN':
            m = self.front.mel(wav, mode=1)
        else:
            m = self.front.mel(wav, mode=0, ref_level_db=0.0, min_level_db=-100.0, scale=8.0, shift=-4.0)
        return m.t()


def wavernn_melspectrogram(wav, hp, device='cuda', _cache={}):
    """dsp.melspectrogram (WaveRNN/utility/dsp.py:71-74): (num_mels, frames) in [0, 1]; hp: the hparams dict."""
    key = (hp['sample_rate'], hp['n_fft'], hp['hop_length'], hp['win_length'], hp['num_mels'], hp['fmin'], str(device))
    if key not in _cache:
        _cache[key] = MelFrontEnd(hp['sample_rate'], hp['n_fft'], hp['hop_length'], hp['win_length'], hp['num_mels'], hp['fmin'],
                                  None, device)
    return _cache[key].mel(wav, mode=0, ref_level_db=0.0, min_level_db=float(hp['min_level_db']), scale=1.0, shift=0.0).t()
