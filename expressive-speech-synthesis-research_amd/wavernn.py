"""WaveRNN vocoder on MI355X.

Mirrors the surface of the reference class (WaveRNN/models/fatchord_version.py:89-415): constructor
arguments, state_dict key names (so existing `.pyt` checkpoints load), `forward`, `generate`,
`restore / load / save / checkpoint / get_step`.  Differences that the hardware path needs:
`generate` also accepts several utterances at once (`generate_batch`), takes an explicit `seed` (the
reference uses torch's global RNG) and optional injected uniform numbers / forced samples for replaying a
recorded trajectory.  All arithmetic on the device runs in libestts.so kernels; the cross-fade, mu-law
expansion and fade-out run on the host in float64 NumPy exactly as the reference does
(fatchord_version.py:222-237).
"""
import math
import os

import numpy as np
import torch

from . import _lib, ops
from ._lib import WavernnCondArgs, WavernnLayerArgs, WavernnPrepArgs, WavernnSampleArgs, check
from .ops import PackedLinear


_IMPULSE_FRAME = 6      # impulse position (frames) of the response below; margin of 6 frames on both sides


def _composite_response(up_weights, scales):
    """Response of the [repeat x s, Conv2d(1, 1, (1, 2 s + 1))] x 3 chain (fatchord_version.py:70-85) to a unit impulse at
    frame _IMPULSE_FRAME, float64, length (2 * _IMPULSE_FRAME + 1) * hop."""
    x = np.zeros(2 * _IMPULSE_FRAME + 1, dtype=np.float64)
    x[_IMPULSE_FRAME] = 1.0
    for s, k in zip(scales, up_weights):
        k = np.asarray(k, dtype=np.float64).reshape(-1)
        x = np.repeat(x, s)
        xp = np.pad(x, (s, s))
        x = sum(k[t] * xp[t:t + x.shape[0]] for t in range(2 * s + 1))
    return x


def _response_to_fir(x, hop, kmin, ntaps):
    fir = np.zeros((hop, ntaps), dtype=np.float64)
    d = np.arange(hop)[:, None] + (kmin + np.arange(ntaps))[None, :] * hop + _IMPULSE_FRAME * hop
    ok = (d >= 0) & (d < x.shape[0])
    fir[ok] = x[d[ok]]
    return fir


def composite_filter_support(scales):
    """(kmin, ntaps) of the composite filter for kernels without zeros (depends on the kernel sizes only)."""
    hop = int(np.prod(scales))
    x = _composite_response([np.ones(2 * s + 1) for s in scales], scales)
    nz = np.nonzero(x > 0)[0]
    lo, hi = int(nz.min()) - _IMPULSE_FRAME * hop, int(nz.max()) - _IMPULSE_FRAME * hop
    kmin = math.ceil((lo - hop + 1) / hop)
    return kmin, math.floor(hi / hop) - kmin + 1


def composite_upsample_filter(up_weights, scales):
    """Collapse the reference's [repeat x s, Conv2d(1, 1, (1, 2 s + 1))] x 3 chain (fatchord_version.py:70-85) into
    one periodic FIR over mel frames:  up[s] = sum_j fir[s % hop, j] * mel_padded[s // hop + pad - (kmin + j)].

    Away from the sequence ends (which the reference trims by `indent`) the chain is linear and invariant to shifts
    by whole frames, so its response to a unit impulse at one frame determines it.  Returns (fir [hop, ntaps], kmin).
    """
    hop = int(np.prod(scales))
    kmin, ntaps = composite_filter_support(scales)
    return _response_to_fir(_composite_response(up_weights, scales), hop, kmin, ntaps).astype(np.float32), kmin


def composite_filter_gradients(up_weights, scales, dfir):
    """Gradients of a loss w.r.t. the three upsampling kernels given dL/dfir [hop, ntaps]: the composite filter is linear
    in each kernel separately, so dL/dw_i[a] = <dfir, composite with w_i replaced by the unit vector e_a>."""
    hop = int(np.prod(scales))
    kmin, ntaps = composite_filter_support(scales)
    dfir = np.asarray(dfir, np.float64)
    ws = [np.asarray(k, np.float64).reshape(-1) for k in up_weights]
    grads = []
    for i, w in enumerate(ws):
        g = np.zeros_like(w)
        for a in range(w.shape[0]):
            e = np.zeros_like(w)
            e[a] = 1.0
            trial = ws[:i] + [e] + ws[i + 1:]
            g[a] = float((_response_to_fir(_composite_response(trial, scales), hop, kmin, ntaps) * dfir).sum())
        grads.append(g)
    return grads


def fold_starts(total_len, target, overlap):
    """Start offsets of the folds fold_with_overlap (fatchord_version.py:298-317) produces for `total_len` samples."""
    num_folds = (total_len - overlap) // (target + overlap)
    extended_len = num_folds * (overlap + target) + overlap
    if total_len - extended_len != 0:
        num_folds += 1
    return [i * (target + overlap) for i in range(num_folds)]


def xfade_and_unfold(y, target, overlap):
    """fatchord_version.py:321-383 (host, float64)."""
    y = np.array(y, dtype=np.float64)
    num_folds, length = y.shape
    target = length - 2 * overlap
    total_len = num_folds * (target + overlap) + overlap
    silence_len = overlap // 2
    fade_len = overlap - silence_len
    t = np.linspace(-1, 1, fade_len, dtype=np.float64)
    fade_in = np.concatenate([np.zeros(silence_len), np.sqrt(0.5 * (1 + t))])
    fade_out = np.concatenate([np.sqrt(0.5 * (1 - t)), np.zeros(silence_len)])
    y[:, :overlap] *= fade_in
    y[:, -overlap:] *= fade_out
    unfolded = np.zeros(total_len, dtype=np.float64)
    for i in range(num_folds):
        start = i * (target + overlap)
        unfolded[start:start + target + 2 * overlap] += y[i]
    return unfolded


def decode_mu_law(y, mu, from_labels=True):
    """WaveRNN/utility/dsp.py:100-105."""
    if from_labels:
        y = 2 * y / (2 ** math.log2(mu) - 1.) - 1.
    mu = mu - 1
    return np.sign(y) / mu * ((1 + mu) ** np.abs(y) - 1)


def _fragment_blocks(w, kp):
    """float [N, K] -> (hi, lo) int16 tensors [N / 16, kp / 32, 512]: the 1 KiB B-operand images of
    v_mfma_f32_16x16x32_bf16 (lane l holds W[16 nb + (l & 15)][32 ks + 8 (l >> 4) .. + 7]), K zero-padded to kp."""
    w = np.asarray(w, np.float32)
    n, k = w.shape
    wp = torch.zeros(n, kp, dtype=torch.float32)
    wp[:, :k] = torch.from_numpy(w)
    hi = wp.to(torch.bfloat16)
    lo = (wp - hi.to(torch.float32)).to(torch.bfloat16)
    img = lambda t: t.view(n // 16, 16, kp // 32, 4, 8).permute(0, 2, 3, 1, 4).reshape(n // 16, kp // 32, 512).view(torch.int16)
    return img(hi), img(lo)


def _with_slack(stream):
    """Flat weight stream followed by 32 KiB of zeros: the layer kernel requests whole chunks unconditionally, so its last
    workgroup reads (and ignores) up to one chunk past the end."""
    return torch.cat([stream.reshape(-1), torch.zeros(16384, dtype=stream.dtype)]).contiguous()


def pack_gru_stream(x_types, h_types):
    """Weight stream of a GRU estts_wavernn_layer: for every 16-unit block the fragment images in the order the kernel's K loop
    consumes them, (x phase: K-step, type, hi | lo), then (h phase: K-step, type, hi | lo).  x_types: float arrays [H, kx]
    (rnn1: xI, r, z, n; rnn2: r, z, n); h_types: the three [H, H] recurrent matrices (r, z, n)."""
    kx = x_types[0].shape[1]
    H = h_types[0].shape[0]
    if H % 32:
        raise ValueError('rnn_dims must be a multiple of 32')

    def phase(mats, kp):
        fr = [_fragment_blocks(m, kp) for m in mats]                          # per type (hi, lo) [nub, nks, 512]
        t = torch.stack([torch.stack(hl) for hl in fr])                      # [type, 2, nub, nks, 512]
        return t.permute(2, 3, 0, 1, 4).reshape(H // 16, -1)                # [nub, nks * type * 2 * 512]
    out = torch.cat([phase(x_types, (kx + 31) // 32 * 32), phase(h_types, H)], dim=1)
    return _with_slack(out)


def pack_dense_stream(w, unit_blocks):
    """Weight stream of a dense estts_wavernn_layer: rows padded with zeros to whole groups of unit_blocks * 16 units;
    per group the fragment images in the order (K-step, unit block, hi | lo).  Returns (stream [groups, ...], padded rows)."""
    w = np.asarray(w, np.float32)
    n, k = w.shape
    per = 16 * unit_blocks
    npad = (n + per - 1) // per * per
    wp = np.zeros((npad, k), np.float32)
    wp[:n] = w
    hi, lo = _fragment_blocks(wp, (k + 31) // 32 * 32)                        # [npad / 16, nks, 512]
    t = torch.stack([hi, lo], dim=2)                                          # [nub, nks, 2, 512]
    t = t.view(npad // per, unit_blocks, t.shape[1], 2, 512).permute(0, 2, 1, 3, 4)
    return _with_slack(t.reshape(npad // per, -1)), npad


def panel_alloc(n_seq, kpitch, device):
    """A zeroed activation panel for n_seq sequences and 32 * kpitch columns (see estts_wavernn_layer)."""
    return torch.zeros((n_seq + 15) // 16, kpitch, 2, 512, dtype=torch.int16, device=device)


def panel_pack(x, kpitch=None):
    """float [n, K] -> activation panel (bf16 hi + lo fragment blocks) on x's device."""
    n, k = x.shape
    kp = (kpitch or (k + 31) // 32) * 32
    npad = (n + 15) // 16 * 16
    xp = torch.zeros(npad, kp, dtype=torch.float32, device=x.device)
    xp[:n, :k] = x
    hi = xp.to(torch.bfloat16)
    lo = (xp - hi.to(torch.float32)).to(torch.bfloat16)
    img = lambda t: t.view(npad // 16, 16, kp // 32, 4, 8).permute(0, 2, 3, 1, 4).reshape(npad // 16, kp // 32, 512).view(torch.int16)
    return torch.stack([img(hi), img(lo)], dim=2).contiguous()


def panel_unpack(panel, n, k):
    """Activation panel -> float32 [n, k] (hi + lo)."""
    rb, kb = panel.shape[0], panel.shape[1]
    f = panel.view(torch.bfloat16).to(torch.float32).sum(dim=2)              # [rb, kb, 512]
    return f.view(rb, kb, 4, 16, 8).permute(0, 3, 1, 2, 4).reshape(rb * 16, kb * 32)[:n, :k]


def _fold_bn(w, p, eps=1e-5):
    s = np.asarray(w[p + '.weight'], np.float64) / np.sqrt(np.asarray(w[p + '.running_var'], np.float64) + eps)
    b = np.asarray(w[p + '.bias'], np.float64) - np.asarray(w[p + '.running_mean'], np.float64) * s
    return s.astype(np.float32), b.astype(np.float32)


class WaveRNN:
    def __init__(self, rnn_dims, fc_dims, bits, pad, upsample_factors, feat_dims, compute_dims, res_out_dims, res_blocks,
                 hop_length, sample_rate, mode='RAW', device='cuda', precision='bf16x3'):
        if mode == 'RAW':
            self.n_classes = 2 ** bits
        elif mode == 'MOL':
            self.n_classes = 30
        else:
            raise RuntimeError(f'Unknown model mode value - {mode}')
        if int(np.prod(upsample_factors)) != hop_length:
            raise ValueError('upsample_factors must multiply to hop_length')
        if not torch.cuda.is_available():
            raise RuntimeError('WaveRNN needs a GPU: the hot path has no CPU implementation')
        _lib.load()
        self.mode, self.pad = mode, pad
        self.rnn_dims, self.fc_dims = rnn_dims, fc_dims
        self.aux_dims = res_out_dims // 4
        self.feat_dims, self.compute_dims, self.res_out_dims, self.res_blocks = feat_dims, compute_dims, res_out_dims, res_blocks
        self.upsample_factors = tuple(upsample_factors)
        self.hop_length, self.sample_rate = hop_length, sample_rate
        self.device = torch.device(device)
        self.precision = precision
        self.step = 0
        self._sd = None
        self.steps_per_graph = 50        # even: the GRU states ping-pong between two buffers
        # one launch per layer over activation panels (estts_wavernn_layer); False = separate products and gate launches
        self.fused = rnn_dims % 32 == 0 and fc_dims == rnn_dims and (res_out_dims // 4) % 8 == 0 and res_out_dims // 4 <= 32 \
            and 8 + feat_dims + res_out_dims // 4 <= 128
        ops.gemm_workspace(self.device)

    @classmethod
    def from_hparams(cls, hp, **kw):
        return cls(rnn_dims=hp['voc_rnn_dims'], fc_dims=hp['voc_fc_dims'], bits=hp['bits'], pad=hp['voc_pad'],
                   upsample_factors=hp['voc_upsample_factors'], feat_dims=hp['num_mels'],
                   compute_dims=hp['voc_compute_dims'], res_out_dims=hp['voc_res_out_dims'],
                   res_blocks=hp['voc_res_blocks'], hop_length=hp['hop_length'], sample_rate=hp['sample_rate'],
                   mode=hp['voc_mode'], **kw)

    # ------------------------------------------------------------------ weights
    def load_state_dict(self, sd, strict=False):
        w = {}
        for k, v in sd.items():
            w[k] = v.detach().cpu().numpy() if torch.is_tensor(v) else np.asarray(v)
        if 'step' in w:
            self.step = int(np.asarray(w['step']).reshape(-1)[0])
        dev = self.device
        p = 'upsample.resnet'
        s, b = _fold_bn(w, p + '.batch_norm')
        wi = np.asarray(w[p + '.conv_in.weight'])                          # (C, feat, k)
        k = wi.shape[2]
        if k != 2 * self.pad + 1:
            raise ValueError('conv_in kernel size does not match pad')
        self.conv_in = PackedLinear(wi.transpose(0, 2, 1).reshape(wi.shape[0], -1), bias=b, scale=s, device=dev, taps=k,
                                    lpad=0, act='relu')
        self.res = []
        for i in range(self.res_blocks):
            q = f'{p}.layers.{i}'
            s1, b1 = _fold_bn(w, q + '.batch_norm1')
            s2, b2 = _fold_bn(w, q + '.batch_norm2')
            self.res.append((PackedLinear(np.asarray(w[q + '.conv1.weight'])[:, :, 0], bias=b1, scale=s1, device=dev, act='relu'),
                             PackedLinear(np.asarray(w[q + '.conv2.weight'])[:, :, 0], bias=b2, scale=s2, device=dev)))
        self.conv_out = PackedLinear(np.asarray(w[p + '.conv_out.weight'])[:, :, 0], bias=w[p + '.conv_out.bias'], device=dev)
        ups = [w[f'upsample.up_layers.{2 * j + 1}.weight'] for j in range(len(self.upsample_factors))]
        fir, kmin = composite_upsample_filter(ups, self.upsample_factors)
        if self.pad - kmin - (fir.shape[1] - 1) < 0 or self.pad + kmin < 0:
            raise ValueError('pad is too small for the upsampling filters')
        self.fir = torch.from_numpy(fir).to(dev)
        self.fir_kmin = kmin
        n_in = 1 + self.feat_dims + self.aux_dims
        self.k_in = (n_in + 3) // 4 * 4
        wI = np.zeros((self.rnn_dims, self.k_in), np.float32)
        wI[:, :n_in] = w['I.weight']
        # rnn1's input product taken straight from the step input: W_ih (W_I x + b_I) + b_ih = (W_ih W_I) x + (W_ih b_I + b_ih),
        # composed in float64.  K drops from rnn_dims to 1 + feat + aux, and the product no longer waits for I's output
        # (which is still computed, as the residual `x + h1` needs it).  Both come out of one launch, see in_proj.
        w_ih, w_i = np.asarray(w['rnn1.weight_ih_l0'], np.float64), wI.astype(np.float64)
        b_comp = w_ih @ np.asarray(w['I.bias'], np.float64) + np.asarray(w['rnn1.bias_ih_l0'], np.float64)
        # both products of the step input in one launch: rows [0, R) give I's output, rows [R, 4R) rnn1's input gates
        self.in_proj = PackedLinear(np.concatenate([wI.astype(np.float64), w_ih @ w_i]).astype(np.float32),
                                    bias=np.concatenate([np.asarray(w['I.bias'], np.float64), b_comp]).astype(np.float32), device=dev)
        self.ih2 = PackedLinear(w['rnn2.weight_ih_l0'], bias=w['rnn2.bias_ih_l0'], device=dev)
        # the one-launch GRU layers (estts_wavernn_layer): one weight stream and one bias table per GRU
        R = self.rnn_dims
        wc = (w_ih @ w_i).astype(np.float32)
        w_hh1, w_hh2 = np.asarray(w['rnn1.weight_hh_l0'], np.float32), np.asarray(w['rnn2.weight_hh_l0'], np.float32)
        w_ih2 = np.asarray(w['rnn2.weight_ih_l0'], np.float32)
        thirds = lambda m: [m[i * R:(i + 1) * R] for i in range(3)]
        # (column order of the step-input panel: x_prev alone in the first group of eight, then mel and the aux slice)
        spread = lambda m: np.concatenate([m[:, :1], np.zeros((m.shape[0], 7), np.float32), m[:, 1:n_in]], axis=1)
        self.gru1_w = pack_gru_stream([spread(wI)] + [spread(t) for t in thirds(wc)], thirds(w_hh1)).to(dev)
        self.gru2_w = pack_gru_stream(thirds(w_ih2), thirds(w_hh2)).to(dev)
        b_hh1, b_hh2 = np.asarray(w['rnn1.bias_hh_l0'], np.float64), np.asarray(w['rnn2.bias_hh_l0'], np.float64)
        b_ih2 = np.asarray(w['rnn2.bias_ih_l0'], np.float64)
        f32 = lambda rows: torch.from_numpy(np.stack(rows).astype(np.float32)).to(dev)
        self.gru1_b = f32([np.asarray(w['I.bias'], np.float64), b_comp[:R] + b_hh1[:R], b_comp[R:2 * R] + b_hh1[R:2 * R],
                           b_comp[2 * R:], b_hh1[2 * R:]])
        self.gru2_b = f32([b_ih2[:R] + b_hh2[:R], b_ih2[R:2 * R] + b_hh2[R:2 * R], b_ih2[2 * R:], b_hh2[2 * R:]])
        self.n_in = n_in
        self.k_in_panel = 8 + self.feat_dims + self.aux_dims
        self.alternate_step_counters = True
        self.fc_unit_blocks, self.fc_seq_tile = 2, 64     # dense layers: 64 sequences x 32 units per workgroup
        self._pack_fc(w)
        # the two recurrent products depend only on the previous step's states: one batched launch (weights and biases of
        # rnn2 follow rnn1's in one packed buffer; 3R is a multiple of 64, so the second matrix starts on a block boundary)
        self.hh12 = PackedLinear(np.concatenate([np.asarray(w['rnn1.weight_hh_l0']), np.asarray(w['rnn2.weight_hh_l0'])]),
                                 bias=np.concatenate([np.asarray(w['rnn1.bias_hh_l0']), np.asarray(w['rnn2.bias_hh_l0'])]), device=dev)
        self.hh12.N = 3 * self.rnn_dims          # one product's width; the batch stride leads to the other
        self.fc1 = PackedLinear(w['fc1.weight'], bias=w['fc1.bias'], device=dev, act='relu')
        self.fc2 = PackedLinear(w['fc2.weight'], bias=w['fc2.bias'], device=dev, act='relu')
        self.fc3 = PackedLinear(w['fc3.weight'], bias=w['fc3.bias'], device=dev)
        if self.fc3.N != self.n_classes:
            raise ValueError(f'fc3 has {self.fc3.N} outputs, mode {self.mode} expects {self.n_classes}')
        self._sd = {k: np.array(v) for k, v in w.items()}
        return self

    def _pack_fc(self, w):
        """Weight streams and padded biases of fc1, fc2, fc3 for estts_wavernn_layer (re-run after changing fc_unit_blocks)."""
        dev = self.device
        self.fc_streams = []
        for name in ('fc1', 'fc2', 'fc3'):
            stream, npad = pack_dense_stream(w[name + '.weight'], self.fc_unit_blocks)
            b = np.zeros(npad, np.float32)
            b[:w[name + '.bias'].shape[0]] = w[name + '.bias']
            self.fc_streams.append((stream.to(dev), torch.from_numpy(b).to(dev)))

    def state_dict(self):
        sd = {k: torch.from_numpy(np.array(v)) for k, v in self._sd.items()}
        sd['step'] = torch.tensor([self.step], dtype=torch.long)
        return sd

    def load(self, path):
        self.load_state_dict(torch.load(path, map_location='cpu', weights_only=True), strict=False)

    def save(self, path):
        torch.save(self.state_dict(), path)

    def restore(self, path):
        if not os.path.exists(path):
            print('\nNew WaveRNN Training Session...\n')
            self.save(path)
        else:
            print(f'\nLoading Weights: "{path}"\n')
            self.load(path)

    def checkpoint(self, path):
        self.save(f'{path}/checkpoint_{self.get_step() // 1000}k_steps.pyt')

    def get_step(self):
        return self.step

    def num_params(self, print_out=True):
        n = sum(v.size for k, v in self._sd.items() if 'running' not in k and k != 'step' and 'num_batches' not in k) / 1e6
        if print_out:
            print('Trainable Parameters: %.3fM' % n)
        return n

    # ------------------------------------------------------------------ conditioning network
    def _aux_frames(self, mel_padded, seg_len=None):
        """mel_padded: [n (T + 2 pad), feat] float32 device tensor, n utterances of seg_len = T + 2 pad padded frames each
        -> MelResNet output [n (T + 2 pad), res_out] (fatchord_version.py:39-45); the first T rows of every utterance count."""
        Tp = mel_padded.shape[0]
        seg_len = Tp if seg_len is None else seg_len
        f32 = dict(dtype=torch.float32, device=self.device)
        x = torch.empty(Tp, self.compute_dims, **f32)
        h = torch.empty(Tp, self.compute_dims, **f32)
        y = torch.empty(Tp, self.compute_dims, **f32)
        ops.gemm(mel_padded, self.conv_in, x, seg_len=seg_len, precision=self.precision)   # rows >= T are partial sums, unused
        for c1, c2 in self.res:
            ops.gemm(x, c1, h, precision=self.precision)
            ops.gemm(h, c2, y, residual=x, precision=self.precision)
            x, y = y, x
        out = torch.empty(Tp, self.res_out_dims, **f32)
        ops.gemm(x, self.conv_out, out, precision=self.precision)
        return out

    def _pack_conditioning(self, padded):
        """padded: list of [T + 2 pad, feat] float32 host tensors.  Runs the residual network once per group of utterances of
        equal length (all rows of a group in one launch per layer: 22 launches for the whole batch instead of 22 per
        utterance) and returns the packed padded mel frames, aux frames and per-utterance offsets on the device."""
        frames = [int(mp.shape[0]) - 2 * self.pad for mp in padded]
        mel_all = torch.cat(padded).contiguous().to(self.device)
        mel_off = np.concatenate([[0], np.cumsum([t + 2 * self.pad for t in frames])[:-1]]).astype(np.int64)
        aux_off = np.concatenate([[0], np.cumsum(frames)[:-1]]).astype(np.int64)
        aux_all = torch.empty(int(sum(frames)), self.res_out_dims, dtype=torch.float32, device=self.device)
        groups = {}
        for i, t in enumerate(frames):
            groups.setdefault(t, []).append(i)
        for T, idx in groups.items():
            Tp = T + 2 * self.pad
            if len(idx) == len(frames):
                stacked = mel_all                                   # one group: the packed frames are the stacked batch
            else:
                stacked = torch.cat([mel_all[mel_off[i]:mel_off[i] + Tp] for i in idx])
            out = self._aux_frames(stacked, seg_len=Tp).view(len(idx), Tp, self.res_out_dims)
            for j, i in enumerate(idx):
                aux_all[aux_off[i]:aux_off[i] + T].copy_(out[j, :T])
        i32 = dict(dtype=torch.int32, device=self.device)
        return {'mel': mel_all, 'aux': aux_all,
                'mel_off': torch.tensor(mel_off, **i32), 'aux_off': torch.tensor(aux_off, **i32), 'frames': frames,
                'utt_samples': torch.tensor([t * self.hop_length for t in frames], **i32)}

    def _prepare(self, mel_list):
        """mel_list: list of [feat, T] arrays/tensors (values as the reference expects): zero-padded by `pad` frames on both
        sides (fatchord_version.py:153-156), then _pack_conditioning."""
        padded = []
        for m in mel_list:
            m = torch.as_tensor(m, dtype=torch.float32)
            if m.dim() != 2 or m.shape[0] != self.feat_dims:
                raise ValueError(f'each mel must be [feat={self.feat_dims}, T], got {tuple(m.shape)}')
            T = m.shape[1]
            mp = torch.zeros(T + 2 * self.pad, self.feat_dims, dtype=torch.float32)
            mp[self.pad:self.pad + T] = m.t().cpu()
            padded.append(mp)
        return self._pack_conditioning(padded)

    # ------------------------------------------------------------------ step machinery
    def _alloc_state(self, n_seq, L):
        f32 = dict(dtype=torch.float32, device=self.device)
        R, d = self.rnn_dims, self.aux_dims
        return {'xin': torch.zeros(n_seq, (self.k_in + 31) // 32 * 32, **f32),   # zero beyond the inputs (the fused layer reads whole K-steps)
                'parity': 0,
                'panels': self._alloc_panels(n_seq) if self._use_layers() else None,
                'xg': torch.empty(n_seq, 4 * R, **f32),                       # [I's output | rnn1's input gates]
                'gi2': torch.empty(n_seq, 3 * R, **f32), 'gh12': torch.empty(2, n_seq, 3 * R, **f32),
                'h12': torch.zeros(2, n_seq, R, **f32),
                # (row pitch rounded up to whole K-steps of 32 and zero there: the fused layer reads whole K-steps)
                'x1cat': torch.zeros(n_seq, (R + d + 31) // 32 * 32, **f32)[:, :R + d],
                'x2cat': torch.zeros(n_seq, R + d, **f32),
                'f1cat': torch.zeros(n_seq, self.fc_dims + d, **f32), 'f2': torch.empty(n_seq, self.fc_dims, **f32),
                'logits': torch.empty(n_seq, self.n_classes, **f32), 'x_prev': torch.zeros(n_seq, **f32),
                'wave': torch.zeros(n_seq, L, **f32), 'step': torch.zeros(2, dtype=torch.int32, device=self.device),
                'done': torch.zeros(1, dtype=torch.int32, device=self.device)}

    def _use_layers(self):
        """One launch per layer over activation panels (split-bf16 only); else separate products and gate launches."""
        return self.fused and self.precision == 'bf16x3'

    def _alloc_panels(self, n_seq):
        R, F, d, dev = self.rnn_dims, self.fc_dims, self.aux_dims, self.device
        kx = (R + d + 31) // 32
        native = (n_seq + 127) // 128 * (R // 16) * 2048
        two = lambda kp: [panel_alloc(n_seq, kp, dev) for _ in range(2)]
        # (two sets of the panels the conditioning writes: a step's conditioning is computed beside the previous step's layers)
        return {'xin': two(4), 'x1': two(kx), 'x2': two(kx), 'f1': two(kx), 'f2': panel_alloc(n_seq, F // 32, dev),
                'cstep': torch.zeros(2, dtype=torch.int32, device=dev),
                'h1': [panel_alloc(n_seq, R // 32, dev) for _ in range(2)],
                'h2': [panel_alloc(n_seq, R // 32, dev) for _ in range(2)],
                # float32 states and rnn1's output in the GRU layers' own order (see estts_wavernn_layer)
                'h1_f32': [torch.zeros(native, dtype=torch.float32, device=dev) for _ in range(2)],
                'h2_f32': [torch.zeros(native, dtype=torch.float32, device=dev) for _ in range(2)],
                'x1_f32': torch.zeros(native, dtype=torch.float32, device=dev)}

    def _xprev_to_panel(self, st):
        """First column of the step-input panel from st['x_prev'] (later steps: the sampler writes it)."""
        x = st['x_prev']
        hi = x.to(torch.bfloat16)
        lo = (x - hi.to(torch.float32)).to(torch.bfloat16)
        b = torch.arange(x.shape[0], device=x.device)
        pn = st['panels']['xin'][0]
        pn[b // 16, 0, 0, (b % 16) * 8] = hi.view(torch.int16)
        pn[b // 16, 0, 1, (b % 16) * 8] = lo.view(torch.int16)

    def _cond_args(self, cond, st, seq_utt, seq_start, n_seq):
        return [self._cond_args_set(cond, st, seq_utt, seq_start, n_seq, q) for q in (0, 1)]

    def _cond_args_set(self, cond, st, seq_utt, seq_start, n_seq, q):
        a, pn = WavernnCondArgs(), st['panels']
        if 'seq_info' not in st:
            # per sequence: (first mel row + pad - fir_kmin, first aux row, start sample, samples of its utterance)
            u = seq_utt.long()
            st['seq_info'] = torch.stack([cond['mel_off'][u] + (self.pad - self.fir_kmin), cond['aux_off'][u], seq_start,
                                          cond['utt_samples'][u]], dim=1).to(torch.int32).contiguous()
        a.mel, a.aux, a.fir, a.ntaps = cond['mel'].data_ptr(), cond['aux'].data_ptr(), self.fir.data_ptr(), self.fir.shape[1]
        a.seq_info = st['seq_info'].data_ptr()
        a.step, a.done = pn['cstep'][0:].data_ptr(), pn['cstep'][1:].data_ptr()
        a.xin_panel, a.a2_panel, a.a3_panel, a.a4_panel = (pn[k][q].data_ptr() for k in ('xin', 'x1', 'x2', 'f1'))
        a.aux_ks, a.aux_kpitch = self.rnn_dims // 32, pn['x1'][q].shape[1]
        a.n_seq, a.feat, a.d, a.hop = n_seq, self.feat_dims, self.aux_dims, self.hop_length
        a._keep = (cond, st, seq_utt, seq_start)
        return a

    def _prep_args(self, cond, st, seq_utt, seq_start, n_seq):
        a = WavernnPrepArgs()
        a.mel, a.mel_off, a.aux, a.aux_off = cond['mel'].data_ptr(), cond['mel_off'].data_ptr(), cond['aux'].data_ptr(), cond['aux_off'].data_ptr()
        a.fir, a.ntaps, a.fir_kmin = self.fir.data_ptr(), self.fir.shape[1], self.fir_kmin
        a.seq_utt, a.seq_start, a.utt_samples = seq_utt.data_ptr(), seq_start.data_ptr(), cond['utt_samples'].data_ptr()
        a.x_prev, a.step = st['x_prev'].data_ptr(), st['step'].data_ptr()
        a.xin, a.ld_xin = st['xin'].data_ptr(), st['xin'].stride(0)
        R, F = self.rnn_dims, self.fc_dims
        a.a2, a.ld_a2 = st['x1cat'][:, R:].data_ptr(), st['x1cat'].stride(0)
        a.a3, a.ld_a3 = st['x2cat'][:, R:].data_ptr(), st['x2cat'].stride(0)
        a.a4, a.ld_a4 = st['f1cat'][:, F:].data_ptr(), st['f1cat'].stride(0)
        a.n_seq, a.feat, a.d, a.hop, a.pad = n_seq, self.feat_dims, self.aux_dims, self.hop_length, self.pad
        return a

    def _sample_args(self, st, n_seq, seed, u, u_mix, forced, logits=None):
        a = WavernnSampleArgs()
        lg = st['logits'] if logits is None else logits
        a.logits, a.ld_logits, a.n_classes = lg.data_ptr(), lg.stride(0), self.n_classes
        a.mode = 0 if self.mode == 'RAW' else 1
        a.u, a.u_mix, a.forced = ops._ptr(u), ops._ptr(u_mix), ops._ptr(forced)
        a.seed = seed & 0xFFFFFFFF
        a.wave, a.ld_wave, a.x_prev = st['wave'].data_ptr(), st['wave'].stride(0), st['x_prev'].data_ptr()
        a.step, a.done, a.n_seq = st['step'].data_ptr(), st['done'].data_ptr(), n_seq
        if st.get('panels') is not None:
            # the value fed back also goes to the next step's input panel (the other set); step t uses set t & 1
            # step t (panel set t & 1) also prepares step t + 1's conditioning, in the other set
            b = WavernnSampleArgs.from_buffer_copy(a)
            for args, nxt in ((a, 1), (b, 0)):
                args.with_cond = 1
                args.cond = _lib.WavernnCondArgs.from_buffer_copy(st['cond_args'][nxt])
                # two step counters, like the panel sets: step t reads counter t & 1 and writes t + 1 into the other
                if self.alternate_step_counters:
                    args.step, args.step_next = st['step'][nxt ^ 1:].data_ptr(), st['step'][nxt:].data_ptr()
            a._keep = b._keep = (lg, u, u_mix, forced, st)
            return [a, b]
        return a

    def _one_step(self, st, prep_args, sample_args, logits_out=None):
        """fatchord_version.py:180-216 for every sequence at once, 11 launches: conditioning, [I | rnn1 input gates] from the
        step input, both recurrent products (they depend only on the previous step's states) as one batched launch, gates,
        rnn2's input product, gates, fc1, fc2, fc3, sampling."""
        lib, R, F, pr = _lib.load(), self.rnn_dims, self.fc_dims, self.precision
        n = st['h12'].shape[1]
        if self._use_layers():
            # Step t works on panel set q = t & 1.  6 launches: rnn1 (with the input layer), rnn2, fc1, fc2, fc3, sampling; the
            # sampler also writes the next step's conditioning (and x_prev) into the other set, its loads overlapping the
            # softmax.  (The conditioning as a launch of its own cost 10 us + a launch boundary per step; on a second
            # stream beside the previous step's layers it was slower still, 106 vs 89 us per step: the fork and join edges
            # of the captured graph cost more than the kernel they take off the path.)
            pn = st['panels']
            q = st['parity']
            st['parity'] ^= 1
            n_seq = st['h12'].shape[1]
            if not st.get('cond_primed'):        # the first step's own conditioning
                check(lib.estts_wavernn_cond(st['cond_args'][q], ops._stream()), 'estts_wavernn_cond')
                st['cond_primed'] = True
            self._layer(1, n_seq, R, pn['xin'][q], (self.k_in_panel + 31) // 32, self.gru1_w, self.gru1_b, a1=pn['h1'][q],
                        h_prev=pn['h1_f32'][q], h_out=pn['h1_f32'][q ^ 1], h_panel=pn['h1'][q ^ 1], out_f32=pn['x1_f32'],
                        out_panel=pn['x1'][q])
            self._layer(2, n_seq, R, pn['x1'][q], pn['x1'][q].shape[1], self.gru2_w, self.gru2_b, a1=pn['h2'][q],
                        h_prev=pn['h2_f32'][q], h_out=pn['h2_f32'][q ^ 1], h_panel=pn['h2'][q ^ 1], x_res=pn['x1_f32'],
                        out_panel=pn['x2'][q])
            (w1, b1), (w2, b2), (w3, b3) = self.fc_streams
            self._layer(0, n_seq, F, pn['x2'][q], pn['x2'][q].shape[1], w1, b1, act='relu', out_panel=pn['f1'][q])
            self._layer(0, n_seq, F, pn['f1'][q], pn['f1'][q].shape[1], w2, b2, act='relu', out_panel=pn['f2'])
            self._layer(0, n_seq, self.n_classes, pn['f2'], pn['f2'].shape[1], w3, b3,
                        out_f32=st['logits'] if logits_out is None else logits_out)
            check(lib.estts_wavernn_sample(sample_args[q], ops._stream()), 'estts_wavernn_sample')
            return
        check(lib.estts_wavernn_prep(prep_args, ops._stream()), 'estts_wavernn_prep')
        h1, h2 = st['h12'][0], st['h12'][1]
        xI, gi1 = st['xg'][:, :R], st['xg'][:, R:]
        ops.gemm(st['xin'], self.in_proj, st['xg'], precision=pr)
        ops.gemm(h1, self.hh12, st['gh12'][0], precision=pr, split=False,
                 batch=(2, n * R, 3 * R * self.hh12.ldw, n * 3 * R, 3 * R))
        self._gates(gi1, st['gh12'][0], h1, xI, st['x1cat'][:, :R])
        ops.gemm(st['x1cat'], self.ih2, st['gi2'], precision=pr)
        self._gates(st['gi2'], st['gh12'][1], h2, st['x1cat'][:, :R], st['x2cat'][:, :R])
        ops.gemm(st['x2cat'], self.fc1, st['f1cat'][:, :F], precision=pr)
        ops.gemm(st['f1cat'], self.fc2, st['f2'], precision=pr)
        ops.gemm(st['f2'], self.fc3, st['logits'] if logits_out is None else logits_out, precision=pr)
        check(lib.estts_wavernn_sample(sample_args, ops._stream()), 'estts_wavernn_sample')

    def _layer(self, mode, n_seq, n_units, a0, ks0, w, b, a1=None, h_prev=None, h_out=None, h_panel=None, x_res=None,
               out_f32=None, out_panel=None, act=None):
        a = WavernnLayerArgs()
        a.a0, a.ks0, a.kpitch0 = a0.data_ptr(), ks0, a0.shape[1]
        if a1 is not None:
            a.a1, a.ks1, a.kpitch1 = a1.data_ptr(), n_units // 32, a1.shape[1]
            a.h_prev, a.h_out = h_prev.data_ptr(), h_out.data_ptr()
            a.h_panel, a.h_kpitch = h_panel.data_ptr(), h_panel.shape[1]
        if x_res is not None:
            a.x_res = x_res.data_ptr()
        a.W, a.bias = w.data_ptr(), b.data_ptr()
        a.n_seq, a.n_units, a.mode, a.act, a.unit_blocks = n_seq, n_units, mode, _lib.ACT[act], self.fc_unit_blocks
        a.seq_tile = self.fc_seq_tile if mode == 0 and self.fc_unit_blocks == 2 else 0
        if out_f32 is not None:
            a.out_f32, a.ldo = out_f32.data_ptr(), (out_f32.stride(0) if out_f32.dim() == 2 else 0)
        if out_panel is not None:
            a.out_panel, a.out_kpitch = out_panel.data_ptr(), out_panel.shape[1]
        check(_lib.load().estts_wavernn_layer(a, ops._stream()), 'estts_wavernn_layer')
        k = 32 * ks0
        macs = n_units * ({0: 1, 1: 4, 2: 3}[mode] * k + (3 * n_units if mode else 0))
        ops._record('wavernn_layer', {'mode': mode, 'n_seq': n_seq, 'n_units': n_units, 'K': k, 'flops': 2.0 * n_seq * macs,
                                      'bytes': 4.0 * macs + 4.0 * n_seq * (k + (n_units if mode else 0) + n_units)},
                    _lib.load().estts_wavernn_layer, a, keep=(a0, w, b, a1, h_prev, h_out, h_panel, x_res, out_f32, out_panel))

    def _gates(self, gi, gh, h, x_res, out):
        M, H = h.shape
        check(_lib.load().estts_gru_gates(gi.data_ptr(), gi.stride(0), gh.data_ptr(), gh.stride(0), h.data_ptr(), h.stride(0),
                                          x_res.data_ptr(), x_res.stride(0), out.data_ptr(), out.stride(0), M, H,
                                          ops._stream()), 'estts_gru_gates')

    def _run_steps(self, st, prep_args, sample_args, L, use_graph):
        n = self.steps_per_graph
        if n % 2:
            raise ValueError('steps_per_graph must be even (the GRU states alternate between two buffers)')
        done = 0
        if use_graph and L >= 2 * n:
            self._one_step(st, prep_args, sample_args)          # warm-up outside capture (also step 0)
            done = 1
            side = torch.cuda.Stream()
            side.wait_stream(torch.cuda.current_stream())
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g, stream=side):
                for _ in range(n):
                    self._one_step(st, prep_args, sample_args)
            # capture does not execute: the step counter still says `done`
            while L - done >= n:
                g.replay()
                done += n
            torch.cuda.current_stream().wait_stream(side)
            torch.cuda.synchronize()
        for _ in range(L - done):
            self._one_step(st, prep_args, sample_args)

    # ------------------------------------------------------------------ reference surface
    def generate_batch(self, mel_list, batched=True, target=11_000, overlap=550, mu_law=True, seed=0, uniforms=None,
                       forced=None, use_graph=True, return_trace=False):
        """Vocode several utterances at once: every fold of every utterance is one sequence of the step kernels.

        mel_list: list of [feat, T] mels in the reference's [0, 1] normalisation.  Returns a list of float64 waveforms
        (as WaveRNN.generate, fatchord_version.py:150-243).  uniforms: optional injected uniform numbers, RAW: [L, n_seq];
        MOL: {'mix': [L, n_seq, nr_mix], 'logistic': [L, n_seq]}.  forced: optional [L, n_seq] sample values fed back
        instead of the drawn ones.  With return_trace also returns the raw per-sequence samples [n_seq, L]."""
        mu_law = mu_law if self.mode == 'RAW' else False
        cond = self._prepare(mel_list)
        hop = self.hop_length
        seq_utt, seq_start, folds = [], [], []
        for ui, T in enumerate(cond['frames']):
            starts = fold_starts(T * hop, target, overlap) if batched else [0]
            folds.append(len(starts))
            seq_utt += [ui] * len(starts)
            seq_start += starts
        n_seq = len(seq_utt)
        L = target + 2 * overlap if batched else max(cond['frames']) * hop
        i32 = dict(dtype=torch.int32, device=self.device)
        seq_utt_d, seq_start_d = torch.tensor(seq_utt, **i32), torch.tensor(seq_start, **i32)
        if min(cond['frames']) < 21:
            # the reference fails here too, with numpy's broadcast error (fatchord_version.py:235-237)
            raise ValueError(f'generate needs at least 21 mel frames per utterance (the fade-out covers 20 hops), got {min(cond["frames"])}')
        st = self._alloc_state(n_seq, L)
        f32 = dict(dtype=torch.float32, device=self.device)
        u = u_mix = None
        if uniforms is not None:
            if self.mode == 'RAW':
                u = torch.as_tensor(np.asarray(uniforms), **f32).contiguous()
            else:
                u = torch.as_tensor(np.asarray(uniforms['logistic']), **f32).contiguous()
                u_mix = torch.as_tensor(np.asarray(uniforms['mix']), **f32).contiguous()
            if u.shape != (L, n_seq):
                raise ValueError(f'uniforms must have shape {(L, n_seq)}, got {tuple(u.shape)}')
        forced_d = None if forced is None else torch.as_tensor(np.asarray(forced), **f32).contiguous()
        prep_args = self._prep_args(cond, st, seq_utt_d, seq_start_d, n_seq)
        if self._use_layers():
            st['cond_args'] = self._cond_args(cond, st, seq_utt_d, seq_start_d, n_seq)
        sample_args = self._sample_args(st, n_seq, seed, u, u_mix, forced_d)
        self._run_steps(st, prep_args, sample_args, L, use_graph)
        if batched and overlap >= 4 and not return_trace:
            return self._unfold_on_device(st['wave'], cond['frames'], folds, target, overlap, mu_law)
        raw = st['wave'].cpu().numpy().astype(np.float64)
        outs, o = [], 0
        for ui, T in enumerate(cond['frames']):
            y = raw[o:o + folds[ui]]
            o += folds[ui]
            y = xfade_and_unfold(y, target, overlap) if batched else y[0]
            if mu_law:
                y = decode_mu_law(y, self.n_classes, False)
            wave_len = (T - 1) * hop
            y = y[:wave_len]
            y[-20 * hop:] *= np.linspace(1, 0, 20 * hop)
            outs.append(y)
        if return_trace:
            return outs, raw
        return outs

    def _unfold_on_device(self, wave, frames, folds, target, overlap, mu_law):
        """The tail of WaveRNN.generate (fatchord_version.py:222-241) for every utterance of the batch in one float64 launch
        (estts_wavernn_unfold): cross-fade of the folds, mu-law expansion, cut to (T - 1) hop samples, closing fade over the
        last 20 hops; one device-to-host copy of the finished waveforms.  The host forms (xfade_and_unfold, decode_mu_law)
        remain for the unbatched path and as the reference the tests compare this with."""
        hop = self.hop_length
        lens = [(T - 1) * hop for T in frames]
        out_off = np.concatenate([[0], np.cumsum(lens)[:-1]]).astype(np.int64)
        seq_off = np.concatenate([[0], np.cumsum(folds)[:-1]]).astype(np.int32)
        dev = self.device
        out = torch.empty(int(sum(lens)), dtype=torch.float64, device=dev)
        t = {k: torch.tensor(np.asarray(v), dtype=dt, device=dev) for k, v, dt in (
            ('seq_off', seq_off, torch.int32), ('n_folds', folds, torch.int32), ('out_len', lens, torch.int32),
            ('out_off', out_off, torch.int64))}
        a = _lib.WavernnUnfoldArgs()
        a.wave, a.ld_wave = wave.data_ptr(), wave.stride(0)
        a.seq_off, a.n_folds, a.out_len, a.out_off = (t[k].data_ptr() for k in ('seq_off', 'n_folds', 'out_len', 'out_off'))
        a.out, a.n_utt, a.target, a.overlap, a.max_len = out.data_ptr(), len(frames), target, overlap, max(lens)
        a.mu = self.n_classes if mu_law else 0
        a.fade_tail = 20 * hop
        fade_len = overlap - overlap // 2
        a.step_in = 2.0 / (fade_len - 1)                 # numpy.linspace(-1, 1, fade_len): step = (stop - start) / (n - 1)
        a.step_tail = -1.0 / (20 * hop - 1)              # numpy.linspace(1, 0, 20 hop)
        check(_lib.load().estts_wavernn_unfold(a, ops._stream()), 'estts_wavernn_unfold')
        host = out.cpu().numpy()
        return [host[o:o + n] for o, n in zip(out_off, lens)]

    def generate(self, mels, batched, target, overlap, mu_law, **kw):
        """fatchord_version.py:150.  mels: [1, feat, T]."""
        mels = torch.as_tensor(mels)
        if mels.dim() != 3 or mels.shape[0] != 1:
            raise ValueError('generate expects mels of shape [1, feat, T]; use generate_batch for several utterances')
        out = self.generate_batch([mels[0]], batched, target, overlap, mu_law, **kw)
        return out if kw.get('return_trace') else out[0]

    def forward(self, x, mels):
        """Teacher-forced logits, fatchord_version.py:119-148.  x: [B, L] float32 in [-1, 1]; mels: [B, feat, L / hop + 2 pad]
        (already padded, as the training collate produces them).  Returns [B, L, n_classes] on the device."""
        x = torch.as_tensor(x, dtype=torch.float32)
        mels = torch.as_tensor(mels, dtype=torch.float32)
        B, L = x.shape
        T = mels.shape[2] - 2 * self.pad
        if T * self.hop_length < L:
            raise ValueError('mels are too short for x')
        # the mels are padded already: feed the inner frames and let _prepare re-create the same padding values
        cond = self._prepare_padded(mels)
        i32 = dict(dtype=torch.int32, device=self.device)
        seq_utt, seq_start = torch.arange(B, **i32), torch.zeros(B, **i32)
        st = self._alloc_state(B, L)
        xd = x.to(self.device)
        st['x_prev'].copy_(xd[:, 0])
        if self._use_layers():
            self._xprev_to_panel(st)
        forced = torch.zeros(L, B, dtype=torch.float32, device=self.device)
        forced[:L - 1] = xd[:, 1:].t()
        logits = torch.empty(L, B, self.n_classes, dtype=torch.float32, device=self.device)
        prep_args = self._prep_args(cond, st, seq_utt, seq_start, B)
        if self._use_layers():
            st['cond_args'] = self._cond_args(cond, st, seq_utt, seq_start, B)
        for t in range(L):
            sample_args = self._sample_args(st, B, 0, None, None, forced, logits=logits[t])
            self._one_step(st, prep_args, sample_args, logits_out=logits[t])
        self.step += 1
        return logits.permute(1, 0, 2)

    __call__ = forward

    def _prepare_padded(self, mels):
        """mels [B, feat, T + 2 pad] whose edge frames are real context (training windows), not zero padding."""
        return self._pack_conditioning([torch.as_tensor(m, dtype=torch.float32).t().contiguous().cpu() for m in mels])

    def upsample(self, mels):
        """UpsampleNetwork.forward (fatchord_version.py:79-86), materialised: mels [B, feat, T + 2 pad] ->
        (upsampled mels [B, T hop, feat], aux [B, T hop, res_out]).  Generation never materialises these; this entry
        exists for parity checks and runs the same conditioning kernel with one "sequence" per output sample."""
        mels = torch.as_tensor(mels, dtype=torch.float32)
        cond = self._prepare_padded(mels)
        B = mels.shape[0]
        T = cond['frames'][0]
        n = T * self.hop_length
        i32 = dict(dtype=torch.int32, device=self.device)
        seq_utt = torch.arange(B, **i32).repeat_interleave(n)
        seq_start = torch.arange(n, **i32).repeat(B)
        st = self._alloc_state(B * n, 1)
        check(_lib.load().estts_wavernn_prep(self._prep_args(cond, st, seq_utt, seq_start, B * n), ops._stream()),
              'estts_wavernn_prep')
        f, d, R, F = self.feat_dims, self.aux_dims, self.rnn_dims, self.fc_dims
        up = st['xin'][:, 1:1 + f].reshape(B, n, f)
        aux = torch.cat([st['xin'][:, 1 + f:1 + f + d], st['x1cat'][:, R:], st['x2cat'][:, R:], st['f1cat'][:, F:]], dim=1)
        return up.clone(), aux.reshape(B, n, 4 * d)
