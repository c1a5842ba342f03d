"""CPU oracle (NumPy) for the WaveRNN half of the hot path.

TEST INFRASTRUCTURE ONLY (tests/, smoke(), bench.py's cpu_baseline leg).

Restates WaveRNN/models/fatchord_version.py, utility/distribution.py and the mu-law helpers of
utility/dsp.py.  `w` is the model's state_dict as name -> ndarray (torch layouts).

PARITY STATUS: pinned.  tests/golden/wavernn_*.npz hold outputs of the reference module itself
(imported from /root/reference/WaveRNN in the build container by tests/golden/make_wavernn_goldens.py);
tests/test_oracle_wavernn.py checks this restatement against them.

Sampling: the reference draws from torch's global RNG (Categorical.sample, Tensor.uniform_), which cannot be
reproduced outside torch, so every sampler here takes its uniform numbers as an argument: RAW mode picks the
class by inverse CDF, class = #{k : cdf[k] <= u}; MOL mode follows sample_from_discretized_mix_logistic with
the two uniform draws supplied.  The golden generator patches the reference's two draw sites to use the same
rule with recorded uniforms, so whole trajectories can be compared.
"""
import math

import numpy as np


def _bn(x, w, p, eps=1e-5):
    """nn.BatchNorm1d in eval mode over the channel axis (axis 1 of (B, C, T))."""
    s = w[p + '.weight'] / np.sqrt(w[p + '.running_var'] + eps)
    return x * s[None, :, None] + (w[p + '.bias'] - w[p + '.running_mean'] * s)[None, :, None]


def _conv1d(x, weight, bias=None):
    """nn.Conv1d, stride 1, no padding.  x (B, Cin, T), weight (Cout, Cin, k)."""
    k = weight.shape[2]
    T = x.shape[2] - k + 1
    y = np.zeros((x.shape[0], weight.shape[0], T), dtype=np.result_type(x, weight))
    for j in range(k):
        y += np.einsum('oc,bct->bot', weight[:, :, j], x[:, :, j:j + T])
    if bias is not None:
        y = y + bias[None, :, None]
    return y


def mel_resnet(m, w, res_blocks, p='upsample.resnet'):
    """fatchord_version.py:28-45."""
    x = np.maximum(_bn(_conv1d(m, w[p + '.conv_in.weight']), w, p + '.batch_norm'), 0)
    for i in range(res_blocks):
        q = f'{p}.layers.{i}'
        r = x
        x = np.maximum(_bn(_conv1d(x, w[q + '.conv1.weight']), w, q + '.batch_norm1'), 0)
        x = _bn(_conv1d(x, w[q + '.conv2.weight']), w, q + '.batch_norm2')
        x = x + r
    return _conv1d(x, w[p + '.conv_out.weight'], w[p + '.conv_out.bias'])


def upsample_network(m, w, upsample_factors, res_blocks, pad):
    """fatchord_version.py:61-86.  m (B, feat, T + 2 pad) -> mels (B, T*hop, feat), aux (B, T*hop, res_out)."""
    total = int(np.prod(upsample_factors))
    indent = pad * total
    aux = mel_resnet(m, w, res_blocks)
    aux = np.repeat(aux, total, axis=2)
    x = m
    for j, s in enumerate(upsample_factors):
        x = np.repeat(x, s, axis=2)
        k = w[f'upsample.up_layers.{2 * j + 1}.weight'].reshape(-1)           # (2 s + 1,)
        xp = np.pad(x, ((0, 0), (0, 0), (s, s)))
        y = np.zeros_like(x)
        for t in range(2 * s + 1):
            y += k[t] * xp[:, :, t:t + x.shape[2]]
        x = y
    x = x[:, :, indent:-indent]
    return x.transpose(0, 2, 1), aux.transpose(0, 2, 1)


def _sigmoid(x):
    return 1.0 / (1.0 + np.exp(-x))


def gru_cell(x, h, w, p):
    """torch.nn.GRUCell with the l0 weights of nn.GRU (gate order r, z, n)."""
    gi = x @ w[p + '.weight_ih_l0'].T + w[p + '.bias_ih_l0']
    gh = h @ w[p + '.weight_hh_l0'].T + w[p + '.bias_hh_l0']
    H = h.shape[1]
    r = _sigmoid(gi[:, :H] + gh[:, :H])
    z = _sigmoid(gi[:, H:2 * H] + gh[:, H:2 * H])
    n = np.tanh(gi[:, 2 * H:] + r * gh[:, 2 * H:])
    return (1 - z) * n + z * h


def step_logits(x_prev, m_t, aux_t, h1, h2, w):
    """One iteration of the generate loop body, fatchord_version.py:187-202.  Returns logits, h1, h2."""
    d = aux_t.shape[1] // 4
    a1, a2, a3, a4 = (aux_t[:, d * i:d * (i + 1)] for i in range(4))
    x = np.concatenate([x_prev, m_t, a1], axis=1) @ w['I.weight'].T + w['I.bias']
    h1 = gru_cell(x, h1, w, 'rnn1')
    x = x + h1
    h2 = gru_cell(np.concatenate([x, a2], axis=1), h2, w, 'rnn2')
    x = x + h2
    x = np.maximum(np.concatenate([x, a3], axis=1) @ w['fc1.weight'].T + w['fc1.bias'], 0)
    x = np.maximum(np.concatenate([x, a4], axis=1) @ w['fc2.weight'].T + w['fc2.bias'], 0)
    return x @ w['fc3.weight'].T + w['fc3.bias'], h1, h2


def softmax(x):
    e = np.exp(x - x.max(axis=-1, keepdims=True))
    return e / e.sum(axis=-1, keepdims=True)


def sample_raw(logits, u):
    """Inverse-CDF categorical draw: class = number of cdf entries <= u (clipped to the last class)."""
    cdf = np.cumsum(softmax(logits), axis=-1)
    c = (cdf <= u[:, None]).sum(axis=-1)
    return np.minimum(c, logits.shape[-1] - 1)


def sample_mol(logits, u_mix, u_logistic, log_scale_min=None):
    """utility/distribution.py:87-123 with the two uniform draws supplied.
    logits (B, 3*nr_mix); u_mix (B, nr_mix) in [1e-5, 1-1e-5]; u_logistic (B,)."""
    if log_scale_min is None:
        log_scale_min = float(np.log(1e-14))
    nr = logits.shape[1] // 3
    g = logits[:, :nr] - np.log(-np.log(u_mix))
    k = g.argmax(axis=-1)
    idx = np.arange(logits.shape[0])
    means = logits[idx, nr + k]
    log_scales = np.maximum(logits[idx, 2 * nr + k], log_scale_min)
    x = means + np.exp(log_scales) * (np.log(u_logistic) - np.log(1.0 - u_logistic))
    return np.clip(x, -1.0, 1.0)


def fold_with_overlap(x, target, overlap):
    """fatchord_version.py:272-319.  x (1, L, C) -> (folds, target + 2 overlap, C)."""
    _, total_len, features = x.shape
    num_folds = (total_len - overlap) // (target + overlap)
    extended_len = num_folds * (overlap + target) + overlap
    remaining = total_len - extended_len
    if remaining != 0:
        num_folds += 1
        padding = target + 2 * overlap - remaining
        x = np.concatenate([x, np.zeros((1, padding, features), dtype=x.dtype)], axis=1)
    folded = np.zeros((num_folds, target + 2 * overlap, features), dtype=x.dtype)
    for i in range(num_folds):
        start = i * (target + overlap)
        folded[i] = x[0, start:start + target + 2 * overlap]
    return folded


def xfade_and_unfold(y, target, overlap):
    """fatchord_version.py:321-383 (float64)."""
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
    """utility/dsp.py:100-105."""
    if from_labels:
        y = 2 * y / (2 ** math.log2(mu) - 1.) - 1.
    mu = mu - 1
    return np.sign(y) / mu * ((1 + mu) ** np.abs(y) - 1)


def encode_mu_law(x, mu):
    """utility/dsp.py:94-97."""
    mu = mu - 1
    fx = np.sign(x) * np.log(1 + mu * np.abs(x)) / np.log(1 + mu)
    return np.floor((fx + 1) / 2 * mu + 0.5)


def forward(x, mels, w, hp, dtype=np.float64):
    """Teacher-forced WaveRNN.forward, fatchord_version.py:119-148.  x (B, L), mels (B, feat, L/hop + 2 pad)."""
    w = {k: np.asarray(v, dtype=dtype) for k, v in w.items() if not k.endswith('num_batches_tracked') and k != 'step'}
    up, aux = upsample_network(np.asarray(mels, dtype), w, hp['voc_upsample_factors'], hp['voc_res_blocks'], hp['voc_pad'])
    B, L = x.shape
    H = w['rnn1.weight_hh_l0'].shape[1]
    h1 = np.zeros((B, H), dtype)
    h2 = np.zeros((B, H), dtype)
    out = []
    for t in range(L):
        lg, h1, h2 = step_logits(np.asarray(x[:, t:t + 1], dtype), up[:, t], aux[:, t], h1, h2, w)
        out.append(lg)
    return np.stack(out, axis=1)


def generate(mels, w, hp, uniforms, batched=True, target=None, overlap=None, mu_law=None, mode=None, dtype=np.float64,
             return_trace=False, forced=None):
    """WaveRNN.generate, fatchord_version.py:150-243, with the random draws supplied.

    mels (1, feat, T).  uniforms: RAW: array (steps, folds); MOL: dict {'mix': (steps, folds, nr_mix),
    'logistic': (steps, folds)}.  Returns float64 waveform (and the per-step logits / samples when return_trace).
    forced (steps, folds), optional: sample values fed back as the next step's input instead of the oracle's own
    draws (teacher-forced replay of a recorded trajectory; the returned samples are still the oracle's own).
    """
    mode = mode or hp['voc_mode']
    target = hp['voc_target'] if target is None else target
    overlap = hp['voc_overlap'] if overlap is None else overlap
    mu_law = (hp['mu_law'] if mu_law is None else mu_law) if mode == 'RAW' else False
    w = {k: np.asarray(v, dtype=dtype) for k, v in w.items() if not k.endswith('num_batches_tracked') and k != 'step'}
    hop, pad = hp['hop_length'], hp['voc_pad']
    n_classes = w['fc3.weight'].shape[0]
    mels = np.asarray(mels, dtype)
    wave_len = (mels.shape[-1] - 1) * hop
    m = np.pad(mels, ((0, 0), (0, 0), (pad, pad)))
    up, aux = upsample_network(m, w, hp['voc_upsample_factors'], hp['voc_res_blocks'], pad)
    if batched:
        up = fold_with_overlap(up, target, overlap)
        aux = fold_with_overlap(aux, target, overlap)
    B, L, _ = up.shape
    H = w['rnn1.weight_hh_l0'].shape[1]
    h1 = np.zeros((B, H), dtype)
    h2 = np.zeros((B, H), dtype)
    x = np.zeros((B, 1), dtype)
    out, trace = [], []
    for t in range(L):
        lg, h1, h2 = step_logits(x, up[:, t], aux[:, t], h1, h2, w)
        if mode == 'MOL':
            s = sample_mol(lg, uniforms['mix'][t], uniforms['logistic'][t])
        else:
            c = sample_raw(lg, uniforms[t])
            s = 2 * c.astype(dtype) / (n_classes - 1.) - 1.
        out.append(s)
        x = s[:, None] if forced is None else np.asarray(forced[t], dtype)[:, None]
        if return_trace:
            trace.append(lg)
    y = np.stack(out).T.astype(np.float64)
    raw = y.copy()
    y = xfade_and_unfold(y, target, overlap) if batched else y[0]
    if mu_law:
        y = decode_mu_law(y, n_classes, False)
    fade_out = np.linspace(1, 0, 20 * hop)
    y = y[:wave_len]
    y[-20 * hop:] *= fade_out
    if return_trace:
        return y, np.stack(trace), raw
    return y
