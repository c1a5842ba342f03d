"""CPU oracle for one WaveRNN training step (float64 PyTorch autograd over a restatement of the model).

TEST INFRASTRUCTURE ONLY (tests/, bench.py's cpu_baseline leg).

Restates, with plain torch ops and explicit loops, what WaveRNN/train_wavernn.py:27-42 runs per batch:
  * WaveRNN.forward in training mode (WaveRNN/models/fatchord_version.py:119-148): MelResNet with BatchNorm1d batch
    statistics (:39-45, :18-25), the repeat + box-filter upsampling chain (:54-86), I -> GRU -> GRU -> fc1 -> fc2 -> fc3
    with the four aux slices; nn.GRU restated step by step with torch's gate order (r, z, n)
  * F.cross_entropy over the class axis (RAW) or discretized_mix_logistic_loss (WaveRNN/utility/distribution.py:16-84)
  * torch.optim.Adam (betas 0.9 / 0.999, eps 1e-8, train_wavernn.py:104)

PINNED: tests/test_oracle_wavernn.py checks loss, every gradient, the BatchNorm running statistics and the parameters
after the optimizer step against tests/golden/wavernn_train_{raw,mol}_small.npz, which the reference module itself
produced (tests/golden/make_wavernn_train_goldens.py).
"""
import math

import numpy as np
import torch
import torch.nn.functional as F


def _bn_train(x, w, p, new_stats, momentum=0.1, eps=1e-5):
    """nn.BatchNorm1d in training mode on (B, C, L): biased batch variance to normalise, unbiased one into running_var."""
    mean = x.mean(dim=(0, 2))
    var = x.var(dim=(0, 2), unbiased=False)
    n = x.shape[0] * x.shape[2]
    new_stats[p + '.running_mean'] = ((1 - momentum) * w[p + '.running_mean'] + momentum * mean.detach()).numpy()
    new_stats[p + '.running_var'] = ((1 - momentum) * w[p + '.running_var'] + momentum * var.detach() * n / (n - 1)).numpy()
    xh = (x - mean[None, :, None]) / torch.sqrt(var[None, :, None] + eps)
    return xh * w[p + '.weight'][None, :, None] + w[p + '.bias'][None, :, None]


def upsample(w, mels, scales, pad, res_blocks, new_stats):
    """UpsampleNetwork.forward (fatchord_version.py:79-86).  mels (B, feat, T + 2 pad) -> (B, T hop, feat), (B, T hop, res_out)."""
    p = 'upsample.resnet'
    x = F.conv1d(mels, w[p + '.conv_in.weight'])
    x = torch.relu(_bn_train(x, w, p + '.batch_norm', new_stats))
    for i in range(res_blocks):
        q = f'{p}.layers.{i}'
        r = x
        x = F.conv1d(x, w[q + '.conv1.weight'])
        x = torch.relu(_bn_train(x, w, q + '.batch_norm1', new_stats))
        x = F.conv1d(x, w[q + '.conv2.weight'])
        x = _bn_train(x, w, q + '.batch_norm2', new_stats) + r
    aux = F.conv1d(x, w[p + '.conv_out.weight'], w[p + '.conv_out.bias'])
    hop = int(np.prod(scales))
    aux = torch.repeat_interleave(aux, hop, dim=2)
    m = mels
    for j, s in enumerate(scales):
        m = torch.repeat_interleave(m, s, dim=2)
        k = w[f'upsample.up_layers.{2 * j + 1}.weight'].reshape(1, 1, -1)
        B, C, L = m.shape
        m = F.conv1d(m.reshape(B * C, 1, L), k, padding=s).reshape(B, C, L)
    indent = pad * hop
    m = m[:, :, indent:-indent]
    return m.transpose(1, 2), aux.transpose(1, 2)


def _gru(x, w, p, H):
    """nn.GRU(batch_first=True), one layer, zero initial state: r, z, n gate order."""
    B, T, _ = x.shape
    gi = x @ w[p + '.weight_ih_l0'].t() + w[p + '.bias_ih_l0']
    h = torch.zeros(B, H, dtype=x.dtype)
    out = []
    for t in range(T):
        gh = h @ w[p + '.weight_hh_l0'].t() + w[p + '.bias_hh_l0']
        r = torch.sigmoid(gi[:, t, :H] + gh[:, :H])
        z = torch.sigmoid(gi[:, t, H:2 * H] + gh[:, H:2 * H])
        n = torch.tanh(gi[:, t, 2 * H:] + r * gh[:, 2 * H:])
        h = (1 - z) * n + z * h
        out.append(h)
    return torch.stack(out, dim=1)


def forward(w, hp, x, mels, new_stats):
    """WaveRNN.forward (fatchord_version.py:119-148), training mode.  w: dict of tensors."""
    R, d = hp['voc_rnn_dims'], hp['voc_res_out_dims'] // 4
    m, aux = upsample(w, mels, hp['voc_upsample_factors'], hp['voc_pad'], hp['voc_res_blocks'], new_stats)
    a1, a2, a3, a4 = (aux[:, :, i * d:(i + 1) * d] for i in range(4))
    h = torch.cat([x[:, :, None], m, a1], dim=2)
    h = h @ w['I.weight'].t() + w['I.bias']
    res = h
    h = _gru(h, w, 'rnn1', R) + res
    res = h
    h = _gru(torch.cat([h, a2], dim=2), w, 'rnn2', R) + res
    h = torch.relu(torch.cat([h, a3], dim=2) @ w['fc1.weight'].t() + w['fc1.bias'])
    h = torch.relu(torch.cat([h, a4], dim=2) @ w['fc2.weight'].t() + w['fc2.bias'])
    return h @ w['fc3.weight'].t() + w['fc3.bias']


def mol_loss(y_hat, y, num_classes=65536, log_scale_min=None):
    """discretized_mix_logistic_loss (utility/distribution.py:16-84), reduce=True.  y_hat (B, T, 3 nr_mix), y (B, T)."""
    if log_scale_min is None:
        log_scale_min = float(np.log(1e-14))
    nr = y_hat.shape[-1] // 3
    logit_probs, means = y_hat[..., :nr], y_hat[..., nr:2 * nr]
    log_scales = torch.clamp(y_hat[..., 2 * nr:], min=log_scale_min)
    yy = y[..., None].expand_as(means)
    c = yy - means
    inv = torch.exp(-log_scales)
    plus_in = inv * (c + 1. / (num_classes - 1))
    min_in = inv * (c - 1. / (num_classes - 1))
    cdf_delta = torch.sigmoid(plus_in) - torch.sigmoid(min_in)
    log_cdf_plus = plus_in - F.softplus(plus_in)
    log_one_minus_cdf_min = -F.softplus(min_in)
    mid_in = inv * c
    log_pdf_mid = mid_in - log_scales - 2. * F.softplus(mid_in)
    iic = (cdf_delta > 1e-5).to(y_hat.dtype)
    iio = iic * torch.log(torch.clamp(cdf_delta, min=1e-12)) + (1. - iic) * (log_pdf_mid - np.log((num_classes - 1) / 2))
    ic = (yy > 0.999).to(y_hat.dtype)
    io = ic * log_one_minus_cdf_min + (1. - ic) * iio
    cond = (yy < -0.999).to(y_hat.dtype)
    log_probs = cond * log_cdf_plus + (1. - cond) * io + F.log_softmax(logit_probs, -1)
    return -torch.mean(torch.logsumexp(log_probs, dim=-1))


def train_step(weights, hp, x, y, mels, dtype=torch.float64):
    """One forward + backward.  weights: state_dict-style dict of arrays.  Returns loss, logits, grads, new running stats."""
    w = {}
    for k, v in weights.items():
        if k == 'step' or k.endswith('num_batches_tracked'):
            continue
        t = torch.as_tensor(np.asarray(v)).to(dtype)
        w[k] = t.clone().requires_grad_('running' not in k)
    new_stats = {}
    xt = torch.as_tensor(np.asarray(x)).to(dtype)
    mt = torch.as_tensor(np.asarray(mels)).to(dtype)
    logits = forward(w, hp, xt, mt, new_stats)
    if hp['voc_mode'] == 'RAW':
        yt = torch.as_tensor(np.asarray(y)).long()
        loss = F.cross_entropy(logits.reshape(-1, logits.shape[-1]), yt.reshape(-1))
    else:
        loss = mol_loss(logits, torch.as_tensor(np.asarray(y)).to(dtype))
    loss.backward()
    grads = {k: t.grad.numpy() for k, t in w.items() if t.requires_grad and t.grad is not None}
    return {'loss': float(loss.detach()), 'logits': logits.detach().numpy(), 'grads': grads, 'new_stats': new_stats}


def adam_update(weights, grads, state, lr, t, beta1=0.9, beta2=0.999, eps=1e-8):
    """torch.optim.Adam, step t (1-based), state = {name: (m, v)}; returns the new weights."""
    out = dict(weights)
    for k, g in grads.items():
        m, v = state.get(k, (np.zeros_like(g), np.zeros_like(g)))
        m = beta1 * m + (1 - beta1) * g
        v = beta2 * v + (1 - beta2) * g * g
        state[k] = (m, v)
        denom = np.sqrt(v) / math.sqrt(1 - beta2 ** t) + eps
        out[k] = np.asarray(weights[k], np.float64) - lr / (1 - beta1 ** t) * m / denom
    return out
