"""CPU oracle for one ForwardTransformer / AutoregressiveTransformer training step (float64 PyTorch autograd).

TEST INFRASTRUCTURE ONLY (tests/, bench.py's cpu_baseline leg).

Restates ForwardTransformer._train_step (TransformerTTS/pretrained_models/for_text_encoder/models.py:389-404):
`__call__(training=True)` -> 3 * masked MAE(mel) + masked MAE(duration) (utils/losses.py:41-47,71-78) -> gradients ->
Keras Adam(beta_1 0.9, beta_2 0.98, epsilon 1e-9) (utils/config_manager.py:172-176).  Training-mode semantics restated
from Keras: BatchNormalization uses the batch mean / biased variance over (batch, time) -- padded frames included -- and
moves the moving statistics with momentum 0.99; the CNNResNorm inside a SelfAttentionConvBlock is called without
`training=` (layers.py:211) and inherits training=True from the enclosing call; Dropout keeps with probability
1 - rate and scales by 1 / (1 - rate); HeadDrop (layers.py:74-97) zeroes whole heads per sample and rescales.
Random masks are INPUTS here (`masks`: list consumed in call order; `head_keep`: list of [B, H] arrays), so that the
device path and this oracle can be run on identical draws.

autoregressive_train_step restates the TTS half of train_models_step (TransformerTTS/model/traning_steps.py:22-56) with the
three `training=` flags True (the shipped configuration): model/AutoregressiveTransformer.py:124-181 on top of
model/layers.py:258-376 (cross-attention blocks), :526-585 (reference encoder: its BatchNormalization layers are called
without `training=` and inherit the style encoder's), utils/losses.py:4-21,41-47.

PARITY STATUS vs TensorFlow: unpinned (TensorFlow is not installable here); gradients come from autograd over the
same forward restatement as oracle/tts_torch.py.
"""
import math

import numpy as np
import torch
import torch.nn.functional as F

from .tts_torch import _act, _conv1d, _pe


class _Ctx:
    def __init__(self, masks, head_keep, new_stats, gates=None):
        self.masks = list(masks) if masks is not None else None
        self.head_keep = list(head_keep) if head_keep is not None else None
        self.new_stats = new_stats
        self.gates = list(gates) if gates is not None else None
        self.training = True        # False: the enclosing sub-network runs with Keras' training=False

    def relu(self, x):
        """ReLU, or -- when the branches another implementation took are given -- the linear piece x * gate it evaluated, so
        that both differentiate the same piecewise-linear function even where an input sits within rounding of zero."""
        if self.gates is None:
            return torch.relu(x)
        g = torch.as_tensor(np.asarray(self.gates.pop(0)), dtype=x.dtype).reshape(x.shape)
        return x * g

    def act(self, x, name):
        return self.relu(x) if name == 'relu' else _act(x, name)

    def dropout(self, x):
        if self.masks is None:
            return x
        m = self.masks.pop(0)           # (an inference-mode sub-network still owns its place in the mask list)
        if not self.training:
            return x
        return x * torch.as_tensor(np.asarray(m), dtype=x.dtype).reshape(x.shape)

    def heads(self, ctx4):
        if self.head_keep is None or not self.training:
            return ctx4
        k = torch.as_tensor(np.asarray(self.head_keep.pop(0)), dtype=ctx4.dtype)
        return ctx4 * k[:, :, None, None]


def _ln(x, w, p):
    return F.layer_norm(x, (x.shape[-1],), w[p + '.gamma'], w[p + '.beta'], 1e-6)


def _bn_train(x, w, p, c):
    """Keras BatchNormalization(training=True) on (B, T, C): statistics over (B, T), eps 1e-3, momentum 0.99."""
    if not c.training:
        return (x - w[p + '.moving_mean']) / torch.sqrt(w[p + '.moving_variance'] + 1e-3) * w[p + '.gamma'] + w[p + '.beta']
    mean = x.mean(dim=(0, 1))
    var = x.var(dim=(0, 1), unbiased=False)
    c.new_stats[p + '.moving_mean'] = (w[p + '.moving_mean'] * 0.99 + mean.detach() * 0.01).numpy()
    c.new_stats[p + '.moving_variance'] = (w[p + '.moving_variance'] * 0.99 + var.detach() * 0.01).numpy()
    return (x - mean) / torch.sqrt(var + 1e-3) * w[p + '.gamma'] + w[p + '.beta']


def _cnn_res_norm(x, w, p, n_layers, inner, last, padding, kind, c):
    norm = (lambda t, q: _ln(t, w, q)) if kind == 'layer' else (lambda t, q: _bn_train(t, w, q, c))
    y = x
    for i in range(n_layers - 1):
        y = c.act(norm(_conv1d(y, w[f'{p}.conv{i}.kernel'], w[f'{p}.conv{i}.bias'], padding), f'{p}.norm{i}'), inner)
    y = _conv1d(y, w[p + '.last_conv.kernel'], w[p + '.last_conv.bias'], padding)
    y = c.act(norm(y, f'{p}.norm{n_layers - 1}'), last)
    return norm(x + y, f'{p}.norm{n_layers}')


def _mha(x, mask, w, p, H, c):
    B, L, _ = x.shape
    q = F.linear(x, w[p + '.wq.kernel'].t(), w[p + '.wq.bias'])
    k = F.linear(x, w[p + '.wk.kernel'].t(), w[p + '.wk.bias'])
    v = F.linear(x, w[p + '.wv.kernel'].t(), w[p + '.wv.bias'])
    d = q.shape[-1]
    dh = d // H
    q = q.view(B, L, H, dh).permute(0, 2, 1, 3)
    k = k.view(B, L, H, dh).permute(0, 2, 3, 1)
    v = v.view(B, L, H, dh).permute(0, 2, 1, 3)
    logits = torch.matmul(q, k) / math.sqrt(dh) + mask * -1e9
    pw = torch.softmax(logits, dim=-1)
    ctx = c.heads(torch.matmul(pw, v)).permute(0, 2, 1, 3).reshape(B, L, d)
    return F.linear(torch.cat([x, ctx], dim=-1), w[p + '.dense.kernel'].t(), w[p + '.dense.bias']), pw


def _sa_blocks(x, mask, w, p, d, heads, dense_blocks, max_pos, c):
    L = x.shape[1]
    pe = _pe(max_pos, d, x.dtype)
    x = x * math.sqrt(float(np.float32(d))) + w[p + '.pos_encoding_scalar'] * pe[None, :L, :]
    x = c.dropout(x)
    attn = []
    for i, H in enumerate(heads):
        b = f'{p}.block{i}'
        a, pw = _mha(x, mask, w, b + '.sarn.mha', H, c)
        x = _ln(c.dropout(_ln(a, w, b + '.sarn.ln')) + x, w, b + '.sarn.last_ln')
        if i < dense_blocks:
            y = F.linear(x, w[b + '.ffn.d1.kernel'].t(), w[b + '.ffn.d1.bias'])
            y = F.linear(y, w[b + '.ffn.d2.kernel'].t(), w[b + '.ffn.d2.bias'])
            y = c.dropout(c.relu(_ln(y, w, b + '.ffn.ln')))
            x = _ln(y + x, w, b + '.ffn.last_ln')
        else:
            x = _cnn_res_norm(x, w, b + '.conv', 2, 'relu', 'relu', 'same', 'batch', c)
        attn.append(pw)
    return x, attn


def masked_mae(targets, pred, sign=None):
    """utils/losses.py:41-47: MeanAbsoluteError with sample_weight = any(target != 0), SUM_OVER_BATCH_SIZE.
    sign: optional sign(pred - targets) as another implementation evaluated it (see _Ctx.relu)."""
    wgt = (targets != 0).any(dim=-1).to(pred.dtype)
    diff = (targets - pred).abs() if sign is None else (pred - targets) * torch.as_tensor(np.asarray(sign), dtype=pred.dtype).reshape(pred.shape)
    per = diff.mean(dim=-1) * wgt
    return per.sum() / per.numel()


def forward_train_step(w, cfg, tokens, mel, durations, masks=None, head_keep=None, dtype=torch.float64, prenet_masks=None,
                       gates=None):
    """One forward + backward.  Returns dict(loss, mel_loss, dur_loss, mel, duration, grads {name: ndarray},
    new_stats {moving statistic name: ndarray}); weights are NOT updated (see adam_update).
    gates: optional list, in forward order, of the branch taken at every ReLU (arrays of 0 / 1, one per ReLU: encoder blocks,
    duration predictor incl. its final Dense(1, relu), decoder prenet, decoder blocks) followed by the two MAE sign arrays
    (mel, duration): the piecewise-linear pieces a device evaluated, which this oracle then differentiates instead of its own
    (its own differ wherever an input is within the device's rounding error of a kink)."""
    dt = dtype
    wt = {}
    for k, v in w.items():
        t = torch.as_tensor(np.asarray(v)).to(dt)
        trainable = not (k.endswith('moving_mean') or k.endswith('moving_variance') or k.endswith('pos_encoding_scalar'))
        wt[k] = t.clone().requires_grad_(trainable)
    signs = None
    if gates is not None:
        gates, signs = list(gates[:-2]), gates[-2:]
    c = _Ctx(masks, head_keep, {}, gates)
    tokens = torch.as_tensor(np.asarray(tokens)).long()
    B, N = tokens.shape
    pad = (tokens == 0).to(dt)
    x = F.embedding(tokens, wt['encoder_prenet.embeddings'])
    x, enc_attn = _sa_blocks(x, pad[:, None, None, :], wt, 'encoder', cfg['encoder_model_dimension'], cfg['encoder_num_heads'],
                             cfg['encoder_dense_blocks'], cfg['encoder_max_position_encoding'], c)
    dp = _cnn_res_norm(x, wt, 'dur_pred.conv_blocks', 2, 'relu', 'relu', 'same', 'layer', c)
    dur = c.relu(F.linear(dp, wt['dur_pred.linear.kernel'].t(), wt['dur_pred.linear.bias']))
    dur = (1.0 - pad)[:, :, None] * dur
    td = torch.as_tensor(np.asarray(durations)).to(dt).reshape(B, N, 1)
    reps = torch.round(td[..., 0]).long().clamp(min=0)
    T = int(reps.sum(dim=1).max())
    rows = []
    for b in range(B):
        e = torch.repeat_interleave(x[b], reps[b], dim=0)
        rows.append(F.pad(e, (0, 0, 0, T - e.shape[0])))
    mels = torch.stack(rows)
    emask = (mels.detach().abs().sum(-1) == 0).to(dt)[:, None, None, :]
    # DecoderPrenet (layers.py:394-402): dropout after each relu, always on; prenet_masks = two scaled keep masks or None
    y = c.relu(F.linear(mels, wt['decoder_prenet.d1.kernel'].t(), wt['decoder_prenet.d1.bias']))
    if prenet_masks is not None:
        y = y * torch.as_tensor(np.asarray(prenet_masks[0]), dtype=dt).reshape(y.shape)
    y = c.relu(F.linear(y, wt['decoder_prenet.d2.kernel'].t(), wt['decoder_prenet.d2.bias']))
    if prenet_masks is not None:
        y = y * torch.as_tensor(np.asarray(prenet_masks[1]), dtype=dt).reshape(y.shape)
    y, dec_attn = _sa_blocks(y, emask, wt, 'decoder', cfg['decoder_model_dimension'], cfg['decoder_num_heads'],
                             cfg['decoder_dense_blocks'], cfg['decoder_max_position_encoding'], c)
    y = F.linear(y, wt['out.kernel'].t(), wt['out.bias'])
    y = _cnn_res_norm(y, wt, 'decoder_postnet', cfg['postnet_conv_layers'], 'tanh', 'linear', 'same', 'batch', c)
    mel_t = torch.as_tensor(np.asarray(mel)).to(dt)
    l_mel = masked_mae(mel_t, y[:, :mel_t.shape[1]], None if signs is None else signs[0])
    l_dur = masked_mae(td, dur, None if signs is None else signs[1])
    if c.gates:
        raise ValueError(f'{len(c.gates)} gate arrays left over: the gate list does not match this model')
    loss = 3.0 * l_mel + 1.0 * l_dur
    loss.backward()
    grads = {k: (t.grad.numpy() if t.grad is not None else np.zeros(t.shape)) for k, t in wt.items() if t.requires_grad}
    return {'loss': float(loss.detach()), 'mel_loss': float(l_mel.detach()), 'dur_loss': float(l_dur.detach()), 'mel': y.detach().numpy(),
            'duration': dur.detach().numpy(), 'grads': grads, 'new_stats': c.new_stats,
            'encoder_attention': [a.detach().numpy() for a in enc_attn], 'decoder_attention': [a.detach().numpy() for a in dec_attn]}


def _mha_cross(kv_in, q_in, mask, w, p, H, c):
    """MultiHeadAttention.call(v, k, q_in, mask) (layers.py:127-147) with k = v = kv_in, head drop from the context."""
    B, Lq, _ = q_in.shape
    q = F.linear(q_in, w[p + '.wq.kernel'].t(), w[p + '.wq.bias'])
    k = F.linear(kv_in, w[p + '.wk.kernel'].t(), w[p + '.wk.bias'])
    v = F.linear(kv_in, w[p + '.wv.kernel'].t(), w[p + '.wv.bias'])
    d = q.shape[-1]
    dh = d // H
    q = q.view(B, Lq, H, dh).permute(0, 2, 1, 3)
    k = k.view(B, -1, H, dh).permute(0, 2, 3, 1)
    v = v.view(B, -1, H, dh).permute(0, 2, 1, 3)
    logits = torch.matmul(q, k) / math.sqrt(dh)
    if mask is not None:
        logits = logits + mask * -1e9
    pw = torch.softmax(logits, dim=-1)
    ctx = c.heads(torch.matmul(pw, v)).permute(0, 2, 1, 3).reshape(B, Lq, d)
    return F.linear(torch.cat([q_in, ctx], dim=-1), w[p + '.dense.kernel'].t(), w[p + '.dense.bias']), pw


def _bn_train_nd(x, w, p, c, dims):
    """Keras BatchNormalization(training=True) with statistics over `dims` (channels last)."""
    if not c.training:
        return (x - w[p + '.moving_mean']) / torch.sqrt(w[p + '.moving_variance'] + 1e-3) * w[p + '.gamma'] + w[p + '.beta']
    mean = x.mean(dim=dims)
    var = x.var(dim=dims, unbiased=False)
    # (a layer that runs twice in one step, the style loss's second pass, moves its statistics twice)
    mm = c.new_stats.get(p + '.moving_mean', w[p + '.moving_mean'].detach().numpy())
    mv = c.new_stats.get(p + '.moving_variance', w[p + '.moving_variance'].detach().numpy())
    c.new_stats[p + '.moving_mean'] = np.asarray(mm) * 0.99 + mean.detach().numpy() * 0.01
    c.new_stats[p + '.moving_variance'] = np.asarray(mv) * 0.99 + var.detach().numpy() * 0.01
    return (x - mean) / torch.sqrt(var + 1e-3) * w[p + '.gamma'] + w[p + '.beta']


def _gst_train(x, w, p, n_convs, stride, H, c):
    """ReferenceEncoderGST.call (layers.py:561-585) in training mode; x (B, T, mel)."""
    from .tts_torch import _conv2d_same_s2
    B = x.shape[0]
    y = x[:, None, :, :]
    for i in range(n_convs):
        y = _conv2d_same_s2(y, w[f'{p}.conv{i}.kernel'], w[f'{p}.conv{i}.bias'], stride)       # (B, C, H, W)
        y = c.relu(_bn_train_nd(y.permute(0, 2, 3, 1), w, f'{p}.norm{i}', c, (0, 1, 2))).permute(0, 3, 1, 2)   # (gates are NHWC)
    y = y.permute(0, 2, 3, 1).reshape(B, y.shape[2], -1)
    u = w[p + '.gru.recurrent_kernel'].shape[0]
    h = torch.zeros(B, u, dtype=y.dtype)
    for t in range(y.shape[1]):
        xm = y[:, t] @ w[p + '.gru.kernel'] + w[p + '.gru.bias'][0]
        hm = h @ w[p + '.gru.recurrent_kernel'] + w[p + '.gru.bias'][1]
        z = torch.sigmoid(xm[:, :u] + hm[:, :u])
        rr = torch.sigmoid(xm[:, u:2 * u] + hm[:, u:2 * u])
        hh = torch.tanh(xm[:, 2 * u:] + rr * hm[:, 2 * u:])
        h = z * h + (1 - z) * hh
    proj = torch.tanh(F.linear(h, w[p + '.rnn_proj.kernel'].t(), w[p + '.rnn_proj.bias']))[:, None, :]
    tok = torch.tanh(w[p + '.gst_tokens'])[None].expand(B, -1, -1)
    return _mha_cross(tok, proj, None, w, p + '.mha', H, c)


def scaled_crossentropy(targets, logits, index=2, scaling=1.0):
    """utils/losses.py:4-21: SparseCategoricalCrossentropy(from_logits) with sample weights (target != 0) + (target == index)
    (scaling - 1), SUM_OVER_BATCH_SIZE."""
    ce = F.cross_entropy(logits.reshape(-1, logits.shape[-1]), targets.reshape(-1).long(), reduction='none')
    wgt = (targets != 0).to(logits.dtype) + (targets == index).to(logits.dtype) * (scaling - 1.0)
    return (ce * wgt.reshape(-1)).sum() / ce.numel()


def autoregressive_train_step(w, cfg, tokens, tar_mel, tar_real, tar_stop, spk_embed=None, r=1, masks=None, head_keep=None,
                              prenet_masks=None, gates=None, stop_scaling=1.0, dtype=torch.float64, use_style_loss=False,
                              train_text_encoder=True, train_style_encoder=True, train_decoder=True):
    """One forward + backward of the TTS half of train_models_step (use_style_loss: traning_steps.py:41-49).  tar_mel: the decoder input (every r-th frame of
    tar[:, :-1]), tar_real / tar_stop: tar[:, 1:] / stop[:, 1:] (AutoregressiveTransformer.input_reshape).
    masks / head_keep / prenet_masks / gates as in forward_train_step; gates end with the sign arrays of the two MAE losses
    (post-net output, linear output).  Returns losses, outputs, grads {name: ndarray} and the new moving statistics."""
    dt = dtype
    wt = {}
    for k, v in w.items():
        t = torch.as_tensor(np.asarray(v)).to(dt)
        trainable = not (k.endswith('moving_mean') or k.endswith('moving_variance') or k.endswith('pos_encoding_scalar'))
        wt[k] = t.clone().requires_grad_(trainable)
    signs = None
    if gates is not None:
        gates, signs = list(gates[:-2]), gates[-2:]
    c = _Ctx(masks, head_keep, {}, gates)
    st = cfg['system_type']
    tokens = torch.as_tensor(np.asarray(tokens)).long()
    tar_mel = torch.as_tensor(np.asarray(tar_mel)).to(dt)
    tar_real = torch.as_tensor(np.asarray(tar_real)).to(dt)
    tar_stop = torch.as_tensor(np.asarray(tar_stop)).long()
    B, N = tokens.shape
    pad = (tokens == 0).to(dt)
    x = F.embedding(tokens, wt['text_encoder_prenet.embeddings'])
    c.training = train_text_encoder
    text_out, text_attn = _sa_blocks(x, pad[:, None, None, :], wt, 'text_encoder', cfg['encoder_model_dimension'],
                                     cfg['encoder_num_heads'], cfg['encoder_dense_blocks'], cfg['encoder_max_position_encoding'], c)
    parts, gst_out = [text_out], None
    c.training = train_style_encoder
    if st in ('speaker_style_text', 'style_text'):
        gst_out, _ = _gst_train(tar_mel, wt, 'style_encoder', len(cfg['ref_encoder_filters']), cfg['ref_encoder_strides'],
                                cfg['gst_multi_num_heads'], c)
        parts.append(gst_out.expand(-1, N, -1))
    if st in ('speaker_style_text', 'speaker_text'):
        parts.append(torch.as_tensor(np.asarray(spk_embed)).to(dt).reshape(B, 1, -1).expand(-1, N, -1))
    enc = torch.cat(parts, dim=2)
    enc_mask = (enc.detach().abs().sum(-1) == 0).to(dt)[:, None, None, :]
    Tq = tar_mel.shape[1]
    dec_pad = (tar_mel.abs().sum(-1) == 0).to(dt)[:, None, None, :]
    look = torch.triu(torch.ones(Tq, Tq, dtype=dt), diagonal=1)
    combined = torch.maximum(dec_pad, look)
    y = c.relu(F.linear(tar_mel, wt['decoder_prenet.d1.kernel'].t(), wt['decoder_prenet.d1.bias']))
    if prenet_masks is not None:
        y = y * torch.as_tensor(np.asarray(prenet_masks[0]), dtype=dt).reshape(y.shape)
    y = c.relu(F.linear(y, wt['decoder_prenet.d2.kernel'].t(), wt['decoder_prenet.d2.bias']))
    if prenet_masks is not None:
        y = y * torch.as_tensor(np.asarray(prenet_masks[1]), dtype=dt).reshape(y.shape)
    d = cfg['decoder_model_dimension']
    pe = _pe(cfg['decoder_max_position_encoding'], d, dt)
    y = y * math.sqrt(float(np.float32(d))) + wt['decoder.pos_encoding_scalar'] * pe[None, :Tq * r:r, :]
    c.training = train_decoder
    y = c.dropout(y)
    dec_attn = []
    for i, H in enumerate(cfg['decoder_num_heads']):
        b = f'decoder.block{i}'
        a, _ = _mha(y, combined, wt, b + '.sarn.mha', H, c)
        y = _ln(c.dropout(_ln(a, wt, b + '.sarn.ln')) + y, wt, b + '.sarn.last_ln')
        a, pw = _mha_cross(enc, y, enc_mask, wt, b + '.carn.mha', H, c)
        y = _ln(c.dropout(a) + y, wt, b + '.carn.layernorm')
        if i < cfg['decoder_dense_blocks']:
            z = F.linear(y, wt[b + '.ffn.d1.kernel'].t(), wt[b + '.ffn.d1.bias'])
            z = F.linear(z, wt[b + '.ffn.d2.kernel'].t(), wt[b + '.ffn.d2.bias'])
            z = c.dropout(c.relu(_ln(z, wt, b + '.ffn.ln')))
            y = _ln(z + y, wt, b + '.ffn.last_ln')
        else:
            y = _cnn_res_norm(y, wt, b + '.conv', 2, 'relu', 'relu', 'causal', 'batch', c)
        dec_attn.append(pw)
    mc = cfg['mel_channels']
    proj = F.linear(y, wt['final_proj_mel.kernel'].t(), wt['final_proj_mel.bias'])[:, :, :r * mc]
    mel = proj.reshape(B, Tq * r, mc)
    stop = F.linear(mel, wt['decoder_postnet.stop_linear.kernel'].t(), wt['decoder_postnet.stop_linear.bias'])
    final = _cnn_res_norm(mel, wt, 'decoder_postnet.conv_blocks', cfg['postnet_conv_layers'], 'tanh', 'linear', 'causal', 'batch', c)
    l_style = None
    c.training = train_style_encoder
    if use_style_loss:
        gst2, _ = _gst_train(final, wt, 'style_encoder', len(cfg['ref_encoder_filters']), cfg['ref_encoder_strides'],
                             cfg['gst_multi_num_heads'], c)
        l_style = ((gst2 - gst_out) ** 2).mean()
    L = tar_real.shape[1]
    l_out = masked_mae(tar_real, final[:, :L], None if signs is None else signs[0])
    l_stop = scaled_crossentropy(tar_stop, stop[:, :L], 2, stop_scaling)
    l_lin = masked_mae(tar_real, mel[:, :L], None if signs is None else signs[1])
    if c.gates:
        raise ValueError(f'{len(c.gates)} gate arrays left over: the gate list does not match this model')
    loss = l_out + l_stop + l_lin
    if l_style is not None:
        loss = loss + l_style
    loss.backward()
    grads = {k: (t.grad.numpy() if t.grad is not None else np.zeros(t.shape)) for k, t in wt.items() if t.requires_grad}
    return {'loss': float(loss.detach()), 'style_loss': None if l_style is None else float(l_style.detach()),
            'losses': {'output': float(l_out.detach()), 'stop_prob': float(l_stop.detach()), 'mel_linear': float(l_lin.detach())},
            'final_output': final.detach().numpy(), 'mel_linear': mel.detach().numpy(), 'stop_prob': stop.detach().numpy(),
            'gst_output': None if gst_out is None else gst_out.detach().numpy(), 'text_enc_output': text_out.detach().numpy(),
            'decoder_attention': [a.detach().numpy() for a in dec_attn], 'grads': grads, 'new_stats': c.new_stats}


def adam_update(w, grads, state, lr, t, beta1=0.9, beta2=0.98, eps=1e-9):
    """Keras Adam, step t (1-based); state = {name: (m, v)} updated in place; returns the new weights."""
    lr_t = lr * math.sqrt(1.0 - beta2 ** t) / (1.0 - beta1 ** t)
    out = dict(w)
    for k, g in grads.items():
        m, v = state.get(k, (np.zeros_like(g), np.zeros_like(g)))
        m = beta1 * m + (1 - beta1) * g
        v = beta2 * v + (1 - beta2) * g * g
        state[k] = (m, v)
        out[k] = np.asarray(w[k], np.float64) - lr_t * m / (np.sqrt(v) + eps)
    return out
