"""Second, independently written CPU restatement (PyTorch ops) of the ForwardTransformer
and GST teacher-forced graphs.

TEST INFRASTRUCTURE ONLY (tests/, smoke(), bench.py's cpu_baseline leg).

Purpose: (1) cross-check oracle/tts_numpy.py, which was written from the same
reference source but with different primitives (here: F.conv1d / F.conv2d /
F.layer_norm / F.batch_norm / torch.softmax / repeat_interleave, there: explicit
NumPy loops); (2) serve as the timed CPU baseline ("port"): op for op what a
TensorFlow-CPU / PyTorch-CPU run of the reference graph executes, in float32 on
all host threads.

Reference locations: TransformerTTS/model/layers.py, model/transformer_utils.py,
pretrained_models/for_text_encoder/models.py:434-456,
model/AutoregressiveTransformer.py:124-181.  PARITY STATUS vs TensorFlow: unpinned
(see oracle/tts_numpy.py header).
"""
import math

import numpy as np
import torch
import torch.nn.functional as F


def _t(w, dtype):
    return {k: torch.as_tensor(np.asarray(v)).to(dtype) for k, v in w.items()}


def _pe(position, d, dtype):
    pos = torch.arange(position, dtype=torch.float64)[:, None]
    i = torch.arange(d)[None, :]
    rates = 1.0 / torch.pow(torch.tensor(10000.0, dtype=torch.float64),
                            (2 * (i // 2)).to(torch.float64) / float(np.float32(d)))
    ang = pos * rates
    pe = torch.where((i % 2) == 0, torch.sin(ang), torch.cos(ang))
    return pe.to(torch.float32).to(dtype)


def _norm(x, w, p, kind):
    if kind == 'layer':
        return F.layer_norm(x, (x.shape[-1],), w[p + '.gamma'], w[p + '.beta'], 1e-6)
    y = F.batch_norm(x.transpose(1, 2), w[p + '.moving_mean'], w[p + '.moving_variance'],
                     w[p + '.gamma'], w[p + '.beta'], False, 0.0, 1e-3)
    return y.transpose(1, 2)


def _conv1d(x, kernel, bias, padding):
    k = kernel.shape[0]
    left = (k - 1) // 2 if padding == 'same' else k - 1
    right = k - 1 - left
    xc = F.pad(x.transpose(1, 2), (left, right))
    y = F.conv1d(xc, kernel.permute(2, 1, 0).contiguous(), bias)
    return y.transpose(1, 2)


def _act(x, name):
    if name == 'relu':
        return torch.relu(x)
    if name == 'tanh':
        return torch.tanh(x)
    return x


def _cnn_res_norm(x, w, p, n_layers, inner, last, padding, kind):
    y = x
    for i in range(n_layers - 1):
        y = _act(_norm(_conv1d(y, w[f'{p}.conv{i}.kernel'], w[f'{p}.conv{i}.bias'], padding), w, f'{p}.norm{i}', kind), inner)
    y = _conv1d(y, w[p + '.last_conv.kernel'], w[p + '.last_conv.bias'], padding)
    y = _act(_norm(y, w, f'{p}.norm{n_layers - 1}', kind), last)
    return _norm(x + y, w, f'{p}.norm{n_layers}', kind)


def _mha(v, k, q_in, mask, w, p, H):
    B, Lq, _ = q_in.shape
    q = F.linear(q_in, w[p + '.wq.kernel'].t(), w[p + '.wq.bias'])
    k = F.linear(k, w[p + '.wk.kernel'].t(), w[p + '.wk.bias'])
    v = F.linear(v, w[p + '.wv.kernel'].t(), w[p + '.wv.bias'])
    d = q.shape[-1]
    dh = d // H
    q = q.view(B, Lq, H, dh).permute(0, 2, 1, 3)
    k = k.view(B, -1, H, dh).permute(0, 2, 3, 1)
    v = v.view(B, -1, H, dh).permute(0, 2, 1, 3)
    logits = torch.matmul(q, k) / math.sqrt(dh)
    if mask is not None:
        logits = logits + mask * -1e9
    pw = torch.softmax(logits, dim=-1)
    ctx = torch.matmul(pw, v).permute(0, 2, 1, 3).reshape(B, Lq, d)
    out = F.linear(torch.cat([q_in, ctx], dim=-1), w[p + '.dense.kernel'].t(), w[p + '.dense.bias'])
    return out, pw


def _ln(x, w, p):
    return F.layer_norm(x, (x.shape[-1],), w[p + '.gamma'], w[p + '.beta'], 1e-6)


def _sarn(x, mask, w, p, H):
    a, pw = _mha(x, x, x, mask, w, p + '.mha', H)
    return _ln(_ln(a, w, p + '.ln') + x, w, p + '.last_ln'), pw


def _ffn(x, w, p):
    y = F.linear(x, w[p + '.d1.kernel'].t(), w[p + '.d1.bias'])
    y = F.linear(y, w[p + '.d2.kernel'].t(), w[p + '.d2.bias'])
    y = torch.relu(_ln(y, w, p + '.ln'))
    return _ln(y + x, w, p + '.last_ln')


def _sa_blocks(x, mask, w, p, d, heads, dense_blocks, max_pos, r=1):
    L = x.shape[1]
    pe = _pe(max_pos, d, x.dtype)
    x = x * math.sqrt(float(np.float32(d))) + w[p + '.pos_encoding_scalar'] * pe[None, :L * r:r, :]
    attn = []
    for i, H in enumerate(heads):
        b = f'{p}.block{i}'
        x, pw = _sarn(x, mask, w, b + '.sarn', H)
        if i < dense_blocks:
            x = _ffn(x, w, b + '.ffn')
        else:
            x = _cnn_res_norm(x, w, b + '.conv', 2, 'relu', 'relu', 'same', 'batch')
        attn.append(pw)
    return x, attn


def forward_transformer_call(w, cfg, tokens, target_durations=None, durations_scalar=1.0, dtype=torch.float32):
    """pretrained_models/for_text_encoder/models.py:434-456 with training=False."""
    with torch.no_grad():
        w = _t(w, dtype)
        tokens = torch.as_tensor(np.asarray(tokens)).long()
        B, N = tokens.shape
        pad = (tokens == 0).to(dtype)
        mask = pad[:, None, None, :]
        x = F.embedding(tokens, w['encoder_prenet.embeddings'])
        x, enc_attn = _sa_blocks(x, mask, w, 'encoder', cfg['encoder_model_dimension'], cfg['encoder_num_heads'],
                                 cfg['encoder_dense_blocks'], cfg['encoder_max_position_encoding'])
        dp = _cnn_res_norm(x, w, 'dur_pred.conv_blocks', 2, 'relu', 'relu', 'same', 'layer')
        dur = torch.relu(F.linear(dp, w['dur_pred.linear.kernel'].t(), w['dur_pred.linear.bias'])) * durations_scalar
        dur = (1.0 - pad)[:, :, None] * dur
        if target_durations is not None:
            reps = torch.round(torch.as_tensor(np.asarray(target_durations)).to(dtype)).long()
        else:
            reps = torch.round(dur[..., 0]).long()
        T = int(reps.sum(dim=1).max())
        mels = torch.zeros(B, T, x.shape[-1], dtype=dtype)
        for b in range(B):
            e = torch.repeat_interleave(x[b], reps[b].clamp(min=0), dim=0)
            mels[b, :e.shape[0]] = e
        emask = (mels.abs().sum(-1) == 0).to(dtype)[:, None, None, :]
        y = torch.relu(F.linear(mels, w['decoder_prenet.d1.kernel'].t(), w['decoder_prenet.d1.bias']))
        y = torch.relu(F.linear(y, w['decoder_prenet.d2.kernel'].t(), w['decoder_prenet.d2.bias']))
        y, dec_attn = _sa_blocks(y, emask, w, 'decoder', cfg['decoder_model_dimension'], cfg['decoder_num_heads'],
                                 cfg['decoder_dense_blocks'], cfg['decoder_max_position_encoding'])
        y = F.linear(y, w['out.kernel'].t(), w['out.bias'])
        y = _cnn_res_norm(y, w, 'decoder_postnet', cfg['postnet_conv_layers'], 'tanh', 'linear', 'same', 'batch')
        return {'mel': y.numpy(), 'duration': dur.numpy(), 'expanded_mask': emask.numpy(),
                'encoder_attention': [a.numpy() for a in enc_attn],
                'decoder_attention': [a.numpy() for a in dec_attn]}


# ---------------------------------------------------------------------------------------------
# GST reference encoder + teacher-forced autoregressive call
# ---------------------------------------------------------------------------------------------
def _conv2d_same_s2(x, kernel, bias, stride):
    # x (B, C, H, W); TF 'SAME': total pad = max((ceil(n/s)-1)*s + k - n, 0), low side gets floor(total/2)
    kh, kw = kernel.shape[0], kernel.shape[1]
    H, W = x.shape[2], x.shape[3]
    ph = max((-(-H // stride) - 1) * stride + kh - H, 0)
    pw = max((-(-W // stride) - 1) * stride + kw - W, 0)
    x = F.pad(x, (pw // 2, pw - pw // 2, ph // 2, ph - ph // 2))
    return F.conv2d(x, kernel.permute(3, 2, 0, 1).contiguous(), bias, stride=stride)


def _gst(x, w, p, n_convs, stride, H):
    B = x.shape[0]
    y = x[:, None, :, :]                                      # (B, 1, T, 80)
    for i in range(n_convs):
        y = _conv2d_same_s2(y, w[f'{p}.conv{i}.kernel'], w[f'{p}.conv{i}.bias'], stride)
        y = F.batch_norm(y, w[f'{p}.norm{i}.moving_mean'], w[f'{p}.norm{i}.moving_variance'],
                         w[f'{p}.norm{i}.gamma'], w[f'{p}.norm{i}.beta'], False, 0.0, 1e-3)
        y = torch.relu(y)
    y = y.permute(0, 2, 3, 1).reshape(B, y.shape[2], -1)      # (B, T', F'*C) channels-last flatten
    u = w[p + '.gru.recurrent_kernel'].shape[0]
    h = torch.zeros(B, u, dtype=y.dtype)
    for t in range(y.shape[1]):
        xm = y[:, t] @ w[p + '.gru.kernel'] + w[p + '.gru.bias'][0]
        hm = h @ w[p + '.gru.recurrent_kernel'] + w[p + '.gru.bias'][1]
        z = torch.sigmoid(xm[:, :u] + hm[:, :u])
        r = torch.sigmoid(xm[:, u:2 * u] + hm[:, u:2 * u])
        hh = torch.tanh(xm[:, 2 * u:] + r * hm[:, 2 * u:])
        h = z * h + (1 - z) * hh
    proj = torch.tanh(F.linear(h, w[p + '.rnn_proj.kernel'].t(), w[p + '.rnn_proj.bias']))[:, None, :]
    tok = torch.tanh(w[p + '.gst_tokens'])[None].expand(B, -1, -1)
    return _mha(tok, tok, proj, None, w, p + '.mha', H)


def autoregressive_call(w, cfg, tokens, targets, spk_embed=None, r=1, dtype=torch.float32):
    """model/AutoregressiveTransformer.py:243-263 with all training flags False."""
    with torch.no_grad():
        w = _t(w, dtype)
        tokens = torch.as_tensor(np.asarray(tokens)).long()
        targets = torch.as_tensor(np.asarray(targets)).to(dtype)
        st = cfg['system_type']
        pad = (tokens == 0).to(dtype)
        x = F.embedding(tokens, w['text_encoder_prenet.embeddings'])
        text_out, text_attn = _sa_blocks(x, pad[:, None, None, :], w, 'text_encoder', cfg['encoder_model_dimension'],
                                         cfg['encoder_num_heads'], cfg['encoder_dense_blocks'],
                                         cfg['encoder_max_position_encoding'])
        N = text_out.shape[1]
        parts, gst_out, gst_attn = [text_out], None, None
        if st in ('speaker_style_text', 'style_text'):
            gst_out, gst_attn = _gst(targets, w, 'style_encoder', len(cfg['ref_encoder_filters']),
                                     cfg['ref_encoder_strides'], cfg['gst_multi_num_heads'])
            parts.append(gst_out.expand(-1, N, -1))
        if st in ('speaker_style_text', 'speaker_text'):
            parts.append(torch.as_tensor(np.asarray(spk_embed)).to(dtype).expand(-1, N, -1))
        enc = torch.cat(parts, dim=2)
        enc_mask = (enc.abs().sum(-1) == 0).to(dtype)[:, None, None, :]
        Tq = targets.shape[1]
        dec_pad = (targets.abs().sum(-1) == 0).to(dtype)[:, None, None, :]
        look = torch.triu(torch.ones(Tq, Tq, dtype=dtype), diagonal=1)
        combined = torch.maximum(dec_pad, look)
        y = torch.relu(F.linear(targets, w['decoder_prenet.d1.kernel'].t(), w['decoder_prenet.d1.bias']))
        y = torch.relu(F.linear(y, w['decoder_prenet.d2.kernel'].t(), w['decoder_prenet.d2.bias']))
        d = cfg['decoder_model_dimension']
        pe = _pe(cfg['decoder_max_position_encoding'], d, dtype)
        y = y * math.sqrt(float(np.float32(d))) + w['decoder.pos_encoding_scalar'] * pe[None, :Tq * r:r, :]
        dec_attn = []
        for i, H in enumerate(cfg['decoder_num_heads']):
            b = f'decoder.block{i}'
            y, _ = _sarn(y, combined, w, b + '.sarn', H)
            a, pw = _mha(enc, enc, y, enc_mask, w, b + '.carn.mha', H)
            y = _ln(a + y, w, b + '.carn.layernorm')
            if i < cfg['decoder_dense_blocks']:
                y = _ffn(y, w, b + '.ffn')
            else:
                y = _cnn_res_norm(y, w, b + '.conv', 2, 'relu', 'relu', 'causal', 'batch')
            dec_attn.append(pw)
        mc = cfg['mel_channels']
        proj = F.linear(y, w['final_proj_mel.kernel'].t(), w['final_proj_mel.bias'])[:, :, :r * mc]
        mel = proj.reshape(proj.shape[0], proj.shape[1] * r, mc)
        stop = F.linear(mel, w['decoder_postnet.stop_linear.kernel'].t(), w['decoder_postnet.stop_linear.bias'])
        final = _cnn_res_norm(mel, w, 'decoder_postnet.conv_blocks', cfg['postnet_conv_layers'], 'tanh', 'linear',
                              'causal', 'batch')
        return {'final_output': final.numpy(), 'mel_linear': mel.numpy(), 'stop_prob': stop.numpy(),
                'decoder_attention': [a.numpy() for a in dec_attn],
                'text_encoder_attention': [a.numpy() for a in text_attn],
                'gst_encoder_attention': None if gst_attn is None else gst_attn.numpy(),
                'gst_output': None if gst_out is None else gst_out.numpy(),
                'text_enc_output': text_out.numpy(), 'decoder_output': y.numpy()}
