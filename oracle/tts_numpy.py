"""CPU oracle (NumPy) for the TransformerTTS half of the hot path.

TEST INFRASTRUCTURE ONLY.  Nothing in the product package imports this file;
only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may.

This is a restatement, in plain NumPy, of the arithmetic the reference runs
through TensorFlow 2.2 Keras layers.  TensorFlow is not installable here, so
the Keras layer semantics are restated from their documented behaviour:

  * Dense            y = x @ kernel(in,out) + bias
  * Conv1D           kernel (k, in, out); 'same' pads floor((k-1)/2) left,
                     rest right; 'causal' pads k-1 left
  * Conv2D           kernel (kh, kw, in, out); stride-2 'same' pads
                     total = max((ceil(n/2)-1)*2 + k - n, 0), before = total//2
  * BatchNormalization  eps 1e-3; inference uses moving mean/variance
  * LayerNormalization  eps as given (1e-6 in the reference), last axis,
                        biased variance
  * GRUCell (TF2 default reset_after=True) gate order z, r, h
  * tf.math.round    round half to even

PARITY STATUS: the reference's own tests pin only the two cross-entropy
constants (TransformerTTS/tests/test_loss.py:10-24) and the tokenizer
(tests/test_char_tokenizer.py:11-23); both are reproduced in
tests/test_oracle_tts.py.  Every attention / conv / norm / GRU / Expand output
is "parity unpinned" against TensorFlow itself; it is cross-checked against a
second, independently written restatement (oracle/tts_torch.py).

All functions take/return NumPy arrays; `w` is a flat dict name -> array using
the naming scheme of the product package's weight initialiser.
"""
import math

import numpy as np


# --------------------------------------------------------------------------
# primitives
# --------------------------------------------------------------------------
def positional_encoding(position, model_dim):
    """TransformerTTS/model/transformer_utils.py:5-21 (float64 angles -> float32 table)."""
    pos = np.arange(position)[:, np.newaxis]
    i = np.arange(model_dim)[np.newaxis, :]
    angle_rates = 1 / np.power(10000, (2 * (i // 2)) / np.float32(model_dim))
    angle_rads = pos * angle_rates
    angle_rads[:, 0::2] = np.sin(angle_rads[:, 0::2])
    angle_rads[:, 1::2] = np.cos(angle_rads[:, 1::2])
    return angle_rads[np.newaxis, ...].astype(np.float32)


def dense(x, kernel, bias=None):
    y = x @ kernel
    if bias is not None:
        y = y + bias
    return y


def activation(x, name):
    if name in (None, 'linear'):
        return x
    if name == 'relu':
        return np.maximum(x, 0)
    if name == 'tanh':
        return np.tanh(x)
    raise ValueError(name)


def layer_norm(x, gamma, beta, eps=1e-6):
    mean = x.mean(axis=-1, keepdims=True)
    var = ((x - mean) ** 2).mean(axis=-1, keepdims=True)
    return (x - mean) / np.sqrt(var + eps) * gamma + beta


def batch_norm_infer(x, gamma, beta, mean, var, eps=1e-3):
    return (x - mean) / np.sqrt(var + eps) * gamma + beta


def conv1d(x, kernel, bias, padding):
    """x (B, T, Cin), kernel (k, Cin, Cout).  Keras Conv1D, stride 1."""
    k = kernel.shape[0]
    if padding == 'same':
        left = (k - 1) // 2
        right = k - 1 - left
    elif padding == 'causal':
        left, right = k - 1, 0
    elif padding == 'valid':
        left = right = 0
    else:
        raise ValueError(padding)
    B, T, _ = x.shape
    xp = np.pad(x, ((0, 0), (left, right), (0, 0)))
    To = xp.shape[1] - k + 1
    y = np.zeros((B, To, kernel.shape[2]), dtype=np.result_type(x, kernel))
    for j in range(k):
        y += xp[:, j:j + To, :] @ kernel[j]
    if bias is not None:
        y = y + bias
    return y


def softmax(x, axis=-1):
    m = x.max(axis=axis, keepdims=True)
    e = np.exp(x - m)
    return e / e.sum(axis=axis, keepdims=True)


def scaled_dot_product_attention(q, k, v, mask):
    """TransformerTTS/model/transformer_utils.py:24-58."""
    logits = q @ np.swapaxes(k, -1, -2)
    logits = logits / math.sqrt(k.shape[-1])
    if mask is not None:
        logits = logits + mask * -1e9
    weights = softmax(logits, axis=-1)
    return weights @ v, weights


def create_encoder_padding_mask(seq):
    """transformer_utils.py:61-63 -> (B,1,1,N), 1.0 where token == 0."""
    return (seq == 0).astype(np.float64)[:, None, None, :]


def create_mel_padding_mask(seq):
    """transformer_utils.py:66-69 -> (B,1,1,T), 1.0 where the frame is all-zero."""
    s = np.abs(seq).sum(axis=-1)
    return (s == 0).astype(np.float64)[:, None, None, :]


def create_look_ahead_mask(size):
    """transformer_utils.py:72-74 -> strict upper triangle of ones."""
    return 1.0 - np.tril(np.ones((size, size)))


# --------------------------------------------------------------------------
# layers (TransformerTTS/model/layers.py)
# --------------------------------------------------------------------------
def _norm(x, w, prefix, kind):
    if kind == 'layer':
        return layer_norm(x, w[prefix + '.gamma'], w[prefix + '.beta'], 1e-6)
    return batch_norm_infer(x, w[prefix + '.gamma'], w[prefix + '.beta'],
                            w[prefix + '.moving_mean'], w[prefix + '.moving_variance'])


def cnn_res_norm(x, w, prefix, n_layers, inner_activation, last_activation, padding, normalization):
    """layers.py:7-47 (inference: BatchNorm uses moving statistics)."""
    inputs = x
    for i in range(n_layers - 1):
        x = conv1d(x, w[f'{prefix}.conv{i}.kernel'], w[f'{prefix}.conv{i}.bias'], padding)
        x = _norm(x, w, f'{prefix}.norm{i}', normalization)
        x = activation(x, inner_activation)
    x = conv1d(x, w[f'{prefix}.last_conv.kernel'], w[f'{prefix}.last_conv.bias'], padding)
    x = _norm(x, w, f'{prefix}.norm{n_layers - 1}', normalization)
    x = activation(x, last_activation)
    return _norm(inputs + x, w, f'{prefix}.norm{n_layers}', normalization)


def ffn_res_norm(x, w, prefix):
    """layers.py:50-71.  No activation between d1 and d2 (reference quirk)."""
    y = dense(x, w[prefix + '.d1.kernel'], w[prefix + '.d1.bias'])
    y = dense(y, w[prefix + '.d2.kernel'], w[prefix + '.d2.bias'])
    y = layer_norm(y, w[prefix + '.ln.gamma'], w[prefix + '.ln.beta'])
    y = np.maximum(y, 0)
    return layer_norm(y + x, w[prefix + '.last_ln.gamma'], w[prefix + '.last_ln.beta'])


def multi_head_attention(v, k, q_in, mask, w, prefix, num_heads):
    """layers.py:100-147.  Output projection consumes [q_in | context]."""
    B = q_in.shape[0]
    q = dense(q_in, w[prefix + '.wq.kernel'], w[prefix + '.wq.bias'])
    k = dense(k, w[prefix + '.wk.kernel'], w[prefix + '.wk.bias'])
    v = dense(v, w[prefix + '.wv.kernel'], w[prefix + '.wv.bias'])
    model_dim = q.shape[-1]
    depth = model_dim // num_heads

    def split(x):
        return x.reshape(B, -1, num_heads, depth).transpose(0, 2, 1, 3)

    ctx, weights = scaled_dot_product_attention(split(q), split(k), split(v), mask)
    ctx = ctx.transpose(0, 2, 1, 3).reshape(B, -1, model_dim)
    out = dense(np.concatenate([q_in, ctx], axis=-1), w[prefix + '.dense.kernel'], w[prefix + '.dense.bias'])
    return out, weights


def self_attention_res_norm(x, mask, w, prefix, num_heads):
    """layers.py:150-168 (inference: dropout is identity)."""
    a, weights = multi_head_attention(x, x, x, mask, w, prefix + '.mha', num_heads)
    a = layer_norm(a, w[prefix + '.ln.gamma'], w[prefix + '.ln.beta'])
    return layer_norm(a + x, w[prefix + '.last_ln.gamma'], w[prefix + '.last_ln.beta']), weights


def self_attention_blocks(x, padding_mask, w, prefix, model_dim, num_heads, dense_blocks,
                          kernel_size, max_pos, reduction_factor=1):
    """layers.py:215-255.  Returns (x, [attention weights per block])."""
    L = x.shape[1]
    pe = positional_encoding(max_pos, model_dim)
    x = x * math.sqrt(np.float32(model_dim))
    x = x + w[prefix + '.pos_encoding_scalar'] * pe[:, :L * reduction_factor:reduction_factor, :]
    attn = []
    for i, h in enumerate(num_heads):
        p = f'{prefix}.block{i}'
        x, a = self_attention_res_norm(x, padding_mask, w, p + '.sarn', h)
        if i < dense_blocks:
            x = ffn_res_norm(x, w, p + '.ffn')
        else:
            x = cnn_res_norm(x, w, p + '.conv', 2, 'relu', 'relu', 'same', 'batch')
        attn.append(a)
    return x, attn


def cross_attention_res_norm(q, k, v, mask, w, prefix, num_heads):
    """layers.py:258-274."""
    a, weights = multi_head_attention(v, k, q, mask, w, prefix + '.mha', num_heads)
    return layer_norm(a + q, w[prefix + '.layernorm.gamma'], w[prefix + '.layernorm.beta']), weights


def cross_attention_blocks(x, enc_output, look_ahead_mask, enc_padding_mask, w, prefix, model_dim,
                           num_heads, dense_blocks, conv_kernel, conv_padding, max_pos, reduction_factor):
    """layers.py:331-376."""
    L = x.shape[1]
    pe = positional_encoding(max_pos, model_dim)
    x = x * math.sqrt(np.float32(model_dim))
    x = x + w[prefix + '.pos_encoding_scalar'] * pe[:, :L * reduction_factor:reduction_factor, :]
    attn = []
    for i, h in enumerate(num_heads):
        p = f'{prefix}.block{i}'
        x, _ = self_attention_res_norm(x, look_ahead_mask, w, p + '.sarn', h)
        x, a = cross_attention_res_norm(x, enc_output, enc_output, enc_padding_mask, w, p + '.carn', h)
        if i < dense_blocks:
            x = ffn_res_norm(x, w, p + '.ffn')
        else:
            x = cnn_res_norm(x, w, p + '.conv', 2, 'relu', 'relu', conv_padding, 'batch')
        attn.append(a)
    return x, attn


def decoder_prenet(x, w, prefix, masks=None):
    """layers.py:379-402.  Its dropout is always 'training'; masks = (m1, m2) are the two keep masks already scaled by
    1 / (1 - rate) (inputs, so that a device path can be compared on identical draws); None = rate 0."""
    x = np.maximum(dense(x, w[prefix + '.d1.kernel'], w[prefix + '.d1.bias']), 0)
    if masks is not None:
        x = x * masks[0]
    x = np.maximum(dense(x, w[prefix + '.d2.kernel'], w[prefix + '.d2.bias']), 0)
    if masks is not None:
        x = x * masks[1]
    return x


def duration_predictor(x, w, prefix):
    """layers.py:618-642."""
    x = cnn_res_norm(x, w, prefix + '.conv_blocks', 2, 'relu', 'relu', 'same', 'layer')
    return np.maximum(dense(x, w[prefix + '.linear.kernel'], w[prefix + '.linear.bias']), 0)


def expand(x, dimensions):
    """layers.py:645-683: repeat row n of x round(dim[n]) times, right-pad with zeros."""
    dims = np.rint(dimensions[..., 0]).astype(np.int64)
    B, N, D = x.shape
    T = int(dims.sum(axis=1).max()) if B else 0
    out = np.zeros((B, T, D), dtype=x.dtype)
    for b in range(B):
        rows = np.repeat(np.arange(N), np.maximum(dims[b], 0))
        out[b, :len(rows)] = x[b, rows]
    return out


def postnet_ar(x, w, prefix, n_layers):
    """layers.py:588-615."""
    stop = dense(x, w[prefix + '.stop_linear.kernel'], w[prefix + '.stop_linear.bias'])
    conv = cnn_res_norm(x, w, prefix + '.conv_blocks', n_layers, 'tanh', 'linear', 'causal', 'batch')
    return {'mel_linear': x, 'final_output': conv, 'stop_prob': stop}


# --------------------------------------------------------------------------
# ForwardTransformer (pretrained_models/for_text_encoder/models.py:283-477)
# --------------------------------------------------------------------------
def forward_transformer_call(w, cfg, tokens, target_durations=None, durations_scalar=1.0, dtype=np.float64, prenet_masks=None):
    """models.py:434-456, training=False.  prenet_masks: the two scaled keep masks of DecoderPrenet's dropout, which the
    reference applies at inference too (layers.py:394-402); None = rate 0.

    tokens (B,N) int; target_durations (B,N) or None.  Returns dict with 'mel',
    'duration', 'expanded_mask', 'encoder_attention', 'decoder_attention'.
    """
    w = {k: np.asarray(v, dtype=dtype) for k, v in w.items()}
    tokens = np.asarray(tokens)
    padding_mask = create_encoder_padding_mask(tokens).astype(dtype)
    x = w['encoder_prenet.embeddings'][tokens]
    x, enc_attn = self_attention_blocks(
        x, padding_mask, w, 'encoder', cfg['encoder_model_dimension'], cfg['encoder_num_heads'],
        cfg['encoder_dense_blocks'], cfg['encoder_attention_conv_kernel'],
        cfg['encoder_max_position_encoding'])
    durations = duration_predictor(x, w, 'dur_pred') * durations_scalar
    durations = (1.0 - padding_mask.reshape(durations.shape)) * durations
    if target_durations is not None:
        mels = expand(x, np.asarray(target_durations, dtype=dtype)[..., None])
    else:
        mels = expand(x, durations)
    expanded_mask = create_mel_padding_mask(mels).astype(dtype)
    mels = decoder_prenet(mels, w, 'decoder_prenet', prenet_masks)
    mels, dec_attn = self_attention_blocks(
        mels, expanded_mask, w, 'decoder', cfg['decoder_model_dimension'], cfg['decoder_num_heads'],
        cfg['decoder_dense_blocks'], cfg['decoder_attention_conv_kernel'],
        cfg['decoder_max_position_encoding'])
    mels = dense(mels, w['out.kernel'], w['out.bias'])
    mels = cnn_res_norm(mels, w, 'decoder_postnet', cfg['postnet_conv_layers'], 'tanh', 'linear', 'same', 'batch')
    return {'mel': mels, 'duration': durations, 'expanded_mask': expanded_mask,
            'encoder_attention': enc_attn, 'decoder_attention': dec_attn}


# --------------------------------------------------------------------------
# GST reference encoder + AutoregressiveTransformer teacher-forced call
# --------------------------------------------------------------------------
def conv2d_same(x, kernel, bias, stride):
    """Keras Conv2D padding='same'.  x (B,H,W,Cin), kernel (kh,kw,Cin,Cout)."""
    kh, kw, _, cout = kernel.shape
    B, H, W, _ = x.shape
    Ho, Wo = -(-H // stride), -(-W // stride)
    ph = max((Ho - 1) * stride + kh - H, 0)
    pw = max((Wo - 1) * stride + kw - W, 0)
    xp = np.pad(x, ((0, 0), (ph // 2, ph - ph // 2), (pw // 2, pw - pw // 2), (0, 0)))
    y = np.zeros((B, Ho, Wo, cout), dtype=np.result_type(x, kernel))
    for i in range(kh):
        for j in range(kw):
            patch = xp[:, i:i + (Ho - 1) * stride + 1:stride, j:j + (Wo - 1) * stride + 1:stride, :]
            y += patch @ kernel[i, j]
    return y + bias


def sigmoid(x):
    return 1.0 / (1.0 + np.exp(-x))


def gru_cell_keras(x, h, kernel, recurrent_kernel, bias):
    """TF2 GRUCell defaults (reset_after=True): kernel (in,3u), recurrent (u,3u), bias (2,3u); gates z,r,h."""
    u = h.shape[-1]
    xm = x @ kernel + bias[0]
    hm = h @ recurrent_kernel + bias[1]
    z = sigmoid(xm[:, :u] + hm[:, :u])
    r = sigmoid(xm[:, u:2 * u] + hm[:, u:2 * u])
    hh = np.tanh(xm[:, 2 * u:] + r * hm[:, 2 * u:])
    return z * h + (1 - z) * hh


def reference_encoder_gst(x, w, prefix, n_convs, stride, num_heads):
    """layers.py:526-585 (inference).  x (B,T,80) -> (B,1,style_dim), attention (B,H,1,tokens)."""
    B = x.shape[0]
    y = x[..., None]
    for i in range(n_convs):
        y = conv2d_same(y, w[f'{prefix}.conv{i}.kernel'], w[f'{prefix}.conv{i}.bias'], stride)
        y = batch_norm_infer(y, w[f'{prefix}.norm{i}.gamma'], w[f'{prefix}.norm{i}.beta'],
                             w[f'{prefix}.norm{i}.moving_mean'], w[f'{prefix}.norm{i}.moving_variance'])
        y = np.maximum(y, 0)
    y = y.reshape(B, y.shape[1], y.shape[2] * y.shape[3])
    u = w[prefix + '.gru.recurrent_kernel'].shape[0]
    h = np.zeros((B, u), dtype=y.dtype)
    for t in range(y.shape[1]):
        h = gru_cell_keras(y[:, t], h, w[prefix + '.gru.kernel'], w[prefix + '.gru.recurrent_kernel'],
                           w[prefix + '.gru.bias'])
    proj = np.tanh(dense(h, w[prefix + '.rnn_proj.kernel'], w[prefix + '.rnn_proj.bias']))[:, None, :]
    tokens = np.tanh(np.broadcast_to(w[prefix + '.gst_tokens'][None], (B,) + w[prefix + '.gst_tokens'].shape))
    out, attn = multi_head_attention(tokens, tokens, proj, None, w, prefix + '.mha', num_heads)
    return out, attn


def ar_encoder(w, cfg, tokens, style_targets, spk_embed):
    """AutoregressiveTransformer.call_encoder (model/AutoregressiveTransformer.py:124-159), training flags False.
    `w` already cast; returns (enc_output, enc_mask, text_attn, gst_attn, gst_out, text_out)."""
    st = cfg['system_type']
    padding_mask = create_encoder_padding_mask(tokens).astype(w['text_encoder_prenet.embeddings'].dtype)
    x = w['text_encoder_prenet.embeddings'][tokens]
    text_out, text_attn = self_attention_blocks(
        x, padding_mask, w, 'text_encoder', cfg['encoder_model_dimension'], cfg['encoder_num_heads'],
        cfg['encoder_dense_blocks'], cfg['encoder_attention_conv_kernel'],
        cfg['encoder_max_position_encoding'])
    N = text_out.shape[1]
    gst_out = gst_attn = None
    parts = [text_out]
    if st in ('speaker_style_text', 'style_text'):
        gst_out, gst_attn = reference_encoder_gst(style_targets, w, 'style_encoder', len(cfg['ref_encoder_filters']),
                                                  cfg['ref_encoder_strides'], cfg['gst_multi_num_heads'])
        parts.append(np.tile(gst_out, (1, N, 1)))
    if st in ('speaker_style_text', 'speaker_text'):
        parts.append(np.tile(np.asarray(spk_embed, dtype=text_out.dtype), (1, N, 1)))
    enc_output = np.concatenate(parts, axis=2)
    enc_mask = create_mel_padding_mask(enc_output).astype(text_out.dtype)
    return enc_output, enc_mask, text_attn, gst_attn, gst_out, text_out


def ar_decoder(w, cfg, enc_output, enc_mask, targets, r, prenet_masks=None):
    """AutoregressiveTransformer.call_decoder (model/AutoregressiveTransformer.py:161-181), training=False;
    prenet_masks: see decoder_prenet."""
    dec_pad = create_mel_padding_mask(targets).astype(targets.dtype)
    look = create_look_ahead_mask(targets.shape[1])
    combined = np.maximum(dec_pad, look)
    dec_in = decoder_prenet(targets, w, 'decoder_prenet', prenet_masks)
    dec_out, dec_attn = cross_attention_blocks(
        dec_in, enc_output, combined, enc_mask, w, 'decoder', cfg['decoder_model_dimension'],
        cfg['decoder_num_heads'], cfg['decoder_dense_blocks'], cfg['decoder_attention_conv_kernel'],
        'causal', cfg['decoder_max_position_encoding'], r)
    mc = cfg['mel_channels']
    proj = dense(dec_out, w['final_proj_mel.kernel'], w['final_proj_mel.bias'])[:, :, :r * mc]
    mel = proj.reshape(proj.shape[0], proj.shape[1] * r, mc)
    out = postnet_ar(mel, w, 'decoder_postnet', cfg['postnet_conv_layers'])
    out.update({'decoder_attention': dec_attn, 'decoder_output': dec_out, 'linear': mel})
    return out


def autoregressive_call(w, cfg, tokens, targets, spk_embed=None, r=1, dtype=np.float64, prenet_masks=None):
    """AutoregressiveTransformer.call (model/AutoregressiveTransformer.py:243-263), all training flags False.

    targets: (B, T/r, 80) reduced, start-vector-prefixed decoder input (also fed to the style encoder,
    as the reference does).  Returns final_output/mel_linear (B, T, 80), stop_prob (B, T, 3), attentions.
    """
    w = {k: np.asarray(v, dtype=dtype) for k, v in w.items()}
    tokens = np.asarray(tokens)
    targets = np.asarray(targets, dtype=dtype)
    enc_output, enc_mask, text_attn, gst_attn, gst_out, text_out = ar_encoder(w, cfg, tokens, targets, spk_embed)
    out = ar_decoder(w, cfg, enc_output, enc_mask, targets, r, prenet_masks)
    out.update({'text_encoder_attention': text_attn, 'gst_encoder_attention': gst_attn,
                'gst_output': gst_out, 'text_enc_output': text_out})
    return out


def autoregressive_predict(w, cfg, tokens, ref_mel, spk_embed=None, r=1, max_length=1000, dtype=np.float64):
    """AutoregressiveTransformer.predict (model/AutoregressiveTransformer.py:265-296) for already-encoded tokens (N,):
    encoder once, then the whole decoder re-run on the growing output until argmax(stop_prob) == 2.
    ref_mel (T, 80) or None; spk_embed (1, D) or None.  Returns {'mel': (frames, 80), 'steps': int}."""
    w = {k: np.asarray(v, dtype=dtype) for k, v in w.items()}
    mc = cfg['mel_channels']
    style = None
    if ref_mel is not None:
        style = np.asarray(ref_mel, dtype=dtype)[None][:, :-1][:, 0::r, :]
    spk = None if spk_embed is None else np.asarray(spk_embed, dtype=dtype)[None]
    inp = np.asarray(tokens)[None]
    start = np.full((1, 1, mc), float(cfg['mel_start_value']), dtype=dtype)
    output, output_concat = start.copy(), start.copy()
    enc_output, enc_mask, *_ = ar_encoder(w, cfg, inp, style, spk)
    steps = 0
    for i in range(int(max_length // r) + 1):
        out = ar_decoder(w, cfg, enc_output, enc_mask, output, r)
        output = np.concatenate([output, out['final_output'][:1, -1:, :]], axis=-2)
        output_concat = np.concatenate([output_concat, out['final_output'][:1, -r:, :]], axis=-2)
        steps += 1
        if int(np.argmax(out['stop_prob'][0, -1])) == 2:
            break
    return {'mel': output_concat[0, 1:, :], 'steps': steps}


# --------------------------------------------------------------------------
# losses (TransformerTTS/utils/losses.py)
# --------------------------------------------------------------------------
def _sparse_ce_from_logits(targets, logits):
    m = logits.max(axis=-1, keepdims=True)
    lse = m[..., 0] + np.log(np.exp(logits - m).sum(axis=-1))
    picked = np.take_along_axis(logits, targets[..., None], axis=-1)[..., 0]
    return lse - picked


def new_scaled_crossentropy(index=2, scaling=1.0):
    """losses.py:4-21.  Keras SUM_OVER_BATCH_SIZE: weighted sum / number of elements (padding included)."""
    def masked_crossentropy_(targets, logits):
        targets = np.asarray(targets)
        logits = np.asarray(logits, dtype=np.float64)
        pad = (targets != 0).astype(np.float64)
        stop = (targets == index).astype(np.float64) * (scaling - 1.0)
        ce = _sparse_ce_from_logits(targets, logits)
        return float((ce * (pad + stop)).sum() / ce.size)
    return masked_crossentropy_


def masked_crossentropy(targets, logits):
    """losses.py:24-29."""
    return new_scaled_crossentropy(index=2, scaling=1.0)(targets, logits)


def masked_mean_absolute_error(targets, logits):
    """losses.py:41-47: mean |t-p| over channels, weight = any(target channel != 0), sum / (B*T)."""
    targets = np.asarray(targets, dtype=np.float64)
    logits = np.asarray(logits, dtype=np.float64)
    mask = (targets != 0).any(axis=-1).astype(np.float64)
    per = np.abs(targets - logits).mean(axis=-1)
    return float((per * mask).sum() / per.size)


def weighted_sum_losses(targets, pred, loss_functions, coeffs):
    """losses.py:71-78."""
    total, vals = 0.0, []
    for t, p, f, c in zip(targets, pred, loss_functions, coeffs):
        v = f(t, p)
        vals.append(v)
        total += c * v
    return total, vals


# --------------------------------------------------------------------------
# schedules (TransformerTTS/utils/scheduling.py)
# --------------------------------------------------------------------------
def piecewise_linear_schedule(step, schedule):
    """scheduling.py:11-36."""
    s = np.array(schedule, dtype=np.float64)
    X, Y = s[:, 0], s[:, 1]
    if step < X[0]:
        return float(Y[0])
    idx = np.where(step >= X)[0][-1]
    if idx == len(Y) - 1:
        return float(Y[-1])
    m = (Y[idx + 1] - Y[idx]) / (X[idx + 1] - X[idx])
    return float(m * step + (Y[idx] - m * X[idx]))


def reduction_schedule(step, schedule):
    """scheduling.py:39-47."""
    s = np.array(schedule)
    r = s[0, 0]
    for i in range(s.shape[0]):
        if s[i, 0] <= step:
            r = s[i, 1]
        else:
            break
    return int(r)


# --------------------------------------------------------------------------
# tokenizer (TransformerTTS/preprocessing/text/tokenizer.py:4-27)
# --------------------------------------------------------------------------
class Tokenizer:
    def __init__(self, alphabet, start_token='>', end_token='<', pad_token='/', add_start_end=True):
        self.alphabet = alphabet
        self.idx_to_token = {i: s for i, s in enumerate(alphabet, start=1)}
        self.idx_to_token[0] = pad_token
        self.token_to_idx = {s: i for i, s in self.idx_to_token.items()}
        self.vocab_size = len(alphabet) + 1
        self.add_start_end = add_start_end
        if add_start_end:
            self.start_token_index = len(alphabet) + 1
            self.end_token_index = len(alphabet) + 2
            self.vocab_size += 2
            self.idx_to_token[self.start_token_index] = start_token
            self.idx_to_token[self.end_token_index] = end_token

    def __call__(self, sentence):
        seq = [self.token_to_idx[c] for c in sentence if c in self.token_to_idx]
        if self.add_start_end:
            seq = [self.start_token_index] + seq + [self.end_token_index]
        return seq

    def decode(self, sequence):
        return ''.join(self.idx_to_token[int(t)] for t in sequence if int(t) in self.idx_to_token)
