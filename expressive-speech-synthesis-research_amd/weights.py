"""Weight containers: seeded synthetic initialisation in the reference's native layouts.

A model's weights are a flat dict  name -> float32 ndarray.  Layouts follow the
framework the reference builds each model with, so that converted checkpoints
drop in unchanged:

  TransformerTTS (Keras): Dense kernel (in, out); Conv1D kernel (k, in, out);
      Conv2D kernel (kh, kw, in, out); BatchNormalization gamma/beta/
      moving_mean/moving_variance; LayerNormalization gamma/beta;
      GRUCell kernel (in, 3u) / recurrent_kernel (u, 3u) / bias (2, 3u).
  WaveRNN (torch): the state_dict keys of WaveRNN/models/fatchord_version.py:89-116.

Names: '<layer path>.<variable>', e.g. 'encoder.block1.sarn.mha.wq.kernel',
'decoder.block0.conv.norm2.moving_mean' (norm index as in CNNResNorm,
TransformerTTS/model/layers.py:28-31: n_layers + 1 normalisations).
"""
import math

import numpy as np


class _Init:
    def __init__(self, seed):
        self.rng = np.random.default_rng(seed)
        self.w = {}

    def glorot(self, name, shape, fan_in, fan_out):
        lim = math.sqrt(6.0 / (fan_in + fan_out))
        self.w[name] = self.rng.uniform(-lim, lim, size=shape).astype(np.float32)

    def dense(self, p, n_in, n_out, bias_value=None):
        self.glorot(p + '.kernel', (n_in, n_out), n_in, n_out)
        if bias_value is None:
            self.w[p + '.bias'] = self.rng.normal(0, 0.02, size=(n_out,)).astype(np.float32)
        else:
            self.w[p + '.bias'] = np.full((n_out,), bias_value, dtype=np.float32)

    def conv1d(self, p, k, n_in, n_out):
        self.glorot(p + '.kernel', (k, n_in, n_out), k * n_in, k * n_out)
        self.w[p + '.bias'] = self.rng.normal(0, 0.02, size=(n_out,)).astype(np.float32)

    def conv2d(self, p, k, n_in, n_out):
        self.glorot(p + '.kernel', (k, k, n_in, n_out), k * k * n_in, k * k * n_out)
        self.w[p + '.bias'] = self.rng.normal(0, 0.02, size=(n_out,)).astype(np.float32)

    def layer_norm(self, p, n):
        self.w[p + '.gamma'] = self.rng.uniform(0.8, 1.2, size=(n,)).astype(np.float32)
        self.w[p + '.beta'] = self.rng.normal(0, 0.1, size=(n,)).astype(np.float32)

    def batch_norm(self, p, n):
        self.layer_norm(p, n)
        self.w[p + '.moving_mean'] = self.rng.normal(0, 0.1, size=(n,)).astype(np.float32)
        self.w[p + '.moving_variance'] = self.rng.uniform(0.5, 1.5, size=(n,)).astype(np.float32)

    def norm(self, p, n, kind):
        (self.layer_norm if kind == 'layer' else self.batch_norm)(p, n)

    def mha(self, p, model_dim, q_in, k_in, v_in):
        self.dense(p + '.wq', q_in, model_dim)
        self.dense(p + '.wk', k_in, model_dim)
        self.dense(p + '.wv', v_in, model_dim)
        self.dense(p + '.dense', q_in + model_dim, model_dim)

    def sarn(self, p, d):
        self.mha(p + '.mha', d, d, d, d)
        self.layer_norm(p + '.ln', d)
        self.layer_norm(p + '.last_ln', d)

    def ffn(self, p, d, hidden):
        self.dense(p + '.d1', d, hidden)
        self.dense(p + '.d2', hidden, d)
        self.layer_norm(p + '.ln', d)
        self.layer_norm(p + '.last_ln', d)

    def cnn_res_norm(self, p, n_in, hidden, n_out, n_layers, k, kind):
        c = n_in
        for i in range(n_layers - 1):
            self.conv1d(f'{p}.conv{i}', k, c, hidden)
            self.norm(f'{p}.norm{i}', hidden, kind)
            c = hidden
        self.conv1d(p + '.last_conv', k, c, n_out)
        self.norm(f'{p}.norm{n_layers - 1}', n_out, kind)
        self.norm(f'{p}.norm{n_layers}', n_out, kind)

    def self_attention_blocks(self, p, d, heads, dense_blocks, ff, conv_filters, k):
        self.w[p + '.pos_encoding_scalar'] = np.array(1.0, dtype=np.float32)
        for i in range(len(heads)):
            b = f'{p}.block{i}'
            self.sarn(b + '.sarn', d)
            if i < dense_blocks:
                self.ffn(b + '.ffn', d, ff)
            else:
                self.cnn_res_norm(b + '.conv', d, conv_filters, d, 2, k, 'batch')

    def cross_attention_blocks(self, p, d, enc_dim, heads, dense_blocks, ff, conv_filters, k):
        self.w[p + '.pos_encoding_scalar'] = np.array(1.0, dtype=np.float32)
        for i in range(len(heads)):
            b = f'{p}.block{i}'
            self.sarn(b + '.sarn', d)
            self.mha(b + '.carn.mha', d, d, enc_dim, enc_dim)
            self.layer_norm(b + '.carn.layernorm', d)
            if i < dense_blocks:
                self.ffn(b + '.ffn', d, ff)
            else:
                self.cnn_res_norm(b + '.conv', d, conv_filters, d, 2, k, 'batch')


def init_forward_transformer(cfg, seed=1234):
    """Random ForwardTransformer weights (layer structure: pretrained_models/for_text_encoder/models.py:309-359)."""
    it = _Init(seed)
    de, dd = cfg['encoder_model_dimension'], cfg['decoder_model_dimension']
    it.w['encoder_prenet.embeddings'] = it.rng.uniform(-0.05, 0.05, size=(cfg['vocab_size'], de)).astype(np.float32)
    it.self_attention_blocks('encoder', de, cfg['encoder_num_heads'], cfg['encoder_dense_blocks'],
                             cfg['encoder_feed_forward_dimension'], cfg['encoder_attention_conv_filters'],
                             cfg['encoder_attention_conv_kernel'])
    it.cnn_res_norm('dur_pred.conv_blocks', de, de, de, 2, 3, 'layer')
    it.dense('dur_pred.linear', de, 1, bias_value=1.0)
    it.dense('decoder_prenet.d1', de, cfg['decoder_feed_forward_dimension'])
    it.dense('decoder_prenet.d2', cfg['decoder_feed_forward_dimension'], dd)
    it.self_attention_blocks('decoder', dd, cfg['decoder_num_heads'], cfg['decoder_dense_blocks'],
                             cfg['decoder_feed_forward_dimension'], cfg['decoder_attention_conv_filters'],
                             cfg['decoder_attention_conv_kernel'])
    it.dense('out', dd, cfg['mel_channels'])
    it.cnn_res_norm('decoder_postnet', cfg['mel_channels'], cfg['postnet_conv_filters'], cfg['mel_channels'],
                    cfg['postnet_conv_layers'], cfg['postnet_kernel_size'], 'batch')
    return it.w


def encoder_output_dim(cfg):
    """Width of the concatenated encoder output (model/AutoregressiveTransformer.py:133-156)."""
    d = cfg['encoder_model_dimension']
    st = cfg['system_type']
    if st in ('speaker_style_text', 'style_text'):
        d += cfg['gst_style_embed_dim']
    if st in ('speaker_style_text', 'speaker_text'):
        d += cfg['speaker_embed_dim']
    return d


def init_autoregressive_transformer(cfg, seed=1234):
    """Random GST AutoregressiveTransformer weights (model/AutoregressiveTransformer.py:62-117)."""
    it = _Init(seed)
    de, dd = cfg['encoder_model_dimension'], cfg['decoder_model_dimension']
    it.w['text_encoder_prenet.embeddings'] = it.rng.uniform(
        -0.05, 0.05, size=(cfg['vocab_size'], cfg['encoder_prenet_dimension'])).astype(np.float32)
    it.self_attention_blocks('text_encoder', de, cfg['encoder_num_heads'], cfg['encoder_dense_blocks'],
                             cfg['encoder_feed_forward_dimension'], cfg['encoder_attention_conv_filters'],
                             cfg['encoder_attention_conv_kernel'])
    if cfg['system_type'] in ('speaker_style_text', 'style_text'):
        p = 'style_encoder'
        c, f = 1, cfg['mel_channels']
        for i, n in enumerate(cfg['ref_encoder_filters']):
            it.conv2d(f'{p}.conv{i}', cfg['ref_encoder_kernel_size'], c, n)
            it.batch_norm(f'{p}.norm{i}', n)
            c = n
            f = -(-f // cfg['ref_encoder_strides'])
        u = cfg['ref_encoder_gru_cell_units']
        it.glorot(p + '.gru.kernel', (f * c, 3 * u), f * c, 3 * u)
        q, _ = np.linalg.qr(it.rng.normal(size=(3 * u, u)))
        it.w[p + '.gru.recurrent_kernel'] = np.ascontiguousarray(q.T).astype(np.float32)
        it.w[p + '.gru.bias'] = it.rng.normal(0, 0.02, size=(2, 3 * u)).astype(np.float32)
        it.dense(p + '.rnn_proj', u, u)
        sd, nh = cfg['gst_style_embed_dim'], cfg['gst_multi_num_heads']
        it.mha(p + '.mha', sd, u, sd // nh, sd // nh)
        tok = it.rng.normal(0, 0.5, size=(cfg['gst_heads'], sd // nh))
        it.w[p + '.gst_tokens'] = np.clip(tok, -1.0, 1.0).astype(np.float32)
    it.dense('decoder_prenet.d1', cfg['mel_channels'], cfg['decoder_prenet_dimension'])
    it.dense('decoder_prenet.d2', cfg['decoder_prenet_dimension'], dd)
    it.cross_attention_blocks('decoder', dd, encoder_output_dim(cfg), cfg['decoder_num_heads'],
                              cfg['decoder_dense_blocks'], cfg['decoder_feed_forward_dimension'],
                              cfg['decoder_attention_conv_filters'], cfg['decoder_attention_conv_kernel'])
    it.dense('final_proj_mel', dd, cfg['mel_channels'] * cfg['max_r'])
    it.dense('decoder_postnet.stop_linear', cfg['mel_channels'], 3)
    it.cnn_res_norm('decoder_postnet.conv_blocks', cfg['mel_channels'], cfg['postnet_conv_filters'],
                    cfg['mel_channels'], cfg['postnet_conv_layers'], cfg['postnet_kernel_size'], 'batch')
    return it.w


def init_wavernn(hp, seed=1234, n_classes=None):
    """Random WaveRNN state_dict (torch layouts; keys as WaveRNN/models/fatchord_version.py:89-116)."""
    rng = np.random.default_rng(seed)
    w = {}
    feat, comp, res_out = hp['num_mels'], hp['voc_compute_dims'], hp['voc_res_out_dims']
    rnn, fc, pad = hp['voc_rnn_dims'], hp['voc_fc_dims'], hp['voc_pad']
    aux = res_out // 4
    if n_classes is None:
        n_classes = 2 ** hp['bits'] if hp['voc_mode'] == 'RAW' else 30

    def uni(shape, fan_in):
        lim = 1.0 / math.sqrt(fan_in)
        return rng.uniform(-lim, lim, size=shape).astype(np.float32)

    def bn(p, n):
        w[p + '.weight'] = rng.uniform(0.8, 1.2, size=(n,)).astype(np.float32)
        w[p + '.bias'] = rng.normal(0, 0.1, size=(n,)).astype(np.float32)
        w[p + '.running_mean'] = rng.normal(0, 0.1, size=(n,)).astype(np.float32)
        w[p + '.running_var'] = rng.uniform(0.5, 1.5, size=(n,)).astype(np.float32)

    k = 2 * pad + 1
    w['upsample.resnet.conv_in.weight'] = uni((comp, feat, k), feat * k)
    bn('upsample.resnet.batch_norm', comp)
    for i in range(hp['voc_res_blocks']):
        p = f'upsample.resnet.layers.{i}'
        w[p + '.conv1.weight'] = uni((comp, comp, 1), comp)
        w[p + '.conv2.weight'] = uni((comp, comp, 1), comp)
        bn(p + '.batch_norm1', comp)
        bn(p + '.batch_norm2', comp)
    w['upsample.resnet.conv_out.weight'] = uni((res_out, comp, 1), comp)
    w['upsample.resnet.conv_out.bias'] = uni((res_out,), comp)
    for j, s in enumerate(hp['voc_upsample_factors']):
        base = np.full((1, 1, 1, 2 * s + 1), 1.0 / (2 * s + 1), dtype=np.float32)
        w[f'upsample.up_layers.{2 * j + 1}.weight'] = base * rng.uniform(0.9, 1.1, size=base.shape).astype(np.float32)
    w['I.weight'] = uni((rnn, feat + aux + 1), feat + aux + 1)
    w['I.bias'] = uni((rnn,), feat + aux + 1)
    for name, n_in in (('rnn1', rnn), ('rnn2', rnn + aux)):
        w[f'{name}.weight_ih_l0'] = uni((3 * rnn, n_in), rnn)
        w[f'{name}.weight_hh_l0'] = uni((3 * rnn, rnn), rnn)
        w[f'{name}.bias_ih_l0'] = uni((3 * rnn,), rnn)
        w[f'{name}.bias_hh_l0'] = uni((3 * rnn,), rnn)
    w['fc1.weight'] = uni((fc, rnn + aux), rnn + aux)
    w['fc1.bias'] = uni((fc,), rnn + aux)
    w['fc2.weight'] = uni((fc, fc + aux), fc + aux)
    w['fc2.bias'] = uni((fc,), fc + aux)
    w['fc3.weight'] = uni((n_classes, fc), fc)
    w['fc3.bias'] = uni((n_classes,), fc)
    return w


def count_params(w):
    return int(sum(v.size for v in w.values()))
