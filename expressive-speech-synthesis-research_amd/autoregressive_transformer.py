"""GST / speaker-conditioned autoregressive TransformerTTS on MI355X: reference encoder + style tokens,
text encoder, teacher-forced decoder.

Mirrors the surface of the reference class (TransformerTTS/model/AutoregressiveTransformer.py:20-327):
`call(inputs, targets, spk_embed, train_*=False)`, `call_encoder`, `call_decoder`, `set_constants`,
`encode_text`, `encode_ref`, `input_reshape`, same output dict keys.  Tensors are torch CUDA tensors;
all arithmetic runs in libestts.so kernels (torch only moves / broadcasts data).
"""
import math

import numpy as np
import torch

from . import ops
from .blocks import (ConvBNLayer, CrossAttentionStack, FFNLayer, SelfAttentionStack, _dev_vec, pack_dense)
from .ops import PackedLinear, fold_batchnorm, same_pad
from .text import Tokenizer, default_alphabet
from .weights import encoder_output_dim


class ReferenceEncoderGST:
    """ReferenceEncoderGST (TransformerTTS/model/layers.py:526-585), inference.

    conv stack: first layer (one input channel) by a direct HBM-bound kernel, the other layers as implicit GEMMs over
    NHWC pixels; GRU: input projection of all steps in one GEMM, then per step one recurrent GEMM + gate kernel
    (Keras gate order z, r, h is permuted to the kernel's r, z, n at pack time; reset_after=True is exactly that form);
    style-token attention: keys / values depend on parameters only and are projected once at load time."""

    def __init__(self, w, p, cfg, device):
        self.device = device
        self.k = cfg['ref_encoder_kernel_size']
        self.stride = cfg['ref_encoder_strides']
        if self.k != 3:
            raise NotImplementedError('reference encoder kernels other than 3x3')
        filters = cfg['ref_encoder_filters']
        self.filters = filters
        # first layer: kernel (3, 3, 1, C0)
        k0 = np.asarray(w[f'{p}.conv0.kernel'])
        s, b = fold_batchnorm(w[f'{p}.norm0.gamma'], w[f'{p}.norm0.beta'], w[f'{p}.norm0.moving_mean'], w[f'{p}.norm0.moving_variance'])
        self.w0 = _dev_vec(k0.reshape(9, filters[0]), device)
        self.s0 = _dev_vec(s, device)
        self.b0 = _dev_vec(np.asarray(w[f'{p}.conv0.bias'], np.float32) * s + b, device)
        self.convs = []
        for i in range(1, len(filters)):
            kern = np.asarray(w[f'{p}.conv{i}.kernel'])                 # (3, 3, cin, cout)
            s, b = fold_batchnorm(w[f'{p}.norm{i}.gamma'], w[f'{p}.norm{i}.beta'], w[f'{p}.norm{i}.moving_mean'],
                                  w[f'{p}.norm{i}.moving_variance'])
            cin, cout = kern.shape[2], kern.shape[3]
            self.convs.append(PackedLinear(kern.reshape(9 * cin, cout).T, bias=np.asarray(w[f'{p}.conv{i}.bias'], np.float32) * s + b,
                                           scale=s, device=device, taps=9, act='relu'))
        u = cfg['ref_encoder_gru_cell_units']
        self.u = u

        def reorder(a):          # Keras gates [z | r | h] -> [r | z | n]
            return np.concatenate([a[..., u:2 * u], a[..., :u], a[..., 2 * u:]], axis=-1)

        bias = np.asarray(w[p + '.gru.bias'])
        self.gru_ih = PackedLinear(reorder(np.asarray(w[p + '.gru.kernel'])).T, bias=reorder(bias[0]), device=device)
        self.gru_hh = PackedLinear(reorder(np.asarray(w[p + '.gru.recurrent_kernel'])).T, bias=reorder(bias[1]), device=device)
        self.rnn_proj = pack_dense(w, p + '.rnn_proj', device, act='tanh')
        m = p + '.mha'
        self.heads = cfg['gst_multi_num_heads']
        self.sd = cfg['gst_style_embed_dim']
        self.wq = pack_dense(w, m + '.wq', device)
        self.out = pack_dense(w, m + '.dense', device)
        tok = np.tanh(np.asarray(w[p + '.gst_tokens'], np.float64))
        k = tok @ np.asarray(w[m + '.wk.kernel'], np.float64) + np.asarray(w[m + '.wk.bias'], np.float64)
        v = tok @ np.asarray(w[m + '.wv.kernel'], np.float64) + np.asarray(w[m + '.wv.bias'], np.float64)
        self.keys, self.values = _dev_vec(k, device), _dev_vec(v, device)
        self.n_tokens = tok.shape[0]
        self.gst_tokens = np.asarray(w[p + '.gst_tokens'])

    def run(self, x, precision, return_attention=True):
        """x: [B, T, mel] float32 device tensor -> (style embedding [B, sd], attention [B, heads, 1, tokens] or None)."""
        B, H, W = x.shape
        precision = ops.precision_at(precision, 'gst')
        dev, f32 = self.device, dict(dtype=torch.float32, device=self.device)
        Ho, pt = same_pad(H, 3, self.stride)
        Wo, pl = same_pad(W, 3, self.stride)
        y = torch.empty(B, Ho, Wo, self.filters[0], **f32)
        ops.conv2d_c1(x.contiguous(), self.w0, self.s0, self.b0, y, self.stride, pt, pl)
        H, W = Ho, Wo
        for lin in self.convs:
            Ho, pt = same_pad(H, 3, self.stride)
            Wo, pl = same_pad(W, 3, self.stride)
            out = torch.empty(B * Ho * Wo, lin.N, **f32)
            ops.gemm(y.view(B * H * W, -1), lin, out, M=B * Ho * Wo, precision=precision,
                     conv2d=(H, W, Ho, Wo, 3, self.stride, pt, pl))
            y, H, W = out, Ho, Wo
        steps, feat = H, W * self.filters[-1]
        seq = y.view(B * steps, feat)
        u = self.u
        gi = torch.empty(B * steps, 3 * u, **f32)
        ops.gemm(seq, self.gru_ih, gi, precision=precision)
        gi3 = gi.view(B, steps, 3 * u)
        state = (torch.zeros(B, u, **f32), torch.empty(B, u, **f32))
        gh = torch.empty(B, 3 * u, **f32)
        for t in range(steps):       # recurrent product + gates in one launch per step; the state alternates between two buffers
            ops.gru_step(gi3[:, t], gh, state[t & 1], self.gru_hh, state[(t + 1) & 1], precision=precision)
        h = state[steps & 1]
        qcat = torch.empty(B, u + self.sd, **f32)
        ops.gemm(h, self.rnn_proj, qcat[:, :u], precision=precision)
        q = torch.empty(B, self.sd, **f32)
        ops.gemm(qcat[:, :u], self.wq, q, precision=precision)
        attn = torch.empty(B, self.heads, 1, self.n_tokens, **f32) if return_attention else None
        ops.attention(q, self.keys, self.values, qcat[:, u:], B, self.heads, 1, self.n_tokens, self.sd // self.heads,
                      attn=attn, precision=precision, kv_shared=True)
        out = torch.empty(B, self.sd, **f32)
        ops.gemm(qcat, self.out, out, precision=precision)
        return out, attn


class AutoregressiveTransformer:
    def __init__(self, weights=None, config=None, device='cuda', precision='bf16x3', fold_ffn=True, seed=1234, **keywords):
        """weights: dict name -> ndarray (weights.py layouts), config: dict with the keys of config/*/autoregressive_config.yaml
        (what ConfigManager.get_model passes); or the reference's own keyword form
        (AutoregressiveTransformer(system_type=..., mel_channels=..., encoder_model_dimension=..., ..., max_r=...),
        utils/config_manager.py:119-162), in which case the variables are initialised from `seed` as Keras would from its
        initialisers."""
        if not torch.cuda.is_available():
            raise RuntimeError('AutoregressiveTransformer needs a GPU: the hot path has no CPU implementation')
        ops._lib.load()
        if config is None:
            from . import configs, weights as _weights
            config = configs.config_from_keywords('autoregressive', **keywords)
            if weights is None:
                weights = _weights.init_autoregressive_transformer(config, seed=seed)
        elif keywords:
            raise TypeError(f'unexpected keywords beside a config dict: {sorted(keywords)}')
        cfg = self.config = dict(config)
        self.device = torch.device(device)
        self.precision = ops.as_precision(precision)
        self.system_type = cfg['system_type']
        self.mel_channels = cfg['mel_channels']
        self.max_r = cfg['max_r']
        self.r = self.max_r
        self.drop_n_heads = 0
        self.stop_prob_index = 2
        self.fuse_step_norms = True       # incremental decoding: row norms inside the one-row products that consume them
        self.fuse_decoder_block = True    # ... and a position's whole decoder block in one launch (estts_decoder_block_row)
        self.fuse_cross_attention = True  # ... and a position's whole cross-attention block in one launch (estts_cross_attention_row)
        self.decoder_prenet_dropout = 0.0
        # DecoderPrenet's dropout is active at inference too (layers.py:394-402).  Masks come from the counter-based
        # generator keyed by (seed, site, decoder position, channel); the seed advances once per decoder call.
        self.dropout_seed = 0
        w, dev = weights, self.device
        de, dd = cfg['encoder_model_dimension'], cfg['decoder_model_dimension']
        self.de, self.dd = de, dd
        self.start_vec = torch.full((1, self.mel_channels), float(cfg['mel_start_value']), device=dev)
        self.end_vec = torch.full((1, self.mel_channels), float(cfg['mel_end_value']), device=dev)
        self.tokenizer = Tokenizer(default_alphabet(), add_start_end=True)
        self.emb = _dev_vec(w['text_encoder_prenet.embeddings'], dev)
        if self.emb.shape[1] != de:
            raise ValueError('encoder_prenet_dimension must equal encoder_model_dimension')
        self.pe_enc = ops.positional_encoding_table(cfg['encoder_max_position_encoding'], de, dev) * float(
            w['text_encoder.pos_encoding_scalar'])
        self.pe_dec = ops.positional_encoding_table(cfg['decoder_max_position_encoding'], dd, dev) * float(
            w['decoder.pos_encoding_scalar'])
        self.text_encoder = SelfAttentionStack(w, 'text_encoder', 'TextEncoder', de, cfg['encoder_num_heads'],
                                               cfg['encoder_dense_blocks'], dev, fold_ffn=fold_ffn)
        self.style_encoder = None
        if self.system_type in ('speaker_style_text', 'style_text'):
            self.style_encoder = ReferenceEncoderGST(w, 'style_encoder', cfg, dev)
        self.enc_dim = encoder_output_dim(cfg)
        self.prenet1 = pack_dense(w, 'decoder_prenet.d1', dev, act='relu')
        self.prenet2 = pack_dense(w, 'decoder_prenet.d2', dev, act='relu')
        self.decoder = CrossAttentionStack(w, 'decoder', 'Decoder', dd, cfg['decoder_num_heads'], cfg['decoder_dense_blocks'],
                                           'causal', dev, fold_ffn=fold_ffn)
        self._final_w = (np.asarray(w['final_proj_mel.kernel']), np.asarray(w['final_proj_mel.bias']))
        self._final = {}
        self.stop_linear = pack_dense(w, 'decoder_postnet.stop_linear', dev)
        self.postnet = ConvBNLayer(w, 'decoder_postnet.conv_blocks', cfg['postnet_conv_layers'], 'tanh', 'linear', 'causal', dev)

    # ------------------------------------------------------------------ helpers
    def _final_proj(self, r):
        """FinalProj restricted to the first r * mel_channels outputs (AutoregressiveTransformer.py:173)."""
        lin = self._final.get(r)
        if lin is None:
            n = r * self.mel_channels
            lin = PackedLinear(self._final_w[0][:, :n].T, bias=self._final_w[1][:n], device=self.device)
            self._final[r] = lin
        return lin

    def _set_r(self, r):
        self.r = r

    def _set_heads(self, heads):
        self.drop_n_heads = heads

    def set_constants(self, decoder_prenet_dropout=None, learning_rate=None, reduction_factor=None, drop_n_heads=None):
        """AutoregressiveTransformer.py:298-307."""
        if decoder_prenet_dropout is not None:
            self.decoder_prenet_dropout = float(decoder_prenet_dropout)
        if reduction_factor is not None:
            self._set_r(int(reduction_factor))
        if drop_n_heads is not None:
            self._set_heads(drop_n_heads)

    def encode_text(self, text):
        return self.tokenizer(text)

    def encode_ref(self, targets):
        """AutoregressiveTransformer.py:312-316: drop the last frame, keep every r-th."""
        tar = torch.as_tensor(targets, dtype=torch.float32, device=self.device)[None]
        return tar[:, :-1][:, 0::self.r, :]

    def input_reshape(self, tar, stop_prob):
        """AutoregressiveTransformer.py:318-327."""
        tar_inp, tar_real = tar[:, :-1], tar[:, 1:]
        return tar_real, tar_inp[:, 0::self.r, :], stop_prob[:, 1:], int(tar_inp.shape[1])

    # ------------------------------------------------------------------ reference surface
    def call_encoder(self, inputs, targets, spk_embed, training_text_encoder=False, training_style_encoder=False,
                     return_attention=True):
        """AutoregressiveTransformer.py:124-159."""
        if training_text_encoder or training_style_encoder:
            raise NotImplementedError('training runs through estts.autoregressive_training.AutoregressiveTransformerTrainer')
        tokens = torch.as_tensor(inputs, device=self.device).to(torch.int32).contiguous()
        B, N = tokens.shape
        dev, f32, de = self.device, dict(dtype=torch.float32, device=self.device), self.de
        ws = self.text_encoder.alloc(B * N, dev)
        key_mask = torch.empty(B, N, **f32)
        ops.embed_posenc(tokens, self.emb, self.pe_enc, None, ws['xc'], key_mask, math.sqrt(np.float32(de)))
        text_attn = self.text_encoder.run(ws, B, N, key_mask, self.precision, return_attention)
        text_out = ws['xc'][:, :de]
        enc = torch.empty(B * N, self.enc_dim, **f32)
        enc3 = enc.view(B, N, self.enc_dim)
        enc3[:, :, :de].copy_(text_out.reshape(B, N, de))
        col = de
        gst_out = gst_attn = gst_tokens = None
        if self.style_encoder is not None:
            tg = torch.as_tensor(targets, dtype=torch.float32, device=dev)
            gst_out, gst_attn = self.style_encoder.run(tg, self.precision, return_attention)
            sd = gst_out.shape[1]
            enc3[:, :, col:col + sd].copy_(gst_out[:, None, :].expand(B, N, sd))
            col += sd
            gst_tokens = {'GST_tokens': self.style_encoder.gst_tokens}
            gst_attn = {'RefEncoderGST_attention': gst_attn}
            gst_out = gst_out.view(B, 1, sd)
        if self.system_type in ('speaker_style_text', 'speaker_text'):
            spk = torch.as_tensor(spk_embed, dtype=torch.float32, device=dev).reshape(B, 1, -1)
            enc3[:, :, col:col + spk.shape[2]].copy_(spk.expand(B, N, spk.shape[2]))
        enc_mask = torch.empty(B, N, **f32)
        ops.row_zero_mask(enc, enc_mask.view(-1))
        return (enc3, enc_mask.view(B, 1, 1, N), text_attn, gst_attn, gst_tokens, gst_out,
                text_out.reshape(B, N, de))          # a view of this call's own workspace (allocated per call)

    def _next_dropout_seed(self):
        self.dropout_seed = (self.dropout_seed * 1000003 + 7919) & 0xFFFFFFFF
        return self.dropout_seed

    def call_decoder(self, encoder_output, targets, encoder_padding_mask, training=False, return_attention=True,
                     dropout_seed=None):
        """AutoregressiveTransformer.py:161-181.  dropout_seed: seed of the prenet dropout masks (rows of the generator
        are b * Tq + t); None draws the next one of this model's sequence."""
        if training:
            raise NotImplementedError('training runs through estts.autoregressive_training.AutoregressiveTransformerTrainer')
        rate = self.decoder_prenet_dropout
        seed = self._next_dropout_seed() if dropout_seed is None else int(dropout_seed) & 0xFFFFFFFF
        dev, f32, dd, r = self.device, dict(dtype=torch.float32, device=self.device), self.dd, self.r
        tg = torch.as_tensor(targets, dtype=torch.float32, device=dev).contiguous()
        B, Tq, C = tg.shape
        N = encoder_output.shape[1]
        if Tq * r > self.pe_dec.shape[0] + r - 1:
            raise ValueError('decoder input exceeds decoder_max_position_encoding')
        enc = encoder_output.reshape(B * N, -1)
        enc_mask = encoder_padding_mask.reshape(B, N).contiguous()
        ws = self.decoder.alloc(B * Tq, B * N, dev)
        self.decoder.project_kv(ws, enc, self.precision)
        dec_mask = torch.empty(B, Tq, **f32)
        ops.row_zero_mask(tg.view(B * Tq, C), dec_mask.view(-1))
        p1 = torch.empty(B * Tq, self.prenet1.N, **f32)
        ops.gemm(tg.view(B * Tq, C), self.prenet1, p1, precision=self.precision, dropout=(rate, seed, 0, 0), site='prenet')
        pe = self.pe_dec[0:Tq * r:r].contiguous()
        ops.gemm(p1, self.prenet2, ws['xc'][:, :dd], alpha=math.sqrt(np.float32(dd)), residual=pe, r_mod=Tq,
                 precision=self.precision, dropout=(rate, seed, 1, 0), site='prenet')
        attn = self.decoder.run(ws, B, Tq, N, dec_mask, enc_mask, self.precision, return_attention)
        dec_out = ws['xc'][:, :dd]
        lin = self._final_proj(r)
        proj = torch.empty(B * Tq, r * C, **f32)
        ops.gemm(dec_out, lin, proj, precision=self.precision, site='out')
        mel = proj.view(B * Tq * r, C)                                    # (b, t, r, c) -> rows (b, t*r + j)
        T = Tq * r
        stop = torch.empty(B * T, 3, **f32)
        ops.gemm(mel, self.stop_linear, stop, precision=self.precision, site='stop')
        hid = max(self.postnet.hidden(), 4)
        final = torch.empty(B * T, C, **f32)
        self.postnet.run(mel, torch.empty(B * T, hid, **f32), torch.empty(B * T, hid, **f32), final, T, self.precision, site='postnet')
        mel3 = mel.view(B, T, C)
        return {'mel_linear': mel3, 'final_output': final.view(B, T, C), 'stop_prob': stop.view(B, T, 3),
                'decoder_attention': attn, 'decoder_output': dec_out.reshape(B, Tq, dd), 'linear': mel3}      # (a view of this call's workspace)

    def call(self, inputs, targets, spk_embed=None, train_text_encoder=False, train_style_encoder=False, train_decoder=False,
             return_attention=True):
        """AutoregressiveTransformer.py:243-263."""
        enc, mask, text_attn, gst_attn, gst_tokens, gst_out, text_out = self.call_encoder(
            inputs, targets, spk_embed, train_text_encoder, train_style_encoder, return_attention)
        out = self.call_decoder(enc, targets, mask, train_decoder, return_attention)
        out.update({'text_encoder_attention': text_attn, 'gst_encoder_attention': gst_attn, 'gst_tokens': gst_tokens,
                    'gst_output': gst_out, 'text_enc_output': text_out})
        return out

    __call__ = call

    def _compile(self, stop_scaling, optimizer=None):
        """AutoregressiveTransformer.py:224-231: masked MAE on the post-net output, stop-token cross entropy whose stop class
        (index 2) weighs `stop_scaling`, masked MAE on the linear output; all three weights 1."""
        self.loss_weights = [1., 1., 1.]
        self.stop_scaling = float(stop_scaling)
        self.optimizer = optimizer

    def _gta_forward(self, inp, tar, stop_prob, spk_embed=None):
        """AutoregressiveTransformer.py:195-218 with training=False: teacher-forced pass on tar[:, :-1] (every r-th frame), the
        three losses against tar[:, 1:] / stop_prob[:, 1:] over the first mel_len output frames."""
        from . import train_ops
        if not hasattr(self, 'loss_weights'):
            raise RuntimeError('call _compile(stop_scaling) first')
        dev = self.device
        tar = torch.as_tensor(tar, dtype=torch.float32, device=dev)
        stop = torch.as_tensor(stop_prob, device=dev).to(torch.int32)
        tar_real, tar_mel, tar_stop, mel_len = self.input_reshape(tar, stop)
        out = self.call(inp, tar_mel.contiguous(), spk_embed)
        B, C = tar.shape[0], tar.shape[2]
        M = B * mel_len
        losses = torch.zeros(3, dtype=torch.float32, device=dev)
        target2 = tar_real.reshape(M, C)
        ops.masked_mae(target2, out['final_output'][:, :mel_len, :].reshape(M, C), losses[0:1])
        train_ops.scaled_xent(out['stop_prob'][:, :mel_len, :].reshape(M, 3), tar_stop.reshape(M).contiguous(), losses[1:2],
                              index=2, scaling=self.stop_scaling)
        ops.masked_mae(target2, out['mel_linear'][:, :mel_len, :].reshape(M, C), losses[2:3])
        w = self.loss_weights
        out['tts_loss'] = losses[0] * w[0] + losses[1] * w[1] + losses[2] * w[2]
        out['losses'] = {'output': losses[0], 'stop_prob': losses[1], 'mel_linear': losses[2]}
        out['reduced_target'] = tar_mel
        return out

    def val_step(self, inp, tar, stop_prob, spk_embed=None):
        """AutoregressiveTransformer.py:220-222 (what extract_durations.py:125,144 runs over the data set)."""
        return self._gta_forward(inp, tar, stop_prob, spk_embed)

    # ------------------------------------------------------------------ incremental decoding
    @staticmethod
    def _as_dense(lin):
        """The same packed convolution weight seen as a dense layer over a flattened window of `taps` rows."""
        import copy
        d = copy.copy(lin)
        d.taps, d.lpad, d.c_per_tap = 1, 0, lin.K
        return d

    @staticmethod
    def _windows(hist, t0, rows, k, cin):
        """[rows, k * cin] view of a row-major history buffer [(k - 1) + T, cin] (k - 1 leading zero rows): row j is the
        causal window that ends at time t0 + j."""
        return torch.as_strided(hist, (rows, k * cin), (cin, 1), hist.storage_offset() + t0 * cin)

    def _predict_incremental(self, inp, targets, spk_embed, max_length, encode, verbose, program=True, dropout_seed=None,
                             stop_check_every=8):
        rate = self.decoder_prenet_dropout
        seed = self._next_dropout_seed() if dropout_seed is None else int(dropout_seed) & 0xFFFFFFFF
        dev, f32 = self.device, dict(dtype=torch.float32, device=self.device)
        pr = ops.precision_at(self.precision, 'incremental')       # one-row products: latency-bound at either precision
        d, r, C = self.dd, self.r, self.mel_channels
        if targets is not None:
            targets = self.encode_ref(targets)
        if spk_embed is not None:
            spk_embed = torch.as_tensor(spk_embed, dtype=torch.float32, device=dev)[None]
        if encode:
            inp = self.encode_text(inp)
        inp = torch.as_tensor(np.asarray(inp), dtype=torch.int32, device=dev)[None]
        enc, mask, text_attn, gst_attn, gst_tokens, _, _ = self.call_encoder(inp, targets, spk_embed)
        N = enc.shape[1]
        enc2 = enc.reshape(N, -1)
        enc_mask = mask.reshape(1, N).contiguous()
        S = int(max_length // r) + 1                                  # decoder steps at most
        if (S - 1) * r >= self.pe_dec.shape[0]:
            raise ValueError('max_length exceeds decoder_max_position_encoding')
        blocks = self.decoder.blocks
        H = [ca.heads for _, ca, _ in blocks]
        # encoder keys / values once per block; per-position caches
        kv_enc = [torch.empty(N, 2 * d, **f32) for _ in blocks]
        for (_, ca, _), kv in zip(blocks, kv_enc):
            ca.project_kv(enc2, kv, pr)
        qkv = [torch.empty(S, 3 * d, **f32) for _ in blocks]         # row i = [q | k | v] of decoder position i
        slab = [(h * N + 3) // 4 * 4 for h in H]                       # per-step weights [1, h, 1, N], 16-byte aligned slabs
        attn = [torch.empty(S, sl, **f32) for sl in slab]
        conv_hist, conv_dense = [], []
        for _, _, ff in blocks:
            if isinstance(ff, FFNLayer):
                conv_hist.append(None)
                conv_dense.append(None)
            else:
                k = ff.layers[0].taps
                conv_hist.append([torch.zeros(k - 1 + S, lin.c_per_tap, **f32) for lin in ff.layers])
                conv_dense.append([self._as_dense(lin) for lin in ff.layers])
        kp = self.postnet.layers[0].taps
        post_hist = [torch.zeros(kp - 1 + S * r, lin.c_per_tap, **f32) for lin in self.postnet.layers]
        post_dense = [self._as_dense(lin) for lin in self.postnet.layers]
        # rows r - 1 + i r of `frames` feed step i: the start vector, then the last frame of every earlier step
        frames = torch.empty(r + S * r, C, **f32)
        frames[r - 1:r].copy_(self.start_vec)
        final = frames[r:]
        stop = torch.empty(S * r, 3, **f32)
        dec_mask = torch.zeros(1, S, **f32)
        xc, xc2 = torch.empty(1, 2 * d, **f32), torch.empty(1, 2 * d, **f32)
        y, x2, q = torch.empty(1, d, **f32), torch.empty(1, d, **f32), torch.empty(1, d, **f32)
        p1 = torch.empty(1, self.prenet1.N, **f32)
        hid = torch.empty(1, max(self.decoder.hidden, 4), **f32)
        lin_r = self._final_proj(r)
        sqrt_d = math.sqrt(np.float32(d))

        y2 = torch.empty(1, d, **f32)
        y2s, x2s = [torch.empty(1, d, **f32) for _ in range(2)], [torch.empty(1, d, **f32) for _ in range(2)]   # (a block reads one pair, writes the other)
        # A row-wise normalisation is folded into the one-row product that consumes it (gemm(ln=...): every column tile
        # recomputes the norm of the single row and tile 0 writes it out for the later residual uses), which takes three
        # launches per block off the step.  `pending` describes the normalisation whose result xc[:, :d] the next consumer
        # needs: None when the buffer already holds it.
        fuse = pr == 'bf16x3' and d % 8 == 0 and d <= 512 and self.fuse_step_norms
        cross_row = [self.fuse_cross_attention and ops.can_cross_attention_row(ca.wq, ca.out, d, ca.heads, N, pr) for _, ca, _ in blocks]
        # ... and a self-attention block's attention over its cache + output projection (the cache grows to S rows)
        # Both row kernels hold the cache in registers, up to S_head rows: a longer utterance decodes its first S_head positions
        # with them and the rest with the launches per layer (two step programs over the same buffers).
        S_head = min(S, 256)
        self_row = [self.fuse_cross_attention and ops.can_cross_attention_row(None, sa.out, d, sa.heads, S_head, pr) for sa, _, _ in blocks]

        # ... or the whole block (self-attention, cross-attention, folded feed-forward product): four workgroups per position
        # that meet at barriers of their own inside the kernel
        block_row = [fuse and self.fuse_decoder_block and isinstance(ff, FFNLayer) and ff.fold and sa.heads == ca.heads and
                     ops.can_decoder_block_row(sa.qkv, sa.out, ca.wq, ca.out, ff.d12, d, sa.heads, S_head, N, pr) for sa, ca, ff in blocks]
        head_kernels = (self_row, block_row)
        tail_kernels = ([False] * len(blocks), [False] * len(blocks))

        prenet_row = fuse and self.fuse_decoder_block and d == 256 and ops.can_prenet_row(self.prenet1, self.prenet2, C, pr)

        def step(i):
            self_row, block_row = head_kernels if in_head[0] else tail_kernels
            frame = frames[r - 1 + i * r:r + i * r]
            if prenet_row:
                ops.prenet_row(frame, dec_mask[0, i:i + 1], self.prenet1, self.prenet2, xc[:, :d], alpha=sqrt_d,
                               residual=self.pe_dec[i * r:i * r + 1], dropout=(rate, seed, 0, 1, i))
            else:
                ops.row_zero_mask(frame, dec_mask[0, i:i + 1])
                ops.gemm(frame, self.prenet1, p1, precision=pr, dropout=(rate, seed, 0, i))
                ops.gemm(p1, self.prenet2, xc[:, :d], alpha=sqrt_d, residual=self.pe_dec[i * r:i * r + 1], precision=pr,
                         dropout=(rate, seed, 1, i))
            pending = None
            for b, (sa, ca, ff) in enumerate(blocks):
                if block_row[b]:
                    ops.decoder_block_row(sa.qkv, sa.out, sa.ln, sa.last_ln, ca.wq, ca.out, ca.ln, ff.d12, qkv[b], i,
                                          kv_enc[b][:, :d], kv_enc[b][:, d:], y2s[b & 1], x2s[b & 1],
                                          x=None if pending is not None else xc[:, :d], ln=pending, self_mask=dec_mask,
                                          key_mask=enc_mask, attn=attn[b][i:i + 1, :ca.heads * N])
                    pending = dict(y=y2s[b & 1], x=x2s[b & 1], ln_a=ff.ln, act='relu', ln_b=ff.last_ln, out=xc[:, :d])
                    continue
                row = qkv[b][i:i + 1]
                ops.gemm(xc[:, :d], sa.qkv, row, precision=pr, ln=pending)
                pending = None
                if self_row[b]:
                    ops.cross_attention_row(None, sa.out, qkv[b][:i + 1, d:2 * d], qkv[b][:i + 1, 2 * d:], y, sa.heads, x=xc[:, :d],
                                            q=row[:, :d], key_mask=dec_mask[:, :i + 1])
                else:
                    ops.attention(row[:, :d], qkv[b][:i + 1, d:2 * d], qkv[b][:i + 1, 2 * d:], xc[:, d:], 1, sa.heads, 1, i + 1,
                                  d // sa.heads, key_mask=dec_mask[:, :i + 1], precision=pr)
                    ops.gemm(xc, sa.out, y, precision=pr)
                if fuse and cross_row[b]:
                    # the whole cross-attention block of this position in one launch: the pending LN(LN(y) + x), the query
                    # projection, attention over the encoder's keys / values and the output projection (y is read, then written)
                    ops.cross_attention_row(ca.wq, ca.out, kv_enc[b][:, :d], kv_enc[b][:, d:], y, ca.heads,
                                            ln=dict(y=y, x=xc[:, :d], ln_a=sa.ln, ln_b=sa.last_ln, out=xc2[:, :d]),
                                            key_mask=enc_mask, attn=attn[b][i:i + 1, :ca.heads * N])
                else:
                    if fuse:
                        ops.gemm(xc2[:, :d], ca.wq, q, precision=pr,
                                 ln=dict(y=y, x=xc[:, :d], ln_a=sa.ln, ln_b=sa.last_ln, out=xc2[:, :d]))
                    else:
                        ops.rownorm(y, xc2[:, :d], ln_a=sa.ln, residual=xc[:, :d], ln_b=sa.last_ln)
                        ops.gemm(xc2[:, :d], ca.wq, q, precision=pr)
                    ops.attention(q, kv_enc[b][:, :d], kv_enc[b][:, d:], xc2[:, d:], 1, ca.heads, 1, N, d // ca.heads,
                                  key_mask=enc_mask, attn=attn[b][i, :ca.heads * N].view(1, ca.heads, 1, N), precision=pr)
                    ops.gemm(xc2, ca.out, y, precision=pr)
                if isinstance(ff, FFNLayer):
                    first = ff.d12 if ff.fold else ff.d1
                    if fuse:
                        # FFNResNorm on LN(y + x): x2 = LN(y + xc2) inside the first product, which writes y2 (or hid);
                        # its own LN(relu(LN(y2)) + x2) is left to the next consumer of xc[:, :d]
                        ops.gemm(x2, first, y2 if ff.fold else hid, precision=pr, site='ffn',
                                 ln=dict(y=y, x=xc2[:, :d], ln_b=ca.ln, out=x2))
                        if not ff.fold:
                            ops.gemm(hid, ff.d2, y2, precision=pr, site='ffn')
                        pending = dict(y=y2, x=x2, ln_a=ff.ln, act='relu', ln_b=ff.last_ln, out=xc[:, :d])
                    else:
                        ops.rownorm(y, x2, residual=xc2[:, :d], ln_b=ca.ln)
                        ff.run(x2, hid, y, xc[:, :d], pr)
                else:
                    hist, dense = conv_hist[b], conv_dense[b]
                    k = ff.layers[0].taps
                    src_row = hist[0][k - 1 + i:k + i]                 # this position's row of the first history
                    ops.rownorm(y, src_row, residual=xc2[:, :d], ln_b=ca.ln)
                    for li, lin in enumerate(dense[:-1]):
                        ops.gemm(self._windows(hist[li], i, 1, k, lin.K // k), lin, hist[li + 1][k - 1 + i:k + i], precision=pr)
                    last = len(dense) - 1
                    ops.gemm(self._windows(hist[last], i, 1, k, dense[last].K // k), dense[last], xc[:, :d], residual=src_row,
                             precision=pr)
            mel_rows = post_hist[0][kp - 1 + i * r:kp - 1 + (i + 1) * r]          # FinalProj writes the r new mel frames here
            ops.gemm(xc[:, :d], lin_r, mel_rows.view(1, r * C), precision=pr, ln=pending)
            ops.gemm(mel_rows, self.stop_linear, stop[i * r:(i + 1) * r], precision=pr)
            for li, lin in enumerate(post_dense[:-1]):
                ops.gemm(self._windows(post_hist[li], i * r, r, kp, lin.K // kp), lin,
                         post_hist[li + 1][kp - 1 + i * r:kp - 1 + (i + 1) * r], precision=pr)
            last = len(post_dense) - 1
            ops.gemm(self._windows(post_hist[last], i * r, r, kp, post_dense[last].K // kp), post_dense[last],
                     final[i * r:(i + 1) * r], residual=mel_rows, precision=pr)

        # every launch argument is affine in i: record steps 0 and 1 once, then issue each step from one C call
        in_head = [True]
        prog = ops.StepProgram(step) if program and S >= 2 else None
        prog_tail = None
        if S > S_head:
            in_head[0] = False
            prog_tail = ops.StepProgram(step) if program else None
        # The reference asks the host after every step whether the stop token won (AutoregressiveTransformer.py:289-293).  Here the
        # steps are issued in groups and the stop rows of a group are read back together: the answer is the same (the first
        # step whose stop class wins ends the utterance; what was decoded past it is dropped), with one device-to-host wait
        # per group instead of per step.
        group = 1 if verbose else max(1, int(stop_check_every))
        steps, done, i = 0, False, 0
        while i < S and not done:
            n = min(group, S - i)
            for j in range(i, i + n):
                in_head[0] = j < S_head
                pj = prog if in_head[0] else prog_tail
                if pj is not None:
                    pj.run(j)
                else:
                    step(j)
            rows = stop[(i + 1) * r - 1:(i + n) * r:r].cpu()               # the last frame's stop logits of every step of the group
            for j in range(n):
                steps = i + j + 1
                if verbose:
                    print(f'\rpred text mel: {i + j} stop out: {float(rows[j, 2])}', end='')
                if int(torch.argmax(rows[j])) == self.stop_prob_index:
                    if verbose:
                        print('Stopping')
                    done = True
                    break
            i += n
        dec_attn = {name: a[:steps, :h * N].reshape(steps, h, N).permute(1, 0, 2)[None].contiguous()
                    for name, a, h in zip(self.decoder.names, attn, H)}
        return {'mel': final[:steps * r].clone(), 'decoder_attention': dec_attn, 'text_encoder_attention': text_attn,
                'gst_encoder_attention': gst_attn, 'gst_tokens': gst_tokens}

    def predict(self, inp, targets, spk_embed, max_length=1000, encode=True, verbose=True, kv_cache=True, stop_check_every=8):
        """AutoregressiveTransformer.py:265-296.  inp: phoneme string (encode=True) or token ids; targets: reference mel
        (T, 80) for the style encoder or None; spk_embed: (1, D) or None.

        kv_cache=True (default) decodes incrementally: every step computes only the new decoder position against cached
        self-attention keys / values, the once-projected encoder keys / values and the causal-convolution histories,
        which gives the same numbers as the reference's loop (all its masks are causal) in O(T) decoder work.
        kv_cache=False re-runs the whole decoder on the growing output each step, literally as the reference does.
        With decoder_prenet_dropout > 0 the two differ in distribution only: the reference (and kv_cache=False) redraws the
        prenet dropout masks of every earlier position at each step, the cached decoder draws each position's mask once.
        stop_check_every: decoder steps issued between two reads of the stop decision (same output for any value; 1 with
        verbose=True, which prints every step's stop value as the reference does)."""
        if kv_cache:
            return self._predict_incremental(inp, targets, spk_embed, max_length, encode, verbose, stop_check_every=stop_check_every)
        dev = self.device
        if targets is not None:
            targets = self.encode_ref(targets)
        if spk_embed is not None:
            spk_embed = torch.as_tensor(spk_embed, dtype=torch.float32, device=dev)[None]
        if encode:
            inp = self.encode_text(inp)
        inp = torch.as_tensor(np.asarray(inp), dtype=torch.int32, device=dev)[None]
        output = self.start_vec[None].clone()
        output_concat = self.start_vec[None].clone()
        out_dict = {}
        enc, mask, text_attn, gst_attn, gst_tokens, _, _ = self.call_encoder(inp, targets, spk_embed)
        for i in range(int(max_length // self.r) + 1):
            model_out = self.call_decoder(enc, output, mask)
            output = torch.cat([output, model_out['final_output'][:1, -1:, :]], dim=-2)
            output_concat = torch.cat([output_concat, model_out['final_output'][:1, -self.r:, :]], dim=-2)
            stop_pred = model_out['stop_prob'][:, -1].cpu()        # host decision every step, as in the reference
            out_dict = {'mel': output_concat[0, 1:, :], 'decoder_attention': model_out['decoder_attention'],
                        'text_encoder_attention': text_attn, 'gst_encoder_attention': gst_attn, 'gst_tokens': gst_tokens}
            if verbose:
                print(f'\rpred text mel: {i} stop out: {float(stop_pred[0, 2])}', end='')
            if int(torch.argmax(stop_pred, dim=-1)) == self.stop_prob_index:
                if verbose:
                    print('Stopping')
                break
        return out_dict
