"""Training of the AutoregressiveTransformer (GST TransformerTTS) on MI355X: the TTS half of the reference's
`train_models_step` (TransformerTTS/model/traning_steps.py:16-56): teacher-forced forward in training mode, masked MAE on
the post-net output + scaled stop-token cross entropy + masked MAE on the linear output, hand-written backward, Adam.

Built from the layer passes of forward_training.ForwardTransformerTrainer (self-attention / FFN / conv-BN blocks with
their backward closures, flat parameter store, Adam, re-packing) plus what the autoregressive model adds
(model/AutoregressiveTransformer.py:124-181, model/layers.py:258-376): causal decoder self-attention, cross-attention over
the concatenated encoder output [text | style | speaker], the reduction factor (FinalProj restricted to r frames per
decoder step), the stop-token head and the causal post-net.

The reference's three flags train_text_encoder / train_style_encoder / train_decoder select Keras' `training=` mode of the
three sub-networks (dropout, head drop, BatchNorm batch statistics), not which variables receive gradients: the tape
differentiates all trainable variables either way (traning_steps.py:54-55).  The shipped configuration sets all three
True; a False flag runs that sub-network as Keras would with training=False (no dropout, no head drop, BatchNormalization on
its moving statistics) and still differentiates it.
"""
import contextlib
import math

import numpy as np
import torch

from . import ops, parallel, train_ops as T
from .forward_training import (ForwardTransformerTrainer, ParamStore, _CnnResNorm, _Ffn, _Layer, _Norm, _Sarn)
from .train_ops import TrainLinear
from .ops import same_pad
from .weights import encoder_output_dim


class _Carn:
    """CrossAttentionResnorm (layers.py:258-274): wq over the decoder stream, wk | wv fused over the encoder output,
    output projection over [q_in | ctx], LayerNorm(dropout(attn) + q_in)."""

    def __init__(self, store, p, d, heads):
        self.d, self.H = d, heads
        self.wq = _Layer(store, p + '.mha.wq')
        self.wkv = _Layer(store, p + '.mha.wkv')
        self.wo = _Layer(store, p + '.mha.dense', need_bwd=False)
        k = store.p[p + '.mha.dense.kernel']
        self.wo_x = TrainLinear(k[:d], None, need_fwd=False)
        self.wo_c = TrainLinear(k[d:], None, need_fwd=False)
        self.ln = _Norm(store, p + '.layernorm', False)

    def linears(self):
        return [self.wq.lin, self.wkv.lin, self.wo.lin, self.wo_x, self.wo_c]


def gru_gate_swap(a):
    """[..., 3u] with gate blocks [z | r | h] <-> [r | z | n]."""
    u = a.shape[-1] // 3
    return np.concatenate([a[..., u:2 * u], a[..., :u], a[..., 2 * u:]], axis=-1)


class _Gst:
    """ReferenceEncoderGST (layers.py:526-585): the layers of its training pass."""

    def __init__(self, store, p, cfg):
        n = len(cfg['ref_encoder_filters'])
        self.stride = cfg['ref_encoder_strides']
        if cfg['ref_encoder_kernel_size'] != 3:
            raise NotImplementedError('reference encoder kernels other than 3x3')
        self.convs = [_Layer(store, f'{p}.conv{i}') for i in range(n)]      # (the first one's data gradient: style loss only)
        self.norms = [_Norm(store, f'{p}.norm{i}', True) for i in range(n)]
        self.u = cfg['ref_encoder_gru_cell_units']
        self.sd, self.H = cfg['gst_style_embed_dim'], cfg['gst_multi_num_heads']
        self.gru_ih = _bias_layer(store, p + '.gru.kernel', p + '.gru.bias_i')
        self.gru_hh = _bias_layer(store, p + '.gru.recurrent_kernel', p + '.gru.bias_h')
        self.rnn_proj = _Layer(store, p + '.rnn_proj', act='tanh')
        self.wq = _Layer(store, p + '.mha.wq')
        self.wkv = _Layer(store, p + '.mha.wkv')
        self.wo = _Layer(store, p + '.mha.dense')
        self.tokens, self.dtokens = store.p[p + '.gst_tokens'], store.g[p + '.gst_tokens']

    def linears(self):
        return [c.lin for c in self.convs] + [self.gru_ih.lin, self.gru_hh.lin, self.rnn_proj.lin, self.wq.lin, self.wkv.lin,
                                              self.wo.lin]


def _bias_layer(store, kernel_name, bias_name):
    """A dense layer whose kernel and bias have names of their own (the two halves of the Keras GRU bias)."""
    class _L:
        pass
    L = _L()
    L.lin = TrainLinear(store.p[kernel_name], store.p[bias_name])
    L.dk, L.db = store.g[kernel_name], store.g[bias_name]
    L.taps, L.lpad, L.N, L.K = 1, 0, L.lin.N, L.lin.K
    return L


def _register_autoregressive_params(store, w, cfg, shapes):
    """Adds every AutoregressiveTransformer variable to the store in forward order (names of weights.py; wq / wk / wv of a
    self-attention fused, wk / wv of a cross-attention fused, the stop head padded to 4 outputs).  `shapes` receives the
    original shape of every variable whose stored layout differs."""
    def add(name, a, trainable=True):
        store.add(name, a, trainable)

    def dense(p):
        add(p + '.kernel', w[p + '.kernel'])
        add(p + '.bias', w[p + '.bias'])

    def conv(p):
        k = np.asarray(w[p + '.kernel'])
        shapes[p + '.kernel'] = k.shape
        add(p + '.kernel', k.reshape(-1, k.shape[-1]))
        add(p + '.bias', w[p + '.bias'])

    def norm(p, batch):
        add(p + '.gamma', w[p + '.gamma'])
        add(p + '.beta', w[p + '.beta'])
        if batch:
            add(p + '.moving_mean', w[p + '.moving_mean'], False)
            add(p + '.moving_variance', w[p + '.moving_variance'], False)

    def sarn(p):
        m = p + '.mha'
        add(m + '.wqkv.kernel', np.concatenate([w[m + '.wq.kernel'], w[m + '.wk.kernel'], w[m + '.wv.kernel']], axis=1))
        add(m + '.wqkv.bias', np.concatenate([w[m + '.wq.bias'], w[m + '.wk.bias'], w[m + '.wv.bias']]))
        dense(m + '.dense')
        norm(p + '.ln', False)
        norm(p + '.last_ln', False)

    def carn(p):
        m = p + '.mha'
        dense(m + '.wq')
        add(m + '.wkv.kernel', np.concatenate([w[m + '.wk.kernel'], w[m + '.wv.kernel']], axis=1))
        add(m + '.wkv.bias', np.concatenate([w[m + '.wk.bias'], w[m + '.wv.bias']]))
        dense(m + '.dense')
        norm(p + '.layernorm', False)

    def cnn(p, n_layers):
        for i in range(n_layers - 1):
            conv(f'{p}.conv{i}')
            norm(f'{p}.norm{i}', True)
        conv(p + '.last_conv')
        norm(f'{p}.norm{n_layers - 1}', True)
        norm(f'{p}.norm{n_layers}', True)

    def tail(b, dense_block):
        if dense_block:
            dense(b + '.ffn.d1')
            dense(b + '.ffn.d2')
            norm(b + '.ffn.ln', False)
            norm(b + '.ffn.last_ln', False)
        else:
            cnn(b + '.conv', 2)

    add('text_encoder_prenet.embeddings', w['text_encoder_prenet.embeddings'])
    for i in range(len(cfg['encoder_num_heads'])):
        b = f'text_encoder.block{i}'
        sarn(b + '.sarn')
        tail(b, i < cfg['encoder_dense_blocks'])
    if cfg['system_type'] in ('speaker_style_text', 'style_text'):
        p = 'style_encoder'
        for i in range(len(cfg['ref_encoder_filters'])):
            k = np.asarray(w[f'{p}.conv{i}.kernel'])                    # (3, 3, cin, cout)
            shapes[f'{p}.conv{i}.kernel'] = k.shape
            rows = k.shape[0] * k.shape[1] * k.shape[2]
            k2 = np.zeros(((rows + 3) // 4 * 4, k.shape[3]), np.float32)    # rows padded to a multiple of 4 (cin = 1 gives 9)
            k2[:rows] = k.reshape(rows, k.shape[3])
            add(f'{p}.conv{i}.kernel', k2)
            add(f'{p}.conv{i}.bias', w[f'{p}.conv{i}.bias'])
            norm(f'{p}.norm{i}', True)
        # Keras gate order [z | r | h] -> the kernels' [r | z | n] (swapping the first two blocks; the swap undoes itself)
        add(p + '.gru.kernel', gru_gate_swap(np.asarray(w[p + '.gru.kernel'])))
        add(p + '.gru.recurrent_kernel', gru_gate_swap(np.asarray(w[p + '.gru.recurrent_kernel'])))
        bias = gru_gate_swap(np.asarray(w[p + '.gru.bias']))
        add(p + '.gru.bias_i', bias[0])
        add(p + '.gru.bias_h', bias[1])
        dense(p + '.rnn_proj')
        m = p + '.mha'
        dense(m + '.wq')
        add(m + '.wkv.kernel', np.concatenate([w[m + '.wk.kernel'], w[m + '.wv.kernel']], axis=1))
        add(m + '.wkv.bias', np.concatenate([w[m + '.wk.bias'], w[m + '.wv.bias']]))
        dense(m + '.dense')
        add(p + '.gst_tokens', w[p + '.gst_tokens'])
    dense('decoder_prenet.d1')
    dense('decoder_prenet.d2')
    for i in range(len(cfg['decoder_num_heads'])):
        b = f'decoder.block{i}'
        sarn(b + '.sarn')
        carn(b + '.carn')
        tail(b, i < cfg['decoder_dense_blocks'])
    dense('final_proj_mel')
    ks = np.zeros((cfg['mel_channels'], 4), np.float32)
    ks[:, :3] = np.asarray(w['decoder_postnet.stop_linear.kernel'])
    bs = np.zeros(4, np.float32)
    bs[:3] = np.asarray(w['decoder_postnet.stop_linear.bias'])
    add('decoder_postnet.stop_linear.kernel', ks)     # a fourth, all-zero output: the data gradient needs N % 4 == 0
    add('decoder_postnet.stop_linear.bias', bs)
    cnn('decoder_postnet.conv_blocks', cfg['postnet_conv_layers'])


class AutoregressiveTransformerTrainer(ForwardTransformerTrainer):
    """The reference's AutoregressiveTransformer as `train_models_step` uses it: `input_reshape`, `train_step` (forward,
    losses, gradients, optimizer), `set_constants`, `step`, `weights()`."""

    fused_gru = True     # the reference encoder's GRU steps as one launch each (estts_gru_fused_fwd / _bwd); False: two

    def __init__(self, weights, config, device='cuda', precision='bf16x3', seed=0, process_group=None, bucket_mb=25):
        self._init_common(config, device, precision, seed, process_group, bucket_mb)
        cfg, dev = self.config, self.device
        self.system_type = cfg['system_type']
        self.max_r = int(cfg['max_r'])
        self.r = self.max_r
        self.loss_weights = [1., 1., 1.]
        self.stop_scaling = float(cfg.get('stop_loss_scaling', 1.0))
        self._pos_scalars = {k: float(np.asarray(weights[k + '.pos_encoding_scalar'])) for k in ('text_encoder', 'decoder')}
        self._orig_shapes = {}
        st = self.store = ParamStore(dev)
        _register_autoregressive_params(st, weights, cfg, self._orig_shapes)
        st.build()
        de, dd = cfg['encoder_model_dimension'], cfg['decoder_model_dimension']
        self.de, self.dd = de, dd
        self.enc_dim = encoder_output_dim(cfg)
        self.emb, self.demb = st.p['text_encoder_prenet.embeddings'], st.g['text_encoder_prenet.embeddings']
        if self.emb.shape[1] != de:
            raise ValueError('encoder_prenet_dimension must equal encoder_model_dimension')
        self.pe_enc = ops.positional_encoding_table(cfg['encoder_max_position_encoding'], de, dev) * self._pos_scalars['text_encoder']
        self.pe_dec = ops.positional_encoding_table(cfg['decoder_max_position_encoding'], dd, dev) * self._pos_scalars['decoder']
        self.enc_blocks = self._make_stack('text_encoder', de, cfg['encoder_num_heads'], cfg['encoder_dense_blocks'],
                                           cfg['encoder_attention_conv_kernel'])
        self.enc_names = self._attn_names('TextEncoder', cfg['encoder_num_heads'], cfg['encoder_dense_blocks'])
        self.dec_blocks = []
        k = cfg['decoder_attention_conv_kernel']
        for i, h in enumerate(cfg['decoder_num_heads']):
            b = f'decoder.block{i}'
            tail = (_Ffn(st, b + '.ffn') if i < cfg['decoder_dense_blocks'] else
                    _CnnResNorm(st, b + '.conv', 2, k, 'causal', 'relu', 'relu', True))
            self.dec_blocks.append((_Sarn(st, b + '.sarn', dd, h), _Carn(st, b + '.carn', dd, h), tail))
        nd = cfg['decoder_dense_blocks']
        self.dec_names = [f'Decoder_DenseBlock{i + 1}_CrossAttention' if i < nd else f'Decoder_ConvBlock{i - nd + 1}_CrossAttention'
                          for i in range(len(cfg['decoder_num_heads']))]
        self.gst = _Gst(st, 'style_encoder', cfg) if self.system_type in ('speaker_style_text', 'style_text') else None
        if self.gst is not None and (3 * self.gst.u > ops.GRU_FUSED_MAX_H or self.gst.u % 16):
            self.fused_gru = False       # the one-launch backward step holds K = 3 * units <= 1536; larger cells: two launches
        self.prenet1 = _Layer(st, 'decoder_prenet.d1', act='relu')
        self.prenet2 = _Layer(st, 'decoder_prenet.d2', act='relu')
        self.stop = _Layer(st, 'decoder_postnet.stop_linear')
        self.postnet = _CnnResNorm(st, 'decoder_postnet.conv_blocks', cfg['postnet_conv_layers'], cfg['postnet_kernel_size'],
                                   'causal', 'tanh', 'linear', True)
        self.dec_scale = torch.full((dd,), math.sqrt(np.float32(dd)), device=dev)
        self.dec_shift = torch.zeros(dd, device=dev)
        self._final = {}
        self._base_linears = []
        for blk in self.enc_blocks:
            self._base_linears += blk[0].linears() + blk[1].linears()
        for S, C, tail in self.dec_blocks:
            self._base_linears += S.linears() + C.linears() + tail.linears()
        self._base_linears += [self.prenet1.lin, self.prenet2.lin, self.stop.lin] + self.postnet.linears()
        if self.gst is not None:
            self._base_linears += self.gst.linears()
        self._linears = list(self._base_linears)

    # ------------------------------------------------------------------ reference surface
    def _compile(self, stop_scaling=None, optimizer=None):
        """AutoregressiveTransformer.py:224-231."""
        if stop_scaling is not None:
            self.stop_scaling = float(stop_scaling)
        super()._compile(optimizer)
        self.loss_weights = [1., 1., 1.]

    def set_constants(self, decoder_prenet_dropout=None, learning_rate=None, reduction_factor=None, drop_n_heads=None):
        """AutoregressiveTransformer.py:298-307."""
        super().set_constants(decoder_prenet_dropout, learning_rate, drop_n_heads)
        if reduction_factor is not None:
            r = int(reduction_factor)
            if not 1 <= r <= self.max_r:
                raise ValueError(f'reduction_factor must be in [1, {self.max_r}]')
            self.r = r

    def input_reshape(self, tar, stop_prob):
        """AutoregressiveTransformer.py:318-327."""
        tar_inp, tar_real = tar[:, :-1], tar[:, 1:]
        return tar_real, tar_inp[:, 0::self.r, :], stop_prob[:, 1:], int(tar_inp.shape[1])

    def _final_layer(self, r):
        """FinalProj restricted to its first r * mel_channels outputs (AutoregressiveTransformer.py:173): a column view of
        the full kernel, so the unused columns keep a zero gradient."""
        L = self._final.get(r)
        if L is None:
            n = r * self.mel_channels
            st = self.store

            class _View:
                pass
            L = _View()
            L.lin = TrainLinear(st.p['final_proj_mel.kernel'][:, :n], st.p['final_proj_mel.bias'][:n])
            L.dk, L.db = st.g['final_proj_mel.kernel'][:, :n], st.g['final_proj_mel.bias'][:n]
            L.taps, L.lpad, L.N, L.K = 1, 0, n, self.dd
            self._final[r] = L
            self._linears = self._base_linears + [v.lin for v in self._final.values()]
            self._pack_plan = None
        return L

    def weights(self):
        """Current variables as a dict name -> ndarray in the layouts of weights.py."""
        out = self._named(self.store.numpy())
        for k, s in self._pos_scalars.items():
            out[k + '.pos_encoding_scalar'] = np.array(s, np.float32)
        return out

    def gradients(self):
        """The gradients of the last step under the same names and layouts as weights() (tests, logging)."""
        out = self._named(self.store.numpy(self.store.grad))
        return {k: v for k, v in out.items() if not k.endswith(('moving_mean', 'moving_variance'))}

    def _named(self, raw):
        out = {}
        for k, v in raw.items():
            if k.endswith('.mha.wqkv.kernel') or k.endswith('.mha.wkv.kernel'):
                names = ('wq', 'wk', 'wv') if 'wqkv' in k else ('wk', 'wv')
                d = v.shape[1] // len(names)
                for j, n in enumerate(names):
                    out[k.replace('wqkv' if 'wqkv' in k else 'wkv', n)] = v[:, j * d:(j + 1) * d].copy()
            elif k.endswith('.mha.wqkv.bias') or k.endswith('.mha.wkv.bias'):
                names = ('wq', 'wk', 'wv') if 'wqkv' in k else ('wk', 'wv')
                d = v.shape[0] // len(names)
                for j, n in enumerate(names):
                    out[k.replace('wqkv' if 'wqkv' in k else 'wkv', n)] = v[j * d:(j + 1) * d].copy()
            elif k.startswith('style_encoder.conv') and k.endswith('.kernel'):
                sh = self._orig_shapes[k]
                out[k] = v[:sh[0] * sh[1] * sh[2]].reshape(sh)
            elif k in ('style_encoder.gru.kernel', 'style_encoder.gru.recurrent_kernel'):
                out[k] = gru_gate_swap(v)
            elif k == 'style_encoder.gru.bias_i':
                out['style_encoder.gru.bias'] = gru_gate_swap(np.stack([v, raw['style_encoder.gru.bias_h']]))
            elif k == 'style_encoder.gru.bias_h':
                pass
            elif k == 'decoder_postnet.stop_linear.kernel':
                out[k] = v[:, :3].copy()
            elif k == 'decoder_postnet.stop_linear.bias':
                out[k] = v[:3].copy()
            elif k in self._orig_shapes:
                out[k] = v.reshape(self._orig_shapes[k])
            else:
                out[k] = v
        return out

    # ------------------------------------------------------------------ layer passes
    def _carn_fwd(self, Cx, cat, enc, d_enc, B, Lq, Lk, enc_mask, out, tape):
        """cat [Mq, 2d]: left half holds q_in, the right half receives the attention context; enc [Mk, enc_dim]."""
        Mq, Mk, d, H = B * Lq, B * Lk, Cx.d, Cx.H
        q = self._empty(Mq, d)
        self._lin_fwd(cat[:, :d], Cx.wq, q, Mq)
        kv = self._empty(Mk, 2 * d)
        self._lin_fwd(enc, Cx.wkv, kv, Mk)
        attn = self._empty(B, H, Lq, Lk)
        ctx = cat[:, d:]
        ops.attention(q, kv[:, :d], kv[:, d:], ctx, B, H, Lq, Lk, d // H, key_mask=enc_mask, attn=attn, precision=self.precision)
        hscale = hinv = None
        if self.drop_n_heads > 0 and H > 1:
            keep = np.ones((B, H), np.float32)
            for b in range(B):
                keep[b, self._host_rng.permutation(H)[:self.drop_n_heads]] = 0.0
            sc = H / (H - self.drop_n_heads)
            hscale = torch.from_numpy(keep * sc).to(self.device).view(-1)
            hinv = torch.from_numpy(keep / sc).to(self.device).view(-1)
            T.head_scale(ctx, hscale, B, Lq, H, d // H)
        a = self._empty(Mq, d)
        self._lin_fwd(cat, Cx.wo, a, Mq)
        drop = (self.dropout_rate, self._seed_now, self._next_site())
        fa = ops.rownorm_args(a, out, residual=cat[:, :d], ln_b=Cx.ln.pair, dropout=drop)
        ops.check(ops._lib.load().estts_rownorm(fa, ops._stream()), 'estts_rownorm')

        def backward(dout):
            da, dx = self._empty(Mq, d), self._empty(Mq, d)
            T.rownorm_bwd(fa, dout=dout, dy=da, dx=dx, dln_b=Cx.ln.dpair)
            T.wgrad(cat, da, Cx.wo.dk, M=Mq, precision=self.backward_precision)
            T.colreduce(1, Cx.wo.db, g=da, M=Mq, C=d)
            dctx = self._empty(Mq, d)
            ops.gemm(da, Cx.wo_c.bwd, dctx, M=Mq, precision=self.backward_precision)
            ops.gemm(da, Cx.wo_x.bwd, dx, M=Mq, precision=self.backward_precision, residual=dx)
            delta = self._empty(B * H * Lq)
            dq, dkv = self._empty(Mq, d), self._empty(Mk, 2 * d)
            ctx_b = ctx
            if hscale is not None:          # see _attention_bwd_headdrop
                ctx_b = ctx.clone()
                T.head_scale(ctx_b, hinv, B, Lq, H, d // H)
                T.head_scale(dctx, hscale, B, Lq, H, d // H)
            T.attention_bwd(q, kv[:, :d], kv[:, d:], ctx_b, dctx, attn, delta, dq, dkv[:, :d], dkv[:, d:], B, H, Lq, Lk, d // H,
                            precision=self.backward_precision)
            self._lin_bwd(cat[:, :d], Cx.wq, dq, Mq, dx=dx, accumulate=True)
            self._lin_bwd(enc, Cx.wkv, dkv, Mk, dx=d_enc, accumulate=True)
            return dx

        tape.append(backward)
        return attn

    def _gst_fwd(self, x, B, Tq, tape, input_grad=False):
        """ReferenceEncoderGST.call(training=True) (layers.py:561-585) on x [B, Tq, mel] -> (style embedding [B, sd],
        attention [B, heads, 1, tokens]); appends ONE closure to `tape` that takes the gradient of the embedding (and returns
        the gradient of x when input_grad, else None).
        Every Conv2D runs as patches (estts_im2col) times a dense kernel, which its two gradients need anyway."""
        G, s = self.gst, self.gst.stride
        y, H, W, Cin = x.contiguous(), Tq, self.mel_channels, 1
        saved = []
        for conv, norm in zip(G.convs, G.norms):
            Ho, pt = same_pad(H, 3, s)
            Wo, pl = same_pad(W, 3, s)
            M = B * Ho * Wo
            patches = self._empty(M, conv.K)
            T.im2col(y, patches, B, H, W, Cin, Ho, Wo, 3, s, pt, pl)
            z = self._empty(M, conv.N)
            kw, summed = self._stat_request(z, norm, conv.N)
            self._lin_fwd(patches, conv, z, M, **kw)
            st = self._bn_stats(z, norm, M, conv.N, summed)
            nxt = self._empty(M, conv.N)
            T.affine_act(z, nxt, st[2], st[3], 'relu', M=M, C=conv.N)
            if self.gate_log is not None:
                self.gate_log.append(nxt > 0)
            saved.append((patches, z, st, (H, W, Cin, Ho, Wo, pt, pl)))
            y, H, W, Cin = nxt, Ho, Wo, conv.N
        steps, feat, u, sd, Hh = H, W * Cin, G.u, G.sd, G.H
        batch = self._bn_batch
        Ms = B * steps
        seq = y.view(Ms, feat)                                   # rows (b, t); features (w, c) as Keras' reshape gives them
        gi = self._empty(Ms, 3 * u)
        self._lin_fwd(seq, G.gru_ih, gi, Ms)
        gi3 = gi.view(B, steps, 3 * u)
        hs = self._empty(steps + 1, B, u)
        hs[0].zero_()
        ghs = self._empty(steps, B, 3 * u)
        for t in range(steps):
            if self.fused_gru:       # recurrent product + gates in one launch
                T.gru_fused_fwd(gi3[:, t], ghs[t], hs[t], G.gru_hh.lin.fwd, hs[t + 1], precision=self.precision)
                continue
            self._lin_fwd(hs[t], G.gru_hh, ghs[t], B)
            T.gru_step_fwd(gi3[:, t], ghs[t], hs[t], hs[t + 1])
        cat = self._empty(B, u + sd)
        proj, ctx = cat[:, :u], cat[:, u:]
        self._lin_fwd(hs[steps], G.rnn_proj, proj, B)
        q = self._empty(B, sd)
        self._lin_fwd(proj, G.wq, q, B)
        n_tok, tokdim = G.tokens.shape
        tok_t = self._empty(n_tok, tokdim)
        T.affine_act(G.tokens, tok_t, None, None, 'tanh', M=n_tok, C=tokdim)
        kv = self._empty(n_tok, 2 * sd)
        self._lin_fwd(tok_t, G.wkv, kv, n_tok)
        kvb = kv[None].expand(B, n_tok, 2 * sd).contiguous().view(B * n_tok, 2 * sd)     # every sample attends to the same tokens
        attn = self._empty(B, Hh, 1, n_tok)
        ops.attention(q, kvb[:, :sd], kvb[:, sd:], ctx, B, Hh, 1, n_tok, sd // Hh, attn=attn, precision=self.precision)
        hscale = hinv = None
        if self.drop_n_heads > 0 and Hh > 1:
            keep = np.ones((B, Hh), np.float32)
            for b in range(B):
                keep[b, self._host_rng.permutation(Hh)[:self.drop_n_heads]] = 0.0
            sc = Hh / (Hh - self.drop_n_heads)
            hscale = torch.from_numpy(keep * sc).to(self.device).view(-1)
            hinv = torch.from_numpy(keep / sc).to(self.device).view(-1)
            T.head_scale(ctx, hscale, B, 1, Hh, sd // Hh)
        out = self._empty(B, sd)
        self._lin_fwd(cat, G.wo, out, B)

        def backward(d_out):
            dcat = self._empty(B, u + sd)
            self._lin_bwd(cat, G.wo, d_out, B, dx=dcat)
            dproj, dctx = dcat[:, :u], dcat[:, u:]
            ctx_b = ctx
            if hscale is not None:
                ctx_b = ctx.clone()
                T.head_scale(ctx_b, hinv, B, 1, Hh, sd // Hh)
                T.head_scale(dctx, hscale, B, 1, Hh, sd // Hh)
            delta = self._empty(B * Hh)
            dq, dkvb = self._empty(B, sd), self._empty(B * n_tok, 2 * sd)
            T.attention_bwd(q, kvb[:, :sd], kvb[:, sd:], ctx_b, dctx, attn, delta, dq, dkvb[:, :sd], dkvb[:, sd:], B, Hh, 1, n_tok,
                            sd // Hh, precision=self.backward_precision)
            dkv = torch.zeros(n_tok * 2 * sd, dtype=torch.float32, device=self.device)
            T.colreduce(1, dkv, g=dkvb.view(B, n_tok * 2 * sd), M=B, C=n_tok * 2 * sd)      # the tokens are shared: sum over the batch
            dtok_t = self._empty(n_tok, tokdim)
            self._lin_bwd(tok_t, G.wkv, dkv.view(n_tok, 2 * sd), n_tok, dx=dtok_t)
            # (this encoder may run twice in a step, see use_style_loss: whatever is written rather than accumulated goes
            # through a buffer of this call and is then added to the gradient)
            unused = torch.zeros(tokdim, dtype=torch.float32, device=self.device)
            dtok = self._empty(n_tok, tokdim)
            T.colreduce(1, unused, g=dtok_t, a=tok_t, act='tanh', dy_out=dtok, M=n_tok, C=tokdim)
            T.axpby(dtok, G.dtokens, alpha=1.0, beta=1.0, M=n_tok, D=tokdim)
            self._lin_bwd(proj, G.wq, dq, B, dx=dproj, accumulate=True)
            dpre = self._empty(B, u)
            T.colreduce(1, G.rnn_proj.db, g=dproj, a=proj, act='tanh', dy_out=dpre, M=B, C=u)
            T.wgrad(hs[steps], dpre, G.rnn_proj.dk, M=B, precision=self.backward_precision)
            dh = self._empty(B, u)
            ops.gemm(dpre, G.rnn_proj.lin.bwd, dh, M=B, precision=self.backward_precision)
            # GRU through time: only the last state feeds the output
            dgi = self._empty(B, steps, 3 * u)
            dgh = self._empty(steps, B, 3 * u)
            zero = torch.zeros(B, u, dtype=torch.float32, device=self.device)
            rec = None
            if self.fused_gru:
                # the last step's gate gradients from the plain kernel; then one launch per step: the product that carries the
                # state gradient from step t to step t - 1, with the gate gradients of step t - 1 as its epilogue
                dhd = self._empty(B, u)
                T.gru_step_bwd(gi3[:, steps - 1], ghs[steps - 1], hs[steps - 1], dh, None, dgi[:, steps - 1], dgh[steps - 1], dhd)
                for t in range(steps - 1, 0, -1):
                    T.gru_fused_bwd(gi3[:, t - 1], ghs[t - 1], hs[t - 1], zero, dgh[t], G.gru_hh.lin.bwd, dgi[:, t - 1], dgh[t - 1],
                                    dhd, precision=self.backward_precision)
            for t in reversed(range(steps if not self.fused_gru else 0)):
                dprev = self._empty(B, u)
                T.gru_step_bwd(gi3[:, t], ghs[t], hs[t], dh if t == steps - 1 else zero, rec, dgi[:, t], dgh[t], dprev)
                ops.gemm(dgh[t], G.gru_hh.lin.bwd, dprev, M=B, precision=self.backward_precision, residual=dprev)
                rec = dprev
            T.wgrad(hs[:steps].view(steps * B, u), dgh.view(steps * B, 3 * u), G.gru_hh.dk, M=steps * B, precision=self.backward_precision,
                    db=G.gru_hh.db)
            dseq = self._empty(Ms, feat)
            self._lin_bwd(seq, G.gru_ih, dgi.view(Ms, 3 * u), Ms, dx=dseq)
            g = dseq                                             # NHWC gradient of the last convolution's output
            for i in reversed(range(len(G.convs))):
                conv, norm = G.convs[i], G.norms[i]
                patches, z, st, (Hi, Wi, Ci, Ho, Wo, pt, pl) = saved[i]
                M = B * Ho * Wo
                dz = self._empty(M, conv.N)
                if batch and self.fold_sums and conv.N <= T.SumSlots.MAX_C:
                    # this call's own sum(dy) and sum(dy * xhat) go to bn_bwd in their copies; it adds them to the gradients
                    sl = T.colreduce(2, z=z, g=g.view(M, conv.N), act='relu', mean=st[0], rstd=st[1], M=M, C=conv.N,
                                     scale=st[2], shift=st[3], defer=True)
                    self._bn_input_grad(batch, g.view(M, conv.N), z, st, norm, (norm.dbeta, norm.dgamma), dz, M, conv.N, slots=sl,
                                        act='relu')
                else:
                    sums = self.store.zeros(2, conv.N)       # this call's own sums: bn_bwd reads them
                    T.colreduce(2, sums[0], sums[1], z=z, g=g.view(M, conv.N), act='relu', mean=st[0], rstd=st[1], dy_out=dz,
                                M=M, C=conv.N, scale=st[2], shift=st[3])
                    self._bn_input_grad(batch, dz, z, st, norm, sums, dz, M, conv.N)
                    T.axpby(sums[0].view(1, -1), norm.dbeta.view(1, -1), alpha=1.0, beta=1.0, M=1, D=conv.N)
                    T.axpby(sums[1].view(1, -1), norm.dgamma.view(1, -1), alpha=1.0, beta=1.0, M=1, D=conv.N)
                if i == 0 and not input_grad:
                    self._lin_bwd(patches, conv, dz, M)
                else:
                    dp = self._empty(M, conv.K)
                    self._lin_bwd(patches, conv, dz, M, dx=dp)
                    g = self._empty(B * Hi * Wi, Ci)
                    T.col2im(dp, g, B, Hi, Wi, Ci, Ho, Wo, 3, s, pt, pl)
            return g if input_grad else None

        tape.append(backward)
        return out, attn

    # ------------------------------------------------------------------ the training step
    @contextlib.contextmanager
    def _submode(self, training):
        """Keras' `training=` of a sub-network (traning_steps.py:24-29): False switches its dropout and head drop off and its
        BatchNormalization layers to their moving statistics; the gradients flow either way."""
        saved = (self.dropout_rate, self.drop_n_heads, self._bn_batch)
        if not training:
            self.dropout_rate, self.drop_n_heads, self._bn_batch = 0.0, 0, False
        try:
            yield
        finally:
            self.dropout_rate, self.drop_n_heads, self._bn_batch = saved

    def _encode(self, tokens, B, N, tape, enc_attn):
        """Text encoder in training mode -> (text_out [Me, de] view, key_mask)."""
        de, Me = self.de, B * N
        key_mask = self._empty(B, N)
        cat0 = self._empty(Me, 2 * de)
        x0 = cat0[:, :de]
        ops.embed_posenc(tokens, self.emb, self.pe_enc, None, x0, key_mask, math.sqrt(np.float32(de)))
        site0, rate0 = self._next_site(), self.dropout_rate
        T.dropout(x0, x0, rate0, self._seed_now, site0, M=Me, D=de)

        def embed_backward(dx):
            T.dropout(dx, dx, rate0, self._seed_now, site0, M=Me, D=de)
            T.embed_bwd(tokens, dx, self.demb, math.sqrt(np.float32(de)))
            return None

        tape.append(embed_backward)
        return self._stack_fwd(self.enc_blocks, cat0, B, N, key_mask, de, tape, enc_attn, self.enc_names), key_mask

    def train_step(self, tts_mel, tts_phonemes, tts_stop, tts_speaker=None, train_text_encoder=True, train_style_encoder=True,
                   train_decoder=True, apply=True, use_style_loss=False, global_frames=None):
        """The TTS half of train_models_step (traning_steps.py:22-56).  use_style_loss: the style encoder runs a second time
        over the model's own post-net output and the mean squared distance of the two style embeddings joins the loss
        (traning_steps.py:41-49).
        tts_mel [B, T + 1, C] (start / end vectors included), tts_phonemes [B, N] int, tts_stop [B, T + 1] int,
        tts_speaker [B, 1, speaker_embed_dim] or None.  Returns the reference's model_out entries that exist after the TTS
        half: final_output, mel_linear, stop_prob, decoder_attention, text_encoder_attention, text_enc_output, tts_loss,
        loss, losses, reduced_target.
        global_frames (data-parallel runs): the padded number of target frames (T) of the GLOBAL batch, i.e. the largest over
        all ranks.  The three losses divide by the padded element count, so a rank whose shard pads to fewer frames would
        weigh its utterances more; with global_frames every rank divides by the global batch's count and the summed
        gradient is that of the loss over the concatenated batch.  The returned losses are scaled the same way (their mean
        over the ranks is the global loss)."""
        dev, C, r = self.device, self.mel_channels, self.r
        tokens = torch.as_tensor(tts_phonemes, device=dev).to(torch.int32).contiguous()
        B, N = tokens.shape
        tar = torch.as_tensor(tts_mel, dtype=torch.float32, device=dev)
        stop_t = torch.as_tensor(tts_stop, device=dev).to(torch.int32)
        tar_real, tar_mel, tar_stop, mel_len = self.input_reshape(tar, stop_t)
        tar_mel = tar_mel.contiguous()
        Tq = tar_mel.shape[1]
        Tt = Tq * r
        if N > self.pe_enc.shape[0] or Tt > self.pe_dec.shape[0] + r - 1:
            raise ValueError('sequence exceeds the positional-encoding table')
        self._infer = None
        self._site = 0
        T.SumSlots.step_begin(self.device, self)      # (clears sums a failed step may have left without a consumer)
        self._seed_now = (self.seed * 1000003 + self.iterations * 7919 + 12345 + self.rank * 0x9E3779B1) & 0xFFFFFFFF
        self._host_rng = np.random.default_rng(self._seed_now)
        st, de, dd = self.store, self.de, self.dd
        Me, Md, Mt = B * N, B * Tq, B * Tt
        enc_tape, enc_attn, dec_attn = [], {}, {}
        st.clear_gradients()

        # ---- encoder output [text | speaker]
        with self._submode(train_text_encoder):
            text_out, _ = self._encode(tokens, B, N, enc_tape, enc_attn)
        enc = self._empty(Me, self.enc_dim)
        enc[:, :de].copy_(text_out)
        gst_tape, gst_out, gst_attn, sd = [], None, None, 0
        if self.gst is not None:
            with self._submode(train_style_encoder):
                gst_out, gst_attn = self._gst_fwd(tar_mel, B, Tq, gst_tape)
            sd = self.gst.sd
            enc.view(B, N, self.enc_dim)[:, :, de:de + sd].copy_(gst_out[:, None, :].expand(B, N, sd))
        if self.system_type in ('speaker_style_text', 'speaker_text'):
            spk = torch.as_tensor(tts_speaker, dtype=torch.float32, device=dev).reshape(B, 1, -1)
            if spk.shape[2] != self.enc_dim - de - sd:
                raise ValueError(f'speaker embedding has {spk.shape[2]} values, the model expects {self.enc_dim - de - sd}')
            enc.view(B, N, self.enc_dim)[:, :, de + sd:].copy_(spk.expand(B, N, spk.shape[2]))
        enc_mask = self._empty(B, N)
        ops.row_zero_mask(enc, enc_mask.view(-1))
        d_enc = torch.zeros(Me, self.enc_dim, dtype=torch.float32, device=dev)

        # ---- decoder prenet (dropout always on, layers.py:394-402), positional encoding of every r-th frame, dropout
        dec_mask = self._empty(B, Tq)
        tg2 = tar_mel.view(Md, C)
        ops.row_zero_mask(tg2, dec_mask.view(-1))
        rp = self.decoder_prenet_dropout
        inv_keep = 1.0 / (1.0 - rp)
        p1 = self._empty(Md, self.prenet1.N)
        self._lin_fwd(tg2, self.prenet1, p1, Md, dropout=(rp, self._seed_now, 1000, 0))
        p2 = self._empty(Md, dd)
        self._lin_fwd(p1, self.prenet2, p2, Md, dropout=(rp, self._seed_now, 1001, 0))
        if self.gate_log is not None:
            self.gate_log += [p1 > 0, p2 > 0]
        pe = self.pe_dec[0:Tt:r].contiguous()
        cat = self._empty(Md, 2 * dd)
        y0 = cat[:, :dd]
        T.affine_act(p2, y0, self.dec_scale, self.dec_shift, None, residual=pe, r_mod=Tq, M=Md, C=dd)
        with self._submode(train_decoder):
            site1, rate1 = self._next_site(), self.dropout_rate
        T.dropout(y0, y0, rate1, self._seed_now, site1, M=Md, D=dd)

        def prenet_backward(dy0):
            T.dropout(dy0, dy0, rate1, self._seed_now, site1, M=Md, D=dd)
            T.axpby(dy0, dy0, alpha=math.sqrt(np.float32(dd)) * inv_keep, beta=0.0, M=Md, D=dd)
            T.colreduce(1, self.prenet2.db, g=dy0, a=p2, act='relu', dy_out=dy0, M=Md, C=dd)
            T.wgrad(p1, dy0, self.prenet2.dk, M=Md, precision=self.backward_precision)
            dp1 = self._empty(Md, self.prenet1.N)
            ops.gemm(dy0, self.prenet2.lin.bwd, dp1, M=Md, precision=self.backward_precision, alpha=inv_keep)
            T.colreduce(1, self.prenet1.db, g=dp1, a=p1, act='relu', dy_out=dp1, M=Md, C=self.prenet1.N)
            T.wgrad(tg2, dp1, self.prenet1.dk, M=Md, precision=self.backward_precision)

        # ---- decoder blocks (layers.py:357-376)
        dec_tape, dec_tape_marks = [], []        # dec_tape_marks[i]: entries of dec_tape before block i
        n_blocks = len(self.dec_blocks)
        with self._submode(train_decoder):
            for i, (S, Cx, tail) in enumerate(self.dec_blocks):
                dec_tape_marks.append(len(dec_tape))
                cat2 = self._empty(Md, 2 * dd)
                self._sarn_fwd(S, cat, B, Tq, dec_mask, cat2[:, :dd], dec_tape, causal=True)
                x2 = self._empty(Md, dd)
                dec_attn[self.dec_names[i]] = self._carn_fwd(Cx, cat2, enc, d_enc, B, Tq, N, enc_mask, x2, dec_tape)
                nxt = self._empty(Md, 2 * dd) if i + 1 < n_blocks else self._empty(Md, dd)
                if isinstance(tail, _Ffn):
                    self._ffn_fwd(tail, x2, Md, nxt[:, :dd], dec_tape)
                else:
                    self._cnn_bn_fwd(tail, x2, Md, Tq, nxt[:, :dd], dec_tape)
                cat = nxt
        dec_out = cat[:, :dd]

        # ---- projections: r frames per decoder step, stop head, causal post-net
        F = self._final_layer(r)
        proj = self._empty(Md, r * C)
        self._lin_fwd(dec_out, F, proj, Md)
        mel = proj.view(Mt, C)
        stop4 = self._empty(Mt, 4)
        self._lin_fwd(mel, self.stop, stop4, Mt)
        final = self._empty(Mt, C)
        post_tape = []
        with self._submode(train_decoder):
            self._cnn_bn_fwd(self.postnet, mel, Mt, Tt, final, post_tape)
        style_tape, style_loss, d_g2 = [], None, None
        if use_style_loss:
            if self.gst is None:
                raise ValueError(f"use_style_loss needs a style system, not '{self.system_type}'")
            with self._submode(train_style_encoder):
                g2, _ = self._gst_fwd(final.view(B, Tt, C), B, Tt, style_tape, input_grad=True)
            style_loss = torch.zeros(1, dtype=torch.float32, device=dev)
            d_g2 = self._empty(B, sd)
            ops.check(ops._lib.load().estts_mse(g2.data_ptr(), gst_out.data_ptr(), B * sd, style_loss.data_ptr(), d_g2.data_ptr(), 1.0,
                                                ops._stream()), 'estts_mse')

        # ---- losses over the first mel_len frames (traning_steps.py:33-40, utils/losses.py)
        Ml = B * mel_len
        target2 = tar_real.reshape(Ml, C)

        def head(x3, width):
            return x3.view(B, Tt, -1)[:, :mel_len, :width].reshape(Ml, width)

        final_l, mel_l, stop_l = head(final, C), head(mel, C), head(stop4, 3)
        labels = tar_stop.reshape(Ml).contiguous()
        rank_share = 1.0
        if global_frames is not None:
            if int(global_frames) < mel_len:
                raise ValueError(f'global_frames {global_frames} is shorter than this rank\'s batch ({mel_len} frames)')
            rank_share = mel_len / float(global_frames)
        losses = torch.zeros(3, dtype=torch.float32, device=dev)
        ops.masked_mae(target2, final_l, losses[0:1])
        dstop_l = torch.zeros(Ml, 4, dtype=torch.float32, device=dev)
        T.scaled_xent(stop_l, labels, losses[1:2], dlogits=dstop_l[:, :3], index=2, scaling=self.stop_scaling,
                      scale=self.loss_weights[1] * rank_share / Ml)
        ops.masked_mae(target2, mel_l, losses[2:3])
        if self.gate_log is not None:
            self.gate_log += [torch.sign(final_l - target2), torch.sign(mel_l - target2)]
        lw = self.loss_weights
        # (scaled_xent was given scale = weight / M for its gradient; its loss value carries the weight too)
        loss_vals = torch.stack([losses[0] * rank_share, losses[1] / lw[1], losses[2] * rank_share])
        tts_loss = loss_vals[0] * lw[0] + loss_vals[1] * lw[1] + loss_vals[2] * lw[2]
        if style_loss is not None:
            tts_loss = tts_loss + style_loss[0]

        # ---- backward
        def spread(g_l, width):
            """[B * mel_len, width] -> [B * Tt, width] with zero rows past mel_len."""
            if mel_len == Tt:
                return g_l
            g = torch.zeros(B, Tt, width, dtype=torch.float32, device=dev)
            g[:, :mel_len].copy_(g_l.view(B, mel_len, width))
            return g.view(Mt, width)

        dfinal_l = self._empty(Ml, C)
        T.masked_mae_bwd(target2, final_l, dfinal_l, lw[0] * rank_share)
        dmel_l = self._empty(Ml, C)
        T.masked_mae_bwd(target2, mel_l, dmel_l, lw[2] * rank_share)
        dfinal = spread(dfinal_l, C)
        if style_loss is not None:             # the second style pass read the whole post-net output
            if dfinal is dfinal_l:
                dfinal = dfinal_l.clone()
            T.axpby(style_tape[0](d_g2).view(Mt, C), dfinal, alpha=1.0, beta=1.0, M=Mt, D=C)
        # the gradient buffer is in forward order (text encoder, style encoder, prenet, decoder blocks, projections, post-net):
        # every finished stage hands its tail to the all-reduce on the communication stream, as in the ForwardTransformer
        # trainer.  With the style loss the style encoder's gradients are complete only after its second backward pass
        # (gst_tape below); its offset lies under everything that is reported before that.
        overlap = self._new_overlap()
        dmel = post_tape[0](dfinal)
        T.axpby(spread(dmel_l, C), dmel, alpha=1.0, beta=1.0, M=Mt, D=C)
        self._lin_bwd(mel, self.stop, spread(dstop_l, 4), Mt, dx=dmel, accumulate=True)
        d_dec = self._empty(Md, dd)
        self._lin_bwd(dec_out, F, dmel.view(Md, r * C), Md, dx=d_dec)
        overlap.ready(self._first_offset('final_proj_mel.'))
        g = d_dec
        marks = self._decoder_tape_marks(dec_tape_marks, len(dec_tape))
        for j, bw in enumerate(reversed(dec_tape)):
            g = bw(g)
            blk = marks.get(len(dec_tape) - 1 - j)
            if blk is not None:
                overlap.ready(self._first_offset(f'decoder.block{blk}.'))
        prenet_backward(g)
        overlap.ready(self._first_offset('decoder_prenet.'))
        if self.gst is not None:           # the style embedding was tiled over the tokens: its gradient is the sum over them
            d_gst = torch.zeros(B, sd, dtype=torch.float32, device=dev)
            d_enc3 = d_enc.view(B, N, self.enc_dim)
            for b in range(B):
                T.colreduce(1, d_gst[b], g=d_enc3[b][:, de:de + sd], M=N, C=sd)
            if style_loss is not None:         # d style_loss / d (first embedding) = -d style_loss / d (second embedding)
                T.axpby(d_g2, d_gst, alpha=-1.0, beta=1.0, M=B, D=sd)
            gst_tape[0](d_gst)
            overlap.ready(self._first_offset('style_encoder.'))
        g = d_enc[:, :de].contiguous()
        for bw in reversed(enc_tape):
            g = bw(g)
        overlap.finish()
        if apply:
            self._apply_gradients()
        T.SumSlots.step_end(self.device)
        return {'final_output': final.view(B, Tt, C), 'mel_linear': mel.view(B, Tt, C), 'stop_prob': stop4.view(B, Tt, 4)[:, :, :3],
                'decoder_attention': dec_attn, 'text_encoder_attention': enc_attn, 'text_enc_output': text_out.reshape(B, N, de),
                'gst_output': None if gst_out is None else gst_out.view(B, 1, sd),
                'gst_encoder_attention': None if gst_attn is None else {'RefEncoderGST_attention': gst_attn}, 'decoder_output': dec_out.reshape(B, Tq, dd), 'linear': mel.view(B, Tt, C),
                'tts_loss': tts_loss, 'style_loss': 0 if style_loss is None else style_loss[0], 'loss': tts_loss,
                'losses': {'output': loss_vals[0], 'stop_prob': loss_vals[1], 'mel_linear': loss_vals[2]},
                'reduced_target': tar_mel}

    @staticmethod
    def _decoder_tape_marks(starts, n):
        """{index of the first tape entry of decoder block i: i}: once the backward pass has run that entry, block i's
        variables (and everything after them in the buffer) are final."""
        return {s: i for i, s in enumerate(starts) if s < n}

    def encode(self, phonemes, reduced_mel, speaker=None):
        """call_encoder in training mode without gradients (train_models_step's separate batch for the mutual-information
        heads, traning_steps.py:62-68): -> (gst_output [B, 1, sd] or None, text_enc_output [B, N, de])."""
        dev = self.device
        tokens = torch.as_tensor(phonemes, device=dev).to(torch.int32).contiguous()
        B, N = tokens.shape
        self._site = 0
        self._seed_now = (self.seed * 1000003 + self.iterations * 7919 + 54321 + self.rank * 0x9E3779B1) & 0xFFFFFFFF
        self._host_rng = np.random.default_rng(self._seed_now)
        log, self.gate_log = self.gate_log, None
        text_out, _ = self._encode(tokens, B, N, [], {})
        gst_out = None
        if self.gst is not None:
            mel = torch.as_tensor(reduced_mel, dtype=torch.float32, device=dev).contiguous()
            gst_out, _ = self._gst_fwd(mel, B, mel.shape[1], [])
            gst_out = gst_out.view(B, 1, -1)
        self.gate_log = log
        return gst_out, text_out.reshape(B, N, self.de)

    def val_step(self, inp, tar, stop_prob, spk_embed=None):
        """AutoregressiveTransformer._val_step (AutoregressiveTransformer.py:220-222): the teacher-forced pass with every
        sub-network in inference mode and the three losses, through the inference class on the current variables."""
        from .autoregressive_transformer import AutoregressiveTransformer
        if self._infer is None:
            self._infer = AutoregressiveTransformer(self.weights(), self.config, device=self.device, precision=self.precision,
                                                    fold_ffn=False)
            self._infer._compile(stop_scaling=self.stop_scaling)
        self._infer.set_constants(reduction_factor=self.r, decoder_prenet_dropout=self.decoder_prenet_dropout)
        return self._infer.val_step(inp, tar, stop_prob, spk_embed)
