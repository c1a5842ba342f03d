"""Training of the ForwardTransformer on MI355X: forward with training=True (dropout, head drop, BatchNorm batch
statistics), masked-MAE losses, hand-written backward, Adam, data-parallel gradient all-reduce.

Mirrors the training surface of the reference class (TransformerTTS/pretrained_models/for_text_encoder/models.py):
`train_step(input_sequence, target_sequence, target_durations)` (:389-404), `val_step` (:413-425), `_compile` (:406-411),
`set_constants(decoder_prenet_dropout, learning_rate, drop_n_heads)` (:458-465), `step` (:430-432); the loop that drives it
is train_forward.py:120-140.  All arithmetic runs in libestts.so kernels; torch supplies memory, views, streams and (for
data parallelism) torch.distributed over RCCL.

Parameters live in ONE flat float32 buffer (Keras layouts: Dense kernel [in, out], Conv1D kernel [k * in, out]; the three
attention projections of a layer are stored side by side as one [d, 3d] matrix), gradients and the two Adam moments in
buffers of the same shape, so that the optimizer is one launch and the gradient exchange a few large collectives.
"""
import math

import numpy as np
import torch

from . import ops, parallel, train_ops as T
from .train_ops import TrainLinear


def _ceil4(n):
    return (n + 3) // 4 * 4


class ParamStore:
    """Named views into flat parameter / gradient / Adam-moment buffers (trainable) and a flat state buffer
    (BatchNorm moving statistics)."""

    ZERO_POOL = 1 << 16      # floats

    def __init__(self, device):
        self.device = device
        self._train, self._state = [], []

    def add(self, name, array, trainable=True):
        a = np.asarray(array, dtype=np.float32)
        a = np.ascontiguousarray(a).reshape(a.shape)       # (ascontiguousarray alone turns a scalar into a vector of one)
        (self._train if trainable else self._state).append((name, a))

    def build(self):
        def layout(items):
            off, table = 0, {}
            for name, a in items:
                table[name] = (off, a.shape)
                # every tensor starts on a 256-byte boundary: the weight-gradient kernels add 64-byte row segments to these
                # views with float atomics, and a segment that straddles two cache lines costs two of them (19200 x 256 x
                # 1536, three taps: 81 us into a view at offset 16 bytes, 62 us into an aligned one)
                off = (off + a.size + 63) // 64 * 64
            return off, table

        n, self.table = layout(self._train)
        ns, self.state_table = layout(self._state)
        host = np.zeros(n, np.float32)
        for name, a in self._train:
            o = self.table[name][0]
            host[o:o + a.size] = a.reshape(-1)
        hs = np.zeros(max(ns, 4), np.float32)
        for name, a in self._state:
            o = self.state_table[name][0]
            hs[o:o + a.size] = a.reshape(-1)
        dev = self.device
        self.flat = torch.from_numpy(host).to(dev)
        # the gradient buffer and a pool of small accumulators that must start every step at zero share one allocation,
        # so that one fill per step clears both
        self.grad_and_pool = torch.zeros(n + self.ZERO_POOL, dtype=torch.float32, device=dev)
        self.grad = self.grad_and_pool[:n]
        self.pool = self.grad_and_pool[n:]
        self.pool_used = 0
        self.m = torch.zeros_like(self.flat)
        self.v = torch.zeros_like(self.flat)
        self.state = torch.from_numpy(hs).to(dev)
        self.p = {k: self._view(self.flat, o, s) for k, (o, s) in self.table.items()}
        self.g = {k: self._view(self.grad, o, s) for k, (o, s) in self.table.items()}
        self.s = {k: self._view(self.state, o, s) for k, (o, s) in self.state_table.items()}
        del self._train, self._state
        return self

    def clear_gradients(self):
        self.grad_and_pool.zero_()
        self.pool_used = 0

    def zeros(self, *shape):
        """A zeroed float32 tensor that lives until the next clear_gradients() (from the pool when it has room)."""
        n = int(np.prod(shape))
        if self.pool_used + n > self.pool.numel():
            return torch.zeros(*shape, dtype=torch.float32, device=self.device)
        o = self.pool_used
        self.pool_used += _ceil4(n)
        return self.pool[o:o + n].view(*shape)

    @staticmethod
    def _view(buf, off, shape):
        n = int(np.prod(shape)) if len(shape) else 1
        return buf[off:off + n].view(*shape) if len(shape) else buf[off:off + 1]

    def numpy(self, buf=None):
        """Every named tensor as a host array; `buf` selects another flat buffer of the same layout (e.g. self.grad)."""
        f, s = (self.flat if buf is None else buf).cpu().numpy(), self.state.cpu().numpy()
        out = {k: f[o:o + int(np.prod(sh) if len(sh) else 1)].reshape(sh).copy() for k, (o, sh) in self.table.items()}
        out.update({k: s[o:o + int(np.prod(sh) if len(sh) else 1)].reshape(sh).copy() for k, (o, sh) in self.state_table.items()})
        return out


class _Layer:
    """A dense / conv layer: packed operands plus the gradient views of its kernel and bias."""

    def __init__(self, store, p, taps=1, lpad=0, act=None, kernel_name='.kernel', need_bwd=True):
        self.lin = TrainLinear(store.p[p + kernel_name], store.p[p + '.bias'], taps=taps, lpad=lpad, act=act, need_bwd=need_bwd)
        self.dk, self.db = store.g[p + kernel_name], store.g[p + '.bias']
        self.taps, self.lpad, self.N, self.K = taps, lpad, self.lin.N, self.lin.K


class _Norm:
    def __init__(self, store, p, batch):
        self.gamma, self.beta = store.p[p + '.gamma'], store.p[p + '.beta']
        self.dgamma, self.dbeta = store.g[p + '.gamma'], store.g[p + '.beta']
        self.pair, self.dpair = (self.gamma, self.beta), (self.dgamma, self.dbeta)
        if batch:
            self.mm, self.mv = store.s[p + '.moving_mean'], store.s[p + '.moving_variance']


class _Sarn:
    def __init__(self, store, p, d, heads):
        self.d, self.H = d, heads
        self.wqkv = _Layer(store, p + '.mha.wqkv')
        self.wo = _Layer(store, p + '.mha.dense', need_bwd=False)
        k = store.p[p + '.mha.dense.kernel']
        self.wo_x = TrainLinear(k[:d], None, need_fwd=False)     # rows that multiply the block input
        self.wo_c = TrainLinear(k[d:], None, need_fwd=False)     # rows that multiply the attention context
        self.ln, self.last_ln = _Norm(store, p + '.ln', False), _Norm(store, p + '.last_ln', False)

    def linears(self):
        return [self.wqkv.lin, self.wo.lin, self.wo_x, self.wo_c]


class _Ffn:
    def __init__(self, store, p):
        self.d1, self.d2 = _Layer(store, p + '.d1'), _Layer(store, p + '.d2')
        self.ln, self.last_ln = _Norm(store, p + '.ln', False), _Norm(store, p + '.last_ln', False)

    def linears(self):
        return [self.d1.lin, self.d2.lin]


class _CnnResNorm:
    """CNNResNorm (layers.py:7-47): n_layers convolutions, n_layers + 1 normalisations."""

    def __init__(self, store, p, n_layers, k, padding, inner, last, batch):
        lpad = (k - 1) // 2 if padding == 'same' else k - 1
        names = [f'{p}.conv{i}' for i in range(n_layers - 1)] + [p + '.last_conv']
        self.convs = [_Layer(store, n, taps=k, lpad=lpad) for n in names]
        self.norms = [_Norm(store, f'{p}.norm{i}', batch) for i in range(n_layers + 1)]
        self.acts = [inner] * (n_layers - 1) + [last]
        self.batch = batch

    def linears(self):
        return [c.lin for c in self.convs]


def _register_forward_params(store, w, cfg):
    """Adds every ForwardTransformer variable to the store in forward order (names of weights.py; wq / wk / wv fused)."""
    def dense(p):
        store.add(p + '.kernel', w[p + '.kernel'])
        store.add(p + '.bias', w[p + '.bias'])

    def conv(p):
        k = np.asarray(w[p + '.kernel'])
        store.add(p + '.kernel', k.reshape(k.shape[0] * k.shape[1], k.shape[2]))
        store.add(p + '.bias', w[p + '.bias'])

    def norm(p, batch):
        store.add(p + '.gamma', w[p + '.gamma'])
        store.add(p + '.beta', w[p + '.beta'])
        if batch:
            store.add(p + '.moving_mean', w[p + '.moving_mean'], trainable=False)
            store.add(p + '.moving_variance', w[p + '.moving_variance'], trainable=False)

    def sarn(p):
        m = p + '.mha'
        store.add(m + '.wqkv.kernel', np.concatenate([w[m + '.wq.kernel'], w[m + '.wk.kernel'], w[m + '.wv.kernel']], axis=1))
        store.add(m + '.wqkv.bias', np.concatenate([w[m + '.wq.bias'], w[m + '.wk.bias'], w[m + '.wv.bias']]))
        dense(m + '.dense')
        norm(p + '.ln', False)
        norm(p + '.last_ln', False)

    def cnn(p, n_layers, batch):
        for i in range(n_layers - 1):
            conv(f'{p}.conv{i}')
            norm(f'{p}.norm{i}', batch)
        conv(p + '.last_conv')
        norm(f'{p}.norm{n_layers - 1}', batch)
        norm(f'{p}.norm{n_layers}', batch)

    def stack(p, heads, dense_blocks):
        for i in range(len(heads)):
            b = f'{p}.block{i}'
            sarn(b + '.sarn')
            if i < dense_blocks:
                dense(b + '.ffn.d1')
                dense(b + '.ffn.d2')
                norm(b + '.ffn.ln', False)
                norm(b + '.ffn.last_ln', False)
            else:
                cnn(b + '.conv', 2, True)

    store.add('encoder_prenet.embeddings', w['encoder_prenet.embeddings'])
    stack('encoder', cfg['encoder_num_heads'], cfg['encoder_dense_blocks'])
    cnn('dur_pred.conv_blocks', 2, False)
    store.add('dur_pred.linear.kernel', np.asarray(w['dur_pred.linear.kernel'])[:, 0])
    store.add('dur_pred.linear.bias', w['dur_pred.linear.bias'])
    dense('decoder_prenet.d1')
    dense('decoder_prenet.d2')
    stack('decoder', cfg['decoder_num_heads'], cfg['decoder_dense_blocks'])
    dense('out')
    cnn('decoder_postnet', cfg['postnet_conv_layers'], True)


class ForwardTransformerTrainer:
    def _init_common(self, config, device, precision, seed, process_group, bucket_mb):
        """What every trainer built on these layer passes shares: device, random streams, data-parallel group, Adam settings."""
        if not torch.cuda.is_available():
            raise RuntimeError('training needs a GPU: the hot path has no CPU implementation')
        ops._lib.load()
        cfg = self.config = dict(config)
        self.device = torch.device(device)
        self.precision = precision
        # operands of the backward products (data and weight gradients, attention backward); 'bf16' trades gradient accuracy
        # (about 3e-3 per product instead of 1e-5) for one MFMA per product instead of three
        self.backward_precision = precision
        self.seed = int(seed)
        self.pg = process_group
        self.world, self.rank = 1, 0
        if process_group is not None or (torch.distributed.is_available() and torch.distributed.is_initialized()):
            self.world = torch.distributed.get_world_size(process_group)
            self.rank = torch.distributed.get_rank(process_group)
        self.bucket_elems = int(bucket_mb * (1 << 20) // 4)
        self.mel_channels = cfg['mel_channels']
        self.dropout_rate = float(cfg.get('dropout_rate', 0.1))
        self.decoder_prenet_dropout = 0.0
        self._offsets = {}
        self._pack_plan = None
        self.drop_n_heads = 0
        self.learning_rate = 1e-4
        self.beta1, self.beta2, self.eps = 0.9, 0.98, 1e-9          # utils/config_manager.py:172-176
        self.iterations = 0
        self._infer = None
        # tests: set to a list and train_step appends, in forward order, the branch every ReLU took ([rows, channels] bool
        # tensors) and then the signs (pred - target) of the MAE losses, so that an oracle can differentiate the same
        # piecewise-linear function the device did
        self.gate_log = None
        self._bn_batch = True      # BatchNormalization on batch statistics (training=True); False: on its moving statistics
        # BatchNormalization sums (moments forward, dbeta / dgamma backward) go from the launch that builds them -- the
        # layer's own product where that can sum its output -- straight to the launch that uses them; False: every sum is a
        # reduction pass plus a launch that adds its 16 copies up (for comparisons)
        self.fold_sums = True
        self._comm = torch.cuda.Stream(device=self.device) if self.world > 1 else None
        self.exchange = True       # False: the step skips its gradient exchange (timing what the exchange costs)

    def __init__(self, weights, config, device='cuda', precision='bf16x3', seed=0, process_group=None, bucket_mb=25):
        self._init_common(config, device, precision, seed, process_group, bucket_mb)
        cfg = self.config
        self.loss_weights = [3., 1.]
        self._pos_scalars = {k: float(np.asarray(weights[k + '.pos_encoding_scalar'])) for k in ('encoder', 'decoder')}
        dev = self.device
        st = self.store = ParamStore(dev)
        _register_forward_params(st, weights, cfg)
        st.build()
        de, dd = cfg['encoder_model_dimension'], cfg['decoder_model_dimension']
        self.de, self.dd = de, dd
        self.emb, self.demb = st.p['encoder_prenet.embeddings'], st.g['encoder_prenet.embeddings']
        self.pe_enc = ops.positional_encoding_table(cfg['encoder_max_position_encoding'], de, dev) * self._pos_scalars['encoder']
        self.pe_dec = ops.positional_encoding_table(cfg['decoder_max_position_encoding'], dd, dev) * self._pos_scalars['decoder']
        self.enc_blocks = self._make_stack('encoder', de, cfg['encoder_num_heads'], cfg['encoder_dense_blocks'],
                                           cfg['encoder_attention_conv_kernel'])
        self.dec_blocks = self._make_stack('decoder', dd, cfg['decoder_num_heads'], cfg['decoder_dense_blocks'],
                                           cfg['decoder_attention_conv_kernel'])
        self.enc_names = self._attn_names('Encoder', cfg['encoder_num_heads'], cfg['encoder_dense_blocks'])
        self.dec_names = self._attn_names('Decoder', cfg['decoder_num_heads'], cfg['decoder_dense_blocks'])
        self.dur = _CnnResNorm(st, 'dur_pred.conv_blocks', 2, 3, 'same', 'relu', 'relu', False)
        self.dur_w, self.ddur_w = st.p['dur_pred.linear.kernel'], st.g['dur_pred.linear.kernel']
        self.dur_b, self.ddur_b = st.p['dur_pred.linear.bias'], st.g['dur_pred.linear.bias']
        self.prenet1 = _Layer(st, 'decoder_prenet.d1', act='relu')
        self.prenet2 = _Layer(st, 'decoder_prenet.d2', act='relu')
        self.out = _Layer(st, 'out')
        self.postnet = _CnnResNorm(st, 'decoder_postnet', cfg['postnet_conv_layers'], cfg['postnet_kernel_size'], 'same', 'tanh',
                                   'linear', True)
        self.dec_scale = torch.full((dd,), math.sqrt(np.float32(dd)), device=dev)
        self.dec_shift = torch.zeros(dd, device=dev)
        self._linears = []
        for blk in self.enc_blocks + self.dec_blocks:
            self._linears += blk[0].linears() + blk[1].linears()
        self._linears += self.dur.linears() + [self.prenet1.lin, self.prenet2.lin, self.out.lin] + self.postnet.linears()

    # ------------------------------------------------------------------ construction helpers
    def _make_stack(self, p, d, heads, dense_blocks, k):
        blocks = []
        for i, h in enumerate(heads):
            b = f'{p}.block{i}'
            sarn = _Sarn(self.store, b + '.sarn', d, h)
            tail = _Ffn(self.store, b + '.ffn') if i < dense_blocks else _CnnResNorm(self.store, b + '.conv', 2, k, 'same', 'relu',
                                                                                   'relu', True)
            blocks.append((sarn, tail))
        return blocks

    @staticmethod
    def _attn_names(name, heads, dense_blocks):
        return [f'{name}_DenseBlock{i + 1}_SelfAttention' if i < dense_blocks else f'{name}_ConvBlock{i - dense_blocks + 1}_SelfAttention'
                for i in range(len(heads))]

    def _empty(self, *shape):
        return torch.empty(*shape, dtype=torch.float32, device=self.device)

    def _new_overlap(self):
        overlap = parallel.GradientOverlap(self.store.grad, self.bucket_elems, self.pg, self._comm)
        if not self.exchange:
            overlap.active = False
        self._last_overlap = overlap          # (tests look at the collectives it issued)
        return overlap

    def _first_offset(self, prefix):
        """Lowest offset in the flat parameter / gradient buffer of the variables whose names start with `prefix`."""
        o = self._offsets.get(prefix)
        if o is None:
            o = self._offsets[prefix] = min(off for name, (off, _) in self.store.table.items() if name.startswith(prefix))
        return o

    def _next_site(self):
        self._site += 1
        return self._site - 1

    # ------------------------------------------------------------------ layer passes
    def _lin_fwd(self, x, L, out, M, seg=0, **kw):
        return ops.gemm(x, L.lin.fwd, out, M=M, seg_len=seg if L.taps > 1 else 0, precision=self.precision, **kw)

    def _lin_bwd(self, x, L, dy, M, seg=0, dx=None, accumulate=False, residual=None):
        """Kernel and bias gradients of layer L (input x, output gradient dy); optionally the input gradient into dx
        (added to dx when accumulate, or to `residual`)."""
        sl = seg if L.taps > 1 else 0
        T.wgrad(x, dy, L.dk, taps=L.taps, lpad=L.lpad, M=M, seg_len=sl, precision=self.backward_precision, db=L.db)
        if dx is not None:
            ops.gemm(dy, L.lin.bwd, dx, M=M, seg_len=sl, precision=self.backward_precision, residual=dx if accumulate else residual)
        return dx

    def _sarn_fwd(self, S, cat, B, L, key_mask, out, tape, causal=False):
        M, d, H = B * L, S.d, S.H
        qkv = self._empty(M, 3 * d)
        self._lin_fwd(cat[:, :d], S.wqkv, qkv, M)
        attn = self._empty(B, H, L, L)
        ctx = cat[:, d:]
        ops.attention(qkv[:, :d], qkv[:, d:2 * d], qkv[:, 2 * d:], ctx, B, H, L, L, d // H, key_mask=key_mask, causal=causal,
                      attn=attn, precision=self.precision)
        hscale = hinv = None
        if self.drop_n_heads > 0 and H > 1:
            keep = np.ones((B, H), np.float32)
            for b in range(B):          # HeadDrop (layers.py:74-97): drop_n_heads random heads per sample, rescale the rest
                keep[b, self._host_rng.permutation(H)[:self.drop_n_heads]] = 0.0
            sc = H / (H - self.drop_n_heads)
            hscale = torch.from_numpy(keep * sc).to(self.device).view(-1)
            hinv = torch.from_numpy(keep / sc).to(self.device).view(-1)
            T.head_scale(ctx, hscale, B, L, H, d // H)
        a = self._empty(M, d)
        self._lin_fwd(cat, S.wo, a, M)
        drop = (self.dropout_rate, self._seed_now, self._next_site())
        fa = ops.rownorm_args(a, out, ln_a=S.ln.pair, residual=cat[:, :d], ln_b=S.last_ln.pair, dropout=drop)
        ops.check(ops._lib.load().estts_rownorm(fa, ops._stream()), 'estts_rownorm')

        def backward(dout):
            da, dx = self._empty(M, d), self._empty(M, d)
            T.rownorm_bwd(fa, dout=dout, dy=da, dx=dx, dln_a=S.ln.dpair, dln_b=S.last_ln.dpair)
            T.wgrad(cat, da, S.wo.dk, M=M, precision=self.backward_precision)
            T.colreduce(1, S.wo.db, g=da, M=M, C=d)
            dctx = self._empty(M, d)
            ops.gemm(da, S.wo_c.bwd, dctx, M=M, precision=self.backward_precision)
            ops.gemm(da, S.wo_x.bwd, dx, M=M, precision=self.backward_precision, residual=dx)
            delta = self._empty(B * H * L)
            dqkv = self._empty(M, 3 * d)
            if hscale is not None:
                self._attention_bwd_headdrop(qkv, ctx, dctx, attn, delta, dqkv, B, H, L, d, hscale, hinv)
            else:
                T.attention_bwd(qkv[:, :d], qkv[:, d:2 * d], qkv[:, 2 * d:], ctx, dctx, attn, delta, dqkv[:, :d], dqkv[:, d:2 * d],
                                dqkv[:, 2 * d:], B, H, L, L, d // H, precision=self.backward_precision)
            self._lin_bwd(cat[:, :d], S.wqkv, dqkv, M, dx=dx, accumulate=True)
            return dx

        tape.append(backward)
        return attn

    def _attention_bwd_headdrop(self, qkv, ctx, dctx, attn, delta, dqkv, B, H, L, d, hscale, hinv):
        """With ctx_post = s * ctx_pre the products need d(ctx_pre) = s * dctx and delta = <d(ctx_pre), ctx_pre>:
        un-scale a copy of the saved context (hinv = 1 / s for kept heads, 0 for dropped ones) and scale the gradient."""
        ctx_pre = ctx.clone()
        T.head_scale(ctx_pre, hinv, B, L, H, d // H)
        T.head_scale(dctx, hscale, B, L, H, d // H)
        T.attention_bwd(qkv[:, :d], qkv[:, d:2 * d], qkv[:, 2 * d:], ctx_pre, dctx, attn, delta, dqkv[:, :d], dqkv[:, d:2 * d],
                        dqkv[:, 2 * d:], B, H, L, L, d // H, precision=self.backward_precision)

    def _ffn_fwd(self, Fb, x, M, out, tape):
        h1 = self._empty(M, Fb.d1.N)
        self._lin_fwd(x, Fb.d1, h1, M)
        h2 = self._empty(M, Fb.d2.N)
        self._lin_fwd(h1, Fb.d2, h2, M)
        drop = (self.dropout_rate, self._seed_now, self._next_site())
        fa = ops.rownorm_args(h2, out, ln_a=Fb.ln.pair, act='relu', residual=x, ln_b=Fb.last_ln.pair, dropout=drop)
        ops.check(ops._lib.load().estts_rownorm(fa, ops._stream()), 'estts_rownorm')
        if self.gate_log is not None:      # the same kernel up to the ReLU: the same arithmetic, hence the same branches
            r = self._empty(M, Fb.d2.N)
            ops.rownorm(h2, r, ln_a=Fb.ln.pair, act='relu')
            self.gate_log.append(r > 0)

        def backward(dout):
            dh2, dx = self._empty(M, Fb.d2.N), self._empty(M, x.shape[1])
            T.rownorm_bwd(fa, dout=dout, dy=dh2, dx=dx, dln_a=Fb.ln.dpair, dln_b=Fb.last_ln.dpair)
            dh1 = self._empty(M, Fb.d1.N)
            self._lin_bwd(h1, Fb.d2, dh2, M, dx=dh1)
            self._lin_bwd(x, Fb.d1, dh1, M, dx=dx, accumulate=True)
            return dx

        tape.append(backward)

    def _bn_train(self, z, norm, M, C, summed=None):
        """Batch statistics of z [M, C] -> (mean, rstd, scale, shift); updates the moving statistics (momentum 0.99).
        summed: the (pointer, stride) the layer's own launch already left the moments in, taken about _bn_shift(norm)
        (_stat_request); otherwise a reduction pass over z builds them.  Either way the sums stay spread over their 16 copies
        until estts_bn_finalize_slots adds them up."""
        st = self._empty(4, C)
        about_next = None
        if summed is None and not (self.fold_sums and C <= T.SumSlots.MAX_C):
            sums = self.store.zeros(2, C)
            T.colreduce(0, sums[0], sums[1], z=z, M=M, C=C)
            T.bn_finalize(sums[0], sums[1], z, M, C, norm.gamma, norm.beta, norm.mm, norm.mv, st[0], st[1], st[2], st[3])
            return st
        if summed is None:
            summed, about = T.colreduce(0, z=z, M=M, C=C, shift=z[0], defer=True), z
        else:
            about = about_next = self._bn_shift(norm, C)
        T.bn_finalize(None, None, about, M, C, norm.gamma, norm.beta, norm.mm, norm.mv, st[0], st[1], st[2], st[3], slots=summed,
                      about_next=about_next)
        return st

    def _bn_shift(self, norm, C):
        """The vector the layer's launch takes the moments of its output about: the batch mean of the layer's previous step
        (zero before the first), kept per normalisation layer.  Any shift is exact in exact arithmetic; one within a few
        deviations of the mean keeps E[(x - k)^2] - E[x - k]^2 from cancelling in float32."""
        shifts = self.__dict__.setdefault('_bn_shifts', {})
        k = shifts.get(id(norm))
        if k is None:
            k = shifts[id(norm)] = torch.zeros(C, dtype=torch.float32, device=self.device)
        return k

    def _stat_request(self, z, norm, C):
        """(gemm keyword arguments, slots) that make the layer's own launch sum the moments of its output z for `norm`, about
        the batch mean of the previous step (_bn_shift)."""
        if not (self._bn_batch and self.fold_sums) or C > T.SumSlots.MAX_C:
            return {}, None
        ptr, stride = T.SumSlots.on(z.device).forward_region(C)
        return {'stats': (ptr, self._bn_shift(norm, C))}, (ptr, stride)

    def _bn_eval(self, norm, C):
        """(mean, rstd, scale, shift) of a BatchNormalization layer in inference mode: its moving statistics, left untouched
        (estts_bn_finalize on one "row" whose value is the moving mean and whose squared deviation is the moving variance)."""
        st = self._empty(4, C)
        T.bn_finalize(self.store.zeros(C), norm.mv, norm.mm, 1, C, norm.gamma, norm.beta, None, None, st[0], st[1], st[2], st[3])
        return st

    def _bn_stats(self, z, norm, M, C, summed=None):
        return self._bn_train(z, norm, M, C, summed) if self._bn_batch else self._bn_eval(norm, C)

    def _bn_input_grad(self, batch, dy, z, st, norm, sums, dz, M, C, slots=None, act=None):
        """Gradient with respect to the normalised tensor given sum(dy) and sum(dy * xhat) in `sums` (beta, gamma order), or
        still spread over the copies at `slots` (then also added to `sums`, the parameter gradients): the batch-statistics
        form, or dy * scale when the layer ran on its moving statistics."""
        if batch:
            if act is not None:      # dy is still the gradient w.r.t. the activation output: bn_bwd takes it through (slots only)
                T.bn_bwd(dy, z, st[0], st[1], norm.gamma, sums[1], sums[0], dz, M=M, C=C, slots=slots, act=act, scale=st[2], shift=st[3])
            else:
                T.bn_bwd(dy, z, st[0], st[1], norm.gamma, sums[1], sums[0], dz, M=M, C=C, slots=slots)
        else:
            T.affine_act(dy, dz, st[2], self.store.zeros(C), None, M=M, C=C)

    def _cnn_bn_fwd(self, Cb, x, M, seg, out, tape):
        """CNNResNorm with BatchNormalization in training mode (layers.py:35-47); x [M, cin] view, out [M, cout] view."""
        n = len(Cb.convs)
        saved, h, s = [], x, None
        for i, (conv, norm, act) in enumerate(zip(Cb.convs, Cb.norms[:n], Cb.acts)):
            z = self._empty(M, conv.N)
            kw, summed = self._stat_request(z, norm, conv.N)
            self._lin_fwd(h, conv, z, M, seg, **kw)
            st = self._bn_stats(z, norm, M, conv.N, summed)
            nxt = self._empty(M, conv.N)
            T.affine_act(z, nxt, st[2], st[3], act, residual=x if i == n - 1 else None, M=M, C=conv.N)
            if self.gate_log is not None and act == 'relu':
                r = nxt
                if i == n - 1:                      # the residual is already added: the activation alone, same kernel
                    r = self._empty(M, conv.N)
                    T.affine_act(z, r, st[2], st[3], act, M=M, C=conv.N)
                self.gate_log.append(r > 0)
            saved.append((h, z, st))
            h = nxt
        s = h                                           # inputs + last activation
        last = Cb.norms[n]
        C = Cb.convs[-1].N
        st_last = self._bn_stats(s, last, M, C)
        T.affine_act(s, out, st_last[2], st_last[3], None, M=M, C=C)
        batch = self._bn_batch

        def backward(dout):
            ds = self._empty(M, C)
            defer = batch and self.fold_sums and C <= T.SumSlots.MAX_C      # the sums go straight to the launch that uses them
            sl = T.colreduce(2, last.dbeta, last.dgamma, z=s, g=dout, mean=st_last[0], rstd=st_last[1], M=M, C=C, defer=defer)
            self._bn_input_grad(batch, dout, s, st_last, last, (last.dbeta, last.dgamma), ds, M, C, slots=sl)
            g, dx = ds, None
            for i in reversed(range(n)):
                conv, norm, act = Cb.convs[i], Cb.norms[i], Cb.acts[i]
                hin, z, st = saved[i]
                dz = self._empty(M, conv.N)
                defer = batch and self.fold_sums and conv.N <= T.SumSlots.MAX_C
                if defer and act is not None:
                    # the reduction does not store g * act': bn_bwd recomputes it from the same operands (one store and one
                    # load of the tensor less)
                    sl = T.colreduce(2, z=z, g=g, act=act, mean=st[0], rstd=st[1], M=M, C=conv.N, scale=st[2], shift=st[3], defer=True)
                    self._bn_input_grad(batch, g, z, st, norm, (norm.dbeta, norm.dgamma), dz, M, conv.N, slots=sl, act=act)
                else:
                    sl = T.colreduce(2, norm.dbeta, norm.dgamma, z=z, g=g, act=act, mean=st[0], rstd=st[1], dy_out=dz, M=M, C=conv.N,
                                     scale=st[2], shift=st[3], defer=defer)
                    self._bn_input_grad(batch, dz, z, st, norm, (norm.dbeta, norm.dgamma), dz, M, conv.N, slots=sl)
                if i > 0:
                    g = self._empty(M, conv.lin.cin)
                    self._lin_bwd(hin, conv, dz, M, seg, dx=g)
                else:
                    dx = self._empty(M, conv.lin.cin)
                    self._lin_bwd(hin, conv, dz, M, seg, dx=dx, residual=ds)
            return dx

        tape.append(backward)

    def _stack_fwd(self, blocks, cat, B, L, key_mask, d, tape, attn_out, names):
        """SelfAttentionBlocks.call after the positional encoding and its dropout (layers.py:247-255).  `cat` is the
        [M, 2d] buffer whose left half holds the stack input; returns the [M, d] view holding the stack output."""
        M = B * L
        for i, (S, tail) in enumerate(blocks):
            x1 = self._empty(M, d)
            attn_out[names[i]] = self._sarn_fwd(S, cat, B, L, key_mask, x1, tape)
            nxt = self._empty(M, 2 * d) if i + 1 < len(blocks) else self._empty(M, d)
            out = nxt[:, :d]
            if isinstance(tail, _Ffn):
                self._ffn_fwd(tail, x1, M, out, tape)
            else:
                self._cnn_bn_fwd(tail, x1, M, L, out, tape)
            cat = nxt
        return cat[:, :d]

    # ------------------------------------------------------------------ reference surface
    def _compile(self, optimizer=None):
        """models.py:406-411: loss weights 3 (mel) and 1 (duration); `optimizer` may be a dict with learning_rate / beta_1 /
        beta_2 / epsilon (the reference passes a Keras Adam built by ConfigManager.new_adam)."""
        self.loss_weights = [3., 1.]
        if isinstance(optimizer, dict):
            self.learning_rate = float(optimizer.get('learning_rate', self.learning_rate))
            self.beta1 = float(optimizer.get('beta_1', self.beta1))
            self.beta2 = float(optimizer.get('beta_2', self.beta2))
            self.eps = float(optimizer.get('epsilon', self.eps))

    def set_constants(self, decoder_prenet_dropout=None, learning_rate=None, drop_n_heads=None, **kwargs):
        """models.py:458-465."""
        if decoder_prenet_dropout is not None:
            if not 0.0 <= float(decoder_prenet_dropout) < 1.0:
                raise ValueError('decoder_prenet_dropout must be in [0, 1)')
            self.decoder_prenet_dropout = float(decoder_prenet_dropout)
        if learning_rate is not None:
            self.learning_rate = float(learning_rate)
        if drop_n_heads is not None:
            self.drop_n_heads = int(drop_n_heads)

    @property
    def step(self):
        """models.py:430-432 (optimizer iterations)."""
        return self.iterations

    def optimizer_state(self):
        """Adam's state for a checkpoint: the first and second moments (flat, in the parameter buffer's order), the step count
        and the dropout seed (the reference checkpoints its optimizer: tf.train.Checkpoint(step, optimizer, net),
        train_forward.py:105-111)."""
        st = self.store
        return {'__adam_m__': st.m.cpu().numpy(), '__adam_v__': st.v.cpu().numpy(),
                '__iterations__': np.asarray(self.iterations, np.int64), '__seed__': np.asarray(self.seed, np.int64)}

    def load_optimizer_state(self, state):
        st = self.store
        m, v = np.asarray(state['__adam_m__'], np.float32), np.asarray(state['__adam_v__'], np.float32)
        if m.shape != tuple(st.m.shape) or v.shape != tuple(st.v.shape):
            raise ValueError(f'optimizer state of {m.shape[0]} elements does not fit this model ({st.m.shape[0]})')
        st.m.copy_(torch.from_numpy(m))
        st.v.copy_(torch.from_numpy(v))
        self.iterations = int(state['__iterations__'])
        self.seed = int(state['__seed__'])

    def broadcast_moving_statistics(self, src=0):
        """BatchNorm moving statistics are per replica during data-parallel training (the reference is single-GPU; there is
        no SyncBN): before a checkpoint every rank takes rank `src`'s, so that all replicas resume from what was saved."""
        if self.world > 1:
            parallel.broadcast_(self.store.state, src=src, group=self.pg)

    def weights(self):
        """Current variables as a dict name -> ndarray in the layouts of weights.py (for ForwardTransformer / checkpoints)."""
        cfg, raw, out = self.config, self.store.numpy(), {}
        for k, v in raw.items():
            if k.endswith('.mha.wqkv.kernel'):
                d = v.shape[0]
                for j, n in enumerate(('wq', 'wk', 'wv')):
                    out[k.replace('wqkv', n)] = v[:, j * d:(j + 1) * d].copy()
            elif k.endswith('.mha.wqkv.bias'):
                d = v.shape[0] // 3
                for j, n in enumerate(('wq', 'wk', 'wv')):
                    out[k.replace('wqkv', n)] = v[j * d:(j + 1) * d].copy()
            elif k == 'dur_pred.linear.kernel':
                out[k] = v.reshape(-1, 1)
            elif k.endswith('.kernel') and ('.conv' in k or 'last_conv' in k):
                kk = (cfg['postnet_kernel_size'] if k.startswith('decoder_postnet') else
                      3 if k.startswith('dur_pred') else
                      cfg['encoder_attention_conv_kernel'] if k.startswith('encoder') else cfg['decoder_attention_conv_kernel'])
                out[k] = v.reshape(kk, v.shape[0] // kk, v.shape[1])
            else:
                out[k] = v
        for k, s in self._pos_scalars.items():
            out[k + '.pos_encoding_scalar'] = np.array(s, np.float32)
        return out

    def _prepare_batch(self, input_sequence, target_sequence, target_durations):
        dev = self.device
        tokens = torch.as_tensor(input_sequence, device=dev).to(torch.int32).contiguous()
        B, N = tokens.shape
        if torch.is_tensor(target_durations):
            tdn = target_durations.detach().cpu().numpy()
        else:
            tdn = np.asarray(target_durations)
        tdn = np.rint(tdn.reshape(B, N).astype(np.float64))
        Tm = int(np.maximum(tdn, 0).sum(axis=1).max())
        mel = torch.as_tensor(target_sequence, device=dev).to(torch.float32).contiguous()
        if mel.shape[0] != B or mel.shape[2] != self.mel_channels:
            raise ValueError('target_sequence must be [B, T, mel_channels]')
        if mel.shape[1] != Tm:
            raise ValueError(f'target mel has {mel.shape[1]} frames but the target durations expand to {Tm}: pad the batch so '
                             'that the longest utterance defines both')
        if N > self.pe_enc.shape[0] or Tm > self.pe_dec.shape[0]:
            raise ValueError('sequence exceeds the positional-encoding table')
        td = torch.from_numpy(tdn.astype(np.float32)).to(dev)
        return tokens, mel, td, B, N, Tm

    def train_step(self, input_sequence, target_sequence, target_durations, global_lengths=None):
        """models.py:389-404: forward (training=True), 3 * MAE(mel) + MAE(duration), gradients, Adam update.

        global_lengths = (T, N) (data-parallel runs): the padded mel and token lengths of the GLOBAL batch, i.e. the largest
        over all ranks.  The reference's losses divide by the padded element count (Keras SUM_OVER_BATCH_SIZE over a padded
        batch), so a rank that pads its shard to a shorter length weighs its utterances more; with global_lengths every rank
        divides by the global batch's count instead and the N-rank step equals the one-GPU step on the concatenated batch
        whatever the ranks pad to.  The returned losses are scaled the same way (their mean over ranks is the global loss)."""
        tokens, mel_t, td, B, N, Tm = self._prepare_batch(input_sequence, target_sequence, target_durations)
        mel_w, dur_w = self.loss_weights
        if global_lengths is not None:
            Tg, Ng = global_lengths
            if Tg < Tm or Ng < N:
                raise ValueError(f'global_lengths {(Tg, Ng)} are shorter than this rank\'s batch {(Tm, N)}')
            mel_w, dur_w = mel_w * Tm / Tg, dur_w * N / Ng
        self._infer = None
        self._site = 0
        T.SumSlots.step_begin(self.device, self)      # (clears sums a failed step may have left without a consumer)
        # (the rank enters the seed: replicas draw different dropout masks and head drops for their shards)
        self._seed_now = (self.seed * 1000003 + self.iterations * 7919 + 12345 + self.rank * 0x9E3779B1) & 0xFFFFFFFF
        self._host_rng = np.random.default_rng(self._seed_now)
        st, de, dd, mc = self.store, self.de, self.dd, self.mel_channels
        Me, Md = B * N, B * Tm
        tape, enc_attn, dec_attn = [], {}, {}
        st.clear_gradients()

        # ---- encoder
        key_mask = self._empty(B, N)
        cat0 = self._empty(Me, 2 * de)
        x0 = cat0[:, :de]
        ops.embed_posenc(tokens, self.emb, self.pe_enc, None, x0, key_mask, math.sqrt(np.float32(de)))
        site0 = self._next_site()
        T.dropout(x0, x0, self.dropout_rate, self._seed_now, site0, M=Me, D=de)

        def embed_backward(dx):
            T.dropout(dx, dx, self.dropout_rate, self._seed_now, site0, M=Me, D=de)
            T.embed_bwd(tokens, dx, self.demb, math.sqrt(np.float32(de)))
            return None

        tape.append(embed_backward)
        enc_out = self._stack_fwd(self.enc_blocks, cat0, B, N, key_mask, de, tape, enc_attn, self.enc_names)

        # ---- duration predictor (CNNResNorm with LayerNorm + Dense(1, relu), layers.py:618-642)
        D = self.dur
        z0 = self._empty(Me, de)
        self._lin_fwd(enc_out, D.convs[0], z0, Me, N)
        a0 = self._empty(Me, de)
        fa0 = ops.rownorm_args(z0, a0, ln_a=D.norms[0].pair, act='relu')
        ops.check(ops._lib.load().estts_rownorm(fa0, ops._stream()), 'estts_rownorm')
        if self.gate_log is not None:
            self.gate_log.append(a0 > 0)
        z1 = self._empty(Me, de)
        self._lin_fwd(a0, D.convs[1], z1, Me, N)
        dur = self._empty(Me)
        fa1 = ops.rownorm_args(z1, None, ln_a=D.norms[1].pair, act='relu', residual=enc_out, ln_b=D.norms[2].pair,
                               dot_w=self.dur_w, dot_b=self.dur_b, dot_scale=1.0, dot_mask=key_mask.view(-1), dot_out=dur)
        ops.check(ops._lib.load().estts_rownorm(fa1, ops._stream()), 'estts_rownorm')
        if self.gate_log is not None:
            r = self._empty(Me, de)
            ops.rownorm(z1, r, ln_a=D.norms[1].pair, act='relu')
            self.gate_log += [r > 0, (dur > 0).view(Me, 1)]

        # ---- length regulator, decoder prenet
        cum = torch.empty(B, N, dtype=torch.int32, device=self.device)
        totals = torch.empty(B, dtype=torch.int32, device=self.device)
        ops.durations_cumsum(td, cum, totals)
        mels_in = self._empty(Md, de)
        frame_mask = self._empty(B, Tm)
        ops.expand(enc_out, cum, mels_in, frame_mask, B, N, Tm, de)
        # DecoderPrenet (layers.py:394-402): relu -> dropout after each dense layer, in the GEMM epilogue (sites 1000 / 1001
        # of the step's generator, apart from the numbered sites of the regular dropouts).  p1 / p2 hold the dropped
        # outputs, so `output > 0` is mask * relu' and the backward pass only adds the 1 / (1 - rate) factor.
        rp = self.decoder_prenet_dropout
        inv_keep = 1.0 / (1.0 - rp)
        p1 = self._empty(Md, self.prenet1.N)
        self._lin_fwd(mels_in, self.prenet1, p1, Md, dropout=(rp, self._seed_now, 1000, 0))
        p2 = self._empty(Md, dd)
        self._lin_fwd(p1, self.prenet2, p2, Md, dropout=(rp, self._seed_now, 1001, 0))
        if self.gate_log is not None:
            self.gate_log += [p1 > 0, p2 > 0]
        dcat0 = self._empty(Md, 2 * dd)
        y0 = dcat0[:, :dd]
        T.affine_act(p2, y0, self.dec_scale, self.dec_shift, None, residual=self.pe_dec, r_mod=Tm, M=Md, C=dd)
        site1 = self._next_site()
        T.dropout(y0, y0, self.dropout_rate, self._seed_now, site1, M=Md, D=dd)

        def prenet_backward(dy0):
            T.dropout(dy0, dy0, self.dropout_rate, self._seed_now, site1, M=Md, D=dd)
            T.axpby(dy0, dy0, alpha=math.sqrt(np.float32(dd)) * inv_keep, beta=0.0, M=Md, D=dd)
            T.colreduce(1, self.prenet2.db, g=dy0, a=p2, act='relu', dy_out=dy0, M=Md, C=dd)
            T.wgrad(p1, dy0, self.prenet2.dk, M=Md, precision=self.backward_precision)
            dp1 = self._empty(Md, self.prenet1.N)
            ops.gemm(dy0, self.prenet2.lin.bwd, dp1, M=Md, precision=self.backward_precision, alpha=inv_keep)
            T.colreduce(1, self.prenet1.db, g=dp1, a=p1, act='relu', dy_out=dp1, M=Md, C=self.prenet1.N)
            T.wgrad(mels_in, dp1, self.prenet1.dk, M=Md, precision=self.backward_precision)
            dmels = self._empty(Md, de)
            ops.gemm(dp1, self.prenet1.lin.bwd, dmels, M=Md, precision=self.backward_precision)
            d_enc = self._empty(Me, de)
            T.expand_bwd(dmels, cum, d_enc, B, N, Tm, de)
            return d_enc

        dec_tape = []
        dec_out = self._stack_fwd(self.dec_blocks, dcat0, B, Tm, frame_mask, dd, dec_tape, dec_attn, self.dec_names)
        mel0 = self._empty(Md, mc)
        self._lin_fwd(dec_out, self.out, mel0, Md)
        mel = self._empty(Md, mc)
        post_tape = []
        self._cnn_bn_fwd(self.postnet, mel0, Md, Tm, mel, post_tape)

        # ---- losses (utils/losses.py:41-47,71-78)
        losses = torch.zeros(2, dtype=torch.float32, device=self.device)
        mel_t2 = mel_t.view(Md, mc)
        ops.masked_mae(mel_t2, mel, losses[0:1])
        ops.masked_mae(td.view(Me, 1), dur.view(Me, 1), losses[1:2])
        if self.gate_log is not None:
            self.gate_log += [torch.sign(mel - mel_t2), torch.sign(dur.view(Me, 1) - td.view(Me, 1))]

        # ---- backward
        dmel = self._empty(Md, mc)
        T.masked_mae_bwd(mel_t2, mel, dmel, mel_w)
        ddur = self._empty(Me)
        T.masked_mae_bwd(td.view(Me, 1), dur.view(Me, 1), ddur.view(Me, 1), dur_w)
        # the gradient buffer is in forward order: every finished stage hands its tail to the all-reduce, which runs on
        # the communication stream beside the rest of the backward pass
        overlap = self._new_overlap()
        g = post_tape[0](dmel)
        d_dec = self._empty(Md, dd)
        self._lin_bwd(dec_out, self.out, g, Md, dx=d_dec)
        overlap.ready(self._first_offset('out.'))
        g = d_dec
        for j, bw in enumerate(reversed(dec_tape)):
            g = bw(g)
            idx = len(dec_tape) - 1 - j            # dec_tape = [attention 0, tail 0, attention 1, tail 1, ...]
            if idx % 2 == 0:
                overlap.ready(self._first_offset(f'decoder.block{idx // 2}.'))
        d_enc = prenet_backward(g)
        overlap.ready(self._first_offset('decoder_prenet.'))
        # duration branch adds into d_enc
        dz1 = self._empty(Me, de)
        T.rownorm_bwd(fa1, ddot=ddur, dy=dz1, dx=d_enc, dx_accumulate=True, dln_a=D.norms[1].dpair, dln_b=D.norms[2].dpair,
                      ddot_w=self.ddur_w, ddot_b=self.ddur_b)
        da0 = self._empty(Me, de)
        self._lin_bwd(a0, D.convs[1], dz1, Me, N, dx=da0)
        dz0 = self._empty(Me, de)
        T.rownorm_bwd(fa0, dout=da0, dy=dz0, dln_a=D.norms[0].dpair)
        self._lin_bwd(enc_out, D.convs[0], dz0, Me, N, dx=d_enc, accumulate=True)
        overlap.ready(self._first_offset('dur_pred.'))
        g = d_enc
        for j, bw in enumerate(reversed(tape)):
            g = bw(g)
            idx = len(tape) - 1 - j                # tape = [embedding, attention 0, tail 0, attention 1, tail 1, ...]
            if idx >= 1 and (idx - 1) % 2 == 0:
                overlap.ready(self._first_offset(f'encoder.block{(idx - 1) // 2}.'))
        overlap.finish()
        self._apply_gradients()
        T.SumSlots.step_end(self.device)
        if global_lengths is not None:
            losses = losses * torch.tensor([Tm / global_lengths[0], N / global_lengths[1]], dtype=torch.float32, device=self.device)
        loss = losses[0] * self.loss_weights[0] + losses[1] * self.loss_weights[1]
        return {'mel': mel.view(B, Tm, mc), 'duration': dur.view(B, N, 1), 'expanded_mask': frame_mask.view(B, 1, 1, Tm),
                'encoder_attention': enc_attn, 'decoder_attention': dec_attn, 'loss': loss,
                'losses': {'mel': losses[0], 'duration': losses[1]}}

    def _apply_gradients(self):
        """Gradient exchange (sum over ranks, bucketed) and the Adam update (optimizer.apply_gradients, models.py:403)."""
        st = self.store                      # (gradients were summed over the ranks during the backward pass)
        self.iterations += 1
        t = self.iterations
        lr_t = self.learning_rate * math.sqrt(1.0 - self.beta2 ** t) / (1.0 - self.beta1 ** t)
        T.adam(st.flat, st.grad, st.m, st.v, lr_t, self.beta1, self.beta2, self.eps, 1.0 / self.world)
        if self._pack_plan is None:
            self._pack_plan = T.PackPlan(self._linears, self.device)
        self._pack_plan.run()

    def val_step(self, input_sequence, target_sequence, target_durations):
        """models.py:413-425: forward with training=False (moving statistics, no dropout) and the same losses."""
        from .forward_transformer import ForwardTransformer
        tokens, mel_t, td, B, N, Tm = self._prepare_batch(input_sequence, target_sequence, target_durations)
        if self._infer is None:
            self._infer = ForwardTransformer(self.weights(), self.config, device=self.device, precision=self.precision,
                                             fold_ffn=False)
        # (DecoderPrenet's dropout is on in the reference's validation pass too; its seed follows the optimizer step)
        self._infer.set_constants(decoder_prenet_dropout=self.decoder_prenet_dropout)
        self._infer.dropout_seed, self._infer._calls = self.seed, self.iterations
        out = self._infer.call(tokens, target_durations=td, max_frames=Tm)
        losses = torch.zeros(2, dtype=torch.float32, device=self.device)
        ops.masked_mae(mel_t.view(B * Tm, -1), out['mel'].reshape(B * Tm, -1), losses[0:1])
        ops.masked_mae(td.view(B * N, 1), out['duration'].reshape(B * N, 1), losses[1:2])
        out['loss'] = losses[0] * self.loss_weights[0] + losses[1] * self.loss_weights[1]
        out['losses'] = {'mel': losses[0], 'duration': losses[1]}
        return out
