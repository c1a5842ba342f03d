"""ForwardTransformer (non-autoregressive TransformerTTS) on MI355X.

Mirrors the surface of the reference class
(TransformerTTS/pretrained_models/for_text_encoder/models.py:283-477): `call`, `forward`,
`predict`, `set_constants`, `encode_text`, same argument meaning and the same output dict keys;
tensors are torch CUDA tensors instead of tf tensors.  All arithmetic runs in libestts.so kernels.
"""
import math

import numpy as np
import torch

from . import ops
from .blocks import ConvBNLayer, SelfAttentionStack, _dev_vec, pack_conv, pack_dense, pack_ln
from .text import Tokenizer, default_alphabet


class _Lane:
    """One of the sentences ForwardTransformer.predict_many keeps in flight: a replica of the model that shares the packed
    weights (read-only) and owns a stream, its workspaces, a split-K scratch and its graphs."""

    MAX_GRAPHS = 24

    def __init__(self, parent):
        import copy
        m = copy.copy(parent)
        m._ws, m._graphs = {}, {}
        m._ws_limit = 1 << 30        # a captured graph owns its workspaces: they leave the cache together with the graphs
        m.__dict__.pop('_lane_list', None)
        self.model = m
        self.stream = torch.cuda.Stream(device=parent.device)
        self.scratch = ops.new_gemm_workspace(parent.device)
        self.seen = set()            # shapes run once launch by launch; the next visit captures them
        self.graphs = {}             # key -> (graph, static inputs, outputs)
        self.state = None

    def _run(self, key, fn, copy_in=None):
        """fn() on this lane's stream: from its graph when the key has one, captured when the key comes for the second time,
        launch by launch when it is new.  Returns fn's result (a graph's own buffers when replayed)."""
        entry = self.graphs.get(key)
        if entry is None and key in self.seen:
            self.stream.synchronize()
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g, stream=self.stream):
                out = fn()
            entry = self.graphs[key] = (g, out, fn)      # (fn's closure keeps the workspaces the graph writes to alive)
        if entry is not None:
            entry[0].replay()
            return entry[1]
        self.seen.add(key)
        return fn()

    def begin(self, tokens, scalar, return_attention):
        m = self.model
        if len(self.graphs) >= self.MAX_GRAPHS:          # between two sentences: graphs, their workspaces and the shape memory go
            self.stream.synchronize()
            self.graphs.clear()
            self.seen.clear()
            m._ws.clear()
        with torch.cuda.stream(self.stream), ops.private_workspace(self.scratch):
            N = tokens.shape[1]
            if N > m.pe_enc.shape[0]:
                raise ValueError(f'{N} tokens exceed encoder_max_position_encoding={m.pe_enc.shape[0]}')
            ws = m._workspace(1, N, 0)
            tok = ws.setdefault('tokens_in', torch.empty(1, N, dtype=torch.int32, device=m.device))
            tok.copy_(tokens, non_blocking=True)

            def encode():
                attn = m._encode(tok, ws, 1, N, scalar, return_attention)
                ops.durations_cumsum(ws['dur'], ws['cum'], ws['totals'])
                return attn
            enc_attn = self._run(('enc', N, float(scalar), bool(return_attention)), encode)
            host_T = ws.setdefault('totals_host', torch.empty(1, dtype=torch.int32).pin_memory())
            host_T.copy_(ws['totals'], non_blocking=True)
            done = torch.cuda.Event()
            done.record(self.stream)
        self.state = (ws, enc_attn, N, host_T, done, float(scalar), bool(return_attention))

    def finish(self):
        m = self.model
        ws, enc_attn, N, host_T, done, scalar, return_attention = self.state
        done.synchronize()
        T = int(host_T[0])
        with torch.cuda.stream(self.stream), ops.private_workspace(self.scratch):
            ws_t = m._workspace(1, N, T)                   # (made here, not under a capture; kept alive by the closure below)
            out = self._run(('dec', N, T, scalar, return_attention),
                            lambda: m._finish_call(ws, enc_attn, 1, N, T, return_attention, keep=ws_t))
            # a replayed graph hands back its own buffers, which this lane's next sentence overwrites
            res = {k: (v.clone() if torch.is_tensor(v) else {n: a.clone() for n, a in v.items()}) for k, v in out.items()}
        return res


class ForwardTransformer:
    def __init__(self, weights=None, config=None, device='cuda', precision='bf16x3', fold_ffn=True, seed=1234, **keywords):
        """weights: dict name -> ndarray in Keras layouts (see weights.py); config: dict with the keys of
        config/*/forward_config.yaml; or the reference's own keyword form (ForwardTransformer(encoder_model_dimension=...,
        decoder_model_dimension=..., dropout_rate=..., ...), models.py:284-309), in which case the variables are initialised
        from `seed` as Keras would from its initialisers."""
        if not torch.cuda.is_available():
            raise RuntimeError('ForwardTransformer needs a GPU: the hot path has no CPU implementation')
        ops._lib.load()
        if config is None:
            from . import configs, weights as _weights
            config = configs.config_from_keywords('forward', **keywords)
            if weights is None:
                weights = _weights.init_forward_transformer(config, seed=seed)
        elif keywords:
            raise TypeError(f'unexpected keywords beside a config dict: {sorted(keywords)}')
        self.config = dict(config)
        self.device = torch.device(device)
        self.precision = ops.as_precision(precision)
        self.drop_n_heads = 0
        self.decoder_prenet_dropout = 0.0
        self.dropout_seed = 0
        self._calls = 0
        self.mel_channels = config['mel_channels']
        w, dev = weights, self.device
        de, dd = config['encoder_model_dimension'], config['decoder_model_dimension']
        self.de, self.dd = de, dd
        self.tokenizer = Tokenizer(default_alphabet(), add_start_end=False)
        self.emb = _dev_vec(w['encoder_prenet.embeddings'], dev)
        self.pe_enc = ops.positional_encoding_table(config['encoder_max_position_encoding'], de, dev) * float(
            w['encoder.pos_encoding_scalar'])
        self.pe_dec = ops.positional_encoding_table(config['decoder_max_position_encoding'], dd, dev) * float(
            w['decoder.pos_encoding_scalar'])
        self.encoder = SelfAttentionStack(w, 'encoder', 'Encoder', de, config['encoder_num_heads'],
                                          config['encoder_dense_blocks'], dev, fold_ffn=fold_ffn)
        self.decoder = SelfAttentionStack(w, 'decoder', 'Decoder', dd, config['decoder_num_heads'],
                                          config['decoder_dense_blocks'], dev, fold_ffn=fold_ffn)
        # duration predictor: CNNResNorm with LayerNorm (layers.py:618-642)
        self.dp_conv0 = pack_conv(w, 'dur_pred.conv_blocks.conv0', 'same', dev)
        self.dp_conv1 = pack_conv(w, 'dur_pred.conv_blocks.last_conv', 'same', dev)
        self.dp_ln = [pack_ln(w, f'dur_pred.conv_blocks.norm{i}', dev) for i in range(3)]
        self.dp_w = _dev_vec(np.asarray(w['dur_pred.linear.kernel'])[:, 0], dev)
        self.dp_b = float(np.asarray(w['dur_pred.linear.bias'])[0])
        self.prenet1 = pack_dense(w, 'decoder_prenet.d1', dev, act='relu')
        self.prenet2 = pack_dense(w, 'decoder_prenet.d2', dev, act='relu')
        self.out = pack_dense(w, 'out', dev)
        self.postnet = ConvBNLayer(w, 'decoder_postnet', config['postnet_conv_layers'], 'tanh', 'linear', 'same', dev)
        self._ws = {}
        self._ws_limit = 8       # cached workspace shapes before the cache is emptied

    # ------------------------------------------------------------------ workspace
    def _workspace(self, B, N, T):
        key = (B, N, T)
        ws = self._ws.get(key)
        if ws is None:
            dev, f32 = self.device, dict(dtype=torch.float32, device=self.device)
            Me, Md = B * N, B * T
            ws = {'enc': self.encoder.alloc(Me, dev), 'dec': self.decoder.alloc(Md, dev),
                  'key_mask': torch.empty(B, N, **f32), 'dp_a': torch.empty(Me, self.de, **f32),
                  'dp_b': torch.empty(Me, self.de, **f32), 'dp_c': torch.empty(Me, self.de, **f32),
                  'dur': torch.empty(B, N, **f32),
                  'cum': torch.empty(B, N, dtype=torch.int32, device=dev),
                  'totals': torch.empty(B, dtype=torch.int32, device=dev),
                  'mels_in': torch.empty(Md, self.de, **f32), 'frame_mask': torch.empty(B, T, **f32),
                  'p1': torch.empty(Md, self.prenet1.N, **f32), 'mel0': torch.empty(Md, self.mel_channels, **f32),
                  'pn_a': torch.empty(Md, max(self.postnet.hidden(), 4), **f32),
                  'pn_b': torch.empty(Md, max(self.postnet.hidden(), 4), **f32)}
            if len(self._ws) > self._ws_limit:
                self._ws.clear()
            self._ws[key] = ws
        return ws

    # ------------------------------------------------------------------ stages
    def _encode(self, tokens, ws, B, N, durations_scalar, return_attention):
        de = self.de
        e = ws['enc']
        ops.embed_posenc(tokens, self.emb, self.pe_enc, None, e['xc'], ws['key_mask'], math.sqrt(np.float32(de)))
        attn = self.encoder.run(e, B, N, ws['key_mask'], self.precision, return_attention)
        x = e['xc'][:, :de]
        ops.gemm(x, self.dp_conv0, ws['dp_a'], seg_len=N, precision=self.precision, site='dur_pred')
        if ops.can_fuse_ln(self.dp_conv1, B * N, precision=self.precision, site='dur_pred'):
            # relu(LN(conv0)) computed by the second convolution on its way in (one launch fewer)
            ops.gemm(ws['dp_b'], self.dp_conv1, ws['dp_c'], seg_len=N, precision=self.precision, site='dur_pred',
                     ln=dict(y=ws['dp_a'], ln_a=self.dp_ln[0], act='relu', out=ws['dp_b']))
        else:
            ops.rownorm(ws['dp_a'], ws['dp_b'], ln_a=self.dp_ln[0], act='relu')
            ops.gemm(ws['dp_b'], self.dp_conv1, ws['dp_c'], seg_len=N, precision=self.precision, site='dur_pred')
        ops.rownorm(ws['dp_c'], None, ln_a=self.dp_ln[1], act='relu', residual=x, ln_b=self.dp_ln[2],
                    dot_w=self.dp_w, dot_b=self.dp_b, dot_scale=float(durations_scalar),
                    dot_mask=ws['key_mask'].view(-1), dot_out=ws['dur'].view(-1))
        return attn

    def _decode(self, ws, B, N, T, return_attention):
        de, dd = self.de, self.dd
        d = ws['dec']
        ops.expand(ws['enc']['xc'][:, :de], ws['cum'], ws['mels_in'], ws['frame_mask'], B, N, T, de)
        # DecoderPrenet's dropout runs with training=True at inference too (layers.py:394-402): counter-based masks in the
        # GEMM epilogue, keyed by (seed, site 1000 / 1001, row, column), a new seed for every call
        rate = self.decoder_prenet_dropout
        drop1 = drop2 = None
        if rate > 0.0:
            if torch.cuda.is_current_stream_capturing():
                raise NotImplementedError('decoder_prenet_dropout > 0 cannot be captured into a graph: every replay would '
                                          'draw the same mask (the seed is a launch argument)')
            seed = (self.dropout_seed * 1000003 + self._calls * 7919 + 12345) & 0xFFFFFFFF
            self._calls += 1
            drop1, drop2 = (rate, seed, 1000, 0), (rate, seed, 1001, 0)
        ops.gemm(ws['mels_in'], self.prenet1, ws['p1'], precision=self.precision, dropout=drop1, site='prenet')
        ops.gemm(ws['p1'], self.prenet2, d['xc'][:, :dd], alpha=math.sqrt(np.float32(dd)), residual=self.pe_dec,
                 r_mod=T, precision=self.precision, dropout=drop2, site='prenet')
        attn = self.decoder.run(d, B, T, ws['frame_mask'], self.precision, return_attention)
        ops.gemm(d['xc'][:, :dd], self.out, ws['mel0'], precision=self.precision, site='out')
        mel = torch.empty(B * T, self.mel_channels, dtype=torch.float32, device=self.device)
        self.postnet.run(ws['mel0'], ws['pn_a'], ws['pn_b'], mel, T, self.precision, site='postnet')
        return mel.view(B, T, self.mel_channels), attn

    # ------------------------------------------------------------------ reference surface
    def call(self, x, target_durations=None, training=False, durations_scalar=1., return_attention=True,
             max_frames=None):
        """models.py:434-456.  x: int32 [B,N] tokens (0 = padding); target_durations: [B,N] or None.

        With target_durations=None the predicted durations decide the number of frames, which costs one
        device-to-host read of B integers (the reference has the same dependency inside Expand).
        `max_frames` (optional) fixes T without that read: frames beyond it are dropped, missing ones are padding.
        """
        if training:
            raise NotImplementedError('training=True runs through estts.forward_training.ForwardTransformerTrainer (dropout, batch statistics)')
        tokens = torch.as_tensor(x, device=self.device).to(torch.int32).contiguous()
        B, N = tokens.shape
        if N > self.pe_enc.shape[0]:
            raise ValueError(f'{N} tokens exceed encoder_max_position_encoding={self.pe_enc.shape[0]}')
        T_known = max_frames
        td = None
        if target_durations is not None:
            if not torch.is_tensor(target_durations) or not target_durations.is_cuda:
                tdn = np.rint(np.asarray(target_durations, dtype=np.float64))
                if T_known is None:
                    T_known = int(np.maximum(tdn, 0).sum(axis=1).max())
                td = torch.from_numpy(tdn.astype(np.float32)).to(self.device)
            else:
                td = target_durations.to(torch.float32)
            td = td.reshape(B, N).contiguous()
        ws = self._workspace(B, N, T_known or 0)
        enc_attn = self._encode(tokens, ws, B, N, durations_scalar, return_attention)
        ops.durations_cumsum(td if td is not None else ws['dur'], ws['cum'], ws['totals'])
        T = T_known
        if T is None:
            T = int(ws['totals'].max().item())
        return self._finish_call(ws, enc_attn, B, N, T, return_attention)

    def _finish_call(self, ws, enc_attn, B, N, T, return_attention, keep=None):
        """The part of call() behind the point where the number of frames is known: length regulator, decoder, post-net."""
        if ws is not self._ws.get((B, N, T)):
            ws2 = self._workspace(B, N, T)
            for k in ('enc', 'key_mask', 'dur', 'cum', 'totals'):
                ws2[k] = ws[k]
            ws = ws2
        if T > self.pe_dec.shape[0]:
            raise ValueError(f'{T} frames exceed decoder_max_position_encoding={self.pe_dec.shape[0]}')
        duration = ws['dur'].clone().view(B, N, 1)
        if T == 0:
            mel = torch.zeros(B, 0, self.mel_channels, device=self.device)
            return {'mel': mel, 'duration': duration, 'expanded_mask': torch.zeros(B, 1, 1, 0, device=self.device),
                    'encoder_attention': enc_attn, 'decoder_attention': {}}
        mel, dec_attn = self._decode(ws, B, N, T, return_attention)
        return {'mel': mel, 'duration': duration, 'expanded_mask': ws['frame_mask'].clone().view(B, 1, 1, T),
                'encoder_attention': enc_attn, 'decoder_attention': dec_attn}

    __call__ = call

    def graph_call(self, x, target_durations=None, max_frames=None, durations_scalar=1., return_attention=True):
        """`call` replayed from a hipGraph: the ~60 launches of a forward are captured once per shape (B, N, max_frames) and every
        later call of that shape is one graph launch (0.41 ms instead of 0.76 ms at B=1, N=100, T=600, where issuing the
        launches one by one from Python is slower than the GPU runs them).  `max_frames` is required: a graph cannot wait for
        the predicted durations to decide its own shapes, so the decoder always runs over max_frames frames (frames past an
        utterance's total are padding, frames beyond max_frames are dropped; `expanded_mask` tells which).  The returned
        tensors are the graph's own buffers: they are overwritten by the next graph_call of the same shape.  No counterpart in
        the reference (its tf.function input signatures play this role)."""
        if max_frames is None:
            raise ValueError('graph_call needs max_frames (the captured decoder has a fixed number of frames)')
        if self.decoder_prenet_dropout > 0.0:
            raise NotImplementedError('decoder_prenet_dropout > 0 cannot be captured into a graph: every replay would draw the same mask')
        tokens = torch.as_tensor(x, device=self.device).to(torch.int32)
        B, N = tokens.shape
        key = (B, N, int(max_frames), target_durations is not None, bool(return_attention), float(durations_scalar))
        if not hasattr(self, '_graphs'):
            self._graphs = {}
        entry = self._graphs.get(key)
        td = None
        if target_durations is not None:
            td = torch.as_tensor(target_durations, device=self.device).to(torch.float32).reshape(B, N)
            td = torch.round(td)                     # as call() does for host durations (round half to even)
        if entry is None:
            if len(self._graphs) >= 8:
                self._graphs.clear()
            tok_s = tokens.clone().contiguous()
            td_s = td.clone().contiguous() if td is not None else None
            model = self

            def run():
                return model.call(tok_s, target_durations=td_s, durations_scalar=durations_scalar,
                                  return_attention=return_attention, max_frames=int(max_frames))
            side = torch.cuda.Stream(device=self.device)
            side.wait_stream(torch.cuda.current_stream(self.device))
            with torch.cuda.stream(side):
                run()                                 # warm-up outside the capture: workspaces, library load
            torch.cuda.current_stream(self.device).wait_stream(side)
            torch.cuda.synchronize(self.device)
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g):
                out = run()
            entry = self._graphs[key] = (g, tok_s, td_s, out)
        g, tok_s, td_s, out = entry
        tok_s.copy_(tokens)
        if td_s is not None:
            td_s.copy_(td)
        g.replay()
        return out

    def forward(self, input_sequence, durations_scalar=1., **kw):
        """models.py:427-428."""
        return self.call(input_sequence, target_durations=None, training=False, durations_scalar=durations_scalar, **kw)

    def set_constants(self, decoder_prenet_dropout=None, learning_rate=None, drop_n_heads=None, **kwargs):
        """models.py:458-465.  The decoder-prenet dropout rate takes effect at inference too, as in the reference
        (layers.py:394-402); head drop is recorded only (it is inactive outside training)."""
        if decoder_prenet_dropout is not None:
            if not 0.0 <= float(decoder_prenet_dropout) < 1.0:
                raise ValueError('decoder_prenet_dropout must be in [0, 1)')
            self.decoder_prenet_dropout = float(decoder_prenet_dropout)
        if drop_n_heads is not None:
            self.drop_n_heads = drop_n_heads

    def encode_text(self, text):
        """models.py:467-468.  The phonemizer (espeak) is an external program; `text` must already be phonemes."""
        return self.tokenizer(text)

    def predict_many(self, inputs, encode=True, speed_regulator=1., in_flight=4, return_attention=False):
        """predict() for a list of sentences with up to `in_flight` of them on the GPU at once (no counterpart in the
        reference, whose synthesize_sentences*.py loop is one sentence at a time: synthesize_sentences_speaker.py:56-75).

        A batch-1 forward is a chain of ~55 latency-bound launches that leaves most of the chip idle, so independent
        sentences overlap almost for free (2 in flight: 1.55x, 4: 1.95x the one-at-a-time rate at N=100, T=600).  Every lane
        is a replica of this model that shares its packed weights and has its own HIP stream, workspaces, split-K scratch and
        hipGraphs: the encoder + duration predictor of a sentence is one graph per token count, the decoder one graph per
        (token count, frame count); a shape a lane meets for the first time runs launch by launch and is captured when it
        comes again.  The kernels and shapes are those of predict(), so every result is bit-identical to predict() on the
        same sentence.  Returns a list of predict()-style dicts in input order (attention maps only with return_attention)."""
        if self.decoder_prenet_dropout > 0.0:
            raise NotImplementedError('predict_many replays captured graphs: decoder_prenet_dropout must be 0')
        reqs = [torch.tensor([self.encode_text(t)], dtype=torch.int32) if encode else
                torch.as_tensor(np.asarray(t), dtype=torch.int32).reshape(1, -1) for t in inputs]
        lanes = self._lanes(max(1, min(int(in_flight), len(reqs) or 1)))
        scalar = 1. / speed_regulator
        results = [None] * len(reqs)
        cur = torch.cuda.current_stream(self.device)
        for lane in lanes:
            lane.stream.wait_stream(cur)
        for w0 in range(0, len(reqs), len(lanes)):
            wave = list(zip(lanes, range(w0, min(w0 + len(lanes), len(reqs)))))
            for lane, i in wave:                       # encoders and duration predictors of the wave, one per lane
                lane.begin(reqs[i], scalar, return_attention)
            for lane, i in wave:                       # frame counts (one small read per lane), then the decoders
                results[i] = lane.finish()
        for lane in lanes:
            cur.wait_stream(lane.stream)
        for out in results:
            out['mel'] = out['mel'].squeeze(0)
            for v in out.values():                     # allocated on a lane's stream, consumed on the caller's
                for t in (v.values() if isinstance(v, dict) else [v]):
                    t.record_stream(cur)
        return results

    def _lanes(self, k):
        lanes = self.__dict__.setdefault('_lane_list', [])
        while len(lanes) < k:
            lanes.append(_Lane(self))
        return lanes[:k]

    def predict(self, inp, encode=True, speed_regulator=1.):
        """models.py:470-477."""
        if encode:
            inp = torch.tensor([self.encode_text(inp)], dtype=torch.int32)
        out = self.forward(inp, durations_scalar=1. / speed_regulator)
        out['mel'] = out['mel'].squeeze()
        return out
