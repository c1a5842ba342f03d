"""WaveRNN training on MI355X: teacher-forced forward over whole windows with every GRU state kept, cross-entropy (RAW)
or mixture-of-logistics (MOL) loss, back-propagation through time, torch-style Adam, data-parallel gradient all-reduce.

Mirrors what WaveRNN/train_wavernn.py:27-42 does per batch with the reference module (models/fatchord_version.py:119-148
in train() mode, torch.optim.Adam at :104): `train_step(x, y, mels)` = forward, loss, backward, optimizer step;
`forward(x, mels)` returns the logits; `state_dict()` / `load_state_dict()` use the reference's key names and layouts, so a
checkpoint moves freely between this class, estts.wavernn.WaveRNN (generation) and the reference.

All arithmetic runs in libestts.so kernels.  The two recurrences are 2 x T dependent (small GEMM + gate kernel) pairs in
each direction; buffers are allocated once per (batch, window) shape so that each direction is captured into one hipGraph and
replayed every step.  Within a direction the two layers' chains overlap on two streams (rnn2 follows rnn1 a chunk of steps
behind; backward, rnn1 follows rnn2): each chain is latency-bound at 32 rows per product, side by side they cost little more
than one.  Master weights are float32 in one flat buffer ([in, out] layout, the transpose of torch's), with
gradients and Adam moments in buffers of the same shape.  The three upsampling kernels enter through their composite
per-phase filter (see wavernn.composite_upsample_filter): its gradient is reduced on the device, the chain rule back to the
three small kernels is a few hundred multiply-adds on the host.
"""
import math
import os

import numpy as np
import torch

from . import _lib, ops, parallel
from . import train_ops as T_
from ._lib import WavernnPrepArgs, check
from .forward_training import ParamStore
from .train_ops import TrainLinear
from .wavernn import composite_filter_gradients, composite_filter_support, composite_upsample_filter


class _Lin:
    def __init__(self, store, name, bias_name=None, need_bwd=True, act=None):
        self.lin = TrainLinear(store.p[name], store.p[bias_name] if bias_name else None, act=act, need_bwd=need_bwd)
        self.dk = store.g[name]
        self.db = store.g[bias_name] if bias_name else None
        self.N, self.K = self.lin.N, self.lin.K


class _BN:
    def __init__(self, store, p):
        self.gamma, self.beta = store.p[p + '.weight'], store.p[p + '.bias']
        self.dgamma, self.dbeta = store.g[p + '.weight'], store.g[p + '.bias']
        self.mm, self.mv = store.s[p + '.running_mean'], store.s[p + '.running_var']


class WaveRNNTrainer:
    def __init__(self, rnn_dims, fc_dims, bits, pad, upsample_factors, feat_dims, compute_dims, res_out_dims, res_blocks,
                 hop_length, sample_rate, mode='RAW', device='cuda', precision='bf16x3', process_group=None, bucket_mb=25,
                 use_graph=True):
        if mode == 'RAW':
            self.n_classes = 2 ** bits
        elif mode == 'MOL':
            self.n_classes = 30
        else:
            raise RuntimeError(f'Unknown model mode value - {mode}')
        if int(np.prod(upsample_factors)) != hop_length:
            raise ValueError('upsample_factors must multiply to hop_length')
        if not torch.cuda.is_available():
            raise RuntimeError('WaveRNN training needs a GPU: the hot path has no CPU implementation')
        _lib.load()
        self.mode, self.pad, self.bits = mode, pad, bits
        self.rnn_dims, self.fc_dims = rnn_dims, fc_dims
        self.aux_dims = res_out_dims // 4
        self.feat_dims, self.compute_dims, self.res_out_dims, self.res_blocks = feat_dims, compute_dims, res_out_dims, res_blocks
        self.upsample_factors = tuple(upsample_factors)
        self.hop_length, self.sample_rate = hop_length, sample_rate
        self.device = torch.device(device)
        self.precision = precision
        # a recurrent step as ONE launch (product + gates, estts_gru_fused_fwd / _bwd) instead of two; False: estts_gemm followed
        # by estts_gru_step_fwd / _bwd (for comparisons)
        # (the one-launch backward step holds K = 3 * rnn_dims in flight: up to 1536, i.e. rnn_dims <= 512 -- the shipped
        # size; larger layers take the two-launch steps, which have no such limit)
        self.fused_gru = 3 * rnn_dims <= ops.GRU_FUSED_MAX_H and rnn_dims % 16 == 0
        # with one-launch steps: both layers' recurrences as ONE chain of launches (step t of one layer and step t -+ chunk of
        # the other in the same launch) instead of two chains on two streams
        self.one_chain = self.fused_gru
        self.use_graph = use_graph
        self.step = 0
        self.lr, self.beta1, self.beta2, self.eps = 1e-4, 0.9, 0.999, 1e-8      # torch.optim.Adam defaults, hp.voc_lr
        self.iterations = 0
        self.pg = process_group
        self.world = 1
        if process_group is not None or (torch.distributed.is_available() and torch.distributed.is_initialized()):
            self.world = torch.distributed.get_world_size(process_group)
        self.bucket_elems = int(bucket_mb * (1 << 20) // 4)
        self._comm = torch.cuda.Stream(device=self.device) if self.world > 1 else None
        self.n_in = 1 + feat_dims + self.aux_dims
        self.k_in = (self.n_in + 3) // 4 * 4
        self.nc_pad = (self.n_classes + 3) // 4 * 4
        self.fir_kmin, self.ntaps = composite_filter_support(self.upsample_factors)
        if self.pad - self.fir_kmin - (self.ntaps - 1) < 0 or self.pad + self.fir_kmin < 0:
            raise ValueError('pad is too small for the upsampling filters')
        self.store = None
        self._ws = None
        ops.gemm_workspace(self.device)

    @classmethod
    def from_hparams(cls, hp, **kw):
        return cls(rnn_dims=hp['voc_rnn_dims'], fc_dims=hp['voc_fc_dims'], bits=hp['bits'], pad=hp['voc_pad'],
                   upsample_factors=hp['voc_upsample_factors'], feat_dims=hp['num_mels'],
                   compute_dims=hp['voc_compute_dims'], res_out_dims=hp['voc_res_out_dims'],
                   res_blocks=hp['voc_res_blocks'], hop_length=hp['hop_length'], sample_rate=hp['sample_rate'],
                   mode=hp['voc_mode'], **kw)

    # ------------------------------------------------------------------ weights (reference key names and layouts)
    def load_state_dict(self, sd, strict=False):
        w = {k: (v.detach().cpu().numpy() if torch.is_tensor(v) else np.asarray(v)) for k, v in sd.items()}
        if 'step' in w:
            self.step = int(np.asarray(w['step']).reshape(-1)[0])
        st = ParamStore(self.device)
        p = 'upsample.resnet'

        def bn(q):
            st.add(q + '.weight', w[q + '.weight'])
            st.add(q + '.bias', w[q + '.bias'])
            st.add(q + '.running_mean', w[q + '.running_mean'], trainable=False)
            st.add(q + '.running_var', w[q + '.running_var'], trainable=False)

        def lin(name, pad_k=None, pad_n=None):
            m = np.asarray(w[name + '.weight'], np.float32).T               # torch [out, in] -> master [in, out]
            K, N = m.shape
            full = np.zeros((pad_k or K, pad_n or N), np.float32)
            full[:K, :N] = m
            st.add(name + '.weight', full)
            if name + '.bias' in w:
                b = np.zeros(pad_n or N, np.float32)
                b[:N] = w[name + '.bias']
                st.add(name + '.bias', b)

        wi = np.asarray(w[p + '.conv_in.weight'], np.float32)               # (C, feat, k) -> [k * feat, C], tap-major
        self.k_conv = wi.shape[2]
        if self.k_conv != 2 * self.pad + 1:
            raise ValueError('conv_in kernel size does not match pad')
        st.add(p + '.conv_in.weight', wi.transpose(2, 1, 0).reshape(-1, wi.shape[0]))
        bn(p + '.batch_norm')
        for i in range(self.res_blocks):
            q = f'{p}.layers.{i}'
            st.add(q + '.conv1.weight', np.asarray(w[q + '.conv1.weight'], np.float32)[:, :, 0].T)
            bn(q + '.batch_norm1')
            st.add(q + '.conv2.weight', np.asarray(w[q + '.conv2.weight'], np.float32)[:, :, 0].T)
            bn(q + '.batch_norm2')
        st.add(p + '.conv_out.weight', np.asarray(w[p + '.conv_out.weight'], np.float32)[:, :, 0].T)
        st.add(p + '.conv_out.bias', w[p + '.conv_out.bias'])
        for j in range(len(self.upsample_factors)):
            st.add(f'upsample.up_layers.{2 * j + 1}.weight', np.asarray(w[f'upsample.up_layers.{2 * j + 1}.weight']).reshape(-1))
        lin('I', pad_k=self.k_in)
        for r in ('rnn1', 'rnn2'):
            st.add(r + '.weight_ih_l0', np.asarray(w[r + '.weight_ih_l0'], np.float32).T)
            st.add(r + '.bias_ih_l0', w[r + '.bias_ih_l0'])
            st.add(r + '.weight_hh_l0', np.asarray(w[r + '.weight_hh_l0'], np.float32).T)
            st.add(r + '.bias_hh_l0', w[r + '.bias_hh_l0'])
        lin('fc1')
        lin('fc2')
        lin('fc3', pad_n=self.nc_pad)
        st.build()
        self.store = st
        self.conv_in = _Lin(st, p + '.conv_in.weight', need_bwd=False)
        self.bn_in = _BN(st, p + '.batch_norm')
        self.blocks = []
        for i in range(self.res_blocks):
            q = f'{p}.layers.{i}'
            self.blocks.append((_Lin(st, q + '.conv1.weight'), _BN(st, q + '.batch_norm1'),
                                _Lin(st, q + '.conv2.weight'), _BN(st, q + '.batch_norm2')))
        self.conv_out = _Lin(st, p + '.conv_out.weight', p + '.conv_out.bias')
        self.I = _Lin(st, 'I.weight', 'I.bias')
        self.ih1 = _Lin(st, 'rnn1.weight_ih_l0', 'rnn1.bias_ih_l0')
        self.hh1 = _Lin(st, 'rnn1.weight_hh_l0', 'rnn1.bias_hh_l0')
        self.ih2 = _Lin(st, 'rnn2.weight_ih_l0', 'rnn2.bias_ih_l0')
        self.hh2 = _Lin(st, 'rnn2.weight_hh_l0', 'rnn2.bias_hh_l0')
        self.fc1 = _Lin(st, 'fc1.weight', 'fc1.bias', act='relu')
        self.fc2 = _Lin(st, 'fc2.weight', 'fc2.bias', act='relu')
        self.fc3 = _Lin(st, 'fc3.weight', 'fc3.bias')
        self._linears = ([self.conv_in.lin, self.conv_out.lin, self.I.lin, self.ih1.lin, self.hh1.lin, self.ih2.lin, self.hh2.lin,
                          self.fc1.lin, self.fc2.lin, self.fc3.lin] + [b[0].lin for b in self.blocks] + [b[2].lin for b in self.blocks])
        self._up_names = [f'upsample.up_layers.{2 * j + 1}.weight' for j in range(len(self.upsample_factors))]
        self._up_shapes = [np.asarray(w[n]).shape for n in self._up_names]
        self._refresh_fir()
        self._ws = None
        return self

    def _refresh_fir(self):
        ups = [self.store.p[n].cpu().numpy() for n in self._up_names]
        fir, kmin = composite_upsample_filter(ups, self.upsample_factors)
        assert kmin == self.fir_kmin and fir.shape[1] == self.ntaps
        self._up_host = ups
        if getattr(self, 'fir', None) is None or self.fir.shape != fir.shape:
            self.fir = torch.from_numpy(fir).to(self.device)
        else:
            self.fir.copy_(torch.from_numpy(fir))

    def state_dict(self):
        out = self._export(self.store.numpy())
        out['step'] = torch.tensor([self.step], dtype=torch.long)
        return out

    def gradients(self):
        """The gradients of the last train_step under the reference's parameter names and layouts (summed over ranks when
        data parallel, before the 1 / world scaling the optimizer applies)."""
        raw = self.store.numpy(self.store.grad)
        return {k: v for k, v in self._export(raw).items() if 'running' not in k}

    def _export(self, raw):
        sd = {}
        p = 'upsample.resnet'
        C, feat, k = self.compute_dims, self.feat_dims, self.k_conv
        for name, v in raw.items():
            if name == p + '.conv_in.weight':
                sd[name] = v.reshape(k, feat, C).transpose(2, 1, 0).copy()
            elif name.endswith(('.conv1.weight', '.conv2.weight', 'conv_out.weight')):
                sd[name] = v.T.copy()[:, :, None]
            elif 'up_layers' in name:
                sd[name] = v.reshape(self._up_shapes[self._up_names.index(name)])
            elif name == 'I.weight':
                sd[name] = v[:self.n_in].T.copy()
            elif name == 'fc3.weight':
                sd[name] = v[:, :self.n_classes].T.copy()
            elif name == 'fc3.bias':
                sd[name] = v[:self.n_classes].copy()
            elif name.endswith('.weight') and not ('batch_norm' in name) or 'weight_ih' in name or 'weight_hh' in name:
                sd[name] = v.T.copy()
            else:
                sd[name] = v
        return {k_: torch.from_numpy(np.ascontiguousarray(v)) for k_, v in sd.items()}

    def get_step(self):
        return self.step

    def save(self, path):
        torch.save(self.state_dict(), path)

    def load(self, path):
        self.load_state_dict(torch.load(path, map_location='cpu', weights_only=True))

    def broadcast_running_statistics(self, src=0):
        """BatchNorm running statistics are per replica during data-parallel training; before a checkpoint every rank takes
        rank `src`'s."""
        if self.world > 1:
            parallel.broadcast_(self.store.state, src=src, group=self.pg)

    def restore(self, path):
        """fatchord_version.py:396-402: a new session writes its initial weights, an old one is loaded."""
        if not os.path.exists(path):
            print('\nNew WaveRNN Training Session...\n')
            self.save(path)
        else:
            print(f'\nLoading Weights: "{path}"\n')
            self.load(path)

    def checkpoint(self, path):
        """fatchord_version.py:385-387."""
        self.save(os.path.join(path, f'checkpoint_{self.get_step() // 1000}k_steps.pyt'))

    def log(self, path, msg):
        """fatchord_version.py:392-394."""
        with open(path, 'a') as f:
            print(msg, file=f)

    # ------------------------------------------------------------------ workspace and captured loops
    def _workspace(self, B, Tf):
        key = (B, Tf)
        if self._ws is not None and self._ws['key'] == key:
            return self._ws
        dev, f32 = self.device, dict(dtype=torch.float32, device=self.device)
        R, F, d, hop = self.rnn_dims, self.fc_dims, self.aux_dims, self.hop_length
        Tm, T = Tf + 2 * self.pad, Tf * hop
        M, Mr = B * T, B * Tf
        i32 = dict(dtype=torch.int32, device=dev)
        ws = {'key': key, 'B': B, 'Tf': Tf, 'T': T, 'M': M, 'Mr': Mr,
              'mel': torch.empty(B * Tm, self.feat_dims, **f32), 'unf': torch.empty(Mr, self.k_conv * self.feat_dims, **f32),
              'aux': torch.empty(Mr, self.res_out_dims, **f32),
              'mel_off': torch.arange(B, **i32) * Tm, 'aux_off': torch.arange(B, **i32) * Tf,
              'utt_samples': torch.full((B,), T, **i32),
              'seq_utt': torch.arange(B, **i32).repeat_interleave(T).contiguous(),
              'seq_start': torch.arange(T, **i32).repeat(B).contiguous(),
              'step0': torch.zeros(1, **i32), 'x': torch.empty(M, **f32),
              'xin': torch.zeros(M, self.k_in, **f32), 'xI': torch.empty(M, R, **f32),
              'gi1': torch.empty(M, 3 * R, **f32), 'gh1': torch.empty(M, 3 * R, **f32), 'H1': torch.empty(M, R, **f32),
              'gi2': torch.empty(M, 3 * R, **f32), 'gh2': torch.empty(M, 3 * R, **f32), 'H2': torch.empty(M, R, **f32),
              'x1cat': torch.zeros(M, R + d, **f32), 'x2cat': torch.zeros(M, R + d, **f32),
              'f1cat': torch.zeros(M, F + d, **f32), 'f2': torch.empty(M, F, **f32),
              'logits': torch.zeros(M, self.nc_pad, **f32), 'dlogits': torch.zeros(M, self.nc_pad, **f32),
              'h0': torch.zeros(B, R, **f32),
              # backward
              'df2': torch.empty(M, F, **f32), 'df1cat': torch.empty(M, F + d, **f32), 'dx2cat': torch.empty(M, R + d, **f32),
              'dx1cat': torch.empty(M, R + d, **f32), 'dgi': torch.empty(M, 3 * R, **f32), 'dgh': torch.empty(M, 3 * R, **f32),
              'dhd': torch.empty(B, R, **f32), 'dhr': torch.empty(B, R, **f32), 'dxI': torch.empty(M, R, **f32),
              # rnn1's own copies: its backward recurrence runs beside rnn2's
              'dgi1': torch.empty(M, 3 * R, **f32), 'dgh1': torch.empty(M, 3 * R, **f32),
              'dhd1': torch.empty(B, R, **f32), 'dhr1': torch.empty(B, R, **f32), 'res2': torch.zeros(M, R + d, **f32),
              'stream_b': torch.cuda.Stream(device=dev), 'gemm_ws_b': ops.new_gemm_workspace(dev),
              'dxin': torch.empty(M, self.k_in, **f32), 'd_aux': torch.empty(Mr, self.res_out_dims, **f32),
              'dfir': torch.zeros(hop, self.ntaps, **f32), 'loss': torch.zeros(1, **f32),
              'labels': torch.zeros(M, **i32), 'yf': torch.zeros(M, **f32), 'graphs': {}}
        self._ws = ws
        return ws

    def _run(self, ws, name, fn):
        """Runs fn() (a fixed sequence of launches on buffers of `ws`) through a hipGraph captured on first use."""
        if not self.use_graph:
            return fn()
        g = ws['graphs'].get(name)
        if g is None:
            fn()                                      # first use: run eagerly (also warms every kernel up) ...
            torch.cuda.synchronize()
            g = torch.cuda.CUDAGraph()
            side = torch.cuda.Stream()
            side.wait_stream(torch.cuda.current_stream())
            traced, ops._trace = ops._trace, None     # (a launch trace, if one is running, has seen them already)
            try:
                with torch.cuda.graph(g, stream=side):   # ... and record the same launches for the following steps
                    fn()
            finally:
                ops._trace = traced
            torch.cuda.current_stream().wait_stream(side)
            ws['graphs'][name] = g
            return None
        g.replay()

    def _gru_fwd(self, ws, gi, gh, H, hh, x_res, out, out_cols, t0=0, t1=None, gemm_ws=None):
        """Steps t0 .. t1 - 1 of one GRU layer over every window (one recurrent product and one gate kernel per step)."""
        B, T, R = ws['B'], ws['T'], self.rnn_dims
        gi3, gh3, H3 = gi.view(B, T, 3 * R), gh.view(B, T, 3 * R), H.view(B, T, R)
        xr3, out3 = x_res.view(B, T, -1), out.view(B, T, -1)
        for t in range(t0, T if t1 is None else t1):
            hp = ws['h0'] if t == 0 else H3[:, t - 1]
            if self.fused_gru:       # the recurrent product with the gates as its epilogue: one launch per step
                T_.gru_fused_fwd(gi3[:, t], gh3[:, t], hp, hh.lin.fwd, H3[:, t], x_res=xr3[:, t, :out_cols],
                                 out=out3[:, t, :out_cols], precision=self.precision)
                continue
            ops.gemm(hp, hh.lin.fwd, gh3[:, t], precision=self.precision, workspace=gemm_ws)
            T_.gru_step_fwd(gi3[:, t], gh3[:, t], hp, H3[:, t], x_res=xr3[:, t, :out_cols], out=out3[:, t, :out_cols])

    def _gru_bwd(self, ws, gi, gh, H, hh, dout, dgi, dgh, dhd, dhr, t_hi=None, t_lo=0, gemm_ws=None):
        """Steps t_hi - 1 .. t_lo (descending) of one layer's backward recurrence.  dout: [M, >= R] gradient w.r.t. the layer
        output (columns 0..R-1); fills dgi, dgh; dhd / dhr carry the hidden-state gradient between steps."""
        B, T, R = ws['B'], ws['T'], self.rnn_dims
        gi3, gh3, H3 = gi.view(B, T, 3 * R), gh.view(B, T, 3 * R), H.view(B, T, R)
        dgi3, dgh3 = dgi.view(B, T, 3 * R), dgh.view(B, T, 3 * R)
        do3 = dout.view(B, T, -1)
        t_hi = T if t_hi is None else t_hi
        if self.fused_gru:
            # one launch per step: the product that carries the state gradient from step t to step t - 1 has the gate
            # gradients of step t - 1 as its epilogue.  Steps t_hi - 1 .. t_lo are therefore the launches t = t_hi .. t_lo + 1
            # (each reads dout of step t - 1 only, which may not exist yet below t_lo); the last step of the window has no
            # successor: its gate gradients come from the plain kernel
            if t_hi == T:
                hp = ws['h0'] if T == 1 else H3[:, T - 2]
                T_.gru_step_bwd(gi3[:, T - 1], gh3[:, T - 1], hp, do3[:, T - 1, :R], None, dgi3[:, T - 1], dgh3[:, T - 1], dhd)
            for t in range(min(t_hi, T - 1), t_lo, -1):
                hp = ws['h0'] if t == 1 else H3[:, t - 2]
                T_.gru_fused_bwd(gi3[:, t - 1], gh3[:, t - 1], hp, do3[:, t - 1, :R], dgh3[:, t], hh.lin.bwd, dgi3[:, t - 1],
                                 dgh3[:, t - 1], dhd, precision=self.precision)
            return
        for t in range(t_hi - 1, t_lo - 1, -1):
            hp = ws['h0'] if t == 0 else H3[:, t - 1]
            T_.gru_step_bwd(gi3[:, t], gh3[:, t], hp, do3[:, t, :R], None if t == T - 1 else dhr, dgi3[:, t], dgh3[:, t], dhd)
            if t > 0:
                ops.gemm(dgh3[:, t], hh.lin.bwd, dhr, precision=self.precision, residual=dhd, workspace=gemm_ws)

    CHUNK = None     # steps between hand-overs from one recurrence to the other; None: a quarter of the window (measured at
                     # 32 windows x 1000 steps: one chunk 57.6 ms per training step, 500: 47.8, 250: 42.8, 125: 44.1, 50: 46.3)

    def _chunks(self, T):
        # one chain of launches for both layers: the hand-over is one product in the same stream, so shorter chunks pay
        # (measured, 32 windows x 1000 steps: 25.0 ms per training step at 250, 23.3 at 125, 22.9 at 63, 23.3 at 32, 24.6 at 16)
        c = self.CHUNK or max(-(-T // (16 if self.fused_gru and self.one_chain else 4)), 1)
        return [(c0, min(c0 + c, T)) for c0 in range(0, T, c)]

    def _fwd_blocks(self, ws, gi, gh, H, hh, x_res, out, out_cols, t0, t1):
        """Argument blocks of the one-launch forward steps t0 .. t1 - 1 of a layer."""
        B, T, R = ws['B'], ws['T'], self.rnn_dims
        gi3, gh3, H3 = gi.view(B, T, 3 * R), gh.view(B, T, 3 * R), H.view(B, T, R)
        xr3, out3 = x_res.view(B, T, -1), out.view(B, T, -1)
        return [T_.gru_fused_fwd(gi3[:, t], gh3[:, t], ws['h0'] if t == 0 else H3[:, t - 1], hh.lin.fwd, H3[:, t],
                                 x_res=xr3[:, t, :out_cols], out=out3[:, t, :out_cols], precision=self.precision, launch=False)
                for t in range(t0, t1)]

    def _bwd_blocks(self, ws, gi, gh, H, hh, dout, dgi, dgh, dhd, t_hi, t_lo):
        """The launches of the backward steps t_hi - 1 .. t_lo of a layer (see _gru_bwd): a callable for the plain kernel of the
        window's last step, argument blocks for the one-launch steps."""
        B, T, R = ws['B'], ws['T'], self.rnn_dims
        gi3, gh3, H3 = gi.view(B, T, 3 * R), gh.view(B, T, 3 * R), H.view(B, T, R)
        dgi3, dgh3, do3 = dgi.view(B, T, 3 * R), dgh.view(B, T, 3 * R), dout.view(B, T, -1)
        blocks = []
        if t_hi == T:
            hp = ws['h0'] if T == 1 else H3[:, T - 2]
            blocks.append(lambda: T_.gru_step_bwd(gi3[:, T - 1], gh3[:, T - 1], hp, do3[:, T - 1, :R], None, dgi3[:, T - 1],
                                                  dgh3[:, T - 1], dhd))
        for t in range(min(t_hi, T - 1), t_lo, -1):
            blocks.append(T_.gru_fused_bwd(gi3[:, t - 1], gh3[:, t - 1], ws['h0'] if t == 1 else H3[:, t - 2], do3[:, t - 1, :R],
                                           dgh3[:, t], hh.lin.bwd, dgi3[:, t - 1], dgh3[:, t - 1], dhd, precision=self.precision,
                                           launch=False))
        return blocks

    @staticmethod
    def _issue(first, second, forward):
        """Two lists of launches side by side: entry i of both in one launch where both are one-launch GRU steps."""
        for i in range(max(len(first), len(second))):
            a = first[i] if i < len(first) else None
            b = second[i] if i < len(second) else None
            if callable(a):
                a()
                a = None
            if callable(b):
                b()
                b = None
            if a is not None or b is not None:
                T_.gru_fused_pair(a, b, forward)

    def _gru12_fwd_one_chain(self, ws):
        """Both GRU layers as ONE chain of launches: rnn2 runs a chunk of steps behind rnn1 (its input gates come from one
        product per chunk over rnn1's outputs), and step t of rnn1 shares its launch with step t - chunk of rnn2.  Two chains side
        by side on two streams take about as long as one and a quarter; this takes one and one chunk."""
        B, T, R, d = ws['B'], ws['T'], self.rnn_dims, self.aux_dims
        x1 = ws['x1cat'].view(B, T, R + d)
        g2 = ws['gi2'].view(B, T, 3 * R)
        chunks = self._chunks(T)
        for k in range(len(chunks) + 1):
            one = self._fwd_blocks(ws, ws['gi1'], ws['gh1'], ws['H1'], self.hh1, ws['xI'], ws['x1cat'], R, *chunks[k]) if k < len(chunks) else []
            two = self._fwd_blocks(ws, ws['gi2'], ws['gh2'], ws['H2'], self.hh2, ws['x1cat'], ws['x2cat'], R, *chunks[k - 1]) if k else []
            self._issue(one, two, True)
            if k < len(chunks):
                c0, c1 = chunks[k]
                ops.gemm(x1[0, c0:c1], self.ih2.lin.fwd, g2[0, c0:c1], precision=self.precision, split=False,
                         batch=(B, T * (R + d), 0, T * 3 * R, 0))

    def _gru21_bwd_one_chain(self, ws):
        """The backward recurrences as one chain, from the last step down: rnn1 a chunk behind rnn2 (the gradient of rnn1's
        outputs comes from one product per chunk over rnn2's input-gate gradients plus the residual path)."""
        B, T, R, d = ws['B'], ws['T'], self.rnn_dims, self.aux_dims
        dg2 = ws['dgi'].view(B, T, 3 * R)
        dx1 = ws['dx1cat'].view(B, T, R + d)
        res = ws['res2'].view(B, T, R + d)
        chunks = list(reversed(self._chunks(T)))
        for k in range(len(chunks) + 1):
            two = self._bwd_blocks(ws, ws['gi2'], ws['gh2'], ws['H2'], self.hh2, ws['dx2cat'], ws['dgi'], ws['dgh'], ws['dhd'],
                                   chunks[k][1], chunks[k][0]) if k < len(chunks) else []
            one = self._bwd_blocks(ws, ws['gi1'], ws['gh1'], ws['H1'], self.hh1, ws['dx1cat'], ws['dgi1'], ws['dgh1'], ws['dhd1'],
                                   chunks[k - 1][1], chunks[k - 1][0]) if k else []
            self._issue(two, one, False)
            if k < len(chunks):
                c0, c1 = chunks[k]
                ops.gemm(dg2[0, c0:c1], self.ih2.lin.bwd, dx1[0, c0:c1], precision=self.precision, split=False,
                         residual=res[0, c0:c1], batch=(B, T * 3 * R, 0, T * (R + d), 0))

    def _gru12_fwd(self, ws):
        """Both GRU layers of the window as two chains that overlap: rnn2 only needs rnn1's outputs up to its own step, so
        after every CHUNK steps of rnn1 (this stream) a second stream computes rnn2's input gates for those steps (one product
        batched over the windows) and runs rnn2 over them while rnn1 goes on.  The chains are latency-bound (32 rows per
        product), so side by side they take little more than one of them."""
        if self.fused_gru and self.one_chain:
            return self._gru12_fwd_one_chain(ws)
        B, T, R, d = ws['B'], ws['T'], self.rnn_dims, self.aux_dims
        cur, sb = torch.cuda.current_stream(), ws['stream_b']
        x1 = ws['x1cat'].view(B, T, R + d)
        g2 = ws['gi2'].view(B, T, 3 * R)
        for c0, c1 in self._chunks(T):
            self._gru_fwd(ws, ws['gi1'], ws['gh1'], ws['H1'], self.hh1, ws['xI'], ws['x1cat'], R, c0, c1)
            sb.wait_stream(cur)
            with torch.cuda.stream(sb):
                ops.gemm(x1[0, c0:c1], self.ih2.lin.fwd, g2[0, c0:c1], precision=self.precision, split=False,
                         batch=(B, T * (R + d), 0, T * 3 * R, 0))
                self._gru_fwd(ws, ws['gi2'], ws['gh2'], ws['H2'], self.hh2, ws['x1cat'], ws['x2cat'], R, c0, c1,
                              gemm_ws=ws['gemm_ws_b'])
        cur.wait_stream(sb)

    def _gru21_bwd(self, ws):
        """The backward recurrences the same way, from the last step down: rnn2 on this stream; after every CHUNK steps the
        second stream turns rnn2's input-gate gradients of those steps into the gradient of rnn1's output (product with
        W_ih2 plus the residual path, ws['res2']) and runs rnn1's backward recurrence over them."""
        if self.fused_gru and self.one_chain:
            return self._gru21_bwd_one_chain(ws)
        B, T, R, d = ws['B'], ws['T'], self.rnn_dims, self.aux_dims
        cur, sb = torch.cuda.current_stream(), ws['stream_b']
        dg2 = ws['dgi'].view(B, T, 3 * R)
        dx1 = ws['dx1cat'].view(B, T, R + d)
        res = ws['res2'].view(B, T, R + d)
        for c0, c1 in reversed(self._chunks(T)):
            self._gru_bwd(ws, ws['gi2'], ws['gh2'], ws['H2'], self.hh2, ws['dx2cat'], ws['dgi'], ws['dgh'], ws['dhd'], ws['dhr'],
                          c1, c0)
            sb.wait_stream(cur)
            with torch.cuda.stream(sb):
                ops.gemm(dg2[0, c0:c1], self.ih2.lin.bwd, dx1[0, c0:c1], precision=self.precision, split=False,
                         residual=res[0, c0:c1], batch=(B, T * 3 * R, 0, T * (R + d), 0))
                self._gru_bwd(ws, ws['gi1'], ws['gh1'], ws['H1'], self.hh1, ws['dx1cat'], ws['dgi1'], ws['dgh1'], ws['dhd1'],
                              ws['dhr1'], c1, c0, gemm_ws=ws['gemm_ws_b'])
        cur.wait_stream(sb)

    # ------------------------------------------------------------------ pieces of the step
    def _bn_gemm(self, x, conv, bn, z):
        """z = conv(x) with the moments of z summed by the same call, about the batch mean of the layer's previous step (zero
        before the first); returns the statistics of _bn_train."""
        C = conv.lin.fwd.N
        if not hasattr(bn, 'shift_est'):
            bn.shift_est = torch.zeros(C, dtype=torch.float32, device=self.device)
        ptr, stride = T_.SumSlots.on(self.device).forward_region(C)
        ops.gemm(x, conv.lin.fwd, z, precision=self.precision, stats=(ptr, bn.shift_est))
        return self._bn_train(z, bn, z.shape[0], C, summed=(ptr, stride))

    def _bn_train(self, z, bn, M, C, summed=None):
        """(mean, rstd, scale, shift) from batch statistics; moving statistics updated (torch.nn.BatchNorm1d: momentum 0.1,
        unbiased moving variance).  summed: (pointer, stride) of moments already taken about bn.shift_est."""
        st = torch.empty(4, C, dtype=torch.float32, device=self.device)
        about = about_next = getattr(bn, 'shift_est', None)
        if summed is None:
            summed, about, about_next = T_.colreduce(0, z=z, M=M, C=C, shift=z[0], defer=True), z, None
        T_.bn_finalize(None, None, about, M, C, bn.gamma, bn.beta, bn.mm, bn.mv, st[0], st[1], st[2], st[3], eps=1e-5,
                       momentum=0.9, moving_var_scale=M / max(M - 1, 1), slots=summed, about_next=about_next)
        return st

    def _lin_bwd(self, x, L, dy, M, dx=None, residual=None):
        T_.wgrad(x, dy, L.dk, M=M, precision=self.precision)
        if L.db is not None:
            T_.colreduce(1, L.db, g=dy, M=M, C=L.N)
        if dx is not None:
            ops.gemm(dy, L.lin.bwd, dx, M=M, precision=self.precision, residual=residual)
        return dx

    def _resnet_fwd(self, ws, tape):
        """MelResNet in training mode (fatchord_version.py:39-45 with BatchNorm batch statistics)."""
        Mr, C, pr = ws['Mr'], self.compute_dims, self.precision
        e = lambda *s: torch.empty(*s, dtype=torch.float32, device=self.device)   # noqa: E731
        z = e(Mr, C)
        st0 = self._bn_gemm(ws['unf'], self.conv_in, self.bn_in, z)
        x = e(Mr, C)
        T_.affine_act(z, x, st0[2], st0[3], 'relu')
        saved = []
        for c1, b1, c2, b2 in self.blocks:
            z1 = e(Mr, C)
            s1 = self._bn_gemm(x, c1, b1, z1)
            a = e(Mr, C)
            T_.affine_act(z1, a, s1[2], s1[3], 'relu')
            z2 = e(Mr, C)
            s2 = self._bn_gemm(a, c2, b2, z2)
            nx = e(Mr, C)
            T_.affine_act(z2, nx, s2[2], s2[3], None, residual=x)
            saved.append((x, z1, s1, a, z2, s2))
            x = nx
        ops.gemm(x, self.conv_out.lin.fwd, ws['aux'], precision=pr)
        x_last = x

        def backward(d_aux):
            dx = e(Mr, C)
            self._lin_bwd(x_last, self.conv_out, d_aux, Mr, dx=dx)
            for (c1, b1, c2, b2), (xin, z1, s1, a, z2, s2) in zip(reversed(self.blocks), reversed(saved)):
                dz2 = e(Mr, C)
                sl = T_.colreduce(2, z=z2, g=dx, mean=s2[0], rstd=s2[1], M=Mr, C=C, defer=True)
                T_.bn_bwd(dx, z2, s2[0], s2[1], b2.gamma, b2.dgamma, b2.dbeta, dz2, M=Mr, C=C, slots=sl)
                da = e(Mr, C)
                self._lin_bwd(a, c2, dz2, Mr, dx=da)
                dz1 = e(Mr, C)
                sl = T_.colreduce(2, z=z1, g=da, a=a, act='relu', mean=s1[0], rstd=s1[1], dy_out=dz1, M=Mr, C=C, defer=True)
                T_.bn_bwd(dz1, z1, s1[0], s1[1], b1.gamma, b1.dgamma, b1.dbeta, dz1, M=Mr, C=C, slots=sl)
                ndx = e(Mr, C)
                self._lin_bwd(xin, c1, dz1, Mr, dx=ndx, residual=dx)
                dx = ndx
            dz = e(Mr, C)
            x0 = saved[0][0] if saved else x_last
            sl = T_.colreduce(2, z=z, g=dx, a=x0, act='relu', mean=st0[0], rstd=st0[1], dy_out=dz, M=Mr, C=C, defer=True)
            T_.bn_bwd(dz, z, st0[0], st0[1], self.bn_in.gamma, self.bn_in.dgamma, self.bn_in.dbeta, dz, M=Mr, C=C, slots=sl)
            T_.wgrad(ws['unf'], dz, self.conv_in.dk, M=Mr, precision=self.precision)

        tape.append(backward)

    def _load_batch(self, ws, x, mels):
        B, Tf, hop = ws['B'], ws['Tf'], self.hop_length
        xd = torch.as_tensor(x, dtype=torch.float32, device=self.device)
        md = torch.as_tensor(mels, dtype=torch.float32, device=self.device)
        if tuple(xd.shape) != (B, Tf * hop) or tuple(md.shape) != (B, self.feat_dims, Tf + 2 * self.pad):
            raise ValueError(f'expected x {(B, Tf * hop)} and mels {(B, self.feat_dims, Tf + 2 * self.pad)}, got '
                             f'{tuple(xd.shape)} and {tuple(md.shape)}')
        rows = md.transpose(1, 2).contiguous()                                  # [B, Tm, feat]
        ws['mel'].copy_(rows.view(-1, self.feat_dims))
        # valid convolution of conv_in as a dense layer over unfolded windows (pure data movement)
        unf = rows.unfold(1, self.k_conv, 1).permute(0, 1, 3, 2).reshape(B * Tf, self.k_conv * self.feat_dims)
        ws['unf'].copy_(unf)
        ws['x'].copy_(xd.reshape(-1))

    def _forward(self, ws, tape):
        R, F, pr, M = self.rnn_dims, self.fc_dims, self.precision, ws['M']
        self._resnet_fwd(ws, tape)
        a = WavernnPrepArgs()
        a.mel, a.mel_off, a.aux, a.aux_off = ws['mel'].data_ptr(), ws['mel_off'].data_ptr(), ws['aux'].data_ptr(), ws['aux_off'].data_ptr()
        a.fir, a.ntaps, a.fir_kmin = self.fir.data_ptr(), self.ntaps, self.fir_kmin
        a.seq_utt, a.seq_start, a.utt_samples = ws['seq_utt'].data_ptr(), ws['seq_start'].data_ptr(), ws['utt_samples'].data_ptr()
        a.x_prev, a.step = ws['x'].data_ptr(), ws['step0'].data_ptr()
        a.xin, a.ld_xin = ws['xin'].data_ptr(), ws['xin'].stride(0)
        a.a2, a.ld_a2 = ws['x1cat'][:, R:].data_ptr(), ws['x1cat'].stride(0)
        a.a3, a.ld_a3 = ws['x2cat'][:, R:].data_ptr(), ws['x2cat'].stride(0)
        a.a4, a.ld_a4 = ws['f1cat'][:, F:].data_ptr(), ws['f1cat'].stride(0)
        a.n_seq, a.feat, a.d, a.hop, a.pad = M, self.feat_dims, self.aux_dims, self.hop_length, self.pad
        check(_lib.load().estts_wavernn_prep(a, ops._stream()), 'estts_wavernn_prep')
        ops.gemm(ws['xin'], self.I.lin.fwd, ws['xI'], precision=pr)
        ops.gemm(ws['xI'], self.ih1.lin.fwd, ws['gi1'], precision=pr)
        self._run(ws, 'gru12_fwd', lambda: self._gru12_fwd(ws))
        ops.gemm(ws['x2cat'], self.fc1.lin.fwd, ws['f1cat'][:, :F], precision=pr)
        ops.gemm(ws['f1cat'], self.fc2.lin.fwd, ws['f2'], precision=pr)
        ops.gemm(ws['f2'], self.fc3.lin.fwd, ws['logits'], precision=pr)

    def forward(self, x, mels):
        """Teacher-forced logits in training mode (fatchord_version.py:119-148): [B, L, n_classes] on the device."""
        B, Tf = int(np.shape(x)[0]), int(np.shape(mels)[2]) - 2 * self.pad
        ws = self._workspace(B, Tf)
        self._load_batch(ws, x, mels)
        self._forward(ws, [])
        self.step += 1
        return ws['logits'].view(B, ws['T'], self.nc_pad)[:, :, :self.n_classes]

    def train_step(self, x, y, mels, lr=None):
        """One iteration of voc_train_loop (train_wavernn.py:27-42).  x [B, L] float inputs, y [B, L] targets (int class
        labels for RAW, floats in [-1, 1] for MOL), mels [B, feat, L / hop + 2 pad].  Returns the loss (device scalar)."""
        if lr is not None:
            self.lr = float(lr)
        B, Tf = int(np.shape(x)[0]), int(np.shape(mels)[2]) - 2 * self.pad
        ws = self._workspace(B, Tf)
        M, Mr, R, F, d, pr = ws['M'], ws['Mr'], self.rnn_dims, self.fc_dims, self.aux_dims, self.precision
        self._load_batch(ws, x, mels)
        T_.SumSlots.step_begin(self.device, self)     # (clears sums a failed step may have left without a consumer)
        st = self.store
        st.grad.zero_()
        ws['loss'].zero_()
        ws['dfir'].zero_()
        tape = []
        self._forward(ws, tape)
        self.step += 1
        # ---- loss (mean over B * L) and its gradient
        if self.mode == 'RAW':
            ws['labels'].copy_(torch.as_tensor(y, device=self.device).reshape(-1))
            T_.softmax_xent(ws['logits'], ws['labels'], ws['loss'], ws['dlogits'], 1.0 / M, M=M, C=self.n_classes)
        else:
            ws['yf'].copy_(torch.as_tensor(y, device=self.device).reshape(-1))
            T_.mol_loss(ws['logits'], ws['yf'], ws['loss'], ws['dlogits'], 1.0 / M, self.n_classes // 3, M=M)
        # ---- backward
        self._lin_bwd(ws['f2'], self.fc3, ws['dlogits'], M, dx=ws['df2'])
        T_.colreduce(1, self.fc2.db, g=ws['df2'], a=ws['f2'], act='relu', dy_out=ws['df2'], M=M, C=F)
        T_.wgrad(ws['f1cat'], ws['df2'], self.fc2.dk, M=M, precision=pr)
        ops.gemm(ws['df2'], self.fc2.lin.bwd, ws['df1cat'], precision=pr)
        g1 = ws['df1cat'][:, :F]
        T_.colreduce(1, self.fc1.db, g=g1, a=ws['f1cat'][:, :F], act='relu', dy_out=g1, M=M, C=F)
        T_.wgrad(ws['x2cat'], g1, self.fc1.dk, M=M, precision=pr)
        ops.gemm(g1, self.fc1.lin.bwd, ws['dx2cat'], precision=pr)
        # x2cat[:, :R] = H2 + x1: the GRU output gradient is dx2cat[:, :R]; the residual path joins dx1cat below
        # (dx1cat = dgi2 W_ih2^T + [dx2cat[:, :R] | 0]; the second term, a copy, is ws['res2'], whose aux columns stay zero)
        ws['res2'][:, :R].copy_(ws['dx2cat'][:, :R])
        self._run(ws, 'gru21_bwd', lambda: self._gru21_bwd(ws))
        self._gru_param_grads(ws, ws['x1cat'], ws['H2'], self.ih2, self.hh2, ws['dgi'], ws['dgh'])
        self._gru_param_grads(ws, ws['xI'], ws['H1'], self.ih1, self.hh1, ws['dgi1'], ws['dgh1'])
        ops.gemm(ws['dgi1'], self.ih1.lin.bwd, ws['dxI'], precision=pr, residual=ws['dx1cat'][:, :R])
        self._lin_bwd(ws['xin'], self.I, ws['dxI'], M, dx=ws['dxin'])
        T_.wavernn_prep_bwd(ws['dxin'], ws['dx1cat'][:, R:], ws['dx2cat'][:, R:], ws['df1cat'][:, F:], ws['mel'], ws['mel_off'],
                            ws['d_aux'], ws['dfir'], self.fir_kmin, ws['B'], ws['Tf'], self.feat_dims, d, self.hop_length, self.pad)
        tape[0](ws['d_aux'])
        # three upsampling kernels: chain rule through the composite filter on the host
        ups = composite_filter_gradients(self._up_host, self.upsample_factors, ws['dfir'].cpu().numpy())
        for n, g in zip(self._up_names, ups):
            st.g[n].copy_(torch.from_numpy(g.astype(np.float32)))
        self._apply_gradients()
        T_.SumSlots.step_end(self.device)
        return ws['loss'][0]

    def _gru_param_grads(self, ws, x_in, H, ih, hh, dgi, dgh):
        M, T = ws['M'], ws['T']
        T_.wgrad(x_in, dgi, ih.dk, M=M, precision=self.precision, db=ih.db)
        # h_{t-1} is H shifted by one step inside each window: a one-tap "convolution" with one row of left padding
        T_.wgrad(H, dgh, hh.dk, taps=1, lpad=1, M=M, seg_len=T, precision=self.precision, db=hh.db)

    def _apply_gradients(self):
        st = self.store
        if self.world > 1:
            parallel.allreduce_buckets(st.grad, self.bucket_elems, self.pg, self._comm)
        self.iterations += 1
        t = self.iterations
        bc2 = math.sqrt(1.0 - self.beta2 ** t)
        # torch: p -= lr / (1 - b1^t) * m / (sqrt(v) / sqrt(1 - b2^t) + eps)  ==  lr_t * m / (sqrt(v) + eps * sqrt(1 - b2^t))
        T_.adam(st.flat, st.grad, st.m, st.v, self.lr * bc2 / (1.0 - self.beta1 ** t), self.beta1, self.beta2, self.eps * bc2,
                1.0 / self.world)
        if getattr(self, '_pack_plan', None) is None:
            self._pack_plan = T_.PackPlan(self._linears, self.device)
        self._pack_plan.run()
        self._refresh_fir()

