"""MINE and CLUB mutual-information heads of the reference (TransformerTTS/model/MINE.py:15-200 over the statistics networks of
TransformerTTS/model/layers.py:471-523) on MI355X, with the update `train_models_step` applies to them
(TransformerTTS/model/traning_steps.py:58-84): the heads see the text / style / speaker embeddings as constants, estimate the
mutual information of a pair, and ascend it with their own Adam optimizer.

The dense stacks run on estts_gemm / estts_gemm_wgrad like every other layer; the estimators and their gradients are one
single-workgroup launch each (estts_mine_divergence, estts_club_terms).  torch supplies memory, row gathers and
concatenation.  The random choices (which character, the batch shuffles) come from a seeded host generator and are kept in
`last_draws`, so a checker can repeat them.

Variables are created at the first call, when the input widths are known (as Keras builds them), Glorot-uniform kernels and
zero biases from `seed`; `weights()` / `load_weights()` exchange them by name.
"""
import math

import numpy as np
import torch

from . import ops, train_ops as T
from .forward_training import ParamStore
from .train_ops import TrainLinear


class _DenseStack:
    """Dense(h) + ReLU for every h in `hidden`, then Dense(out) (+ tanh): MineNetFirstOrder / CLUBNet (layers.py:471-485,
    507-523).  The last layer is stored with its outputs padded to a multiple of 4 (the data gradient needs that)."""

    def __init__(self, store_items, prefix, in_dim, hidden, out_dim, out_act, rng):
        self.prefix, self.dims, self.out_dim, self.out_act = prefix, [in_dim] + list(hidden), out_dim, out_act
        self.out_pad = (out_dim + 3) // 4 * 4
        for i, (a, b) in enumerate(zip(self.dims[:-1], self.dims[1:])):
            lim = math.sqrt(6.0 / (a + b))
            store_items.append((f'{prefix}.fc{i}.kernel', rng.uniform(-lim, lim, size=(a, b)).astype(np.float32)))
            store_items.append((f'{prefix}.fc{i}.bias', np.zeros(b, np.float32)))
        a = self.dims[-1]
        lim = math.sqrt(6.0 / (a + out_dim))
        k = np.zeros((a, self.out_pad), np.float32)
        k[:, :out_dim] = rng.uniform(-lim, lim, size=(a, out_dim))
        store_items.append((f'{prefix}.fc_proj.kernel', k))
        store_items.append((f'{prefix}.fc_proj.bias', np.zeros(self.out_pad, np.float32)))

    def bind(self, store):
        self.layers = []
        names = [f'{self.prefix}.fc{i}' for i in range(len(self.dims) - 1)] + [self.prefix + '.fc_proj']
        for j, n in enumerate(names):
            last = j == len(names) - 1
            lin = TrainLinear(store.p[n + '.kernel'], store.p[n + '.bias'], act=(self.out_act if last else 'relu'))
            self.layers.append((lin, store.g[n + '.kernel'], store.g[n + '.bias']))
        return [l[0] for l in self.layers]

    def forward(self, x, precision):
        """x [M, in] -> (out [M, out_pad], saved activations)."""
        acts = [x]
        for lin, _, _ in self.layers:
            y = torch.empty(x.shape[0], lin.N, dtype=torch.float32, device=x.device)
            ops.gemm(acts[-1], lin.fwd, y, precision=precision)
            acts.append(y)
        return acts[-1], acts

    def backward(self, acts, dout, precision):
        """Accumulates the kernel / bias gradients; dout [M, out_pad] is the gradient of the (activated) output."""
        g = dout
        M = g.shape[0]
        for j in reversed(range(len(self.layers))):
            lin, dk, db = self.layers[j]
            act = self.out_act if j == len(self.layers) - 1 else 'relu'
            dz = torch.empty_like(g)
            if act is None:
                dz = g
                T.colreduce(1, db, g=g, M=M, C=lin.N)
            else:
                T.colreduce(1, db, g=g, a=acts[j + 1], act=act, dy_out=dz, M=M, C=lin.N)
            T.wgrad(acts[j], dz, dk, M=M, precision=precision)
            if j > 0:
                g = torch.empty(M, lin.K, dtype=torch.float32, device=dz.device)
                ops.gemm(dz, lin.bwd, g, M=M, precision=precision)


class _Head:
    """What MINE and CLUB share: lazily built variables in a flat store, Adam, the random draws."""

    def __init__(self, pair_type, device, precision, seed, name):
        if pair_type not in ('style_text', 'style_speaker', 'text_speaker', 'style_text_speaker'):
            raise ValueError(f'pair_type {pair_type!r} is not supported')
        if not torch.cuda.is_available():
            raise RuntimeError('the mutual-information heads need a GPU: the hot path has no CPU implementation')
        ops._lib.load()
        self.pair_type, self.device, self.precision, self.name = pair_type, torch.device(device), precision, name
        self.seed = int(seed)
        self._rng = np.random.default_rng(self.seed)
        self.store = None
        self.learning_rate, self.beta1, self.beta2, self.eps = 1e-3, 0.9, 0.999, 1e-7     # tf.keras.optimizers.Adam defaults
        self.iterations = 0
        self.last_draws = None
        self._pending = None
        self._pending_opt = None

    @property
    def step(self):
        return self.iterations

    def _compile(self, optimizer=None):
        """`optimizer`: dict with learning_rate / beta_1 / beta_2 / epsilon (the reference compiles every head with
        Adam(learning_rate_mine_schedule[0][1], beta_1=0.9, beta_2=0.98, epsilon=1e-9), train_autoregressive.py:97-101)."""
        if isinstance(optimizer, dict):
            self.learning_rate = float(optimizer.get('learning_rate', self.learning_rate))
            self.beta1 = float(optimizer.get('beta_1', self.beta1))
            self.beta2 = float(optimizer.get('beta_2', self.beta2))
            self.eps = float(optimizer.get('epsilon', self.eps))

    def _finish_build(self, items, stacks):
        st = self.store = ParamStore(self.device)
        for n, a in items:
            st.add(n, a)
        st.build()
        if self._pending is not None:
            self.load_weights(self._pending)
            self._pending = None
        if self._pending_opt is not None:
            self.load_optimizer_state(self._pending_opt)
            self._pending_opt = None
        self._linears = []
        for s in stacks:
            self._linears += s.bind(st)
        self._pack_plan = T.PackPlan(self._linears, self.device)

    def weights(self):
        if self.store is None:
            raise RuntimeError('the head has no variables before its first call')
        out = {}
        for k, v in self.store.numpy().items():
            stack = self._stack_of(k)
            if k.endswith('fc_proj.kernel'):
                v = v[:, :stack.out_dim]
            elif k.endswith('fc_proj.bias'):
                v = v[:stack.out_dim]
            out[k] = v.copy()
        return out

    def gradients(self):
        out = {}
        for k, v in self.store.numpy(self.store.grad).items():
            stack = self._stack_of(k)
            out[k] = (v[:, :stack.out_dim] if k.endswith('fc_proj.kernel') else v[:stack.out_dim] if k.endswith('fc_proj.bias') else v).copy()
        return out

    def load_weights(self, w):
        """Variables by name (weights() layout); before the first call they are kept until the variables exist."""
        if self.store is None:
            self._pending = dict(w)
            return
        for k, v in w.items():
            dst = self.store.p[k]
            v = torch.as_tensor(np.asarray(v, np.float32), device=self.device)
            if dst.dim() == 2:
                dst[:, :v.shape[1]].copy_(v)
            else:
                dst[:v.shape[0]].copy_(v)
        if getattr(self, '_linears', None):
            self._pack_plan.run()

    def optimizer_state(self):
        """Adam's state for a checkpoint (the reference checkpoints every head with its optimizer,
        train_autoregressive.py:119-131): moments in the flat store's order, the step count, and the state of the generator
        behind the head's random pairings, so that a resumed run draws what the uninterrupted one would have."""
        if self.store is None:
            raise RuntimeError('the head has no variables before its first call')
        st = self.store
        rng = np.frombuffer(repr(self._rng.bit_generator.state).encode(), dtype=np.uint8)
        return {'__adam_m__': st.m.cpu().numpy(), '__adam_v__': st.v.cpu().numpy(),
                '__iterations__': np.asarray(self.iterations, np.int64), '__rng__': rng}

    def load_optimizer_state(self, state):
        """Before the first call (the variables are built lazily from the first batch's widths) the state is kept until they
        exist, like load_weights()."""
        if self.store is None:
            self._pending_opt = dict(state)
            self.iterations = int(state['__iterations__'])
            return
        st = self.store
        m, v = np.asarray(state['__adam_m__'], np.float32), np.asarray(state['__adam_v__'], np.float32)
        if m.shape != tuple(st.m.shape) or v.shape != tuple(st.v.shape):
            raise ValueError(f'optimizer state of {m.shape[0]} elements does not fit this head ({st.m.shape[0]})')
        st.m.copy_(torch.from_numpy(m))
        st.v.copy_(torch.from_numpy(v))
        self.iterations = int(state['__iterations__'])
        if '__rng__' in state:
            import ast
            self._rng.bit_generator.state = ast.literal_eval(bytes(np.asarray(state['__rng__'], np.uint8)).decode())

    def _apply(self):
        st = self.store
        self.iterations += 1
        t = self.iterations
        lr_t = self.learning_rate * math.sqrt(1.0 - self.beta2 ** t) / (1.0 - self.beta1 ** t)
        T.adam(st.flat, st.grad, st.m, st.v, lr_t, self.beta1, self.beta2, self.eps, 1.0)
        self._pack_plan.run()

    def _draw(self, text_embed, speaker_embed, draws):
        """The reference's random choices: one character position for the whole batch (tf.random.shuffle(range(N))[:1]), a
        batch permutation for the shuffled text and another for the shuffled speakers."""
        B, N = text_embed.shape[0], text_embed.shape[1]
        if draws is None:
            draws = {'char': int(self._rng.permutation(N)[0]), 'text_perm': self._rng.permutation(B),
                     'speaker_perm': self._rng.permutation(B)}
        self.last_draws = draws
        dev = self.device
        text = text_embed[:, draws['char'], :].contiguous()
        text_sh = text[torch.as_tensor(draws['text_perm'], device=dev)]
        spk = spk_sh = None
        if speaker_embed is not None:
            spk = torch.as_tensor(speaker_embed, dtype=torch.float32, device=dev).reshape(B, -1)
            spk_sh = spk[torch.as_tensor(draws['speaker_perm'], device=dev)]
        return text, text_sh, spk, spk_sh


class MINE(_Head):
    """MINE(conv_filters, conv_kernel, dense_hidden_units, beta_values, divergence_type, pair_type) (MINE.py:15-37) with the
    first-order statistics network the reference instantiates (MineNetFirstOrder; conv_filters / conv_kernel belong to the
    second-order variant it leaves commented out and are accepted and ignored)."""

    def __init__(self, conv_filters=None, conv_kernel=None, dense_hidden_units=(512, 64), beta_values=(0, 0.5, 1),
                 divergence_type='KL', pair_type='style_text', device='cuda', precision='bf16x3', seed=0, name='MINE'):
        super().__init__(pair_type, device, precision, seed, name)
        if divergence_type not in ('KL', 'reyni'):
            raise ValueError("divergence_type must be 'KL' or 'reyni'")
        self.hidden, self.beta, self.div_type = list(dense_hidden_units), [float(b) for b in beta_values], divergence_type
        self.betas_d = torch.tensor(self.beta, dtype=torch.float32, device=self.device)

    def _stack_of(self, name):
        return self.net

    def _pairs(self, text_embed, style_embed, speaker_embed, draws):
        text, text_sh, spk, spk_sh = self._draw(text_embed, speaker_embed, draws)
        style = None if style_embed is None else style_embed.reshape(style_embed.shape[0], -1)
        pt = self.pair_type
        if pt == 'style_text':
            j, m = [style, text], [style, text_sh]
        elif pt == 'style_speaker':
            j, m = [style, spk], [style, spk_sh]
        elif pt == 'text_speaker':
            j, m = [text, spk], [text, spk_sh]
        else:
            j, m = [style, text, spk], [style, text_sh, spk_sh]
        if any(t is None for t in j):
            raise ValueError(f"pair_type '{pt}' needs embeddings this system type does not produce")
        return torch.cat(j, dim=1).contiguous(), torch.cat(m, dim=1).contiguous()

    def call(self, text_embed, style_embed, speaker_embed, exp_terms_holder, draws=None, train=False):
        """MINE.call (MINE.py:97-149): (mi, exp_terms).  text_embed [B, N, d], style_embed [B, 1, sd], speaker_embed
        [B, 1, ds] device tensors (constants); exp_terms_holder: the reference's mi_holder (smoothing_factor, exp_terms
        [n_beta, 2]).  train=True also ascends mi by one Adam step (traning_steps.py:76-81)."""
        joint, marginal = self._pairs(text_embed, style_embed, speaker_embed, draws)
        B = joint.shape[0]
        if self.store is None:
            items = []
            self.net = _DenseStack(items, 'mine_net', joint.shape[1], self.hidden, 1, None, np.random.default_rng(self.seed + 1))
            self._finish_build(items, [self.net])
        x = torch.cat([joint, marginal], dim=0)                       # one pass of the shared network over both samples
        scores, acts = self.net.forward(x, self.precision)
        s = scores[:, 0].contiguous()
        prev = torch.as_tensor(exp_terms_holder['exp_terms'], dtype=torch.float32, device=self.device).contiguous()
        n_rows = 1 if self.div_type == 'KL' else len(self.beta)
        if prev.numel() < 2 * n_rows:
            raise ValueError(f'exp_terms must hold {n_rows} rows of 2 values')
        nxt = torch.zeros(n_rows, 2, dtype=torch.float32, device=self.device)
        mi = torch.zeros(1, dtype=torch.float32, device=self.device)
        ds = torch.empty(2 * B, dtype=torch.float32, device=self.device)
        ops.check(ops._lib.load().estts_mine_divergence(
            s.data_ptr(), s[B:].data_ptr(), B, 0 if self.div_type == 'KL' else 1, self.betas_d.data_ptr(), len(self.beta),
            float(exp_terms_holder['smoothing_factor']), prev.data_ptr(), nxt.data_ptr(), mi.data_ptr(), ds.data_ptr(),
            ds[B:].data_ptr(), -1.0, ops._stream()), 'estts_mine_divergence')
        if train:
            self.store.clear_gradients()
            dout = torch.zeros(2 * B, self.net.out_pad, dtype=torch.float32, device=self.device)
            dout[:, 0].copy_(ds)
            self.net.backward(acts, dout, self.precision)
            self._apply()
        return mi[0], nxt


class CLUB(_Head):
    """CLUB(dense_hidden_units, pair_type) (MINE.py:152-200): two CLUBNets (mean and tanh-bounded log-variance, 256 outputs
    each, layers.py:507-523); call returns (lld, bound) and training ascends lld."""

    OUT = 256

    def __init__(self, dense_hidden_units=(512, 64), pair_type='style_text', device='cuda', precision='bf16x3', seed=0, name='CLUB'):
        super().__init__(pair_type, device, precision, seed, name)
        if pair_type == 'style_text_speaker':
            raise ValueError("CLUB has no 'style_text_speaker' pair")
        self.hidden = list(dense_hidden_units)

    def _stack_of(self, name):
        return self.net_mu if name.startswith('net_mu') else self.net_log_var

    def call(self, text_embed, style_embed, speaker_embed, exp_terms_holder=None, draws=None, train=False):
        text, text_sh, spk, spk_sh = self._draw(text_embed, speaker_embed, draws)
        style = None if style_embed is None else style_embed.reshape(style_embed.shape[0], -1)
        x, y, ys = {'style_text': (style, text, text_sh), 'style_speaker': (style, spk, spk_sh),
                    'text_speaker': (text, spk, spk_sh)}[self.pair_type]
        if x is None or y is None:
            raise ValueError(f"pair_type '{self.pair_type}' needs embeddings this system type does not produce")
        if y.shape[1] != self.OUT:
            raise ValueError(f'CLUB predicts {self.OUT} values: the target embedding has {y.shape[1]}')
        x, y, ys = x.contiguous(), y.contiguous(), ys.contiguous()
        B = x.shape[0]
        if self.store is None:
            items, rng = [], np.random.default_rng(self.seed + 1)
            self.net_mu = _DenseStack(items, 'net_mu', x.shape[1], self.hidden, self.OUT, None, rng)
            self.net_log_var = _DenseStack(items, 'net_log_var', x.shape[1], self.hidden, self.OUT, 'tanh', rng)
            self._finish_build(items, [self.net_mu, self.net_log_var])
        mu, acts_mu = self.net_mu.forward(x, self.precision)
        lv, acts_lv = self.net_log_var.forward(x, self.precision)
        out = torch.zeros(2, dtype=torch.float32, device=self.device)
        dmu, dlv = torch.empty_like(mu), torch.empty_like(lv)
        ops.check(ops._lib.load().estts_club_terms(mu.data_ptr(), lv.data_ptr(), y.data_ptr(), ys.data_ptr(), B, self.OUT, out.data_ptr(),
                                                   dmu.data_ptr(), dlv.data_ptr(), -1.0, ops._stream()), 'estts_club_terms')
        if train:
            self.store.clear_gradients()
            self.net_mu.backward(acts_mu, dmu, self.precision)
            self.net_log_var.backward(acts_lv, dlv, self.precision)
            self._apply()
        return out[0], out[1]
