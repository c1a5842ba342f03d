"""Packed building blocks shared by the acoustic models: every reference Keras layer group
(TransformerTTS/model/layers.py) becomes a short fixed sequence of HIP kernel launches.

Buffer convention: a stack works on a "cat buffer" xc [rows, 2*d]; columns [0, d) hold the block
input x, columns [d, 2d) receive the attention context, so that the output projection, which the
reference feeds with concat([q_in, context]) (layers.py:144-145), is one GEMM over the whole row.
"""
import math

import numpy as np
import torch

from . import ops
from .ops import PackedLinear, fold_batchnorm


def _dev_vec(a, device):
    return torch.from_numpy(np.ascontiguousarray(np.asarray(a, dtype=np.float32))).to(device)


def pack_dense(w, p, device, act=None):
    """Keras Dense: kernel (in, out)."""
    return PackedLinear(np.asarray(w[p + '.kernel']).T, bias=w[p + '.bias'], device=device, act=act)


def pack_conv_bn(w, conv_p, norm_p, padding, device, act=None, last_norm_p=None):
    """Keras Conv1D kernel (k, in, out) followed by inference BatchNormalization (folded into s1/b1),
    optionally followed (after activation and residual add) by a second BatchNormalization (s2/b2)."""
    kern = np.asarray(w[conv_p + '.kernel'])
    k, cin, cout = kern.shape
    s, b = fold_batchnorm(w[norm_p + '.gamma'], w[norm_p + '.beta'], w[norm_p + '.moving_mean'],
                          w[norm_p + '.moving_variance'])
    bias = np.asarray(w[conv_p + '.bias'], np.float32) * s + b
    s2 = b2 = None
    if last_norm_p is not None:
        s2, b2 = fold_batchnorm(w[last_norm_p + '.gamma'], w[last_norm_p + '.beta'], w[last_norm_p + '.moving_mean'],
                                w[last_norm_p + '.moving_variance'])
    lpad = (k - 1) // 2 if padding == 'same' else k - 1
    return PackedLinear(kern.reshape(k * cin, cout).T, bias=bias, scale=s, device=device, taps=k, lpad=lpad, act=act,
                        s2=s2, b2=b2)


def pack_conv(w, conv_p, padding, device):
    kern = np.asarray(w[conv_p + '.kernel'])
    k, cin, cout = kern.shape
    lpad = (k - 1) // 2 if padding == 'same' else k - 1
    return PackedLinear(kern.reshape(k * cin, cout).T, bias=w[conv_p + '.bias'], device=device, taps=k, lpad=lpad)


def pack_ln(w, p, device):
    return (_dev_vec(w[p + '.gamma'], device), _dev_vec(w[p + '.beta'], device))


class SelfAttentionLayer:
    """SelfAttentionResNorm (layers.py:150-168): fused QKV projection, attention, output projection over
    [x | ctx], then LN(LN(y) + x)."""

    def __init__(self, w, p, d, heads, device):
        m = p + '.mha'
        wqkv = np.concatenate([np.asarray(w[m + '.wq.kernel']).T, np.asarray(w[m + '.wk.kernel']).T,
                               np.asarray(w[m + '.wv.kernel']).T], axis=0)
        bqkv = np.concatenate([w[m + '.wq.bias'], w[m + '.wk.bias'], w[m + '.wv.bias']])
        self.qkv = PackedLinear(wqkv, bias=bqkv, device=device)
        self.out = pack_dense(w, m + '.dense', device)
        self.ln = pack_ln(w, p + '.ln', device)
        self.last_ln = pack_ln(w, p + '.last_ln', device)
        self.d, self.heads = d, heads

    def run(self, xc, qkv, y, x1, B, L, key_mask, causal, attn, precision, defer_norm=False):
        """defer_norm: leave LN(LN(y) + x) to the convolution that consumes it (ops.gemm(ln=...)); returns its description."""
        d = self.d
        ops.gemm(xc[:, :d], self.qkv, qkv, precision=precision, site='qkv')
        ops.attention(qkv[:, :d], qkv[:, d:2 * d], qkv[:, 2 * d:], xc[:, d:], B, self.heads, L, L, d // self.heads,
                      key_mask=key_mask, causal=causal, attn=attn, precision=precision)
        if not defer_norm and ops.can_fuse_ln_epilogue(self.out, B * L):   # batch sizes: the norms inside the projection's launch
            ops.gemm(xc, self.out, x1, precision=precision, site='attn_out',
                     norm=dict(ln_a=self.ln, residual=xc[:, :d], ln_b=self.last_ln))
            return None
        ops.gemm(xc, self.out, y, precision=precision, site='attn_out')
        if defer_norm:
            return dict(y=y, x=xc[:, :d], ln_a=self.ln, ln_b=self.last_ln, out=x1)
        ops.rownorm(y, x1, ln_a=self.ln, residual=xc[:, :d], ln_b=self.last_ln)
        return None


class FFNLayer:
    """FFNResNorm (layers.py:50-71): LN(relu(LN(d2(d1(x)))) + x).  The reference applies no activation
    between d1 and d2, so for inference the two are folded into one d x d product at pack time
    (W = W1 W2, b = b1 W2 + b2, in float64); fold=False keeps the two GEMMs."""

    def __init__(self, w, p, device, fold=True):
        self.fold = fold
        if fold:
            w1 = np.asarray(w[p + '.d1.kernel'], np.float64)
            w2 = np.asarray(w[p + '.d2.kernel'], np.float64)
            b = np.asarray(w[p + '.d1.bias'], np.float64) @ w2 + np.asarray(w[p + '.d2.bias'], np.float64)
            self.d12 = PackedLinear((w1 @ w2).T, bias=b, device=device)
        else:
            self.d1 = pack_dense(w, p + '.d1', device)
            self.d2 = pack_dense(w, p + '.d2', device)
        self.ln = pack_ln(w, p + '.ln', device)
        self.last_ln = pack_ln(w, p + '.last_ln', device)

    def hidden(self):
        return 0 if self.fold else self.d1.N

    def run(self, x1, h, y, out, precision):
        if self.fold and ops.can_fuse_ln_epilogue(self.d12, x1.shape[0]):
            ops.gemm(x1, self.d12, out, precision=precision, site='ffn',
                     norm=dict(ln_a=self.ln, act='relu', residual=x1, ln_b=self.last_ln))
            return
        if self.fold:
            ops.gemm(x1, self.d12, y, precision=precision, site='ffn')
        else:
            ops.gemm(x1, self.d1, h, precision=precision, site='ffn')
            ops.gemm(h, self.d2, y, precision=precision, site='ffn')
        ops.rownorm(y, out, ln_a=self.ln, act='relu', residual=x1, ln_b=self.last_ln)


class ConvBNLayer:
    """CNNResNorm with BatchNormalization (layers.py:7-47), inference mode: every Conv1D + BN + activation is
    one implicit-GEMM launch; the last one also adds the block input and applies the trailing BN."""

    def __init__(self, w, p, n_layers, inner_act, last_act, padding, device):
        self.layers = []
        for i in range(n_layers - 1):
            self.layers.append(pack_conv_bn(w, f'{p}.conv{i}', f'{p}.norm{i}', padding, device, act=inner_act))
        self.layers.append(pack_conv_bn(w, p + '.last_conv', f'{p}.norm{n_layers - 1}', padding, device, act=last_act,
                                        last_norm_p=f'{p}.norm{n_layers}'))

    def hidden(self):
        return max([l.N for l in self.layers[:-1]] + [0])

    def can_take_norm(self, M, precision, site='conv'):
        """True when the block's first convolution can compute its input as a deferred LayerNorm combination (it must not be
        the layer that adds the block input back: that one reads rows other workgroups are still writing)."""
        return len(self.layers) >= 2 and ops.can_fuse_ln(self.layers[0], M, precision=precision, site=site)

    def run(self, x, h_a, h_b, out, seg_len, precision, site='conv', ln=None):
        """x: block input view; h_a/h_b: two scratch buffers wide enough for the hidden size; out: output view.
        ln: the block input is still to be computed (SelfAttentionLayer.run(defer_norm=True)): the first convolution does it."""
        src = x
        bufs = (h_a, h_b)
        for i, lin in enumerate(self.layers[:-1]):
            dst = bufs[i & 1][:, :lin.N]
            ops.gemm(src, lin, dst, seg_len=seg_len, precision=precision, site=site, ln=ln if i == 0 else None)
            src = dst
        ops.gemm(src, self.layers[-1], out, seg_len=seg_len, residual=x, precision=precision, site=site)


class SelfAttentionStack:
    """SelfAttentionBlocks (layers.py:215-255) without its positional front end (the caller fuses
    x*sqrt(d) + PE into whatever produces the stack input)."""

    def __init__(self, w, p, name, d, heads, dense_blocks, device, fold_ffn=True):
        self.d, self.name = d, name
        self.blocks = []
        self.names = []
        for i, h in enumerate(heads):
            b = f'{p}.block{i}'
            sa = SelfAttentionLayer(w, b + '.sarn', d, h, device)
            if i < dense_blocks:
                ff = FFNLayer(w, b + '.ffn', device, fold=fold_ffn)
                self.names.append(f'{name}_DenseBlock{i + 1}_SelfAttention')
            else:
                ff = ConvBNLayer(w, b + '.conv', 2, 'relu', 'relu', 'same', device)
                self.names.append(f'{name}_ConvBlock{i - dense_blocks + 1}_SelfAttention')
            self.blocks.append((sa, ff))
        self.hidden = max(ff.hidden() for _, ff in self.blocks)
        self.fuse_norm = True        # the attention block's LN(LN(y) + x) inside the next convolution where that launch can

    def alloc(self, rows, device):
        d = self.d
        f32 = dict(dtype=torch.float32, device=device)
        return {'xc': torch.empty(rows, 2 * d, **f32), 'qkv': torch.empty(rows, 3 * d, **f32),
                'y': torch.empty(rows, d, **f32), 'x1': torch.empty(rows, d, **f32),
                'h': torch.empty(rows, max(self.hidden, 4), **f32)}

    def run(self, ws, B, L, key_mask, precision, return_attention, causal=False):
        """Input in ws['xc'][:, :d]; output left there.  Returns {name: attention weights} (empty if not requested)."""
        d = self.d
        precision = ops.scoped(precision, 'encoder' if 'Encoder' in self.name else 'decoder')
        xc, attn_out = ws['xc'], {}
        for (sa, ff), name in zip(self.blocks, self.names):
            attn = None
            if return_attention:
                attn = torch.empty(B, sa.heads, L, L, dtype=torch.float32, device=xc.device)
                attn_out[name] = attn
            defer = isinstance(ff, ConvBNLayer) and self.fuse_norm and ff.can_take_norm(B * L, precision)
            ln = sa.run(xc, ws['qkv'], ws['y'], ws['x1'], B, L, key_mask, causal, attn, precision, defer_norm=defer)
            if isinstance(ff, FFNLayer):
                ff.run(ws['x1'], ws['h'], ws['y'], xc[:, :d], precision)
            else:
                ff.run(ws['x1'], ws['h'], ws['h'], xc[:, :d], L, precision, ln=ln)
        return attn_out


class CrossAttentionLayer:
    """CrossAttentionResnorm (layers.py:258-274): q from the decoder stream, k / v from the encoder output,
    output projection over [q_in | ctx], then LN(attn + q)."""

    def __init__(self, w, p, d, heads, device):
        m = p + '.mha'
        self.wq = pack_dense(w, m + '.wq', device)
        wkv = np.concatenate([np.asarray(w[m + '.wk.kernel']).T, np.asarray(w[m + '.wv.kernel']).T], axis=0)
        self.wkv = PackedLinear(wkv, bias=np.concatenate([w[m + '.wk.bias'], w[m + '.wv.bias']]), device=device)
        self.out = pack_dense(w, m + '.dense', device)
        self.ln = pack_ln(w, p + '.layernorm', device)
        self.d, self.heads = d, heads

    def project_kv(self, enc, kv, precision):
        """enc [B*N, E] -> kv [B*N, 2d] (keys | values); depends on the encoder output only."""
        ops.gemm(enc, self.wkv, kv, precision=precision, site='cross_kv')

    def run(self, xc, q, kv, y, out, B, Lq, Lk, enc_mask, attn, precision):
        d = self.d
        ops.gemm(xc[:, :d], self.wq, q, precision=precision, site='cross_q')
        ops.attention(q, kv[:, :d], kv[:, d:], xc[:, d:], B, self.heads, Lq, Lk, d // self.heads, key_mask=enc_mask,
                      attn=attn, precision=precision)
        if ops.can_fuse_ln_epilogue(self.out, B * Lq):
            ops.gemm(xc, self.out, out, precision=precision, site='attn_out', norm=dict(residual=xc[:, :d], ln_b=self.ln))
            return
        ops.gemm(xc, self.out, y, precision=precision, site='attn_out')
        ops.rownorm(y, out, residual=xc[:, :d], ln_b=self.ln)


class CrossAttentionStack:
    """CrossAttentionBlocks (layers.py:331-376) without its positional front end."""

    def __init__(self, w, p, name, d, heads, dense_blocks, conv_padding, device, fold_ffn=True):
        self.d, self.name = d, name
        self.blocks, self.names = [], []
        for i, h in enumerate(heads):
            b = f'{p}.block{i}'
            sa = SelfAttentionLayer(w, b + '.sarn', d, h, device)
            ca = CrossAttentionLayer(w, b + '.carn', d, h, device)
            if i < dense_blocks:
                ff = FFNLayer(w, b + '.ffn', device, fold=fold_ffn)
                self.names.append(f'{name}_DenseBlock{i + 1}_CrossAttention')
            else:
                ff = ConvBNLayer(w, b + '.conv', 2, 'relu', 'relu', conv_padding, device)
                self.names.append(f'{name}_ConvBlock{i - dense_blocks + 1}_CrossAttention')
            self.blocks.append((sa, ca, ff))
        self.hidden = max(ff.hidden() for _, _, ff in self.blocks)

    def alloc(self, rows, enc_rows, device):
        d = self.d
        f32 = dict(dtype=torch.float32, device=device)
        return {'xc': torch.empty(rows, 2 * d, **f32), 'xc2': torch.empty(rows, 2 * d, **f32),
                'qkv': torch.empty(rows, 3 * d, **f32), 'q': torch.empty(rows, d, **f32), 'y': torch.empty(rows, d, **f32),
                'x2': torch.empty(rows, d, **f32), 'h': torch.empty(rows, max(self.hidden, 4), **f32),
                'kv': [torch.empty(enc_rows, 2 * d, **f32) for _ in self.blocks]}

    def project_kv(self, ws, enc, precision):
        precision = ops.scoped(precision, 'encoder' if 'Encoder' in self.name else 'decoder')
        for (_, ca, _), kv in zip(self.blocks, ws['kv']):
            ca.project_kv(enc, kv, precision)

    def run(self, ws, B, Lq, Lk, dec_key_mask, enc_mask, precision, return_attention):
        """Input in ws['xc'][:, :d]; output left there.  Self-attention is causal (look-ahead mask) and masks padded
        decoder frames; returns {name: cross-attention weights}."""
        d = self.d
        precision = ops.scoped(precision, 'encoder' if 'Encoder' in self.name else 'decoder')
        xc, xc2, attn_out = ws['xc'], ws['xc2'], {}
        for (sa, ca, ff), kv, name in zip(self.blocks, ws['kv'], self.names):
            attn = None
            if return_attention:
                attn = torch.empty(B, ca.heads, Lq, Lk, dtype=torch.float32, device=xc.device)
                attn_out[name] = attn
            sa.run(xc, ws['qkv'], ws['y'], xc2[:, :d], B, Lq, dec_key_mask, True, None, precision)
            ca.run(xc2, ws['q'], kv, ws['y'], ws['x2'], B, Lq, Lk, enc_mask, attn, precision)
            if isinstance(ff, FFNLayer):
                ff.run(ws['x2'], ws['h'], ws['y'], xc[:, :d], precision)
            else:
                ff.run(ws['x2'], ws['h'], ws['h'], xc[:, :d], Lq, precision)
        return attn_out
