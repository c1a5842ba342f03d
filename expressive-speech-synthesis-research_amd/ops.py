"""Host-side op layer: torch tensors in, HIP kernels (through the C ABI) out.

torch is used for device memory and the current stream only; no torch op
computes anything on the hot path.
"""
import contextlib
import ctypes
import math
import os
import sys

import numpy as np
import torch

from . import _lib
from ._lib import ACT, AttentionArgs, GemmArgs, RownormArgs, check

PRECISION = {'bf16': 1, 'bf16x3': 3}


class PrecisionPolicy:
    """Operand precision per call site.  `table` maps a site ('qkv', 'attention', 'attn_out', 'ffn', 'conv', 'cross_q',
    'cross_kv', 'dur_pred', 'prenet', 'out', 'stop', 'postnet', 'gst'), optionally prefixed by the stack it sits in
    ('encoder.qkv', 'decoder.conv'), to 'bf16' or 'bf16x3'; 'default' covers the rest.  Look-up order:
    '<scope>.<site>', '<site>', 'default'."""

    def __init__(self, table, scope=''):
        bad = {k: v for k, v in table.items() if v not in PRECISION}
        if bad:
            raise ValueError(f'PrecisionPolicy: unknown precision in {bad}')
        self.table, self.scope = dict(table), scope
        self.table.setdefault('default', 'bf16x3')

    def scoped(self, scope):
        return PrecisionPolicy(self.table, scope.lower())

    def at(self, site):
        t = self.table
        return t.get(f'{self.scope}.{site}') or t.get(site) or t['default']

    def __repr__(self):
        return f'PrecisionPolicy({self.table})'


def as_precision(precision):
    """'bf16' / 'bf16x3' stay strings; a dict becomes a PrecisionPolicy."""
    if isinstance(precision, dict):
        return PrecisionPolicy(precision)
    if not isinstance(precision, PrecisionPolicy) and precision not in PRECISION:
        raise ValueError(f'precision must be one of {sorted(PRECISION)} or a site table, got {precision!r}')
    return precision


def scoped(precision, scope):
    return precision.scoped(scope) if isinstance(precision, PrecisionPolicy) else precision


def precision_at(precision, site):
    return precision.at(site) if isinstance(precision, PrecisionPolicy) else precision

# When set to a list, every op appends (kind, info dict, relaunch callable): used by bench.py to time
# individual kernels with events and to price them against the roofline.
_trace = None
# ESTTS_DEBUG_SYNC=1: name every launch on stderr and wait for it, so that a faulting kernel is the last line printed
_debug_sync = os.environ.get('ESTTS_DEBUG_SYNC', '') not in ('', '0')


def start_trace():
    global _trace
    _trace = []
    return _trace


def stop_trace():
    global _trace
    t, _trace = _trace, None
    return t


def _record(kind, info, fn, *args, keep=()):
    """Remember a launch; the relaunch callable passes whatever stream is current when it is called.  `keep` holds the
    tensors whose device pointers are baked into `args`, so that a replay can never touch freed memory."""
    if _debug_sync:
        print(f'[estts] {kind} {info}', file=sys.stderr, flush=True)
        torch.cuda.synchronize()
    if _trace is not None:
        _trace.append((kind, info, lambda _keep=tuple(keep): fn(*args, _stream())))


# When set to a list, gemm / attention / rownorm / row_zero_mask append (kind, argument block, tensors) instead of
# launching: see StepProgram.
_program = None


class StepProgram:
    """A fixed sequence of launches whose arguments depend affinely on a step index, issued from one C call per step
    (estts_run_program).  Built by recording the same Python code for step 0 and step 1: every argument field that
    differs between the two recordings becomes a patch `field += step * (value1 - value0)`."""

    KINDS = {'gemm': 0, 'attention': 1, 'rownorm': 2, 'row_zero_mask': 3, 'cross_attention_row': 4, 'decoder_block_row': 5, 'prenet_row': 6}

    def __init__(self, record_step):
        """record_step(i): issues the launches of step i through this module's op functions."""
        global _program
        recs = []
        for i in (0, 1):
            _program = []
            try:
                record_step(i)
            finally:
                recs.append(_program)
                _program = None
        r0, r1 = recs
        if len(r0) != len(r1) or any(a[0] != b[0] for a, b in zip(r0, r1)):
            raise RuntimeError('StepProgram: steps 0 and 1 issue different launch sequences')
        self._keep = (r0, r1)
        self.n = len(r0)
        self.ops = (_lib.ProgOp * self.n)()
        for j, ((kind, a0, _), (_, a1, _)) in enumerate(zip(r0, r1)):
            op = self.ops[j]
            op.kind = self.KINDS[kind]
            op.args = ctypes.addressof(a0)
            k = 0
            for name, ftype in a0._fields_:
                v0, v1 = getattr(a0, name) or 0, getattr(a1, name) or 0
                if v0 == v1:
                    continue
                if ftype not in (ctypes.c_void_p, ctypes.c_int32) or k >= _lib.PROG_MAX_PATCH:
                    raise RuntimeError(f'StepProgram: field {name} of a {kind} launch cannot be patched')
                op.patch_offset[k] = getattr(type(a0), name).offset
                op.patch_is_ptr[k] = int(ftype is ctypes.c_void_p)
                op.patch_stride[k] = int(v1 - v0)
                k += 1
            op.n_patch = k

    def run(self, step):
        check(_lib.load().estts_run_program(self.ops, self.n, step, _stream()), 'estts_run_program')


def _stream():
    return torch.cuda.current_stream().cuda_stream


def _ptr(t):
    return 0 if t is None else t.data_ptr()


def _ceil(a, b):
    return (a + b - 1) // b * b


def _check_f32(t, name):
    if t.dtype != torch.float32 or not t.is_cuda:
        raise TypeError(f'{name}: expected a float32 CUDA tensor, got {t.dtype} on {t.device}')
    if t.stride(-1) != 1:
        raise ValueError(f'{name}: last dimension must be contiguous')


class PackedLinear:
    """A dense / conv1d weight in the kernel's layout: bf16 hi (+ lo) [N_pad][K_pad], K contiguous, plus the
    float32 epilogue vectors (s1, b1: scale and bias applied before the activation; s2, b2: trailing affine).

    `w_nk` is a float array [N, K] (for a convolution K = taps * c_per_tap, tap-major).  N is padded to a multiple
    of 64 and K to a multiple of 64 with zeros; storage order is fragment-major (see _fragment_major)."""

    def __init__(self, w_nk, bias=None, scale=None, device='cuda', taps=1, lpad=0, act=None, s2=None, b2=None):
        w_nk = np.ascontiguousarray(np.asarray(w_nk, dtype=np.float32))
        self.N, self.K = w_nk.shape
        if self.K % 4:
            raise ValueError('K must be a multiple of 4')
        self.taps, self.lpad = taps, lpad
        self.c_per_tap = self.K // taps
        self.act = ACT[act]
        Np, Kp = _ceil(self.N, 64), _ceil(self.K, 64)
        wp = torch.zeros(Np, Kp, dtype=torch.float32)
        wp[:self.N, :self.K] = torch.from_numpy(w_nk)
        hi = wp.to(torch.bfloat16)
        lo = (wp - hi.to(torch.float32)).to(torch.bfloat16)
        self.ldw = Kp
        self.w_hi = self._fragment_major(hi).to(device)
        self.w_lo = self._fragment_major(lo).to(device)

        def vec(v, fill):
            out = torch.full((Np,), fill, dtype=torch.float32)
            if v is not None:
                out[:self.N] = torch.from_numpy(np.asarray(v, dtype=np.float32))
            return out.to(device)

        self.s1 = vec(scale, 1.0) if scale is not None else None
        self.b1 = vec(bias, 0.0) if bias is not None else None
        self.s2 = vec(s2, 1.0) if s2 is not None else None
        self.b2 = vec(b2, 0.0) if s2 is not None else None

    @staticmethod
    def _fragment_major(w):
        """[Np][Kp] -> blocks of 64 lanes x 8 values: block (n // 16, k // 32), lane (n % 16) + 16 * ((k % 32) // 8),
        i.e. the B-operand register image of v_mfma_f32_16x16x32_bf16, so that one wave reads a fragment as 1 KiB
        of contiguous memory."""
        Np, Kp = w.shape
        return w.view(Np // 16, 16, Kp // 32, 4, 8).permute(0, 2, 3, 1, 4).contiguous()

    @property
    def nbytes(self):
        return self.w_hi.numel() * 2


_WS_BYTES = 64 << 20
_N_COUNTERS = 4096
_workspaces = {}


def gemm_workspace(device):
    """Per-device split-K scratch (float32 slabs + zeroed tickets) shared by every gemm() call on the
    current stream.  Launches that may run concurrently on different streams must not share it."""
    key = torch.device(device).index if torch.device(device).index is not None else torch.cuda.current_device()
    ws = _workspaces.get(key)
    if ws is None:
        dev = torch.device('cuda', key)
        ws = (torch.empty(_WS_BYTES // 4, dtype=torch.float32, device=dev),
              torch.zeros(_N_COUNTERS, dtype=torch.int32, device=dev))
        _workspaces[key] = ws
    return ws


_ws_override = None


@contextlib.contextmanager
def private_workspace(ws):
    """Every gemm() issued inside the block uses the split-K scratch `ws` (new_gemm_workspace()) instead of the device's shared
    one: for launch sequences that run beside others on a stream of their own."""
    global _ws_override
    prev, _ws_override = _ws_override, ws
    try:
        yield ws
    finally:
        _ws_override = prev


def new_gemm_workspace(device, nbytes=4 << 20):
    """A private split-K scratch for gemm(workspace=...)."""
    return (torch.empty(nbytes // 4, dtype=torch.float32, device=device), torch.zeros(_N_COUNTERS, dtype=torch.int32, device=device))


def can_fuse_ln(lin, M, N_out=None, precision='bf16x3', site=None, act=None):
    """True when gemm(..., ln=...) can compute its input rows as a LayerNorm combination on the way in (estts_gemm_args.ln_y):
    a 1-D convolution of 2..8 taps over at most 1024 rows, c_per_tap <= 512, K <= 1536, split-bf16 operands, no activation or
    ReLU between the two norms, and not while a step program is being recorded (a decoding step convolves a window of history
    rows of which only the last is new: there is nothing to recompute the older ones from); or a dense layer over at most 16
    rows with K <= 512 (the one-row products of incremental decoding: every column tile recomputes the norm of its rows,
    which costs nothing when there is one).  Callers fall back to rownorm() + gemm()."""
    if precision_at(precision, site) != 'bf16x3' or act not in (None, 'relu'):
        return False
    if lin.taps == 1:
        return M <= 16 and lin.K % 8 == 0 and lin.K <= 512
    if _program is not None:
        return False
    C = lin.c_per_tap
    if not (2 <= lin.taps <= 8 and C % 8 == 0 and C <= 512 and lin.K % 8 == 0 and lin.K <= 1536 and M <= 1024):
        return False
    t32 = ((M + 31) // 32) * (((lin.N if N_out is None else N_out) + 31) // 32)
    tm = 16 if t32 < 128 else 32
    if (tm + lin.taps - 1) * (C + 8) * 4 > 96 * 1024:
        return False
    return not (tm == 32 and C > 256 and lin.K > 768)      # (that variant would not fit its registers)


LN_EPILOGUE_MIN_ROWS = 8192     # rows from which gemm(norm=...) beats gemm() + rownorm() (64-row tiles: 128 workgroups of 512 threads)


def can_fuse_ln_epilogue(lin, M, act=None):
    """True when gemm(..., norm=...) can apply the LayerNorm combination that follows the layer inside its own launch
    (estts_gemm_args.ln_epilogue) and that pays: 256 outputs, no epilogue operands of the layer's own beyond scale / bias,
    thousands of rows, and not while a step program is being recorded.  Callers fall back to gemm() + rownorm()."""
    if os.environ.get('ESTTS_LN_EPILOGUE', '1') == '0':      # (measurements: the two-launch form)
        return False
    return (lin.N == 256 and M >= LN_EPILOGUE_MIN_ROWS and lin.act == 0 and act in (None, 'linear', 'none') and lin.s2 is None
            and _program is None)


def gemm(x, lin, out, M=None, seg_len=0, residual=None, r_mod=0, alpha=1.0, precision='bf16x3', act=None, tune=0,
         split=True, conv2d=None, dropout=None, batch=None, workspace=None, site=None, ln=None, stats=None, norm=None):
    """out[:, :N] = epilogue(x[:, :K] (*) lin).  x, out, residual: 2-D float32 views (row stride arbitrary).
    dropout = (rate, seed, site, row0): keep/scale after the activation, generator row = row0 + output row.
    batch = (count, stride_a, stride_w, stride_c, stride_v): that many products of this shape in one launch; x / lin / out
    describe product 0 and the strides (elements) lead to the others (see estts_gemm_args.batch).
    workspace = (float32 scratch, zeroed int32 tickets): split-K scratch of its own for launches on a second stream that may
    run beside launches using the shared one (new_gemm_workspace()).
    precision: 'bf16' / 'bf16x3', or a PrecisionPolicy resolved at `site`.
    ln = dict(y=, x= or None, ln_a=(gamma, beta), ln_b=(gamma, beta) or None, act= or None, out=, eps=1e-6): the input rows
    are LNb(act(LNa(y)) + x), computed inside the launch and also written to `out`, which is what `x` must be (see
    can_fuse_ln; rownorm() fused into the convolution that consumes it).
    stats = (slots pointer, shift [N] tensor or None): also add the column moments of the output about `shift` to the 16
    copies at the pointer (train_ops.SumSlots.forward_region; estts_gemm_args.stat_slots), for bn_finalize(slots=...).
    norm = dict(ln_a=(gamma, beta) or None, act= or None, residual= or None, ln_b=(gamma, beta) or None, eps=1e-6): out =
    LNb(act(LNa(x (*) lin)) + residual), i.e. the rownorm() that follows the layer, inside this launch (see
    can_fuse_ln_epilogue; N = 256 only)."""
    precision = precision_at(precision, site)
    _check_f32(x, 'x'); _check_f32(out, 'out')
    M = x.shape[0] if M is None else M
    a = GemmArgs()
    a.A, a.lda, a.seg_len = x.data_ptr(), x.stride(0), seg_len
    a.taps, a.c_per_tap, a.tap_lpad = lin.taps, lin.c_per_tap, lin.lpad
    a.W_hi, a.W_lo, a.ldw = lin.w_hi.data_ptr(), lin.w_lo.data_ptr(), lin.ldw
    a.M, a.N, a.K = M, lin.N, lin.K
    a.s1, a.b1 = _ptr(lin.s1), _ptr(lin.b1)
    a.act = lin.act if act is None else ACT[act]
    a.alpha = alpha
    if residual is not None:
        _check_f32(residual, 'residual')
        a.R, a.ldr, a.r_mod = residual.data_ptr(), residual.stride(0), r_mod
    a.s2, a.b2 = _ptr(lin.s2), _ptr(lin.b2)
    a.C, a.ldc = out.data_ptr(), out.stride(0)
    a.precision = PRECISION[precision]
    a.tune = tune
    if conv2d is not None:   # (H, W, Ho, Wo, kw, stride, pad_top, pad_left); x rows are NHWC pixels
        H, W, Ho, Wo, kw, stride, pt, pl = conv2d
        a.conv_h, a.conv_w, a.conv_wo, a.conv_howo = H, W, Wo, Ho * Wo
        a.conv_kw, a.conv_stride, a.conv_pt, a.conv_pl = kw, stride, pt, pl
    if dropout is not None and dropout[0] > 0.0:
        if not dropout[0] < 1.0:
            raise ValueError('gemm: dropout rate must be below 1')
        a.drop_rate, a.drop_seed, a.drop_site, a.drop_row0 = dropout
    if batch is not None:
        a.batch, a.batch_stride_a, a.batch_stride_w, a.batch_stride_c, a.batch_stride_v = batch
    if ln is not None:
        y, xr, lo = ln['y'], ln.get('x'), ln['out']
        _check_f32(y, 'ln.y'); _check_f32(lo, 'ln.out')
        if x.data_ptr() != lo.data_ptr():
            raise ValueError('gemm: with ln= the input must be the buffer the normalised rows are written to (ln[\'out\'])')
        if out.data_ptr() in (y.data_ptr(), lo.data_ptr()) or (xr is not None and out.data_ptr() == xr.data_ptr()):
            raise ValueError('gemm: the output must not alias an operand of the LayerNorm prologue')
        a.ln_y, a.ln_ldy, a.ln_out, a.ln_ldo = y.data_ptr(), y.stride(0), lo.data_ptr(), lo.stride(0)
        if xr is not None:
            _check_f32(xr, 'ln.x')
            a.ln_x, a.ln_ldx = xr.data_ptr(), xr.stride(0)
        if ln.get('ln_a') is not None:
            a.ln_ga, a.ln_ba = ln['ln_a'][0].data_ptr(), ln['ln_a'][1].data_ptr()
        elif ln.get('ln_b') is None or ln.get('act') not in (None, 'linear', 'none'):
            raise ValueError('gemm: ln= needs ln_a or ln_b, and an activation only with ln_a')
        if ln.get('ln_b') is not None:
            a.ln_gb, a.ln_bb = ln['ln_b'][0].data_ptr(), ln['ln_b'][1].data_ptr()
        a.ln_act, a.ln_eps = ACT[ln.get('act')], ln.get('eps', 1e-6)
    if stats is not None:
        a.stat_slots, a.stat_shift = stats[0], _ptr(stats[1])
    if norm is not None:
        if ln is not None or residual is not None or stats is not None or batch is not None or (dropout is not None and dropout[0] > 0.0):
            raise ValueError('gemm: norm= not together with ln=, residual=, stats=, batch= or dropout=')
        a.ln_epilogue = 1
        xr = norm.get('residual')
        if xr is not None:
            _check_f32(xr, 'norm.residual')
            a.ln_x, a.ln_ldx = xr.data_ptr(), xr.stride(0)
        if norm.get('ln_a') is not None:
            a.ln_ga, a.ln_ba = norm['ln_a'][0].data_ptr(), norm['ln_a'][1].data_ptr()
        elif norm.get('ln_b') is None or norm.get('act') not in (None, 'linear', 'none'):
            raise ValueError('gemm: norm= needs ln_a or ln_b, and an activation only with ln_a')
        if norm.get('ln_b') is not None:
            a.ln_gb, a.ln_bb = norm['ln_b'][0].data_ptr(), norm['ln_b'][1].data_ptr()
        a.ln_act, a.ln_eps = ACT[norm.get('act')], norm.get('eps', 1e-6)
    if split:
        ws, cnt = workspace if workspace is not None else (_ws_override if _ws_override is not None else gemm_workspace(x.device))
        a.ws, a.ws_bytes, a.counters, a.n_counters = ws.data_ptr(), ws.numel() * 4, cnt.data_ptr(), cnt.numel()
    if lin.taps == 1 and x.shape[1] < lin.K:
        raise ValueError(f'gemm: x has {x.shape[1]} columns, weight expects K={lin.K}')
    if out.shape[1] < lin.N:
        raise ValueError(f'gemm: out has {out.shape[1]} columns, weight produces N={lin.N}')
    if _program is not None:
        _program.append(('gemm', a, (x, out, residual, lin)))
        return out
    check(_lib.load().estts_gemm(a, _stream()), 'estts_gemm')
    nsplit = 2 if precision == 'bf16x3' else 1
    info = {'M': M, 'N': lin.N, 'K': lin.K, 'taps': lin.taps, 'flops': 2.0 * M * lin.N * lin.K,
            'bytes': 4.0 * M * lin.K / lin.taps + 2.0 * nsplit * lin.N * lin.K + 4.0 * M * lin.N}
    if ln is not None:       # another kernel (rows normalised on the way in, read twice, written once): its own line in a kernel table
        info['ln_prologue'] = True
        info['bytes'] += 4.0 * M * lin.c_per_tap * (2 if ln.get('x') is not None else 1)
    if norm is not None:     # the rows are normalised on the way out: the same bytes as the plain layer plus the residual rows
        info['ln_epilogue'] = True
        info['bytes'] += 4.0 * M * lin.N if norm.get('residual') is not None else 0.0
    _record('gemm', info,
            _lib.load().estts_gemm, a, keep=(x, out, residual, lin, ln, norm))
    return out


def attention(q, k, v, ctx, B, H, Lq, Lk, depth, key_mask=None, causal=False, attn=None, precision='bf16x3',
              kv_shared=False, variant=0, site='attention'):
    """q: [B*Lq, >=H*depth] view, k/v: [B*Lk, ...] views, ctx: [B*Lq, ...] view; attn: [B,H,Lq,Lk] or None."""
    precision = precision_at(precision, site)
    for t, n in ((q, 'q'), (k, 'k'), (v, 'v'), (ctx, 'ctx')):
        _check_f32(t, n)
    a = AttentionArgs()
    a.q, a.ldq = q.data_ptr(), q.stride(0)
    a.k, a.ldk = k.data_ptr(), k.stride(0)
    a.v, a.ldv = v.data_ptr(), v.stride(0)
    a.ctx, a.ldo = ctx.data_ptr(), ctx.stride(0)
    if attn is not None:
        if tuple(attn.shape) != (B, H, Lq, Lk) or not attn.is_contiguous():
            raise ValueError('attention: attn must be a contiguous [B,H,Lq,Lk] tensor')
        a.attn = attn.data_ptr()
    a.key_mask = _ptr(key_mask)
    a.B, a.H, a.Lq, a.Lk, a.depth = B, H, Lq, Lk, depth
    a.causal = int(causal)
    a.precision = PRECISION[precision]
    a.kv_shared = int(kv_shared)
    a.variant = variant
    if _program is not None:
        _program.append(('attention', a, (q, k, v, ctx, attn, key_mask)))
        return ctx
    check(_lib.load().estts_attention(a, _stream()), 'estts_attention')
    _record('attention', {'B': B, 'H': H, 'Lq': Lq, 'Lk': Lk, 'depth': depth, 'weights': attn is not None,
                          'flops': 4.0 * B * H * Lq * Lk * depth,
                          'bytes': 4.0 * B * H * depth * (2 * Lq + 2 * Lk) + (4.0 * B * H * Lq * Lk if attn is not None else 0)},
            _lib.load().estts_attention, a, keep=(q, k, v, ctx, attn, key_mask))
    return ctx


def can_cross_attention_row(wq, wo, d, heads, Lk, precision='bf16x3', site=None):
    """True when cross_attention_row() can take the block: width 256, heads of depth 64, at most 256 keys (Lk = the most there
    will be), split operands, plain dense projections (bias only).  wq None: the query rows are given."""
    plain = all(l.taps == 1 and l.act == 0 and l.s1 is None and l.s2 is None for l in (wq, wo) if l is not None)
    return (precision_at(precision, site) == 'bf16x3' and d == 256 and heads * 64 == d and 0 < Lk <= 256 and plain and
            (wq is None or (wq.N == d and wq.K == d)) and wo.N == d and wo.K == 2 * d)


def cross_attention_row(wq, wo, k, v, y, heads, x=None, ln=None, key_mask=None, attn=None, M=None, q=None):
    """One decoder position per row through q projection, attention over the encoder's keys / values and the output projection
    over [x | ctx] in one launch (estts_cross_attention_row; see can_cross_attention_row).  x: [M, d] rows, or ln = dict(y=,
    x= or None, ln_a=, ln_b=, act=, out=, eps=1e-6) as in gemm(ln=...): the rows are that normalisation, also written to
    ln['out'].  k / v: [Lk, >= d] views shared by all rows (M = 1 in the incremental decoder); attn: [M, heads, Lk] view or None;
    y: [M, d] output (may be the buffer ln['y'] points to).  q: [M, d] query rows instead of wq (attention + output projection
    only: a self-attention block after the product that wrote its q | k | v row)."""
    a = _lib.CrossRowArgs()
    src = ln['y'] if ln is not None else x
    M = src.shape[0] if M is None else M
    d = wo.N
    if ln is not None:
        yy, xr, lo = ln['y'], ln.get('x'), ln.get('out')
        _check_f32(yy, 'ln.y')
        a.ln_y, a.ln_ldy = yy.data_ptr(), yy.stride(0)
        if xr is not None:
            _check_f32(xr, 'ln.x')
            a.ln_x, a.ln_ldx = xr.data_ptr(), xr.stride(0)
        if lo is not None:
            _check_f32(lo, 'ln.out')
            a.ln_out, a.ln_ldo = lo.data_ptr(), lo.stride(0)
        if ln.get('ln_a') is not None:
            a.ln_ga, a.ln_ba = ln['ln_a'][0].data_ptr(), ln['ln_a'][1].data_ptr()
        if ln.get('ln_b') is not None:
            a.ln_gb, a.ln_bb = ln['ln_b'][0].data_ptr(), ln['ln_b'][1].data_ptr()
        a.ln_act, a.ln_eps = ACT[ln.get('act')], ln.get('eps', 1e-6)
    else:
        _check_f32(x, 'x')
        a.x, a.ldx = x.data_ptr(), x.stride(0)
    for t, n in ((k, 'k'), (v, 'v'), (y, 'y')):
        _check_f32(t, n)
    if q is not None:
        _check_f32(q, 'q')
        a.q, a.ldq = q.data_ptr(), q.stride(0)
    else:
        a.wq_hi, a.wq_lo, a.ldwq, a.bq = wq.w_hi.data_ptr(), wq.w_lo.data_ptr(), wq.ldw, _ptr(wq.b1)
    a.wo_hi, a.wo_lo, a.ldwo, a.bo = wo.w_hi.data_ptr(), wo.w_lo.data_ptr(), wo.ldw, _ptr(wo.b1)
    a.k, a.v, a.ldk, a.ldv, a.kv_batch_stride = k.data_ptr(), v.data_ptr(), k.stride(0), v.stride(0), 0
    a.key_mask = _ptr(key_mask)
    Lk = k.shape[0]
    if attn is not None:
        _check_f32(attn, 'attn')
        a.attn, a.attn_row_stride = attn.data_ptr(), attn.stride(0)
    a.y, a.ldy = y.data_ptr(), y.stride(0)
    a.M, a.d, a.H, a.Lk = M, d, heads, Lk
    ws, cnt = _ws_override if _ws_override is not None else gemm_workspace(y.device)   # the heads' shares of y meet here
    a.ws, a.ws_bytes, a.counters, a.n_counters = ws.data_ptr(), ws.numel() * 4, cnt.data_ptr(), cnt.numel()
    a._keep = (x, ln, k, v, y, key_mask, attn, wq, wo, ws, cnt, q)
    if _program is not None:
        _program.append(('cross_attention_row', a, a._keep))
        return y
    check(_lib.load().estts_cross_attention_row(a, _stream()), 'estts_cross_attention_row')
    nw = 2 if q is not None else 3     # d x d weight panels read
    _record('cross_attention_row', {'M': M, 'd': d, 'H': heads, 'Lk': Lk, 'projected': q is None,
                                    'flops': 2.0 * M * (nw * d * d + 2 * Lk * d),
                                    'bytes': 4.0 * (nw * d * d + 2 * Lk * d) + 4.0 * M * (3 * d + heads * Lk)},
            _lib.load().estts_cross_attention_row, a, keep=a._keep)
    return y


def can_prenet_row(lin1, lin2, C, precision='bf16x3'):
    """True when prenet_row() can take the pre-net: at most 128 input channels, two plain dense layers of 256 units with ReLU,
    split operands."""
    plain = all(l.taps == 1 and l.act == ACT['relu'] and l.s1 is None and l.s2 is None for l in (lin1, lin2))
    return (plain and precision_at(precision, None) == 'bf16x3' and C % 4 == 0 and 0 < C <= 128 and (lin1.N, lin1.K) == (256, C) and
            (lin2.N, lin2.K) == (256, 256))


def prenet_row(x, mask, lin1, lin2, out, alpha=1.0, residual=None, dropout=None):
    """The decoder's pre-net for one position per row of x in one launch (estts_prenet_row; see can_prenet_row): mask[m] = the
    frame is all zeros, out = alpha * drop(relu(drop(relu(x W1 + b1)) W2 + b2)) + residual.  dropout = (rate, seed, site of the
    first layer, site of the second, row0) with the generator of gemm(dropout=...)."""
    a = _lib.PrenetRowArgs()
    _check_f32(x, 'x')
    _check_f32(out, 'out')
    a.x, a.ldx, a.C = x.data_ptr(), x.stride(0), x.shape[1]
    a.mask = _ptr(mask)
    a.w1_hi, a.w1_lo, a.ldw1, a.b1, a.N1 = lin1.w_hi.data_ptr(), lin1.w_lo.data_ptr(), lin1.ldw, _ptr(lin1.b1), lin1.N
    a.w2_hi, a.w2_lo, a.ldw2, a.b2, a.N2 = lin2.w_hi.data_ptr(), lin2.w_lo.data_ptr(), lin2.ldw, _ptr(lin2.b1), lin2.N
    a.alpha = alpha
    if residual is not None:
        _check_f32(residual, 'residual')
        a.R, a.ldr = residual.data_ptr(), residual.stride(0)
    if dropout is not None and dropout[0] > 0.0:
        a.drop_rate, a.drop_seed, a.drop_site1, a.drop_site2, a.drop_row0 = dropout
    a.out, a.ldo, a.M = out.data_ptr(), out.stride(0), x.shape[0]
    a._keep = (x, mask, lin1, lin2, out, residual)
    if _program is not None:
        _program.append(('prenet_row', a, a._keep))
        return out
    check(_lib.load().estts_prenet_row(a, _stream()), 'estts_prenet_row')
    _record('prenet_row', {'M': a.M, 'C': a.C, 'flops': 2.0 * a.M * 256 * (a.C + 256), 'bytes': 4.0 * 256 * (a.C + 256) + 4.0 * a.M * (a.C + 513)},
            _lib.load().estts_prenet_row, a, keep=a._keep)
    return out


_block_counters = {}


def _decoder_block_counters(device):
    """The barrier counters of decoder_block_row (per device; zero at allocation, every launch leaves them ready for the next)."""
    key = torch.device(device).index if torch.device(device).index is not None else torch.cuda.current_device()
    c = _block_counters.get(key)
    if c is None:
        c = _block_counters[key] = torch.zeros(128, dtype=torch.int32, device=torch.device('cuda', key))
    return c


def can_decoder_block_row(wqkv, wos, wqc, woc, wff, d, heads, max_pos, Lk, precision='bf16x3'):
    """True when decoder_block_row() can take the block: width 256, 4 heads, at most 256 decoder and encoder positions, split
    operands everywhere (also at the feed-forward site), plain dense layers (bias only), the feed-forward pair folded into wff."""
    lins = (wqkv, wos, wqc, woc, wff)
    plain = all(l is not None and l.taps == 1 and l.act == 0 and l.s1 is None and l.s2 is None for l in lins)
    return (plain and precision_at(precision, None) == 'bf16x3' and precision_at(precision, 'ffn') == 'bf16x3' and d == 256 and heads == 4 and
            0 < max_pos <= 256 and 0 < Lk <= 256 and (wqkv.N, wqkv.K) == (3 * d, d) and (wos.N, wos.K) == (d, 2 * d) and
            (wqc.N, wqc.K) == (d, d) and (woc.N, woc.K) == (d, 2 * d) and (wff.N, wff.K) == (d, d))


def decoder_block_row(wqkv, wos, sa_ln, sa_last_ln, wqc, woc, ca_ln, wff, cache, pos, k, v, y2, x2, x=None, ln=None, self_mask=None,
                      key_mask=None, attn=None, eps=1e-6):
    """One decoder position per utterance (the M rows of x, or of the pending normalisation ln = dict(y=, x=, ln_a=, act=, ln_b=)
    as in gemm(ln=...)) through self-attention, cross-attention and the folded feed-forward product of a block in one launch
    (estts_decoder_block_row; see can_decoder_block_row).  cache: [S, 3 d] rows q | k | v of the decoder positions ([M, S, 3 d]
    for M > 1), row `pos` is written; k / v: [Lk, >= d] encoder keys / values ([M, Lk, >= d] views of one tensor for M > 1);
    self_mask: [M, > pos], key_mask: [M, Lk] contiguous, attn: [M, heads * Lk] view or None.  Writes y2 (the product) and x2
    (its residual), [M, d]: the block's output is LN(relu(LN(y2)) + x2), left to the consumer."""
    a = _lib.DecoderBlockArgs()
    d = wos.N
    if ln is not None:
        yy, xr = ln['y'], ln.get('x')
        _check_f32(yy, 'ln.y')
        a.ln_y, a.ln_ldy = yy.data_ptr(), yy.stride(0)
        if xr is not None:
            _check_f32(xr, 'ln.x')
            a.ln_x, a.ln_ldx = xr.data_ptr(), xr.stride(0)
        if ln.get('ln_a') is not None:
            a.ln_ga, a.ln_ba = ln['ln_a'][0].data_ptr(), ln['ln_a'][1].data_ptr()
        if ln.get('ln_b') is not None:
            a.ln_gb, a.ln_bb = ln['ln_b'][0].data_ptr(), ln['ln_b'][1].data_ptr()
        a.ln_act = ACT[ln.get('act')]
    else:
        _check_f32(x, 'x')
        a.x, a.ldx = x.data_ptr(), x.stride(0)
    a.ln_eps = eps
    for t, n in ((cache, 'cache'), (k, 'k'), (v, 'v'), (y2, 'y2'), (x2, 'x2')):
        _check_f32(t, n)
    a.wqkv_hi, a.wqkv_lo, a.ldwqkv, a.bqkv = wqkv.w_hi.data_ptr(), wqkv.w_lo.data_ptr(), wqkv.ldw, _ptr(wqkv.b1)
    M = y2.shape[0]
    if M > 64 or (M > 1 and (cache.dim() != 3 or k.dim() != 3 or v.dim() != 3 or k.stride(0) != v.stride(0))):
        raise ValueError('decoder_block_row: at most 64 rows; with more than one, cache / k / v carry a leading batch dimension')
    if not 0 <= pos < cache.shape[-2] or cache.shape[-1] < 3 * d or (self_mask is not None and self_mask.shape[-1] <= pos):
        raise ValueError(f'decoder_block_row: position {pos} needs a cache of more than {pos} rows of {3 * d} columns and a self_mask '
                         f'of more than {pos} columns')
    a.qkv, a.ldqkv, a.qkv_batch_stride, a.pos = cache.data_ptr(), cache.stride(-2), cache.stride(0) if cache.dim() == 3 else 0, pos
    if self_mask is not None:
        a.self_mask, a.self_mask_stride = self_mask.data_ptr(), self_mask.stride(0)
    a.wos_hi, a.wos_lo, a.ldwos, a.bos = wos.w_hi.data_ptr(), wos.w_lo.data_ptr(), wos.ldw, _ptr(wos.b1)
    a.sa_ga, a.sa_ba, a.sa_gb, a.sa_bb = (sa_ln[0].data_ptr(), sa_ln[1].data_ptr(), sa_last_ln[0].data_ptr(), sa_last_ln[1].data_ptr())
    a.wqc_hi, a.wqc_lo, a.ldwqc, a.bqc = wqc.w_hi.data_ptr(), wqc.w_lo.data_ptr(), wqc.ldw, _ptr(wqc.b1)
    a.k, a.v, a.ldk, a.ldv, a.kv_batch_stride, a.Lk = (k.data_ptr(), v.data_ptr(), k.stride(-2), v.stride(-2),
                                                       k.stride(0) if k.dim() == 3 else 0, k.shape[-2])
    if key_mask is not None and M > 1 and not (key_mask.is_contiguous() and tuple(key_mask.shape) == (M, k.shape[-2])):
        raise ValueError('decoder_block_row: key_mask must be a contiguous [M, Lk] tensor')
    a.key_mask = _ptr(key_mask)
    if attn is not None:
        _check_f32(attn, 'attn')
        a.attn, a.attn_row_stride = attn.data_ptr(), attn.stride(0)
    a.woc_hi, a.woc_lo, a.ldwoc, a.boc = woc.w_hi.data_ptr(), woc.w_lo.data_ptr(), woc.ldw, _ptr(woc.b1)
    a.ca_g, a.ca_b = ca_ln[0].data_ptr(), ca_ln[1].data_ptr()
    a.wff_hi, a.wff_lo, a.ldwff, a.bff = wff.w_hi.data_ptr(), wff.w_lo.data_ptr(), wff.ldw, _ptr(wff.b1)
    a.y2, a.ldy2, a.x2, a.ldx2 = y2.data_ptr(), y2.stride(0), x2.data_ptr(), x2.stride(0)
    ws = (_ws_override if _ws_override is not None else gemm_workspace(y2.device))[0]
    cnt = _decoder_block_counters(y2.device)
    a.ws, a.ws_bytes, a.counters, a.n_counters = ws.data_ptr(), ws.numel() * 4, cnt.data_ptr(), cnt.numel()
    a.M, a.d, a.H = M, d, 4
    a._keep = (x, ln, cache, k, v, y2, x2, self_mask, key_mask, attn, wqkv, wos, sa_ln, sa_last_ln, wqc, woc, ca_ln, wff, ws, cnt)
    if _program is not None:
        _program.append(('decoder_block_row', a, a._keep))
        return y2
    check(_lib.load().estts_decoder_block_row(a, _stream()), 'estts_decoder_block_row')
    Lk = k.shape[-2]
    _record('decoder_block_row', {'M': M, 'd': d, 'H': 4, 'Lk': Lk, 'pos': pos, 'flops': 2.0 * (8 * d * d + 2 * (Lk + pos + 1) * d),
                                  'bytes': 4.0 * (8 * d * d + 2 * (Lk + pos + 1) * d) + 4.0 * (6 * d + 4 * Lk)},
            _lib.load().estts_decoder_block_row, a, keep=a._keep)
    return y2


def rownorm_args(y, out, ln_a=None, act=None, residual=None, ln_b=None, eps=1e-6, M=None, D=None,
                 dot_w=None, dot_b=0.0, dot_scale=1.0, dot_mask=None, dot_out=None, dropout=None):
    """Fills the argument block of estts_rownorm (also embedded in the backward call)."""
    _check_f32(y, 'y')
    a = RownormArgs()
    a.y, a.ldy = y.data_ptr(), y.stride(0)
    if residual is not None:
        a.x, a.ldx = residual.data_ptr(), residual.stride(0)
    if ln_a is not None:
        a.ga, a.ba = ln_a[0].data_ptr(), ln_a[1].data_ptr()
    if ln_b is not None:
        a.gb, a.bb = ln_b[0].data_ptr(), ln_b[1].data_ptr()
    a.act, a.eps = ACT[act], eps
    if out is not None:
        a.out, a.ldo = out.data_ptr(), out.stride(0)
    a.M = y.shape[0] if M is None else M
    a.D = y.shape[1] if D is None else D
    if dot_w is not None:
        a.dot_w, a.dot_scale = dot_w.data_ptr(), dot_scale
        if torch.is_tensor(dot_b):
            a.dot_b_ptr = dot_b.data_ptr()
        else:
            a.dot_b = dot_b
        a.dot_mask, a.dot_out = _ptr(dot_mask), _ptr(dot_out)
    if dropout is not None and dropout[0] > 0.0:
        a.drop_rate, a.drop_seed, a.drop_site = dropout
    # the block holds raw device pointers: keep their tensors alive for as long as the block is (a backward pass
    # re-reads y and the residual long after the forward call returned)
    a._keep = (y, out, residual, ln_a, ln_b, dot_w, dot_b, dot_mask, dot_out)
    return a


def rownorm(y, out, ln_a=None, act=None, residual=None, ln_b=None, eps=1e-6, M=None, D=None,
            dot_w=None, dot_b=0.0, dot_scale=1.0, dot_mask=None, dot_out=None, dropout=None):
    """out = LNb(dropout(act(LNa(y))) + residual); ln_a / ln_b are (gamma, beta) tensor pairs or None;
    dropout = (rate, seed, site) or None."""
    a = rownorm_args(y, out, ln_a, act, residual, ln_b, eps, M, D, dot_w, dot_b, dot_scale, dot_mask, dot_out, dropout)
    if _program is not None:
        _program.append(('rownorm', a, a._keep))
        return out
    check(_lib.load().estts_rownorm(a, _stream()), 'estts_rownorm')
    _record('rownorm', {'M': a.M, 'D': a.D, 'flops': 0.0,
                        'bytes': 4.0 * a.M * a.D * (1 + (residual is not None) + (out is not None))},
            _lib.load().estts_rownorm, a, keep=(y, out, residual, ln_a, ln_b, dot_w, dot_mask, dot_out))
    return out


def embed_posenc(tokens, emb, pe, pos_scalar, out, key_mask, scale, pe_stride=1):
    """tokens int32 [B,N]; emb [V,D]; pe [P,D]; out [B*N, >=D] view; key_mask [B,N] or None."""
    if tokens.dtype != torch.int32:
        raise TypeError('tokens must be int32')
    B, N = tokens.shape
    D = emb.shape[1]
    if N * pe_stride > pe.shape[0] + pe_stride - 1:
        raise ValueError(f'sequence length {N} exceeds the positional table ({pe.shape[0]})')
    check(_lib.load().estts_embed_posenc(tokens.data_ptr(), emb.data_ptr(), pe.data_ptr(), _ptr(pos_scalar),
                                         out.data_ptr(), out.stride(0), _ptr(key_mask), B, N, D, scale, pe_stride,
                                         _stream()), 'estts_embed_posenc')
    _record('embed_posenc', {'B': B, 'N': N, 'D': D, 'flops': 0.0, 'bytes': 12.0 * B * N * D},
            _lib.load().estts_embed_posenc, tokens.data_ptr(), emb.data_ptr(), pe.data_ptr(), _ptr(pos_scalar),
            out.data_ptr(), out.stride(0), _ptr(key_mask), B, N, D, scale, pe_stride,
            keep=(tokens, emb, pe, pos_scalar, out, key_mask))
    return out


def durations_cumsum(dur, cum, totals):
    """dur float32 [B,N] -> cum int32 [B,N] (inclusive prefix sums of round-half-even), totals int32 [B]."""
    B, N = dur.shape
    check(_lib.load().estts_durations_cumsum(dur.data_ptr(), cum.data_ptr(), totals.data_ptr(), B, N, _stream()),
          'estts_durations_cumsum')
    _record('durations_cumsum', {'B': B, 'N': N, 'flops': 0.0, 'bytes': 8.0 * B * N},
            _lib.load().estts_durations_cumsum, dur.data_ptr(), cum.data_ptr(), totals.data_ptr(), B, N,
            keep=(dur, cum, totals))


def expand(x, cum, out, frame_mask, B, N, T, D):
    check(_lib.load().estts_expand(x.data_ptr(), x.stride(0), cum.data_ptr(), out.data_ptr(), out.stride(0),
                                   _ptr(frame_mask), B, N, T, D, _stream()), 'estts_expand')
    _record('expand', {'B': B, 'N': N, 'T': T, 'D': D, 'flops': 0.0, 'bytes': 8.0 * B * T * D},
            _lib.load().estts_expand, x.data_ptr(), x.stride(0), cum.data_ptr(), out.data_ptr(), out.stride(0),
            _ptr(frame_mask), B, N, T, D, keep=(x, cum, out, frame_mask))
    return out


def conv2d_c1(x, w9, scale, bias, out, stride, pad_t, pad_l):
    """x [B,H,W] (one input channel), w9 [9, Cout], out [B,Ho,Wo,Cout] NHWC: conv + folded BN + ReLU."""
    B, H, W = x.shape
    _, Ho, Wo, Cout = out.shape
    check(_lib.load().estts_conv2d_c1(x.data_ptr(), w9.data_ptr(), scale.data_ptr(), bias.data_ptr(), out.data_ptr(), B, H, W,
                                      Ho, Wo, Cout, stride, pad_t, pad_l, _stream()), 'estts_conv2d_c1')
    _record('conv2d_c1', {'B': B, 'H': H, 'W': W, 'Cout': Cout, 'stride': stride, 'flops': 18.0 * B * Ho * Wo * Cout,
                          'bytes': 4.0 * B * (H * W + Ho * Wo * Cout)},
            _lib.load().estts_conv2d_c1, x.data_ptr(), w9.data_ptr(), scale.data_ptr(), bias.data_ptr(), out.data_ptr(), B, H, W,
            Ho, Wo, Cout, stride, pad_t, pad_l, keep=(x, w9, scale, bias, out))
    return out


def gru_gates(gi, gh, h, out, x_res=None, M=None, H=None):
    """GRU cell update from the two gate pre-activations (gate order r, z, n): h is replaced by the new state and
    out = new state (+ x_res).  gi, gh: [M, >=3H] views; h, out, x_res: [M, >=H] views."""
    M = h.shape[0] if M is None else M
    H = h.shape[1] if H is None else H
    args = (gi.data_ptr(), gi.stride(0), gh.data_ptr(), gh.stride(0), h.data_ptr(), h.stride(0), _ptr(x_res),
            0 if x_res is None else x_res.stride(0), out.data_ptr(), out.stride(0), M, H)
    check(_lib.load().estts_gru_gates(*args, _stream()), 'estts_gru_gates')
    _record('gru_gates', {'M': M, 'H': H, 'flops': 0.0, 'bytes': 4.0 * M * H * (8 + (x_res is not None))},
            _lib.load().estts_gru_gates, *args, keep=(gi, gh, h, out, x_res))
    return out


GRU_FUSED_MAX_H = 1536          # estts_gru_fused_fwd: H <= 1536; estts_gru_fused_bwd: 3 H <= 1536 (48 K-steps of 32 per launch)


def gru_step(gi, gh, h_prev, lin, h_out, precision='bf16x3', site=None):
    """One GRU step in one launch (estts_gru_fused_fwd): gh = h_prev (*) lin (W_hh [3H x H] with its bias; stored), gates in the
    order r, z, n, h_out = the new state.  h_out must be another buffer than h_prev (other workgroups still read it)."""
    precision = precision_at(precision, site)
    a = _lib.GruFusedArgs()
    a.gi, a.ldgi, a.gh, a.ldgh = gi.data_ptr(), gi.stride(0), gh.data_ptr(), gh.stride(0)
    a.h_prev, a.ldhp = h_prev.data_ptr(), h_prev.stride(0)
    a.W_hi, a.W_lo, a.ldw = lin.w_hi.data_ptr(), lin.w_lo.data_ptr(), lin.ldw
    a.bias = _ptr(lin.b1)
    a.h_out, a.ldho = h_out.data_ptr(), h_out.stride(0)
    a.M, a.H = h_prev.shape
    a.precision = PRECISION[precision]
    if (lin.N, lin.K) != (3 * a.H, a.H) or lin.s1 is not None:
        raise ValueError('gru_step: the weight must be W_hh [3H x H] without an output scale')
    if a.H > GRU_FUSED_MAX_H or a.H % 16:
        # beyond what one launch holds (a workgroup keeps a whole K = H row block in flight): the product, then the gates
        gemm(h_prev, lin, gh, precision=precision)
        return gru_gates(gi, gh, h_prev, h_out)
    check(_lib.load().estts_gru_fused_fwd(a, _stream()), 'estts_gru_fused_fwd')
    _record('gru_step', {'M': a.M, 'H': a.H, 'flops': 6.0 * a.M * a.H * a.H, 'bytes': 4.0 * a.M * a.H * 9 + 12.0 * a.H * a.H},
            _lib.load().estts_gru_fused_fwd, a, keep=(gi, gh, h_prev, lin, h_out))
    return h_out


def row_zero_mask(x, mask, M=None, D=None):
    M = x.shape[0] if M is None else M
    D = x.shape[1] if D is None else D
    if _program is not None:
        _program.append(('row_zero_mask', _lib.RowmaskArgs(x.data_ptr(), x.stride(0), mask.data_ptr(), M, D), (x, mask)))
        return mask
    check(_lib.load().estts_row_zero_mask(x.data_ptr(), x.stride(0), mask.data_ptr(), M, D, _stream()), 'estts_row_zero_mask')
    _record('row_zero_mask', {'M': M, 'D': D, 'flops': 0.0, 'bytes': 4.0 * M * (D + 1)},
            _lib.load().estts_row_zero_mask, x.data_ptr(), x.stride(0), mask.data_ptr(), M, D, keep=(x, mask))
    return mask


def masked_mae(target, pred, out):
    """target / pred: 2-D [rows, C] float32 views; out: zeroed 1-element float32 tensor (accumulated)."""
    M, C = target.shape
    check(_lib.load().estts_masked_mae(target.data_ptr(), target.stride(0), pred.data_ptr(), pred.stride(0), out.data_ptr(),
                                       M, C, _stream()), 'estts_masked_mae')
    return out


def same_pad(n, k, stride):
    """TensorFlow 'SAME': output size and low-side padding for extent n."""
    out = -(-n // stride)
    total = max((out - 1) * stride + k - n, 0)
    return out, total // 2


def positional_encoding_table(position, model_dim, device):
    """Sinusoidal table, float64 angles rounded to float32 (TransformerTTS/model/transformer_utils.py:5-21)."""
    pos = np.arange(position, dtype=np.float64)[:, None]
    i = np.arange(model_dim)[None, :]
    rates = 1.0 / np.power(10000.0, (2 * (i // 2)) / np.float32(model_dim))
    ang = pos * rates
    ang[:, 0::2] = np.sin(ang[:, 0::2])
    ang[:, 1::2] = np.cos(ang[:, 1::2])
    return torch.from_numpy(ang.astype(np.float32)).to(device)


def fold_batchnorm(gamma, beta, mean, var, eps=1e-3):
    """Inference BatchNormalization as y = x * s + b (Keras epsilon 1e-3)."""
    s = np.asarray(gamma, np.float64) / np.sqrt(np.asarray(var, np.float64) + eps)
    b = np.asarray(beta, np.float64) - np.asarray(mean, np.float64) * s
    return s.astype(np.float32), b.astype(np.float32)
