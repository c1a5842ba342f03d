"""Host-side wrappers of the training kernels (the "Training" block of include/estts.h): torch tensors in,
HIP launches out.  No torch op computes anything here; torch provides memory, views and the stream."""
import torch

from . import _lib, ops
from ._lib import (ACT, AttentionBwdArgs, ColreduceArgs, RownormBwdArgs, WgradArgs, check)
from .ops import PRECISION, _ceil, _ptr, _stream


class TrainLinear:
    """A dense / conv1d layer whose float32 master weight lives in a caller-owned buffer (Keras layout: kernel
    [K = taps * cin, N], row-major view; bias [N] or None).  Holds the two packed bf16 operands estts_gemm needs:
    `fwd` (y = x W) and `bwd` (dx = dy W^T, taps flipped for convolutions); repack() refreshes them from the master
    after an optimizer step.  `fwd` / `bwd` quack like ops.PackedLinear."""

    class _Packed:
        pass

    def __init__(self, kernel, bias, taps=1, lpad=0, act=None, need_fwd=True, need_bwd=True):
        if kernel.dim() != 2 or kernel.stride(1) != 1 or kernel.dtype != torch.float32:
            raise ValueError('kernel must be a float32 [K, N] view with contiguous rows')
        self.kernel, self.bias = kernel, bias
        self.K, self.N = kernel.shape
        if self.K % taps or (self.K // taps) % 4:
            raise ValueError('K must be taps * cin with cin a multiple of 4')
        self.taps, self.lpad = taps, lpad
        self.cin = self.K // taps
        dev = kernel.device
        f, b = self._Packed(), self._Packed()
        f.N, f.K, f.taps, f.lpad, f.c_per_tap = self.N, self.K, taps, lpad, self.cin
        b.N, b.K, b.taps, b.lpad, b.c_per_tap = self.cin, taps * self.N, taps, taps - 1 - lpad, self.N
        for pk, act_, need in ((f, act, need_fwd), (b, None, need_bwd)):
            pk.act = ACT[act_]
            pk.ldw = _ceil(pk.K, 64)
            n = _ceil(pk.N, 64) * pk.ldw if need else 0
            pk.w_hi = torch.empty(n, dtype=torch.int16, device=dev)
            pk.w_lo = torch.empty(n, dtype=torch.int16, device=dev)
            pk.s1 = pk.s2 = pk.b2 = None
            pk.b1 = None
        f.b1 = bias
        self.fwd, self.bwd = (f if need_fwd else None), (b if need_bwd else None)
        if need_bwd and self.N % 4:
            raise ValueError('the data gradient needs N % 4 == 0')
        self.repack()

    def repack(self):
        lib = _lib.load()
        k = self.kernel
        for mode, pk in ((0, self.fwd), (1, self.bwd)):
            if pk is not None:
                check(lib.estts_pack_weight(k.data_ptr(), k.stride(0), self.K, self.N, self.taps, mode, pk.w_hi.data_ptr(),
                                            pk.w_lo.data_ptr(), _stream()), 'estts_pack_weight')


class PackPlan:
    """Re-packs every operand of a list of TrainLinear layers with one launch (estts_pack_weights): the job table is
    built once and lives on the device; the master weights and the packed buffers must not be re-allocated afterwards."""

    def __init__(self, linears, device):
        import ctypes
        jobs, unit = [], 0
        for lin in linears:
            for mode, pk in ((0, lin.fwd), (1, lin.bwd)):
                if pk is None:
                    continue
                rows = lin.N if mode == 0 else lin.K // lin.taps
                cols = lin.K if mode == 0 else lin.taps * lin.N
                j = _lib.PackJob(lin.kernel.data_ptr(), pk.w_hi.data_ptr(), pk.w_lo.data_ptr(), unit, lin.kernel.stride(0), lin.K,
                                 lin.N, lin.taps, mode, 0)
                jobs.append(j)
                unit += (_ceil(rows, 64) // 16) * (_ceil(cols, 64) // 32) * 64
        self.n, self.total = len(jobs), unit
        arr = (_lib.PackJob * self.n)(*jobs)
        host = torch.frombuffer(bytearray(ctypes.string_at(ctypes.addressof(arr), ctypes.sizeof(arr))), dtype=torch.uint8)
        self.table = host.to(device)
        self._keep = linears

    def run(self):
        check(_lib.load().estts_pack_weights(self.table.data_ptr(), self.n, self.total, _stream()), 'estts_pack_weights')


def wgrad(x, dy, dw, taps=1, lpad=0, M=None, seg_len=0, precision='bf16x3', splits=0, db=None):
    """dw[K, N] += conv-unfolded(x)^T dy (float atomics; dw must be zero or hold a partial sum); db[N] += column sums
    of dy (the bias gradient) when given."""
    a = WgradArgs()
    a.A, a.lda = x.data_ptr(), x.stride(0)
    K, N = dw.shape
    a.seg_len, a.taps, a.c_per_tap, a.tap_lpad = seg_len, taps, K // taps, lpad
    a.dY, a.lddy = dy.data_ptr(), dy.stride(0)
    a.M, a.N, a.K = (x.shape[0] if M is None else M), N, K
    a.dW, a.lddw = dw.data_ptr(), dw.stride(0)
    a.precision, a.splits = PRECISION[precision], splits
    a.db = _ptr(db)
    check(_lib.load().estts_gemm_wgrad(a, _stream()), 'estts_gemm_wgrad')
    ops._record('wgrad', {'M': a.M, 'N': N, 'K': K, 'taps': taps, 'flops': 2.0 * a.M * N * K,
                          'bytes': 4.0 * a.M * (K / taps + N) + 4.0 * N * K},
                _lib.load().estts_gemm_wgrad, a, keep=(x, dy, dw, db))


def attention_bwd(q, k, v, ctx, dctx, attn, delta, dq, dk, dv, B, H, Lq, Lk, depth, precision='bf16x3'):
    a = AttentionBwdArgs()
    a.q, a.ldq, a.k, a.ldk, a.v, a.ldv = q.data_ptr(), q.stride(0), k.data_ptr(), k.stride(0), v.data_ptr(), v.stride(0)
    a.ctx, a.ldo, a.dctx, a.lddo = ctx.data_ptr(), ctx.stride(0), dctx.data_ptr(), dctx.stride(0)
    a.attn, a.delta = attn.data_ptr(), delta.data_ptr()
    a.dq, a.lddq, a.dk, a.lddk, a.dv, a.lddv = (dq.data_ptr(), dq.stride(0), dk.data_ptr(), dk.stride(0), dv.data_ptr(),
                                                dv.stride(0))
    a.B, a.H, a.Lq, a.Lk, a.depth, a.precision = B, H, Lq, Lk, depth, PRECISION[precision]
    if tuple(attn.shape) != (B, H, Lq, Lk) or not attn.is_contiguous() or delta.numel() < B * H * Lq:
        raise ValueError('attention_bwd: attn must be contiguous [B,H,Lq,Lk], delta at least B*H*Lq floats')
    check(_lib.load().estts_attention_bwd(a, _stream()), 'estts_attention_bwd')
    ops._record('attention_bwd', {'B': B, 'H': H, 'Lq': Lq, 'Lk': Lk, 'depth': depth, 'flops': 10.0 * B * H * Lq * Lk * depth,
                                  'bytes': 4.0 * B * H * (2.0 * Lq * Lk + depth * (4 * Lq + 4 * Lk))},
                _lib.load().estts_attention_bwd, a, keep=(q, k, v, ctx, dctx, attn, delta, dq, dk, dv))


def rownorm_bwd(fwd_args, dout=None, ddot=None, dy=None, dx=None, dx_accumulate=False, dln_a=None, dln_b=None,
                ddot_w=None, ddot_b=None):
    """fwd_args: the RownormArgs of the forward call (ops.rownorm_args); dln_a / dln_b: (dgamma, dbeta) tensors."""
    a = RownormBwdArgs()
    a.f = fwd_args
    if dout is not None:
        a.dout, a.lddout = dout.data_ptr(), dout.stride(0)
    a.ddot = _ptr(ddot)
    if dy is not None:
        a.dy, a.lddy = dy.data_ptr(), dy.stride(0)
    if dx is not None:
        a.dx, a.lddx, a.dx_accumulate = dx.data_ptr(), dx.stride(0), int(dx_accumulate)
    if dln_a is not None:
        a.dga, a.dba = dln_a[0].data_ptr(), dln_a[1].data_ptr()
    if dln_b is not None:
        a.dgb, a.dbb = dln_b[0].data_ptr(), dln_b[1].data_ptr()
    a.ddot_w, a.ddot_b = _ptr(ddot_w), _ptr(ddot_b)
    ws = colreduce_workspace(fwd_args._keep[0].device)      # both kernels leave it zero, calls on one stream may share it
    a.ws, a.ws_bytes = ws.data_ptr(), ws.numel() * 4
    check(_lib.load().estts_rownorm_bwd(a, _stream()), 'estts_rownorm_bwd')


_colreduce_ws = {}


def colreduce_workspace(device):
    """Per-device scratch over which colreduce spreads its atomics (zero between calls; calls on one stream may share it)."""
    key = torch.device(device).index if torch.device(device).index is not None else torch.cuda.current_device()
    ws = _colreduce_ws.get(key)
    if ws is None:
        ws = _colreduce_ws[key] = torch.zeros(16 * 2 * 8192 + 64, dtype=torch.float32, device=torch.device('cuda', key))
    return ws


class SumSlots:
    """Column sums that stay spread over 16 copies between the launch that builds them and the launch that consumes them
    (estts_colreduce with defer, estts_gemm with stat_slots -> estts_bn_finalize_slots / estts_bn_bwd_slots), one set of
    regions per device; every launch involved must run on one stream.
    Region 2 holds forward moments: estts_bn_finalize_slots clears what it reads.  Regions 0 and 1 take turns for the
    BatchNorm backward sums: estts_bn_bwd_slots reads a region from many workgroups and cannot clear it, so the next
    deferred reduction, which writes the other region, clears it on the side."""
    MAX_C = 4096
    _per_device = {}

    def __init__(self, device):
        self.buf = torch.zeros(3, 16 * 2 * self.MAX_C, dtype=torch.float32, device=device)
        self.used = [0, 0]       # floats of regions 0 / 1 that may be non-zero
        self.turn = 0

    @classmethod
    def on(cls, device):
        dev = torch.device(device)
        key = dev.index if dev.index is not None else torch.cuda.current_device()
        if key not in cls._per_device:
            cls._per_device[key] = cls(torch.device('cuda', key))
        return cls._per_device[key]

    @classmethod
    def step_begin(cls, device, owner):
        """Called by a trainer at the start of train_step: if the previous step on this device did not reach step_end (an
        exception between a producer and its consumer), sums may sit in the regions with nobody left to clear them, so all
        regions are zeroed before the new step adds to them."""
        s = cls.on(device)
        if getattr(s, 'open_step', None) is not None:
            s.buf.zero_()
            s.used, s.turn = [0, 0], 0
        s.open_step = owner

    @classmethod
    def step_end(cls, device):
        cls.on(device).open_step = None

    @classmethod
    def reset_all(cls):
        """Zero every region (after launches that left sums nobody consumed, e.g. a producer replayed on its own for timing)."""
        for s in cls._per_device.values():
            s.buf.zero_()
            s.used, s.turn = [0, 0], 0
            s.open_step = None

    @staticmethod
    def stride(C):
        return (C + 255) // 256 * 256

    @staticmethod
    def _not_under_capture():
        # the turn / used bookkeeping lives in Python: it advances when a launch is issued, not when a captured graph is
        # replayed, so a replayed sequence with an odd number of deferred reductions would add into a region nobody cleared
        if torch.cuda.is_current_stream_capturing():
            raise RuntimeError('SumSlots: deferred column sums cannot be recorded into a hipGraph (their clear / turn bookkeeping '
                               'is host-side); capture with fold_sums = False or keep these launches outside the graph')

    def forward_region(self, C):
        """(pointer, stride) of the zeroed region a producer of forward moments for C columns writes."""
        self._not_under_capture()
        if C > self.MAX_C:
            raise ValueError(f'SumSlots: at most {self.MAX_C} columns')
        return self.buf[2].data_ptr(), self.stride(C)

    def backward_region(self, C):
        """(pointer, stride, clear pointer, clear floats) for the next deferred mode-2 reduction over C columns."""
        self._not_under_capture()
        if C > self.MAX_C:
            raise ValueError(f'SumSlots: at most {self.MAX_C} columns')
        i, o = self.turn, 1 - self.turn
        out = (self.buf[i].data_ptr(), self.stride(C), self.buf[o].data_ptr() if self.used[o] else None, self.used[o])
        self.used[i], self.used[o], self.turn = 16 * 2 * self.stride(C), 0, o
        return out


def colreduce(mode, out0=None, out1=None, z=None, g=None, a=None, act=None, mean=None, rstd=None, dy_out=None, M=None, C=None,
              scale=None, shift=None, defer=False):
    """defer=True (modes 0 / 2): the sums stay spread over 16 copies (SumSlots) and the call returns (pointer, stride) for
    bn_finalize(slots=...) / bn_bwd(slots=...); out0 / out1 are not used.  In mode 0 `shift` [C] is then what the moments
    are taken about (zero when None; z[0] is the usual choice)."""
    r = ColreduceArgs()
    ref = z if z is not None else g
    r.M = ref.shape[0] if M is None else M
    r.C = ref.shape[1] if C is None else C
    if z is not None:
        r.z, r.ldz = z.data_ptr(), z.stride(0)
    if g is not None:
        r.g, r.ldg = g.data_ptr(), g.stride(0)
    if a is not None:
        r.a, r.lda = a.data_ptr(), a.stride(0)
    r.act = ACT[act]
    r.mean, r.rstd = _ptr(mean), _ptr(rstd)
    r.out0, r.out1 = _ptr(out0), _ptr(out1)
    if dy_out is not None:
        r.dy_out, r.lddy = dy_out.data_ptr(), dy_out.stride(0)
    r.mode = mode
    r.scale, r.shift = _ptr(scale), _ptr(shift)
    if defer:
        slots = SumSlots.on(ref.device)
        r.out0 = r.out1 = None
        r.defer = 1
        if mode == 0:
            ptr, stride = slots.forward_region(r.C)
        else:
            ptr, stride, clr, n = slots.backward_region(r.C)
            r.clear, r.clear_floats = clr, n
        r.ws, r.ws_bytes = ptr, 16 * 2 * stride * 4
        check(_lib.load().estts_colreduce(r, _stream()), 'estts_colreduce')
        return ptr, stride
    ws = colreduce_workspace(out0.device)
    r.ws, r.ws_bytes = ws.data_ptr(), ws.numel() * 4
    check(_lib.load().estts_colreduce(r, _stream()), 'estts_colreduce')


def bn_finalize(s0, s1, z, M, C, gamma, beta, moving_mean, moving_var, mean, rstd, scale, shift, eps=1e-3, momentum=0.99,
                moving_var_scale=1.0, slots=None, about_next=None):
    """slots = (pointer, stride) of deferred sums (colreduce(0, defer=True) or gemm(stats=...)): s0 / s1 are then ignored and
    z is the [C] vector (or matrix whose row 0) the moments were taken about; about_next [C] also receives the batch mean (the
    shift of the layer's next step; may be z)."""
    if slots is not None:
        check(_lib.load().estts_bn_finalize_slots(slots[0], slots[1], z.data_ptr(), M, C, eps, momentum, moving_var_scale,
                                                  gamma.data_ptr(), beta.data_ptr(), _ptr(moving_mean), _ptr(moving_var),
                                                  mean.data_ptr(), rstd.data_ptr(), scale.data_ptr(), shift.data_ptr(),
                                                  _ptr(about_next), _stream()),
              'estts_bn_finalize_slots')
        return
    check(_lib.load().estts_bn_finalize(s0.data_ptr(), s1.data_ptr(), z.data_ptr(), M, C, eps, momentum, moving_var_scale, gamma.data_ptr(),
                                        beta.data_ptr(), _ptr(moving_mean), _ptr(moving_var), mean.data_ptr(), rstd.data_ptr(),
                                        scale.data_ptr(), shift.data_ptr(), _stream()), 'estts_bn_finalize')


def affine_act(z, out, scale=None, shift=None, act=None, residual=None, r_mod=0, M=None, C=None):
    M = z.shape[0] if M is None else M
    C = z.shape[1] if C is None else C
    check(_lib.load().estts_affine_act(z.data_ptr(), z.stride(0), _ptr(scale), _ptr(shift), ACT[act], _ptr(residual),
                                       0 if residual is None else residual.stride(0), r_mod, out.data_ptr(), out.stride(0), M, C,
                                       _stream()), 'estts_affine_act')
    return out


def bn_bwd(dy, z, mean, rstd, gamma, dgamma, dbeta, dz, M=None, C=None, slots=None, act=None, scale=None, shift=None):
    """slots = (pointer, stride) of deferred mode-2 sums: they are the dbeta / dgamma used, and their totals are added to the
    dgamma / dbeta tensors (the parameter gradients).  With slots, act / scale / shift: dy is the gradient w.r.t.
    act(z * scale + shift) and goes through the activation inside the launch (the reduction then needs no dy_out)."""
    M = z.shape[0] if M is None else M
    C = z.shape[1] if C is None else C
    if slots is not None:
        check(_lib.load().estts_bn_bwd_slots(dy.data_ptr(), dy.stride(0), z.data_ptr(), z.stride(0), mean.data_ptr(), rstd.data_ptr(),
                                             gamma.data_ptr(), slots[0], slots[1], dgamma.data_ptr(), dbeta.data_ptr(),
                                             dz.data_ptr(), dz.stride(0), M, C, ACT[act], _ptr(scale), _ptr(shift), _stream()),
              'estts_bn_bwd_slots')
        return dz
    if scale is not None:
        raise ValueError('bn_bwd: the activation inside the launch needs slots=')
    check(_lib.load().estts_bn_bwd(dy.data_ptr(), dy.stride(0), z.data_ptr(), z.stride(0), mean.data_ptr(), rstd.data_ptr(),
                                   gamma.data_ptr(), dgamma.data_ptr(), dbeta.data_ptr(), dz.data_ptr(), dz.stride(0), M, C,
                                   _stream()), 'estts_bn_bwd')
    return dz


def dropout(x, out, rate, seed, site, M=None, D=None):
    M = x.shape[0] if M is None else M
    D = x.shape[1] if D is None else D
    check(_lib.load().estts_dropout(x.data_ptr(), x.stride(0), out.data_ptr(), out.stride(0), M, D, rate, seed, site, _stream()),
          'estts_dropout')
    return out


def axpby(x, out, alpha=1.0, beta=1.0, M=None, D=None):
    M = x.shape[0] if M is None else M
    D = x.shape[1] if D is None else D
    check(_lib.load().estts_axpby(x.data_ptr(), x.stride(0), alpha, out.data_ptr(), out.stride(0), beta, M, D, _stream()),
          'estts_axpby')
    return out


def head_scale(x, scale, B, L, H, depth):
    check(_lib.load().estts_head_scale(x.data_ptr(), x.stride(0), scale.data_ptr(), B, L, H, depth, _stream()), 'estts_head_scale')
    return x


def im2col(x, out, B, H, W, C, Ho, Wo, k, stride, pad_t, pad_l):
    """x: NHWC [B, H, W, C] contiguous -> out [B * Ho * Wo, >= k * k * C] patches (estts_im2col)."""
    check(_lib.load().estts_im2col(x.data_ptr(), out.data_ptr(), out.stride(0), B, H, W, C, Ho, Wo, k, k, stride, pad_t, pad_l,
                                   _stream()), 'estts_im2col')
    return out


def col2im(dp, dx, B, H, W, C, Ho, Wo, k, stride, pad_t, pad_l):
    """dp [B * Ho * Wo, >= k * k * C] patch gradients -> dx NHWC [B, H, W, C] contiguous (estts_col2im)."""
    check(_lib.load().estts_col2im(dp.data_ptr(), dp.stride(0), dx.data_ptr(), B, H, W, C, Ho, Wo, k, k, stride, pad_t, pad_l,
                                   _stream()), 'estts_col2im')
    return dx


def embed_bwd(tokens, dx, demb, scale):
    B, N = tokens.shape
    check(_lib.load().estts_embed_bwd(tokens.data_ptr(), dx.data_ptr(), dx.stride(0), demb.data_ptr(), B, N, demb.shape[1], scale,
                                      _stream()), 'estts_embed_bwd')


def expand_bwd(dout, cum, dx, B, N, T, D):
    check(_lib.load().estts_expand_bwd(dout.data_ptr(), dout.stride(0), cum.data_ptr(), dx.data_ptr(), dx.stride(0), B, N, T, D,
                                       _stream()), 'estts_expand_bwd')
    return dx


def masked_mae_bwd(target, pred, dpred, scale, M=None, C=None):
    M = target.shape[0] if M is None else M
    C = target.shape[1] if C is None else C
    check(_lib.load().estts_masked_mae_bwd(target.data_ptr(), target.stride(0), pred.data_ptr(), pred.stride(0), dpred.data_ptr(),
                                           dpred.stride(0), M, C, scale, _stream()), 'estts_masked_mae_bwd')
    return dpred


def adam(p, grad, m, v, lr_t, beta1=0.9, beta2=0.98, eps=1e-9, grad_scale=1.0):
    n = p.numel()
    check(_lib.load().estts_adam(p.data_ptr(), grad.data_ptr(), m.data_ptr(), v.data_ptr(), n, lr_t, beta1, beta2, eps, grad_scale,
                                 _stream()), 'estts_adam')


def gru_step_fwd(gi, gh, h_prev, h_out, x_res=None, out=None):
    M, H = h_prev.shape
    check(_lib.load().estts_gru_step_fwd(gi.data_ptr(), gi.stride(0), gh.data_ptr(), gh.stride(0), h_prev.data_ptr(),
                                         h_prev.stride(0), h_out.data_ptr(), h_out.stride(0), _ptr(x_res),
                                         0 if x_res is None else x_res.stride(0), _ptr(out), 0 if out is None else out.stride(0),
                                         M, H, _stream()), 'estts_gru_step_fwd')


def gru_step_bwd(gi, gh, h_prev, dh, dh_rec, dgi, dgh, dh_prev):
    M, H = h_prev.shape
    check(_lib.load().estts_gru_step_bwd(gi.data_ptr(), gi.stride(0), gh.data_ptr(), gh.stride(0), h_prev.data_ptr(),
                                         h_prev.stride(0), dh.data_ptr(), dh.stride(0), _ptr(dh_rec),
                                         0 if dh_rec is None else dh_rec.stride(0), dgi.data_ptr(), dgi.stride(0),
                                         dgh.data_ptr(), dgh.stride(0), dh_prev.data_ptr(), dh_prev.stride(0), M, H, _stream()),
          'estts_gru_step_bwd')


def _gru_fused_common(gi, gh, h_prev, lin, precision):
    a = _lib.GruFusedArgs()
    a.gi, a.ldgi, a.gh, a.ldgh = gi.data_ptr(), gi.stride(0), gh.data_ptr(), gh.stride(0)
    a.h_prev, a.ldhp = h_prev.data_ptr(), h_prev.stride(0)
    a.W_hi, a.W_lo, a.ldw = lin.w_hi.data_ptr(), lin.w_lo.data_ptr(), lin.ldw
    a.M, a.H = h_prev.shape
    a.precision = PRECISION[precision]
    return a


def gru_fused_fwd(gi, gh, h_prev, lin, h_out, x_res=None, out=None, precision='bf16x3', launch=True):
    """One GRU step forward in one launch: gh = h_prev (*) lin (W_hh packed for estts_gemm, its bias included), gates, h_out
    (and out = h_out + x_res); see estts_gru_fused_fwd.  launch=False: only builds and returns the argument block (for
    gru_fused_pair; the tensors must outlive the launch)."""
    a = _gru_fused_common(gi, gh, h_prev, lin, precision)
    if (lin.N, lin.K) != (3 * a.H, a.H) or lin.s1 is not None:
        raise ValueError('gru_fused_fwd: the weight must be W_hh [3H x H] without an output scale')
    a.bias = _ptr(lin.b1)
    a.h_out, a.ldho = h_out.data_ptr(), h_out.stride(0)
    if x_res is not None:
        a.x_res, a.ldx = x_res.data_ptr(), x_res.stride(0)
    if out is not None:
        a.out, a.ldo = out.data_ptr(), out.stride(0)
    if not launch:
        return a
    check(_lib.load().estts_gru_fused_fwd(a, _stream()), 'estts_gru_fused_fwd')


def gru_fused_bwd(gi, gh, h_prev, dh, dgh_next, lin_bwd, dgi, dgh, dhd, precision='bf16x3', launch=True):
    """The hidden-state gradient reaching a step from the next one (dgh_next (*) lin_bwd + dhd) and the step's gate gradients in
    one launch; see estts_gru_fused_bwd.  launch=False: only builds and returns the argument block (gru_fused_pair)."""
    a = _gru_fused_common(gi, gh, h_prev, lin_bwd, precision)
    if (lin_bwd.N, lin_bwd.K) != (a.H, 3 * a.H) or lin_bwd.s1 is not None or lin_bwd.b1 is not None:
        raise ValueError('gru_fused_bwd: the weight must be the transposed pack [H x 3H] of W_hh without scale or bias')
    a.dgh_next, a.lddghn, a.dh, a.lddh = dgh_next.data_ptr(), dgh_next.stride(0), dh.data_ptr(), dh.stride(0)
    a.dgi, a.lddgi, a.dgh, a.lddgh = dgi.data_ptr(), dgi.stride(0), dgh.data_ptr(), dgh.stride(0)
    a.dhd, a.lddhd = dhd.data_ptr(), dhd.stride(0)
    if not launch:
        return a
    check(_lib.load().estts_gru_fused_bwd(a, _stream()), 'estts_gru_fused_bwd')


def gru_fused_pair(a, b, forward):
    """Argument blocks of gru_fused_fwd / gru_fused_bwd (launch=False) as ONE launch: two independent steps of the same width;
    either may be None (then the other runs alone)."""
    lib = _lib.load()
    if a is None or b is None:
        one = a if b is None else b
        fn, name = (lib.estts_gru_fused_fwd, 'estts_gru_fused_fwd') if forward else (lib.estts_gru_fused_bwd, 'estts_gru_fused_bwd')
        check(fn(one, _stream()), name)
        return
    fn, name = (lib.estts_gru_fused_fwd2, 'estts_gru_fused_fwd2') if forward else (lib.estts_gru_fused_bwd2, 'estts_gru_fused_bwd2')
    check(fn(a, b, _stream()), name)


def wavernn_prep_bwd(dxin, da2, da3, da4, mel, mel_off, d_aux, dfir, fir_kmin, n_utt, frames, feat, d, hop, pad):
    a = _lib.WavernnPrepBwdArgs()
    a.dxin, a.ld_dxin = dxin.data_ptr(), dxin.stride(0)
    a.da2, a.ld_da2, a.da3, a.ld_da3, a.da4, a.ld_da4 = (da2.data_ptr(), da2.stride(0), da3.data_ptr(), da3.stride(0),
                                                         da4.data_ptr(), da4.stride(0))
    a.mel, a.mel_off, a.d_aux, a.dfir = mel.data_ptr(), mel_off.data_ptr(), d_aux.data_ptr(), dfir.data_ptr()
    a.ntaps, a.fir_kmin = dfir.shape[1], fir_kmin
    a.n_utt, a.frames, a.feat, a.d, a.hop, a.pad = n_utt, frames, feat, d, hop, pad
    check(_lib.load().estts_wavernn_prep_bwd(a, _stream()), 'estts_wavernn_prep_bwd')


def softmax_xent(logits, labels, loss, dlogits, scale, M=None, C=None):
    M = logits.shape[0] if M is None else M
    C = logits.shape[1] if C is None else C
    if labels.dtype != torch.int32:
        raise TypeError('labels must be int32')
    check(_lib.load().estts_softmax_xent(logits.data_ptr(), logits.stride(0), labels.data_ptr(), loss.data_ptr(), dlogits.data_ptr(),
                                         dlogits.stride(0), M, C, scale, _stream()), 'estts_softmax_xent')


def mol_loss(y_hat, y, loss, d, scale, nr_mix, num_classes=65536, log_scale_min=-32.23619130191664, M=None):
    M = y_hat.shape[0] if M is None else M
    check(_lib.load().estts_mol_loss(y_hat.data_ptr(), y_hat.stride(0), y.data_ptr(), loss.data_ptr(), d.data_ptr(), d.stride(0), M,
                                     nr_mix, num_classes, log_scale_min, scale, _stream()), 'estts_mol_loss')


def _debug_wrap(fn):
    import functools
    import sys

    @functools.wraps(fn)
    def inner(*a, **kw):
        r = fn(*a, **kw)
        shapes = [tuple(t.shape) for t in a if torch.is_tensor(t)]
        print(f'[estts] {fn.__name__} {shapes}', file=sys.stderr, flush=True)
        torch.cuda.synchronize()
        return r
    return inner


if ops._debug_sync:      # ESTTS_DEBUG_SYNC=1: name every training launch and wait for it (fault localisation)
    for _n in ('wgrad', 'attention_bwd', 'rownorm_bwd', 'colreduce', 'bn_finalize', 'affine_act', 'bn_bwd', 'dropout', 'axpby',
               'head_scale', 'embed_bwd', 'expand_bwd', 'masked_mae_bwd', 'adam', 'gru_step_fwd', 'gru_step_bwd', 'wavernn_prep_bwd',
               'softmax_xent', 'mol_loss'):
        globals()[_n] = _debug_wrap(globals()[_n])


def scaled_xent(logits, labels, loss, dlogits=None, index=2, scaling=1.0, scale=None, M=None, C=None):
    """new_scaled_crossentropy (utils/losses.py:4-21): loss[0] += scale * sum_m w[m] ce[m], w = (label != 0) + (label == index)
    (scaling - 1); scale defaults to 1 / M (Keras SUM_OVER_BATCH_SIZE).  dlogits: optional gradient output."""
    M = logits.shape[0] if M is None else M
    C = logits.shape[1] if C is None else C
    if labels.dtype != torch.int32:
        raise TypeError('scaled_xent: labels must be int32')
    scale = 1.0 / M if scale is None else scale
    check(_lib.load().estts_scaled_xent(logits.data_ptr(), logits.stride(0), labels.data_ptr(), loss.data_ptr(),
                                        dlogits.data_ptr() if dlogits is not None else None,
                                        dlogits.stride(0) if dlogits is not None else 0, M, C, index, scaling, scale, _stream()),
          'estts_scaled_xent')
    return loss
