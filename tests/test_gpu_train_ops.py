"""Training kernels through the C ABI vs PyTorch autograd in float64 on the CPU (the checker only; nothing here is
on the product path).  Tolerances are those of split-bf16 products (relative ~2e-5) unless the op is exact."""
import math

import numpy as np
import pytest
import torch
import torch.nn.functional as F

import estts  # noqa: F401

pytestmark = pytest.mark.gpu


@pytest.fixture(scope='module')
def T():
    from estts import ops, train_ops
    ops._lib.load()
    return train_ops


def dev(a, dtype=torch.float32):
    return torch.as_tensor(np.asarray(a)).to(dtype).cuda()


def rel_err(a, b):
    a = a.detach().cpu().double().numpy() if torch.is_tensor(a) else np.asarray(a, np.float64)
    b = b.detach().cpu().double().numpy() if torch.is_tensor(b) else np.asarray(b, np.float64)
    return np.abs(a - b).max() / max(np.abs(b).max(), 1e-12)


def conv_ref(x, kernel, bias, taps, lpad, seg):
    """x [B*seg, cin] float64, kernel [taps*cin, N] -> [B*seg, N]; zero padding at segment borders."""
    cin = kernel.shape[0] // taps
    B = x.shape[0] // seg
    xp = F.pad(x.view(B, seg, cin), (0, 0, lpad, taps - 1 - lpad))
    y = sum(xp[:, t:t + seg, :] @ kernel[t * cin:(t + 1) * cin] for t in range(taps))
    return (y + bias).reshape(B * seg, -1)


@pytest.mark.parametrize('taps,lpad,cin,N,B,seg', [(1, 0, 72, 40, 1, 50), (3, 1, 24, 52, 3, 37), (5, 2, 80, 64, 2, 33), (5, 4, 16, 20, 2, 19),
                                                   (3, 1, 48, 132, 3, 37), (5, 2, 64, 256, 4, 50)])      # K, N >= 128
def test_linear_forward_dgrad_wgrad(T, taps, lpad, cin, N, B, seg):
    from estts import ops
    rng = np.random.default_rng(taps * 100 + N)
    M, K = B * seg, taps * cin
    kernel = rng.normal(0, 0.2, size=(K, N))
    bias = rng.normal(0, 0.2, size=N)
    x = rng.normal(0, 1, size=(M, cin))
    dy = rng.normal(0, 1, size=(M, N))
    xt = torch.tensor(x, requires_grad=True)
    kt = torch.tensor(kernel, requires_grad=True)
    bt = torch.tensor(bias, requires_grad=True)
    y_ref = conv_ref(xt, kt, bt, taps, lpad, seg)
    y_ref.backward(torch.tensor(dy))

    kd, bd, xd, dyd = dev(kernel), dev(bias), dev(x), dev(dy)
    lin = T.TrainLinear(kd, bd, taps=taps, lpad=lpad)
    y = torch.empty(M, N, device='cuda')
    ops.gemm(xd, lin.fwd, y, seg_len=seg if taps > 1 else 0)
    assert rel_err(y, y_ref) < 3e-5
    dx = torch.empty(M, cin, device='cuda')
    ops.gemm(dyd, lin.bwd, dx, seg_len=seg if taps > 1 else 0)
    assert rel_err(dx, xt.grad) < 3e-5
    dw = torch.zeros(K, N, device='cuda')
    T.wgrad(xd, dyd, dw, taps=taps, lpad=lpad, seg_len=seg if taps > 1 else 0)
    assert rel_err(dw, kt.grad) < 3e-5
    # a second call accumulates
    T.wgrad(xd, dyd, dw, taps=taps, lpad=lpad, seg_len=seg if taps > 1 else 0, splits=3)
    assert rel_err(dw, 2 * kt.grad) < 3e-5
    db = torch.zeros(N, device='cuda')
    T.colreduce(1, db, g=dyd)
    assert rel_err(db, bt.grad) < 1e-5
    # the bias gradient as a by-product of the weight-gradient kernel
    db2 = torch.zeros(N, device='cuda')
    T.wgrad(xd, dyd, torch.zeros(K, N, device='cuda'), taps=taps, lpad=lpad, seg_len=seg if taps > 1 else 0, splits=3, db=db2)
    assert rel_err(db2, bt.grad) < 1e-5


@pytest.mark.parametrize('B,seg,K,N', [(3, 37, 24, 40), (5, 8, 72, 20), (2, 700, 64, 192)])
def test_wgrad_one_tap_shifted(T, B, seg, K, N):
    """One tap with one row of left padding (h_{t-1} against the gate gradients of step t in the GRU backward pass): row m
    meets row m - 1 of its own segment, the first row of every segment meets nothing."""
    rng = np.random.default_rng(seg)
    x, dy = rng.normal(size=(B * seg, K)), rng.normal(size=(B * seg, N))
    prev = np.concatenate([np.zeros((B, 1, K)), x.reshape(B, seg, K)[:, :-1]], axis=1).reshape(B * seg, K)
    dw = torch.zeros(K, N, device='cuda')
    db = torch.zeros(N, device='cuda')
    T.wgrad(dev(x), dev(dy), dw, taps=1, lpad=1, seg_len=seg, db=db)
    assert rel_err(dw, prev.T @ dy) < 3e-5
    assert rel_err(db, dy.sum(0)) < 1e-5


@pytest.mark.parametrize('taps,lpad,cin,N,B,seg,prec', [(3, 1, 64, 72, 5, 37, 'bf16x3'), (3, 1, 128, 256, 7, 100, 'bf16x3'),
                                                        (5, 2, 64, 80, 3, 50, 'bf16x3'), (5, 4, 128, 64, 9, 19, 'bf16x3'),
                                                        (3, 2, 192, 40, 40, 7, 'bf16x3'), (3, 0, 64, 64, 2, 600, 'bf16x3'),
                                                        (5, 2, 256, 256, 32, 600, 'bf16x3'), (3, 1, 64, 72, 5, 37, 'bf16')])
def test_wgrad_all_taps_from_one_staged_tile(T, taps, lpad, cin, N, B, seg, prec):
    """Convolutions over whole 64-channel tiles take the kernel that stages the input rows once for all taps: sequence
    borders inside a 32-row step, sequences shorter than a step, causal padding (lpad = taps - 1) and row counts that are not
    multiples of 32, against float64; the bias gradient rides along."""
    rng = np.random.default_rng(taps * 1000 + cin + seg)
    M, K = B * seg, taps * cin
    x = rng.normal(size=(M, cin))
    dy = rng.normal(size=(M, N))
    xp = np.pad(x.reshape(B, seg, cin), ((0, 0), (lpad, taps - 1 - lpad), (0, 0)))
    ref = np.concatenate([np.einsum('bsc,bsn->cn', xp[:, t:t + seg], dy.reshape(B, seg, N)) for t in range(taps)], 0)
    dw = torch.zeros(K, N, device='cuda')
    db = torch.zeros(N, device='cuda')
    # (an explicit number of row ranges selects that kernel whatever the size; left to itself the library takes it from
    # about 512 rows per workgroup, which the 19200-row case reaches)
    T.wgrad(dev(x), dev(dy), dw, taps=taps, lpad=lpad, seg_len=seg, precision=prec, db=db, splits=0 if M >= 19200 else 2)
    assert rel_err(dw, ref) < (3e-5 if prec == 'bf16x3' else 2e-2)
    assert rel_err(db, dy.sum(0)) < 1e-5


def test_wgrad_large_rows_many_splits(T):
    rng = np.random.default_rng(0)
    M, K, N = 4100, 192, 136
    x, dy = rng.normal(size=(M, K)), rng.normal(size=(M, N))
    dw = torch.zeros(K, N, device='cuda')
    T.wgrad(dev(x), dev(dy), dw)
    assert rel_err(dw, x.T @ dy) < 3e-5


@pytest.mark.parametrize('B,H,Lq,Lk,depth', [(2, 2, 37, 37, 64), (1, 4, 50, 21, 32), (2, 1, 16, 70, 128), (4, 2, 37, 53, 64),
                                            (4, 4, 70, 40, 32),
                                            # enough 64-row workgroups for the batched kernel: on both sides, on the query
                                            # side only, on the key side only, and with Lk not a multiple of 4
                                            (8, 4, 400, 400, 64), (3, 4, 1030, 70, 32), (3, 4, 50, 1030, 128),
                                            (16, 4, 200, 203, 128)])
def test_attention_backward(T, B, H, Lq, Lk, depth):
    from estts import ops
    rng = np.random.default_rng(Lq * 7 + Lk)
    d = H * depth
    q, k, v = (rng.normal(0, 1, size=(B * L, d)) for L in (Lq, Lk, Lk))
    mask = np.zeros((B, Lk))
    mask[0, Lk - 3:] = 1.0
    dctx = rng.normal(0, 1, size=(B * Lq, d))
    qt, kt, vt = (torch.tensor(a, requires_grad=True) for a in (q, k, v))

    def heads(t, L):
        return t.view(B, L, H, depth).permute(0, 2, 1, 3)

    logits = heads(qt, Lq) @ heads(kt, Lk).transpose(-1, -2) / math.sqrt(depth) + torch.tensor(mask)[:, None, None, :] * -1e9
    pw = torch.softmax(logits, -1)
    ctx_ref = (pw @ heads(vt, Lk)).permute(0, 2, 1, 3).reshape(B * Lq, d)
    ctx_ref.backward(torch.tensor(dctx))

    qd, kd_, vd, md, dcd = dev(q), dev(k), dev(v), dev(mask), dev(dctx)
    ctx = torch.empty(B * Lq, d, device='cuda')
    attn = torch.empty(B, H, Lq, Lk, device='cuda')
    ops.attention(qd, kd_, vd, ctx, B, H, Lq, Lk, depth, key_mask=md, attn=attn)
    assert rel_err(ctx, ctx_ref) < 5e-5
    delta = torch.empty(B * H * Lq, device='cuda')
    dq, dk, dv = torch.empty_like(qd), torch.empty_like(kd_), torch.empty_like(vd)
    T.attention_bwd(qd, kd_, vd, ctx, dcd, attn, delta, dq, dk, dv, B, H, Lq, Lk, depth)
    assert rel_err(dq, qt.grad) < 1e-4
    assert rel_err(dk, kt.grad) < 1e-4
    assert rel_err(dv, vt.grad) < 1e-4


@pytest.mark.parametrize('B,H,L,depth', [(2, 2, 45, 32), (16, 4, 200, 32)])
def test_attention_backward_strided_qkv_buffer(T, B, H, L, depth):
    """q, k, v and their gradients as column blocks of one [M, 3d] buffer (the fused QKV projection layout)."""
    from estts import ops
    rng = np.random.default_rng(3)
    d = H * depth
    qkv = rng.normal(size=(B * L, 3 * d))
    dctx = rng.normal(size=(B * L, d))
    t = torch.tensor(qkv, requires_grad=True)

    def heads(x):
        return x.view(B, L, H, depth).permute(0, 2, 1, 3)

    pw = torch.softmax(heads(t[:, :d]) @ heads(t[:, d:2 * d]).transpose(-1, -2) / math.sqrt(depth), -1)
    (pw @ heads(t[:, 2 * d:])).permute(0, 2, 1, 3).reshape(B * L, d).backward(torch.tensor(dctx))
    qd = dev(qkv)
    ctx = torch.empty(B * L, d, device='cuda')
    attn = torch.empty(B, H, L, L, device='cuda')
    ops.attention(qd[:, :d], qd[:, d:2 * d], qd[:, 2 * d:], ctx, B, H, L, L, depth, attn=attn)
    g = torch.empty_like(qd)
    delta = torch.empty(B * H * L, device='cuda')
    T.attention_bwd(qd[:, :d], qd[:, d:2 * d], qd[:, 2 * d:], ctx, dev(dctx), attn, delta, g[:, :d], g[:, d:2 * d], g[:, 2 * d:],
                    B, H, L, L, depth)
    assert rel_err(g, t.grad) < 1e-4


@pytest.mark.parametrize('D,use_a,act,use_res,use_b,rate,dot', [
    (256, True, None, True, True, 0.1, False),      # SelfAttentionResNorm
    (512, True, 'relu', True, True, 0.1, False),    # FFNResNorm
    (64, True, 'relu', False, False, 0.0, False),   # duration predictor inner conv
    (128, True, 'relu', True, True, 0.0, True),     # duration predictor last stage + Dense(1, relu)
    (80, False, 'tanh', True, False, 0.25, False),
    (384, True, None, True, True, 0.0, True),       # two chunks per lane, the second half full
    (1024, True, 'relu', True, True, 0.1, False),   # four chunks per lane: the widest instantiation
])
def test_rownorm_forward_backward(T, D, use_a, act, use_res, use_b, rate, dot):
    from estts import ops
    rng = np.random.default_rng(D)
    M = 203
    y, x = rng.normal(0, 1.5, size=(M, D)), rng.normal(0, 1, size=(M, D))
    ga, ba, gb, bb = (rng.normal(1, 0.2, size=D), rng.normal(0, 0.2, size=D), rng.normal(1, 0.2, size=D), rng.normal(0, 0.2, size=D))
    dout = rng.normal(0, 1, size=(M, D))
    dot_w, dot_b = rng.normal(0, 0.2, size=D), 0.3
    dmask = (rng.uniform(size=M) < 0.2).astype(np.float64)
    ddot = rng.normal(0, 1, size=M)
    seed, site = 77, 5
    keep = torch.ones(M, D, device='cuda')
    if rate > 0:
        T.dropout(keep, keep, rate, seed, site)
        frac = float((keep > 0).float().mean())
        assert abs(frac - (1 - rate)) < 0.02
    keep64 = keep.cpu().double()

    ts = {n: torch.tensor(v, requires_grad=True) for n, v in dict(y=y, x=x, ga=ga, ba=ba, gb=gb, bb=bb, w=dot_w).items()}
    h = ts['y']
    if use_a:
        h = F.layer_norm(h, (D,), ts['ga'], ts['ba'], 1e-6)
    if act == 'relu':
        h = torch.relu(h)
    elif act == 'tanh':
        h = torch.tanh(h)
    h = h * keep64
    if use_res:
        h = h + ts['x']
    if use_b:
        h = F.layer_norm(h, (D,), ts['gb'], ts['bb'], 1e-6)
    loss = (h * torch.tensor(dout)).sum()
    if dot:
        dval = torch.relu(h @ ts['w'] + dot_b) * 0.7 * (1 - torch.tensor(dmask))
        loss = loss + (dval * torch.tensor(ddot)).sum()
    loss.backward()

    yd, xd = dev(y), dev(x)
    lna = (dev(ga), dev(ba)) if use_a else None
    lnb = (dev(gb), dev(bb)) if use_b else None
    out = torch.empty(M, D, device='cuda')
    dot_out = torch.empty(M, device='cuda') if dot else None
    kw = dict(ln_a=lna, act=act, residual=xd if use_res else None, ln_b=lnb, dropout=(rate, seed, site),
              dot_w=dev(dot_w) if dot else None, dot_b=dot_b, dot_scale=0.7, dot_mask=dev(dmask) if dot else None, dot_out=dot_out)
    ops.rownorm(yd, out, **kw)
    assert rel_err(out, h) < 2e-5
    if dot:
        assert rel_err(dot_out, dval) < 2e-5
    fa = ops.rownorm_args(yd, out, **kw)
    dy = torch.empty(M, D, device='cuda')
    dx = torch.full((M, D), 1.0, device='cuda') if use_res else None
    dla = (torch.zeros(D, device='cuda'), torch.zeros(D, device='cuda')) if use_a else None
    dlb = (torch.zeros(D, device='cuda'), torch.zeros(D, device='cuda')) if use_b else None
    dw = torch.zeros(D, device='cuda') if dot else None
    dbias = torch.zeros(1, device='cuda') if dot else None
    T.rownorm_bwd(fa, dout=dev(dout), ddot=dev(ddot) if dot else None, dy=dy, dx=dx, dx_accumulate=True, dln_a=dla, dln_b=dlb,
                  ddot_w=dw, ddot_b=dbias)
    assert rel_err(dy, ts['y'].grad) < 5e-5
    if use_res:
        assert rel_err(dx - 1.0, ts['x'].grad) < 5e-5
    if use_a:
        assert rel_err(dla[0], ts['ga'].grad) < 5e-5 and rel_err(dla[1], ts['ba'].grad) < 5e-5
    if use_b:
        assert rel_err(dlb[0], ts['gb'].grad) < 5e-5 and rel_err(dlb[1], ts['bb'].grad) < 5e-5
    if dot:
        assert rel_err(dw, ts['w'].grad) < 5e-5


@pytest.mark.parametrize('M,C,act', [(300, 80, 'tanh'), (1000, 512, 'relu'), (77, 256, None)])
def test_batchnorm_training_forward_backward(T, M, C, act):
    rng = np.random.default_rng(C + M)
    z = rng.normal(0.5, 2.0, size=(M, C)) + rng.normal(0, 3, size=C)
    res = rng.normal(size=(M, C))
    gamma, beta = rng.normal(1, 0.2, size=C), rng.normal(0, 0.2, size=C)
    mm, mv = rng.normal(0, 0.1, size=C), rng.uniform(0.5, 1.5, size=C)
    dout = rng.normal(size=(M, C))
    zt, gt, bt = (torch.tensor(a, requires_grad=True) for a in (z, gamma, beta))
    rm, rv = torch.tensor(mm.copy()), torch.tensor(mv.copy())
    # Keras: momentum 0.99 (torch's momentum is 1 - that), biased variance in the moving update, eps 1e-3
    bn = F.batch_norm(zt.t()[None], None, None, gt, bt, True, 0.0, 1e-3)[0].t()
    a_ref = torch.tanh(bn) if act == 'tanh' else (torch.relu(bn) if act == 'relu' else bn)
    out_ref = a_ref + torch.tensor(res)
    (out_ref * torch.tensor(dout)).sum().backward()
    mean_ref, var_ref = z.mean(0), z.var(0)
    rm_ref, rv_ref = mm * 0.99 + mean_ref * 0.01, mv * 0.99 + var_ref * 0.01

    zd, gd, bd = dev(z), dev(gamma), dev(beta)
    mmd, mvd = dev(mm), dev(mv)
    s = torch.zeros(2, C, device='cuda')
    T.colreduce(0, s[0], s[1], z=zd)
    mean, rstd, scale, shift = (torch.empty(C, device='cuda') for _ in range(4))
    T.bn_finalize(s[0], s[1], zd, M, C, gd, bd, mmd, mvd, mean, rstd, scale, shift)
    assert rel_err(mean, mean_ref) < 1e-5
    assert rel_err(rstd, 1 / np.sqrt(var_ref + 1e-3)) < 1e-5
    assert rel_err(mmd, rm_ref) < 1e-5 and rel_err(mvd, rv_ref) < 1e-5
    a = torch.empty(M, C, device='cuda')
    T.affine_act(zd, a, scale, shift, act)
    out = torch.empty(M, C, device='cuda')
    T.affine_act(a, out, residual=dev(res))
    assert rel_err(out, out_ref) < 1e-5
    # backward
    g2 = torch.zeros(2, C, device='cuda')
    dy = torch.empty(M, C, device='cuda')
    if C == 512:       # activation output recomputed from z instead of read back
        T.colreduce(2, g2[0], g2[1], z=zd, g=dev(dout), act=act, mean=mean, rstd=rstd, dy_out=dy, scale=scale, shift=shift)
    else:
        T.colreduce(2, g2[0], g2[1], z=zd, g=dev(dout), a=a if act else None, act=act, mean=mean, rstd=rstd, dy_out=dy)
    assert rel_err(g2[0], bt.grad) < 2e-5 and rel_err(g2[1], gt.grad) < 2e-5
    dz = torch.empty(M, C, device='cuda')
    T.bn_bwd(dy, zd, mean, rstd, gd, g2[1], g2[0], dz)
    assert rel_err(dz, zt.grad) < 5e-5


def test_deferred_sums_refuse_graph_capture_and_recover_from_an_unfinished_step(T):
    """The regions' turn / clear bookkeeping is host-side: recording a deferred reduction into a hipGraph must raise instead
    of replaying into a region nobody cleared; and a step that ended between a producer and its consumer must not pollute
    the next one (step_begin clears what the unfinished step left)."""
    rng = np.random.default_rng(3)
    M, C = 300, 128
    z = dev(rng.normal(size=(M, C)))
    g = torch.cuda.CUDAGraph()
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with pytest.raises(RuntimeError, match='hipGraph'):
        with torch.cuda.graph(g, stream=s):
            T.colreduce(0, z=z, shift=z[0], defer=True)
    torch.cuda.synchronize()
    T.SumSlots.reset_all()
    slots = T.SumSlots.on('cuda')
    owner = object()
    T.SumSlots.step_begin('cuda', owner)
    T.colreduce(0, z=z, shift=None, defer=True)             # a producer whose consumer never runs (the step "fails" here)
    assert float(slots.buf[2].abs().max()) > 0.0
    T.SumSlots.step_begin('cuda', owner)                    # the next step starts: leftovers are cleared
    assert float(slots.buf.abs().max()) == 0.0
    gamma, beta = torch.ones(C, device='cuda'), torch.zeros(C, device='cuda')
    st = torch.empty(4, C, device='cuda')
    sl = T.colreduce(0, z=z, shift=None, defer=True)
    T.bn_finalize(None, None, torch.zeros(C, device='cuda'), M, C, gamma, beta, None, None, *st, slots=sl)
    T.SumSlots.step_end('cuda')
    assert rel_err(st[0], z.double().mean(0)) < 1e-5        # the mean of THIS step's tensor only
    T.SumSlots.step_begin('cuda', owner)                    # a finished step leaves nothing to clear
    T.SumSlots.step_end('cuda')


@pytest.mark.parametrize('sizes', [[(1000, 512), (700, 80), (5000, 1024), (77, 256)]])
def test_batchnorm_with_deferred_sums_matches_the_two_launch_path(T, sizes):
    """colreduce(defer=True) -> bn_finalize(slots=) / bn_bwd(slots=) against the path that adds the copies up in a launch of its
    own; the backward regions take turns and are cleared by the next reduction, so the column count changes between calls."""
    rng = np.random.default_rng(11)
    for rnd, (M, C) in enumerate(sizes * 2):
        z = dev(rng.normal(0.5, 2.0, size=(M, C)) + rng.normal(0, 3, size=C))
        gamma, beta = dev(rng.normal(1, 0.2, size=C)), dev(rng.normal(0, 0.2, size=C))
        dout = dev(rng.normal(size=(M, C)))
        stats = []
        for defer in (False, True):
            mm, mv = torch.zeros(C, device='cuda'), torch.ones(C, device='cuda')
            st = torch.empty(4, C, device='cuda')
            if defer:
                sl = T.colreduce(0, z=z, shift=z[0], defer=True)
                T.bn_finalize(None, None, z, M, C, gamma, beta, mm, mv, *st, slots=sl)
            else:
                s = torch.zeros(2, C, device='cuda')
                T.colreduce(0, s[0], s[1], z=z)
                T.bn_finalize(s[0], s[1], z, M, C, gamma, beta, mm, mv, *st)
            g2 = torch.zeros(2, C, device='cuda') + (0.5 if defer else 0.0)    # totals are ADDED to the gradients
            dy, dz = torch.empty(M, C, device='cuda'), torch.empty(M, C, device='cuda')
            sl = T.colreduce(2, g2[0], g2[1], z=z, g=dout, act='relu', mean=st[0], rstd=st[1], dy_out=dy, scale=st[2], shift=st[3],
                             defer=defer)
            T.bn_bwd(dy, z, st[0], st[1], gamma, g2[1], g2[0], dz, slots=sl)
            if defer:       # the same from the raw gradient: bn_bwd takes it through the activation, the reduction stores no dy
                g3 = torch.zeros(2, C, device='cuda')
                dz3 = torch.empty(M, C, device='cuda')
                sl = T.colreduce(2, z=z, g=dout, act='relu', mean=st[0], rstd=st[1], scale=st[2], shift=st[3], defer=True)
                T.bn_bwd(dout, z, st[0], st[1], gamma, g3[1], g3[0], dz3, slots=sl, act='relu', scale=st[2], shift=st[3])
                assert rel_err(dz3, dz) < 1e-6 and rel_err(g3, g2 - 0.5) < 1e-6, (rnd, M, C)
            stats.append((st.clone(), mm, mv, (g2 - (0.5 if defer else 0.0)).clone(), dz))
        a, b = stats
        # (float atomics: the order of the additions differs from launch to launch)
        assert max(rel_err(b[i], a[i]) for i in range(5)) < 2e-6, (rnd, M, C)
    slots = T.SumSlots.on('cuda')
    assert float(slots.buf[2].abs().max()) == 0.0          # bn_finalize cleared what it read
    assert float(slots.buf[slots.turn].abs().max()) == 0.0  # the region the next reduction writes is clear


def test_batch_mean_of_the_previous_step_conditions_the_variance(T):
    """bn_finalize(about_next=k) hands the batch mean to the next step's moments: with a mean 300 deviations off zero the first
    step (k = 0) loses digits of the variance in float32, the second (k = previous mean) does not."""
    rng = np.random.default_rng(5)
    M, C = 20000, 256
    gamma, beta = torch.ones(C, device='cuda'), torch.zeros(C, device='cuda')
    k = torch.zeros(C, device='cuda')
    errs = []
    offs = rng.normal(0, 5, size=C)          # the channel means drift little from step to step
    for step in range(2):
        z = rng.normal(300.0, 1.0, size=(M, C)) + offs + 0.05 * step
        zd = dev(z)
        sl = T.colreduce(0, z=zd, shift=k, defer=True)
        st = torch.empty(4, C, device='cuda')
        T.bn_finalize(None, None, k, M, C, gamma, beta, torch.zeros(C, device='cuda'), torch.ones(C, device='cuda'), *st, slots=sl,
                      about_next=k)
        z32 = zd.double().cpu().numpy()
        errs.append(rel_err(st[1], 1 / np.sqrt(z32.var(0) + 1e-3)))
        assert rel_err(k, z32.mean(0)) < 1e-6
    assert errs[1] < 1e-5 and errs[0] > 10 * errs[1], errs


@pytest.mark.parametrize('M,N,K,taps', [(19200, 256, 768, 3), (4096, 512, 256, 1), (300, 80, 256, 1), (1000, 384, 640, 5)])
def test_gemm_sums_the_moments_of_its_output(T, M, N, K, taps):
    """estts_gemm with stat_slots (large-M tiles: inside the epilogue; other kernels: a reduction pass behind the product) ->
    bn_finalize(slots=): mean / rstd of the layer output about a shift vector, against float64 moments of the stored output."""
    from estts import ops
    rng = np.random.default_rng(M + N)
    cin = K // taps
    seg = 100 if taps > 1 else 0
    x = dev(rng.normal(size=(M, cin)))
    kernel = rng.normal(size=(K, N)) / math.sqrt(K)
    bias = rng.normal(0, 2, size=N)
    lin = T.TrainLinear(dev(kernel), dev(bias), taps=taps, lpad=(taps - 1) // 2).fwd
    shift_v = dev(bias + rng.normal(0, 0.1, size=N))
    gamma, beta = dev(rng.normal(1, 0.2, size=N)), dev(rng.normal(0, 0.2, size=N))
    for about in (shift_v, None):
        out = torch.empty(M, N, device='cuda')
        ptr, stride = T.SumSlots.on('cuda').forward_region(N)
        ops.gemm(x, lin, out, M=M, seg_len=seg, stats=(ptr, about))
        st = torch.empty(4, N, device='cuda')
        mm, mv = torch.zeros(N, device='cuda'), torch.ones(N, device='cuda')
        T.bn_finalize(None, None, about if about is not None else torch.zeros(N, device='cuda'), M, N, gamma, beta, mm, mv, *st,
                      slots=(ptr, stride))
        o = out.double().cpu().numpy()
        assert rel_err(st[0], o.mean(0)) < 1e-5
        assert rel_err(st[1], 1 / np.sqrt(o.var(0) + 1e-3)) < 1e-5
        plain = torch.empty(M, N, device='cuda')
        ops.gemm(x, lin, plain, M=M, seg_len=seg)
        assert torch.equal(plain, out)
        assert float(T.SumSlots.on('cuda').buf[2].abs().max()) == 0.0



@pytest.mark.parametrize('M,H,prec', [(32, 512, 'bf16x3'), (8, 128, 'bf16x3'), (37, 64, 'bf16x3'), (32, 512, 'bf16')])
def test_gru_step_in_one_launch_matches_product_plus_gate_kernel(T, M, H, prec):
    """estts_gru_fused_fwd / _bwd against estts_gemm followed by estts_gru_step_fwd / _bwd on the same operands (row strides of a
    [B, T, .] recurrence buffer), and the forward step against float64."""
    from estts import ops
    rng = np.random.default_rng(M + H)
    Tn = 3                                                        # rows are one time step of [M, Tn, .] buffers
    w_hh = rng.normal(0, 1 / math.sqrt(H), size=(3 * H, H))
    b_hh = rng.normal(0, 0.1, size=3 * H)
    lin = T.TrainLinear(dev(w_hh.T.copy()), dev(b_hh))            # kernel [K = H, N = 3H]
    gi = dev(rng.normal(size=(M, Tn, 3 * H)))
    hs = dev(rng.normal(0, 0.5, size=(M, Tn, H)))
    xr = dev(rng.normal(size=(M, Tn, H + 8)))
    hp = hs[:, 0]
    outs = []
    for fused in (False, True):
        gh = torch.zeros(M, Tn, 3 * H, device='cuda')
        hn = torch.zeros(M, Tn, H, device='cuda')
        out = torch.zeros(M, Tn, H + 8, device='cuda')
        if fused:
            T.gru_fused_fwd(gi[:, 1], gh[:, 1], hp, lin.fwd, hn[:, 1], x_res=xr[:, 1, :H], out=out[:, 1, :H], precision=prec)
        else:
            ops.gemm(hp, lin.fwd, gh[:, 1], precision=prec)
            T.gru_step_fwd(gi[:, 1], gh[:, 1], hp, hn[:, 1], x_res=xr[:, 1, :H], out=out[:, 1, :H])
        outs.append((gh, hn, out))
    tol = 2e-5 if prec == 'bf16x3' else 2e-2
    for a, b in zip(*outs):
        assert rel_err(b, a) < tol
    g64 = hp.double().cpu().numpy() @ w_hh.T + b_hh
    i64 = gi[:, 1].double().cpu().numpy()
    sg = lambda x: 1 / (1 + np.exp(-x))                           # noqa: E731
    r, z = sg(i64[:, :H] + g64[:, :H]), sg(i64[:, H:2 * H] + g64[:, H:2 * H])
    n = np.tanh(i64[:, 2 * H:] + r * g64[:, 2 * H:])
    h_ref = (1 - z) * n + z * hp.double().cpu().numpy()
    assert rel_err(outs[1][1][:, 1], h_ref) < (3e-5 if prec == 'bf16x3' else 2e-2)
    assert float(outs[1][1][:, 0].abs().max()) == 0.0 and float(outs[1][2][:, 1, H:].abs().max()) == 0.0    # nothing beside the step
    # backward: the state gradient from step t + 1 and the gate gradients of step t
    gh = outs[0][0]
    dgh_next = dev(rng.normal(0, 0.3, size=(M, Tn, 3 * H)))
    dh = dev(rng.normal(size=(M, Tn, H + 8)))
    dhd0 = dev(rng.normal(0, 0.3, size=(M, H)))
    res = []
    for fused in (False, True):
        dgi = torch.zeros(M, Tn, 3 * H, device='cuda')
        dgh = dgh_next.clone()
        dhd = dhd0.clone()
        if fused:
            T.gru_fused_bwd(gi[:, 1], gh[:, 1], hp, dh[:, 1, :H], dgh[:, 2], lin.bwd, dgi[:, 1], dgh[:, 1], dhd, precision=prec)
        else:
            dhr = torch.empty(M, H, device='cuda')
            ops.gemm(dgh[:, 2], lin.bwd, dhr, precision=prec, residual=dhd)
            T.gru_step_bwd(gi[:, 1], gh[:, 1], hp, dh[:, 1, :H], dhr, dgi[:, 1], dgh[:, 1], dhd)
        res.append((dgi, dgh, dhd))
    for a, b in zip(*res):
        assert rel_err(b, a) < tol


def test_expand_embed_loss_adam_axpby_headscale(T):
    from estts import ops
    rng = np.random.default_rng(5)
    # length regulator gradient
    B, N, D = 3, 11, 32
    dur = rng.integers(0, 6, size=(B, N)).astype(np.float32)
    dur[1, 7:] = 0
    Tmax = int(dur.sum(1).max())
    cum = torch.empty(B, N, dtype=torch.int32, device='cuda')
    tot = torch.empty(B, dtype=torch.int32, device='cuda')
    ops.durations_cumsum(dev(dur), cum, tot)
    dout = rng.normal(size=(B * Tmax, D))
    dx = torch.empty(B * N, D, device='cuda')
    T.expand_bwd(dev(dout), cum, dx, B, N, Tmax, D)
    ref = np.zeros((B, N, D))
    for b in range(B):
        t = 0
        for n in range(N):
            r = int(dur[b, n])
            ref[b, n] = dout.reshape(B, Tmax, D)[b, t:t + r].sum(0)
            t += r
    assert rel_err(dx, ref.reshape(B * N, D)) < 1e-6
    # embedding gradient
    V = 9
    tok = rng.integers(0, V, size=(B, N)).astype(np.int32)
    g = rng.normal(size=(B * N, D))
    demb = torch.zeros(V, D, device='cuda')
    T.embed_bwd(dev(tok, torch.int32), dev(g), demb, 2.5)
    ref = np.zeros((V, D))
    np.add.at(ref, tok.reshape(-1), g * 2.5)
    assert rel_err(demb, ref) < 1e-5
    # masked MAE gradient
    M, C = 50, 80
    tgt = rng.normal(size=(M, C))
    tgt[40:] = 0
    pred = rng.normal(size=(M, C))
    pt = torch.tensor(pred, requires_grad=True)
    w = (torch.tensor(tgt) != 0).any(-1).double()
    loss = ((torch.tensor(tgt) - pt).abs().mean(-1) * w).sum() / M
    (3.0 * loss).backward()
    dp = torch.empty(M, C, device='cuda')
    T.masked_mae_bwd(dev(tgt), dev(pred), dp, 3.0)
    assert rel_err(dp, pt.grad) < 1e-6
    val = torch.zeros(1, device='cuda')
    ops.masked_mae(dev(tgt), dev(pred), val)
    assert abs(float(val) - float(loss)) < 1e-5
    # Adam (Keras form), two steps
    n = 1000
    p0, g1, g2 = rng.normal(size=n), rng.normal(size=n), rng.normal(size=n)
    p, m, v = dev(p0), torch.zeros(n, device='cuda'), torch.zeros(n, device='cuda')
    pr, mr, vr = p0.copy(), np.zeros(n), np.zeros(n)
    for t, gg in enumerate((g1, g2), start=1):
        lr_t = 1e-2 * math.sqrt(1 - 0.98 ** t) / (1 - 0.9 ** t)
        T.adam(p, dev(gg), m, v, lr_t, grad_scale=0.5)
        ge = gg * 0.5
        mr = 0.9 * mr + 0.1 * ge
        vr = 0.98 * vr + 0.02 * ge * ge
        pr = pr - lr_t * mr / (np.sqrt(vr) + 1e-9)
    assert rel_err(p, pr) < 1e-5
    # axpby and head scaling
    a, b = rng.normal(size=(7, 16)), rng.normal(size=(7, 16))
    bd = dev(b)
    T.axpby(dev(a), bd, alpha=2.0, beta=1.0)
    assert rel_err(bd, 2 * a + b) < 1e-6
    B2, L, H, dep = 2, 5, 4, 8
    x = rng.normal(size=(B2 * L, H * dep))
    sc = rng.uniform(size=(B2, H))
    xd = dev(x)
    T.head_scale(xd, dev(sc.reshape(-1)), B2, L, H, dep)
    ref = (x.reshape(B2, L, H, dep) * sc[:, None, :, None]).reshape(B2 * L, H * dep)
    assert rel_err(xd, ref) < 1e-6


def test_scaled_crossentropy_reference_known_answers_and_gradient(T):
    """The stop-token loss against the constants of the reference's own test (TransformerTTS/tests/test_loss.py:10-24), then
    loss and gradient against the oracle / autograd on random logits with padding and stop labels."""
    from oracle import tts_numpy as O
    logits = dev([[.3, .2, .1], [.3, .2, .1], [.3, .2, .1]])
    labels = dev([0, 1, 2], torch.int32)
    for scaling, want in ((5.0, 2.3705523014068604), (1.0, 0.7679619193077087)):
        loss = torch.zeros(1, device='cuda')
        T.scaled_xent(logits, labels, loss, index=2, scaling=scaling)
        assert abs(loss.item() - want) < 1e-6
    rng = np.random.default_rng(0)
    B, L, C = 3, 37, 3
    lg = rng.normal(size=(B, L, C)).astype(np.float32) * 2
    tg = rng.integers(0, 3, size=(B, L)).astype(np.int32)
    loss = torch.zeros(1, device='cuda')
    d = torch.empty(B * L, C, device='cuda')
    T.scaled_xent(dev(lg).view(B * L, C), dev(tg, torch.int32).view(-1), loss, d, index=2, scaling=8.0)
    assert abs(loss.item() - O.new_scaled_crossentropy(2, 8.0)(tg, lg)) < 1e-5
    x = torch.tensor(lg, dtype=torch.float64, requires_grad=True)
    t = torch.tensor(tg).long()
    w = (t != 0).double() + (t == 2).double() * 7.0
    ref = (F.cross_entropy(x.view(-1, C), t.view(-1), reduction='none') * w.view(-1)).sum() / (B * L)
    ref.backward()
    assert abs(loss.item() - ref.item()) < 1e-5
    assert rel_err(d, x.grad.view(-1, C)) < 1e-5
