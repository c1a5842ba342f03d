"""Per-kernel parity tests (GPU): each HIP kernel, called through the C ABI, against a float64/float32
restatement of the same op on identical inputs.  Tolerances are stated per test."""
import math

import numpy as np
import pytest
import torch

import estts
from estts import ops
from estts.ops import PackedLinear

pytestmark = pytest.mark.gpu
DEV = 'cuda'


def _rand(*shape, seed=0, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    return (torch.randn(*shape, generator=g) * scale)


@pytest.mark.parametrize('M,N,K', [(600, 512, 768), (100, 1536, 512), (37, 80, 256), (1, 64, 4), (130, 1, 512)])
@pytest.mark.parametrize('precision,tol', [('bf16x3', 5e-5), ('bf16', 3e-2)])
def test_gemm_dense(M, N, K, precision, tol):
    x = _rand(M, K, seed=1)
    w = _rand(N, K, seed=2, scale=1 / math.sqrt(K))
    b = _rand(N, seed=3)
    lin = PackedLinear(w.numpy(), bias=b.numpy(), device=DEV)
    out = torch.full((M, N), float('nan'), device=DEV)
    ops.gemm(x.to(DEV), lin, out, precision=precision)
    ref = x.double() @ w.double().t() + b.double()
    err = (out.cpu().double() - ref).abs().max().item()
    assert err < tol, err


@pytest.mark.parametrize('tile', [1, 2, 3, 4, 5])
@pytest.mark.parametrize('splitk', [1, 2, 8])
def test_gemm_tiles_and_split_k(tile, splitk):
    # every tile shape x split-K factor against float64, several back-to-back launches sharing the workspace
    M, N, K = 200, 192, 1152
    x = _rand(M, K // 3, seed=1)
    w = _rand(N, K, seed=2, scale=1 / math.sqrt(K))
    b = _rand(N, seed=3)
    lin = PackedLinear(w.numpy(), bias=b.numpy(), device=DEV, taps=3, lpad=1, act='relu')
    xp = torch.nn.functional.pad(x.double(), (0, 0, 1, 1))
    ref = torch.relu(sum(xp[j:j + M] @ w.double()[:, j * (K // 3):(j + 1) * (K // 3)].t() for j in range(3)) + b.double())
    xd = x.to(DEV)
    for rep in range(3):
        out = torch.full((M, N), float('nan'), device=DEV)
        ops.gemm(xd, lin, out, seg_len=M, precision='bf16x3', tune=tile | (splitk << 8))
        assert (out.cpu().double() - ref).abs().max().item() < 5e-5
    _, counters = ops.gemm_workspace(DEV)
    assert int(counters.abs().sum().item()) == 0      # tickets are left at zero


def test_gemm_asymmetric_layout():
    # A = identity picks out W^T exactly; catches transposed / permuted fragment maps (values exact in bf16)
    K = N = 64
    x = torch.eye(K)
    w = torch.arange(N * K, dtype=torch.float32).reshape(N, K) % 251 - 125
    lin = PackedLinear(w.numpy(), device=DEV)
    out = torch.empty(K, N, device=DEV)
    ops.gemm(x.to(DEV), lin, out, precision='bf16')
    assert torch.equal(out.cpu(), w.t())


@pytest.mark.parametrize('taps,lpad,cin,cout,T,B', [(3, 1, 256, 512, 50, 2), (5, 2, 80, 256, 33, 3), (5, 4, 80, 80, 20, 1)])
def test_gemm_conv_epilogue(taps, lpad, cin, cout, T, B):
    # conv1d ('same' or causal) + scale/bias + tanh + residual + trailing affine, strided input/output views
    x_full = _rand(B * T, cin + 8, seed=4)
    x = x_full[:, :cin]
    kern = _rand(taps, cin, cout, seed=5, scale=1 / math.sqrt(taps * cin))
    s1, b1 = _rand(cout, seed=6).abs() + 0.5, _rand(cout, seed=7)
    s2, b2 = _rand(cout, seed=8).abs() + 0.5, _rand(cout, seed=9)
    res = _rand(B * T, cout, seed=10)
    lin = PackedLinear(kern.reshape(taps * cin, cout).t().numpy(), bias=b1.numpy(), scale=s1.numpy(), device=DEV,
                       taps=taps, lpad=lpad, act='tanh', s2=s2.numpy(), b2=b2.numpy())
    out_full = torch.zeros(B * T, cout + 4, device=DEV)
    ops.gemm(x_full.to(DEV)[:, :cin], lin, out_full[:, :cout], seg_len=T, residual=res.to(DEV), precision='bf16x3')
    xp = torch.nn.functional.pad(x.double().view(B, T, cin), (0, 0, lpad, taps - 1 - lpad))
    acc = sum(xp[:, j:j + T] @ kern[j].double() for j in range(taps)).reshape(B * T, cout)
    ref = (torch.tanh(acc * s1.double() + b1.double()) + res.double()) * s2.double() + b2.double()
    # split-bf16 carries ~16 mantissa bits per operand: tolerance relative to the output range
    err = ((out_full[:, :cout].cpu().double() - ref).abs().max() / ref.abs().max()).item()
    assert err < 3e-5, err
    assert (out_full[:, cout:] == 0).all()


@pytest.mark.parametrize('tile', [1, 4, 5, 6, 7, 8, 13, 9, 10, 9 | (4 << 16), 10 | (4 << 16)])
def test_gemm_tile_layouts_identity_and_conv(tile):
    # identity input picks out W^T exactly (values exact in bf16): catches permuted fragment maps of each kernel
    K, N = 96, 160
    w = torch.arange(N * K, dtype=torch.float32).reshape(N, K) % 251 - 125
    lin = PackedLinear(w.numpy(), device=DEV)
    out = torch.empty(K, N, device=DEV)
    ops.gemm(torch.eye(K).to(DEV), lin, out, precision='bf16', tune=tile)
    assert torch.equal(out.cpu(), w.t())
    # causal conv, 3 segments, N = 80 (partial column tile), every epilogue stage
    taps, lpad, cin, cout, T, B = 5, 4, 80, 80, 67, 3
    x = _rand(B * T, cin, seed=4)
    kern = _rand(taps, cin, cout, seed=5, scale=1 / math.sqrt(taps * cin))
    s1, b1, s2, b2 = _rand(cout, seed=6).abs() + 0.5, _rand(cout, seed=7), _rand(cout, seed=8).abs() + 0.5, _rand(cout, seed=9)
    res = _rand(B * T, cout, seed=10)
    lin = PackedLinear(kern.reshape(taps * cin, cout).t().numpy(), bias=b1.numpy(), scale=s1.numpy(), device=DEV,
                       taps=taps, lpad=lpad, act='tanh', s2=s2.numpy(), b2=b2.numpy())
    out = torch.empty(B * T, cout, device=DEV)
    ops.gemm(x.to(DEV), lin, out, seg_len=T, residual=res.to(DEV), precision='bf16x3', tune=tile)
    xp = torch.nn.functional.pad(x.double().view(B, T, cin), (0, 0, lpad, taps - 1 - lpad))
    acc = sum(xp[:, j:j + T] @ kern[j].double() for j in range(taps)).reshape(B * T, cout)
    ref = (torch.tanh(acc * s1.double() + b1.double()) + res.double()) * s2.double() + b2.double()
    assert ((out.cpu().double() - ref).abs().max() / ref.abs().max()).item() < 3e-5


@pytest.mark.parametrize('tile', [9, 10, 11, 12])
@pytest.mark.parametrize('waves', [4, 8])
@pytest.mark.parametrize('taps,cin,lpad', [(1, 32, 0), (1, 264, 0), (3, 96, 1), (5, 80, 2), (3, 256, 2), (1, 1536, 0), (3, 512, 1), (5, 256, 4)])
def test_gemm_oneshot_kernel(tile, waves, taps, cin, lpad):
    # the all-loads-up-front kernel at every K-step count a wave can hold (1..6), K tails inside a 32-wide step, rows past M,
    # partial column tiles, several segments (tiles that straddle a sequence boundary), both precisions; float64 reference.
    # Tiles 11 / 12: the convolution variant that stages the tile's source rows in LDS (always eight waves, taps > 1 only).
    K = taps * cin
    if tile >= 11 and waves == 4:
        pytest.skip('the staged variant has one wave count')
    if (K + 31) // 32 > 6 * waves or (tile >= 11 and taps == 1):
        with pytest.raises(RuntimeError, match='one-shot'):
            ops.gemm(torch.zeros(8, cin, device=DEV), PackedLinear(np.zeros((16, K), np.float32), device=DEV, taps=taps, lpad=lpad),
                     torch.zeros(8, 16, device=DEV), tune=tile | (waves << 16))
        return
    B, T, N = 3, 23, 72
    x = _rand(B * T, cin, seed=4)
    kern = _rand(taps, cin, N, seed=5, scale=1 / math.sqrt(K))
    s1, b1, s2, b2 = _rand(N, seed=6).abs() + 0.5, _rand(N, seed=7), _rand(N, seed=8).abs() + 0.5, _rand(N, seed=9)
    res = _rand(T, N, seed=10)
    lin = PackedLinear(kern.reshape(K, N).t().numpy(), bias=b1.numpy(), scale=s1.numpy(), device=DEV,
                       taps=taps, lpad=lpad, act='relu', s2=s2.numpy(), b2=b2.numpy())
    xp = torch.nn.functional.pad(x.double().view(B, T, cin), (0, 0, lpad, taps - 1 - lpad))
    acc = sum(xp[:, j:j + T] @ kern[j].double() for j in range(taps)).reshape(B * T, N)
    ref = (torch.relu(acc * s1.double() + b1.double()) * 0.5 + res.double().repeat(B, 1)) * s2.double() + b2.double()
    for precision, tol in (('bf16x3', 3e-5), ('bf16', 2e-2)):
        out = torch.full((B * T, N + 3), float('nan'), device=DEV)
        ops.gemm(x.to(DEV), lin, out[:, :N], seg_len=T, residual=res.to(DEV), r_mod=T, alpha=0.5, precision=precision,
                 tune=tile | (waves << 16))
        assert ((out[:, :N].cpu().double() - ref).abs().max() / ref.abs().max()).item() < tol
        assert torch.isnan(out[:, N:]).all()


def test_gemm_oneshot_kernels_agree_with_the_ring_kernel_on_random_shapes():
    # 120 random dense / convolution shapes (row counts from 1, column counts that leave partial tiles, K tails inside a 32-wide
    # step, 1..4 sequences so that tiles straddle sequence boundaries, causal and 'same' padding, padded row strides on input,
    # output and residual, r_mod): every one-shot kernel that accepts the shape against the register-ring kernel (tile 4),
    # which the float64 tests above pin.  Same products in a different summation order: 2e-5 of the output range.
    rng = np.random.default_rng(7)
    checked = 0
    for case in range(120):
        taps = int(rng.choice([1, 1, 3, 5]))
        cin = 8 * int(rng.integers(1, 40 if taps == 1 else 24))
        K = taps * cin
        if K > 1536:
            continue
        B, T = int(rng.integers(1, 5)), int(rng.integers(1, 70))
        N = int(rng.integers(1, 150))
        lpad = int(rng.integers(0, taps))
        M = B * T
        pa, pc, pr = (4 * int(rng.integers(0, 3)) for _ in range(3))
        x = _rand(M, cin + pa, seed=1000 + case)
        w = _rand(N, K, seed=2000 + case, scale=1 / math.sqrt(K))
        b1, s1 = _rand(N, seed=3000 + case), _rand(N, seed=4000 + case).abs() + 0.5
        use_rmod = bool(rng.integers(0, 2)) and T > 1
        res = _rand(T if use_rmod else M, N + pr, seed=5000 + case)
        lin = PackedLinear(w.numpy(), bias=b1.numpy(), scale=s1.numpy(), device=DEV, taps=taps, lpad=lpad, act='tanh')
        xd, rd = x.to(DEV), res.to(DEV)

        def run(tune):
            out = torch.full((M, N + pc), float('nan'), device=DEV)
            ops.gemm(xd[:, :cin], lin, out[:, :N], seg_len=T, residual=rd[:, :N], r_mod=T if use_rmod else 0, alpha=0.7,
                     precision='bf16x3', tune=tune)
            assert torch.isnan(out[:, N:]).all()
            return out[:, :N].clone()

        ref = run(4)
        assert torch.isfinite(ref).all()
        for tune in (9 | (4 << 16), 9 | (8 << 16), 10 | (4 << 16), 10 | (8 << 16), 11, 12, 0):
            try:
                got = run(tune)
            except RuntimeError as e:
                assert 'one-shot' in str(e), e          # outside that kernel's envelope (K-steps per wave, taps == 1)
                continue
            err = ((got - ref).abs().max() / ref.abs().max().clamp_min(1e-3)).item()
            assert err < 2e-5, (case, tune, M, N, K, taps, lpad, T, err)
            checked += 1
    assert checked > 400


def test_gemm_large_tiles_agree_with_the_ring_kernel_on_random_shapes():
    # The kernels for thousands of rows (LDS-staged tiles 5..8, and tile 13, which reads W from memory straight into the matrix
    # instruction's registers) on 80 random dense / convolution shapes: partial row and column tiles, K tails inside a 32-wide
    # step and K-step counts from 1 (every branch of the peeled main loop: fewer than four steps, odd and even counts),
    # 1..4 sequences so that tiles straddle sequence boundaries, causal and 'same' padding, padded row strides, r_mod; against
    # the register-ring kernel (tile 4), which the float64 tests above pin.  Tile 13 does the arithmetic of tile 8 in the same
    # order: those two must agree bit for bit.
    rng = np.random.default_rng(11)
    checked = 0
    for case in range(80):
        taps = int(rng.choice([1, 1, 3, 5]))
        cin = 4 * int(rng.integers(1, 80 if taps == 1 else 40))
        K = taps * cin
        B, T = int(rng.integers(1, 5)), int(rng.integers(1, 160))
        N = int(rng.integers(1, 300))
        lpad = int(rng.integers(0, taps))
        M = B * T
        pa, pc, pr = (4 * int(rng.integers(0, 3)) for _ in range(3))
        x = _rand(M, cin + pa, seed=6000 + case)
        w = _rand(N, K, seed=7000 + case, scale=1 / math.sqrt(K))
        b1, s1 = _rand(N, seed=8000 + case), _rand(N, seed=9000 + case).abs() + 0.5
        use_rmod = bool(rng.integers(0, 2)) and T > 1
        res = _rand(T if use_rmod else M, N + pr, seed=10000 + case)
        lin = PackedLinear(w.numpy(), bias=b1.numpy(), scale=s1.numpy(), device=DEV, taps=taps, lpad=lpad, act='tanh')
        xd, rd = x.to(DEV), res.to(DEV)

        def run(tune):
            out = torch.full((M, N + pc), float('nan'), device=DEV)
            ops.gemm(xd[:, :cin], lin, out[:, :N], seg_len=T, residual=rd[:, :N], r_mod=T if use_rmod else 0, alpha=0.7,
                     precision='bf16x3', tune=tune)
            assert torch.isnan(out[:, N:]).all()
            return out[:, :N].clone()

        ref = run(4)
        assert torch.isfinite(ref).all()
        got = {}
        for tune in (5, 6, 7, 8, 13):
            got[tune] = run(tune)
            err = ((got[tune] - ref).abs().max() / ref.abs().max().clamp_min(1e-3)).item()
            assert err < 2e-5, (case, tune, M, N, K, taps, lpad, T, err)
            checked += 1
        assert torch.equal(got[13], got[8]), (case, M, N, K, taps, lpad, T)
    assert checked == 400


def test_gemm_large_tiles_id_14_agrees_with_id_13():
    # eight waves side by side (64 x 256 tiles): the arithmetic of tile 13 in the same order, on shapes with partial row and
    # column tiles, K tails, convolutions across sequence boundaries, and every epilogue operand
    rng = np.random.default_rng(12)
    for case in range(24):
        taps = int(rng.choice([1, 1, 3, 5]))
        cin = 4 * int(rng.integers(1, 80 if taps == 1 else 40))
        K = taps * cin
        B, T = int(rng.integers(1, 5)), int(rng.integers(1, 200))
        N = int(rng.choice([256, 256, 512, 300, 100, 37]))
        lpad = int(rng.integers(0, taps))
        M = B * T
        x = _rand(M, cin, seed=16000 + case).to(DEV)
        w = _rand(N, K, seed=17000 + case, scale=1 / math.sqrt(K))
        lin = PackedLinear(w.numpy(), bias=_rand(N, seed=18000 + case).numpy(), scale=(_rand(N, seed=19000 + case).abs() + 0.5).numpy(),
                           device=DEV, taps=taps, lpad=lpad, act='relu')
        res = _rand(M, N, seed=20000 + case).to(DEV)
        outs = []
        for tune in (13, 14):
            out = torch.full((M, N + 4), float('nan'), device=DEV)
            ops.gemm(x, lin, out[:, :N], seg_len=T, residual=res, alpha=0.7, precision='bf16x3', tune=tune)
            assert torch.isnan(out[:, N:]).all()
            outs.append(out[:, :N].clone())
        assert torch.isfinite(outs[0]).all()
        assert torch.equal(outs[0], outs[1]), (case, M, N, K, taps, lpad, T)
    # launches of more tiles than CUs whose last round would be at most half full send that round's rows out as 32-row tiles
    # behind the 64-row ones (tune bit 26 turns that off): same values wherever a row lands, ragged ends included
    for case, (M, N, cin, taps, T) in enumerate([(19200, 256, 512, 1, 600), (19203, 256, 64, 3, 6401), (17000, 384, 96, 1, 17000),
                                                  (16640, 256, 128, 1, 16640), (33000, 100, 32, 5, 1100), (9000, 512, 64, 1, 9000)]):
        K = taps * cin
        x = _rand(M, cin, seed=30000 + case).to(DEV)
        lin = PackedLinear(_rand(N, K, seed=31000 + case, scale=1 / math.sqrt(K)).numpy(), bias=_rand(N, seed=32000 + case).numpy(),
                           device=DEV, taps=taps, lpad=(taps - 1) // 2, act='relu')
        res = _rand(M, N, seed=33000 + case).to(DEV)
        outs = []
        for tune in (13 | (1 << 26), 13, 14 | (1 << 26), 14):
            out = torch.full((M + 1, N + 4), float('nan'), device=DEV)
            ops.gemm(x, lin, out[:M, :N], seg_len=T, residual=res, precision='bf16x3', tune=tune)
            assert torch.isnan(out[:, N:]).all() and torch.isnan(out[M:]).all()
            outs.append(out[:M, :N].clone())
        assert torch.isfinite(outs[0]).all()
        for o in outs[1:]:
            assert torch.equal(outs[0], o), (case, M, N, K, taps)


def test_gemm_automatic_choice_agrees_with_the_ring_kernel_at_thousands_of_rows():
    # tune = 0 over the sizes where it picks among ids 6 / 13 / 14 / 15 and the mixed 64- / 32-row launches (1500 .. 40000 rows,
    # 65 .. 1100 columns, K with and without a multiple of 64, 1 / 3 / 5 taps over several sequences, residual, both affine
    # pairs, activation) against the register-ring kernel (id 4), which the float64 tests pin
    rng = np.random.default_rng(14)
    for case in range(28):
        taps = int(rng.choice([1, 1, 3, 5]))
        cin = (64 if rng.integers(0, 2) else 4) * int(rng.integers(1, 9 if taps == 1 else 5))
        K = taps * cin
        B = int(rng.integers(1, 5))
        T = int(rng.integers(1500, 40000)) // B
        M = B * T
        N = int(rng.choice([65, 80, 128, 200, 256, 256, 300, 512, 768, 1100]))
        lpad = int(rng.integers(0, taps))
        x = _rand(M, cin, seed=50000 + case).to(DEV)
        lin = PackedLinear(_rand(N, K, seed=51000 + case, scale=1 / math.sqrt(K)).numpy(), bias=_rand(N, seed=52000 + case).numpy(),
                           scale=(_rand(N, seed=53000 + case).abs() + 0.5).numpy(), device=DEV, taps=taps, lpad=lpad, act='relu')
        res = _rand(M, N, seed=54000 + case).to(DEV)
        outs = []
        for tune in (4, 0):
            out = torch.full((M + 1, N + 4), float('nan'), device=DEV)
            ops.gemm(x, lin, out[:M, :N], seg_len=T, residual=res, alpha=0.7, precision='bf16x3', tune=tune)
            assert torch.isnan(out[:, N:]).all() and torch.isnan(out[M:]).all()
            outs.append(out[:M, :N].clone())
        assert torch.isfinite(outs[0]).all()
        err = ((outs[0] - outs[1]).abs().max() / outs[0].abs().max()).item()
        assert err < 2e-5, (case, M, N, K, taps, lpad, T, err)


def test_gemm_two_k_groups_id_15_agrees_with_id_13():
    # 64 x 128 tiles whose eight waves take the two halves of K in two groups and add their accumulators up at the end: the
    # products of tile 13 in another order of summation (2e-6 of the largest value), with partial tiles, every epilogue operand,
    # convolutions across sequence boundaries, K from 64 (one step per group) up; K not a multiple of 64 is an error
    rng = np.random.default_rng(13)
    for case in range(24):
        taps = int(rng.choice([1, 1, 3, 5]))
        cin = 64 * int(rng.integers(1, 9)) if taps == 1 else 64 * int(rng.integers(1, 4))
        K = taps * cin
        B, T = int(rng.integers(1, 5)), int(rng.integers(1, 200))
        N = int(rng.choice([256, 512, 300, 100, 37, 128]))
        lpad = int(rng.integers(0, taps))
        M = B * T
        x = _rand(M, cin, seed=40000 + case).to(DEV)
        lin = PackedLinear(_rand(N, K, seed=41000 + case, scale=1 / math.sqrt(K)).numpy(), bias=_rand(N, seed=42000 + case).numpy(),
                           scale=(_rand(N, seed=43000 + case).abs() + 0.5).numpy(), device=DEV, taps=taps, lpad=lpad, act='tanh')
        res = _rand(M, N, seed=44000 + case).to(DEV)
        outs = []
        for tune in (13, 15):
            out = torch.full((M + 1, N + 4), float('nan'), device=DEV)
            ops.gemm(x, lin, out[:M, :N], seg_len=T, residual=res, alpha=0.7, precision='bf16x3', tune=tune)
            assert torch.isnan(out[:, N:]).all() and torch.isnan(out[M:]).all()
            outs.append(out[:M, :N].clone())
        assert torch.isfinite(outs[0]).all()
        err = ((outs[0] - outs[1]).abs().max() / outs[0].abs().max()).item()
        assert err < 2e-6, (case, M, N, K, taps, lpad, T, err)
    lin = PackedLinear(_rand(128, 96, seed=1).numpy(), device=DEV)
    with pytest.raises(RuntimeError, match='multiple of 64'):
        ops.gemm(_rand(64, 96, seed=2).to(DEV), lin, torch.empty(64, 128, device=DEV), tune=15)


@pytest.mark.parametrize('precision', ['bf16x3', 'bf16'])
def test_gemm_layernorm_epilogue_is_gemm_then_rownorm(precision):
    # estts_gemm_args.ln_epilogue: the rownorm() that follows a 256-output layer inside the layer's launch -- the three
    # combinations the transformer blocks use (layers.py:65-71 FFNResNorm, :163-168 SelfAttentionResNorm, :270-274
    # CrossAttentionResnorm) and a single norm, on dense layers and a convolution, row counts that end inside a 64-row tile, padded
    # row strides, the output aliasing the residual.  Same products in the same order and the same row arithmetic as
    # rownorm_fast_kernel: equal up to the contraction of multiply-adds (1e-6 of the largest value).
    N = 256
    cases = [  # (B, T, cin, taps, lpad, ln_a, act, residual, ln_b)
        (1, 64, 512, 1, 0, True, None, True, True),
        (2, 100, 256, 1, 0, True, 'relu', True, True),
        (3, 211, 512, 1, 0, False, None, True, True),
        (1, 1000, 96, 1, 0, True, 'relu', False, False),
        (2, 150, 64, 3, 1, True, None, True, True),
        (32, 600, 512, 1, 0, True, None, True, True),     # 300 row tiles: 256 of 64 rows and 88 of 32
        (1, 16999, 64, 1, 0, True, 'relu', True, True),    # ... ending inside a 32-row tile
        (3, 6000, 32, 3, 1, False, None, True, True),
        (1, 1, 256, 1, 0, False, None, True, True),
    ]
    for idx, (B, T, cin, taps, lpad, use_a, act, use_res, use_b) in enumerate(cases):
        M, K = B * T, taps * cin
        x = _rand(M, cin + 4, seed=21000 + idx).to(DEV)
        w = _rand(N, K, seed=22000 + idx, scale=1 / math.sqrt(K))
        lin = PackedLinear(w.numpy(), bias=_rand(N, seed=23000 + idx).numpy(), device=DEV, taps=taps, lpad=lpad)
        ln_a = ((_rand(N, seed=24000 + idx).abs() + 0.5).to(DEV), _rand(N, seed=25000 + idx).to(DEV)) if use_a else None
        ln_b = ((_rand(N, seed=26000 + idx).abs() + 0.5).to(DEV), _rand(N, seed=27000 + idx).to(DEV)) if use_b else None
        res = _rand(M, N + 8, seed=28000 + idx).to(DEV) if use_res else None
        y = torch.empty(M, N, device=DEV)
        want = torch.empty(M, N, device=DEV)
        ops.gemm(x[:, :cin], lin, y, seg_len=T, precision=precision, tune=13)
        ops.rownorm(y, want, ln_a=ln_a, act=act, residual=None if res is None else res[:, :N], ln_b=ln_b)
        got = torch.full((M + 3, N + 4), float('nan'), device=DEV)
        ops.gemm(x[:, :cin], lin, got[:M, :N], seg_len=T, precision=precision,
                 norm=dict(ln_a=ln_a, act=act, residual=None if res is None else res[:, :N], ln_b=ln_b))
        assert torch.isnan(got[M:]).all() and torch.isnan(got[:, N:]).all()
        assert torch.isfinite(want).all()
        err = (got[:M, :N] - want).abs().max().item() / want.abs().max().item()
        assert err < 1e-6, (idx, precision, err)
        if use_res:   # in place over the residual rows, as a block that overwrites its input would
            r2 = res[:, :N].clone()
            ops.gemm(x[:, :cin], lin, r2, seg_len=T, precision=precision, norm=dict(ln_a=ln_a, act=act, residual=r2, ln_b=ln_b))
            assert torch.equal(r2, got[:M, :N]), idx
    # what the launch cannot do is an error, not a silent fallback
    lin = PackedLinear(_rand(128, 64, seed=1).numpy(), device=DEV)
    with pytest.raises(RuntimeError, match='LayerNorm epilogue'):
        ops.gemm(_rand(64, 64, seed=2).to(DEV), lin, torch.empty(64, 128, device=DEV), norm=dict(ln_b=(torch.ones(128, device=DEV), torch.zeros(128, device=DEV))))


@pytest.mark.parametrize('B,T,cin,taps,lpad,N', [(1, 100, 512, 3, 1, 512), (1, 600, 256, 3, 1, 512), (3, 23, 96, 5, 2, 72),
                                                (2, 37, 264, 2, 1, 40), (4, 150, 256, 3, 2, 256)])
@pytest.mark.parametrize('form', ['double', 'single_relu', 'residual_only_a'])
def test_gemm_layernorm_prologue(B, T, cin, taps, lpad, N, form):
    # the convolution that computes its input rows as LNb(act(LNa(y)) + x) on the way in (and writes them out once) against
    # estts_rownorm followed by the same convolution: same kernels downstream of the rows, so 2e-5 of the output range;
    # several sequences (halo rows across sequence boundaries are masked at use), 16 x 16 and 32 x 32 tiles, both row widths
    M, K = B * T, taps * cin
    y, x = _rand(M, cin + 4, seed=21), _rand(M, cin, seed=22)
    ga, ba = _rand(cin, seed=23).abs() + 0.5, _rand(cin, seed=24)
    gb, bb = _rand(cin, seed=25).abs() + 0.5, _rand(cin, seed=26)
    w = _rand(N, K, seed=27, scale=1 / math.sqrt(K))
    lin = PackedLinear(w.numpy(), bias=_rand(N, seed=28).numpy(), device=DEV, taps=taps, lpad=lpad, act='relu')
    yd, xd = y.to(DEV)[:, :cin], x.to(DEV)
    ln_a, ln_b = (ga.to(DEV), ba.to(DEV)), (gb.to(DEV), bb.to(DEV))
    kw = {'double': dict(ln_a=ln_a, ln_b=ln_b, residual=xd, act=None),
          'single_relu': dict(ln_a=ln_a, ln_b=None, residual=None, act='relu'),
          'residual_only_a': dict(ln_a=ln_a, ln_b=None, residual=xd, act='relu')}[form]
    assert ops.can_fuse_ln(lin, M)
    rows_ref = torch.empty(M, cin, device=DEV)
    ops.rownorm(yd, rows_ref, ln_a=kw['ln_a'], act=kw['act'], residual=kw['residual'], ln_b=kw['ln_b'])
    out_ref = torch.empty(M, N, device=DEV)
    ops.gemm(rows_ref, lin, out_ref, seg_len=T, precision='bf16x3', tune=4)
    rows = torch.full((M, cin + 8), float('nan'), device=DEV)
    out = torch.full((M, N + 4), float('nan'), device=DEV)
    ops.gemm(rows[:, :cin], lin, out[:, :N], seg_len=T, precision='bf16x3',
             ln=dict(y=yd, x=kw['residual'], ln_a=kw['ln_a'], ln_b=kw['ln_b'], act=kw['act'], out=rows[:, :cin]))
    assert torch.isnan(rows[:, cin:]).all() and torch.isnan(out[:, N:]).all()
    assert (rows[:, :cin] - rows_ref).abs().max().item() < 2e-5 * max(1.0, rows_ref.abs().max().item())
    assert ((out[:, :N] - out_ref).abs().max() / out_ref.abs().max()).item() < 2e-5
    # what cannot take the prologue says so (rows wider than 512), and aliasing is refused; a dense layer is offered the
    # prologue only for a handful of rows (every column tile recomputes the norms)
    dense = PackedLinear(_rand(N, cin, seed=29).numpy(), device=DEV)
    assert not ops.can_fuse_ln(dense, M) and ops.can_fuse_ln(dense, 16)
    wide = PackedLinear(_rand(N, 1024, seed=30).numpy(), device=DEV)
    wide_rows = torch.empty(M, 1024, device=DEV)
    with pytest.raises(RuntimeError, match='LayerNorm prologue'):
        ops.gemm(wide_rows, wide, out[:, :N],
                 ln=dict(y=torch.zeros(M, 1024, device=DEV), ln_a=(torch.ones(1024, device=DEV), torch.zeros(1024, device=DEV)),
                         out=wide_rows))
    with pytest.raises(ValueError, match='alias'):
        ops.gemm(rows[:, :cin], lin, yd, seg_len=T, ln=dict(y=yd, ln_a=ln_a, out=rows[:, :cin]))


@pytest.mark.parametrize('M,K,N', [(1, 256, 768), (1, 256, 800), (3, 128, 40), (16, 512, 256), (7, 264, 72)])
@pytest.mark.parametrize('form', ['double', 'relu_double', 'second_only'])
def test_dense_layernorm_prologue_for_few_rows(M, K, N, form):
    # the one-row products of incremental decoding: LN(LN(y) + x) (SelfAttentionResNorm), LN(relu(LN(y)) + x) (FFNResNorm) and
    # LN(y + x) (CrossAttentionResnorm) computed inside the dense layer that consumes them, against estts_rownorm + the same
    # layer; the rows are also written out once
    y, x = _rand(M, K + 4, seed=31), _rand(M, K, seed=32)
    ln_a = (_rand(K, seed=33).abs().to(DEV) + 0.5, _rand(K, seed=34).to(DEV))
    ln_b = (_rand(K, seed=35).abs().to(DEV) + 0.5, _rand(K, seed=36).to(DEV))
    lin = PackedLinear(_rand(N, K, seed=37, scale=1 / math.sqrt(K)).numpy(), bias=_rand(N, seed=38).numpy(), device=DEV)
    yd, xd = y.to(DEV)[:, :K], x.to(DEV)
    kw = {'double': dict(ln_a=ln_a, ln_b=ln_b, act=None), 'relu_double': dict(ln_a=ln_a, ln_b=ln_b, act='relu'),
          'second_only': dict(ln_a=None, ln_b=ln_b, act=None)}[form]
    assert ops.can_fuse_ln(lin, M)
    rows_ref = torch.empty(M, K, device=DEV)
    ops.rownorm(yd, rows_ref, ln_a=kw['ln_a'], act=kw['act'], residual=xd, ln_b=kw['ln_b'])
    out_ref = torch.empty(M, N, device=DEV)
    ops.gemm(rows_ref, lin, out_ref, precision='bf16x3')
    rows = torch.full((M, K + 8), float('nan'), device=DEV)
    out = torch.full((M, N + 4), float('nan'), device=DEV)
    ops.gemm(rows[:, :K], lin, out[:, :N], precision='bf16x3', ln=dict(y=yd, x=xd, out=rows[:, :K], **kw))
    assert torch.isnan(rows[:, K:]).all() and torch.isnan(out[:, N:]).all()
    assert (rows[:, :K] - rows_ref).abs().max().item() < 2e-5 * max(1.0, rows_ref.abs().max().item())
    assert ((out[:, :N] - out_ref).abs().max() / out_ref.abs().max()).item() < 2e-5
    with pytest.raises(ValueError, match='ln_a or ln_b'):
        ops.gemm(rows[:, :K], lin, out[:, :N], ln=dict(y=yd, x=xd, out=rows[:, :K], ln_a=None, ln_b=ln_b, act='relu'))


def test_gemm_posenc_residual_shared_rows():
    B, T, K, N = 3, 17, 64, 32
    x, w, pe = _rand(B * T, K, seed=1), _rand(N, K, seed=2, scale=0.1), _rand(40, N, seed=3)
    lin = PackedLinear(w.numpy(), device=DEV, act='relu')
    out = torch.empty(B * T, N, device=DEV)
    ops.gemm(x.to(DEV), lin, out, alpha=4.0, residual=pe.to(DEV), r_mod=T, precision='bf16x3')
    ref = torch.relu(x.double() @ w.double().t()) * 4.0 + pe[:T].double().repeat(B, 1)
    assert ((out.cpu().double() - ref).abs().max() / ref.abs().max()).item() < 3e-5


def test_gemm_rejects_bad_arguments():
    lin = PackedLinear(np.zeros((8, 16), np.float32), device=DEV)
    with pytest.raises(ValueError):
        ops.gemm(torch.zeros(4, 8, device=DEV), lin, torch.zeros(4, 8, device=DEV))
    with pytest.raises(TypeError):
        ops.gemm(torch.zeros(4, 16, device=DEV, dtype=torch.float16), lin, torch.zeros(4, 8, device=DEV))


def _attention_ref(q, k, v, mask, causal):
    B, H, Lq, dh = q.shape
    s = q @ k.transpose(-1, -2) / math.sqrt(dh)
    m = torch.zeros(B, 1, Lq, k.shape[2], dtype=q.dtype)
    if mask is not None:
        m = torch.maximum(m, mask[:, None, None, :].to(q.dtype))
    if causal:
        m = torch.maximum(m, torch.triu(torch.ones(Lq, k.shape[2], dtype=q.dtype), diagonal=1))
    p = torch.softmax(s + m * -1e9, dim=-1)
    return p @ v, p


@pytest.mark.parametrize('B,H,Lq,Lk,dh,causal,masked', [
    (1, 4, 600, 600, 64, False, False), (2, 4, 100, 100, 128, False, True), (3, 2, 37, 37, 32, True, True),
    (2, 4, 1, 10, 64, False, False), (2, 4, 50, 75, 64, False, True)])
@pytest.mark.parametrize('precision,tol', [('bf16x3', 3e-5), ('bf16', 3e-2)])
def test_attention(B, H, Lq, Lk, dh, causal, masked, precision, tol):
    d = H * dh
    q, k, v = _rand(B * Lq, d, seed=1), _rand(B * Lk, d, seed=2), _rand(B * Lk, d, seed=3)
    mask = None
    if masked:
        mask = torch.zeros(B, Lk)
        for b in range(B):
            mask[b, Lk - 1 - 3 * b:] = 1.0
    ctx = torch.full((B * Lq, d + 4), float('nan'), device=DEV)
    attn = torch.full((B, H, Lq, Lk), float('nan'), device=DEV)
    ops.attention(q.to(DEV), k.to(DEV), v.to(DEV), ctx[:, :d], B, H, Lq, Lk, dh,
                  key_mask=None if mask is None else mask.to(DEV), causal=causal, attn=attn, precision=precision)
    split = lambda t, L: t.double().view(B, L, H, dh).permute(0, 2, 1, 3)
    ref, pref = _attention_ref(split(q, Lq), split(k, Lk), split(v, Lk), mask, causal)
    ref = ref.permute(0, 2, 1, 3).reshape(B * Lq, d)
    assert (ctx[:, :d].cpu().double() - ref).abs().max().item() < tol
    assert (attn.cpu().double() - pref).abs().max().item() < tol
    assert torch.isnan(ctx[:, d:]).all()          # columns outside the view untouched
    # without the weights output the context must be identical
    ctx2 = torch.empty(B * Lq, d, device=DEV)
    ops.attention(q.to(DEV), k.to(DEV), v.to(DEV), ctx2, B, H, Lq, Lk, dh,
                  key_mask=None if mask is None else mask.to(DEV), causal=causal, attn=None, precision=precision)
    assert torch.equal(ctx2, ctx[:, :d])


@pytest.mark.parametrize('B,H,Lq,Lk,dh,causal,masked,first_masked', [
    (2, 2, 150, 150, 64, False, True, False), (2, 2, 150, 150, 64, True, True, False), (1, 2, 130, 130, 64, True, False, True),
    (2, 1, 70, 45, 32, False, True, False), (1, 2, 65, 201, 128, False, False, False), (2, 2, 200, 200, 32, True, False, True),
    (4, 2, 150, 150, 64, True, True, False), (8, 4, 130, 97, 64, False, True, False), (2, 4, 330, 330, 64, True, False, True)])
@pytest.mark.parametrize('with_weights', [False, True])
def test_attention_64_query_variant(B, H, Lq, Lk, dh, causal, masked, first_masked, with_weights):
    """The batched kernel (64 queries per workgroup, shared K / V tiles), forced with variant=2: ragged Lq / Lk, key
    masks, the causal early stop, and a query whose visible keys are all masked (key 0 masked under a causal mask):
    float32 absorbs every score into -1e9 there, so that row is uniform over ALL keys, as in the reference."""
    d = H * dh
    q, k, v = _rand(B * Lq, d, seed=11), _rand(B * Lk, d, seed=12), _rand(B * Lk, d, seed=13)
    mask = None
    if masked or first_masked:
        mask = torch.zeros(B, Lk)
        if masked:
            for b in range(B):
                mask[b, Lk - 1 - 3 * b:] = 1.0
        if first_masked:
            mask[:, 0] = 1.0
    ctx = torch.full((B * Lq, d + 4), float('nan'), device=DEV)
    attn = torch.full((B, H, Lq, Lk), float('nan'), device=DEV) if with_weights else None
    ops.attention(q.to(DEV), k.to(DEV), v.to(DEV), ctx[:, :d], B, H, Lq, Lk, dh,
                  key_mask=None if mask is None else mask.to(DEV), causal=causal, attn=attn, variant=2)
    split = lambda t, L: t.double().view(B, L, H, dh).permute(0, 2, 1, 3)
    qs, ks, vs = split(q, Lq), split(k, Lk), split(v, Lk)
    m = torch.zeros(B, 1, Lq, Lk, dtype=torch.float64)
    if mask is not None:
        m = torch.maximum(m, mask[:, None, None, :].double())
    if causal:
        m = torch.maximum(m, torch.triu(torch.ones(Lq, Lk, dtype=torch.float64), diagonal=1))
    logits = ((qs @ ks.transpose(-1, -2)) / math.sqrt(dh)).float() + (m * -1e9).float()      # the float32 sum, as computed
    pref = torch.softmax(logits.double(), dim=-1)
    ref = (pref @ vs).permute(0, 2, 1, 3).reshape(B * Lq, d)
    if first_masked:
        assert abs(pref[0, 0, 0].max().item() - 1.0 / Lk) < 1e-12      # the uniform row is really in the case
    assert (ctx[:, :d].cpu().double() - ref).abs().max().item() < 3e-5
    assert torch.isnan(ctx[:, d:]).all()
    if with_weights:
        assert (attn.cpu().double() - pref).abs().max().item() < 3e-5
    # and the 16-query kernel gives the same numbers
    ctx1 = torch.empty(B * Lq, d, device=DEV)
    ops.attention(q.to(DEV), k.to(DEV), v.to(DEV), ctx1, B, H, Lq, Lk, dh,
                  key_mask=None if mask is None else mask.to(DEV), causal=causal, variant=1)
    assert (ctx1 - ctx[:, :d]).abs().max().item() < 3e-5


@pytest.mark.parametrize('B,H,Lq,Lk,causal,variant', [(2, 4, 96, 96, True, 2), (2, 4, 96, 96, False, 1), (8, 4, 128, 64, False, 0)])
def test_attention_key_mask_not_16_byte_aligned(B, H, Lq, Lk, causal, variant):
    """Lk is a multiple of 4, so whole key tiles read their mask four values at a time -- unless the mask does not start on
    a 16-byte boundary (here: a view one float into a buffer), which must give the same numbers."""
    dh = 64
    d = H * dh
    q, k, v = _rand(B * Lq, d, seed=21), _rand(B * Lk, d, seed=22), _rand(B * Lk, d, seed=23)
    mask = torch.zeros(B, Lk)
    for b in range(B):
        mask[b, Lk - 2 - 5 * b:] = 1.0
    aligned = mask.to(DEV)
    shifted = torch.zeros(B * Lk + 1, device=DEV)[1:].view(B, Lk)
    shifted.copy_(aligned)
    assert shifted.data_ptr() % 16 == 4 and aligned.data_ptr() % 16 == 0
    outs = []
    for m in (aligned, shifted):
        ctx = torch.empty(B * Lq, d, device=DEV)
        attn = torch.empty(B, H, Lq, Lk, device=DEV)
        ops.attention(q.to(DEV), k.to(DEV), v.to(DEV), ctx, B, H, Lq, Lk, dh, key_mask=m, causal=causal, attn=attn, variant=variant)
        outs.append((ctx, attn))
    assert torch.equal(outs[0][0], outs[1][0]) and torch.equal(outs[0][1], outs[1][1])
    split = lambda t, L: t.double().view(B, L, H, dh).permute(0, 2, 1, 3)
    ref, pref = _attention_ref(split(q, Lq), split(k, Lk), split(v, Lk), mask, causal)
    assert (outs[1][0].cpu().double() - ref.permute(0, 2, 1, 3).reshape(B * Lq, d)).abs().max().item() < 3e-5
    assert (outs[1][1].cpu().double() - pref).abs().max().item() < 3e-5


def test_attention_online_softmax_rescale_branch():
    # one key far above the rest, late in the sequence: forces the running-max rescale at a chosen tile
    B, H, L, dh = 1, 1, 96, 32
    q = torch.zeros(L, dh); k = torch.zeros(L, dh); v = _rand(L, dh, seed=5)
    q[:, 0] = 8.0
    k[:, 0] = torch.linspace(-1, 1, L)
    k[70, 0] = 6.0
    ctx = torch.empty(L, dh, device=DEV)
    ops.attention(q.to(DEV), k.to(DEV), v.to(DEV), ctx, B, H, L, L, dh, precision='bf16x3')
    ref, _ = _attention_ref(q.double().view(1, 1, L, dh), k.double().view(1, 1, L, dh), v.double().view(1, 1, L, dh), None, False)
    assert (ctx.cpu().double() - ref[0, 0]).abs().max().item() < 3e-5


@pytest.mark.parametrize('D', [256, 512, 80, 1024])
def test_rownorm_variants(D):
    M = 77
    y, x = _rand(M, D, seed=1), _rand(M, D, seed=2)
    ga, ba, gb, bb = (_rand(D, seed=s) for s in (3, 4, 5, 6))
    dev = lambda t: t.to(DEV)
    F = torch.nn.functional
    out = torch.empty(M, D, device=DEV)
    ops.rownorm(dev(y), out, ln_a=(dev(ga), dev(ba)), act='relu', residual=dev(x), ln_b=(dev(gb), dev(bb)))
    ref = F.layer_norm(torch.relu(F.layer_norm(y.double(), (D,), ga.double(), ba.double(), 1e-6)) + x.double(), (D,),
                       gb.double(), bb.double(), 1e-6)
    assert (out.cpu().double() - ref).abs().max().item() < 2e-5
    ops.rownorm(dev(y), out, ln_a=(dev(ga), dev(ba)), act='relu')
    ref = torch.relu(F.layer_norm(y.double(), (D,), ga.double(), ba.double(), 1e-6))
    assert (out.cpu().double() - ref).abs().max().item() < 2e-5
    # dot output (duration head): relu(out . w + b) * scale * (1 - mask)
    w = _rand(D, seed=7, scale=0.1)
    mask = (torch.arange(M) % 5 == 0).float()
    dot = torch.empty(M, device=DEV)
    ops.rownorm(dev(y), None, ln_a=(dev(ga), dev(ba)), act='relu', residual=dev(x), ln_b=(dev(gb), dev(bb)),
                dot_w=dev(w), dot_b=1.0, dot_scale=0.5, dot_mask=dev(mask), dot_out=dot)
    full = F.layer_norm(torch.relu(F.layer_norm(y.double(), (D,), ga.double(), ba.double(), 1e-6)) + x.double(), (D,),
                        gb.double(), bb.double(), 1e-6)
    ref = torch.relu(full @ w.double() + 1.0) * 0.5 * (1 - mask.double())
    assert (dot.cpu().double() - ref).abs().max().item() < 2e-5


def test_embed_posenc_and_mask():
    B, N, D, V = 3, 19, 64, 30
    tok = torch.randint(1, V, (B, N), generator=torch.Generator().manual_seed(0), dtype=torch.int32)
    tok[1, 12:] = 0
    emb, pe = _rand(V, D, seed=1), _rand(40, D, seed=2)
    out = torch.empty(B * N, 2 * D, device=DEV)
    km = torch.empty(B, N, device=DEV)
    ops.embed_posenc(tok.to(DEV), emb.to(DEV), pe.to(DEV), None, out, km, 8.0)
    ref = emb[tok.long()] * 8.0 + pe[:N][None]
    assert torch.allclose(out[:, :D].cpu().view(B, N, D), ref, atol=1e-6)
    assert torch.equal(km.cpu(), (tok == 0).float())


def test_length_regulator_bit_exact():
    B, N, D = 3, 11, 32
    dur = torch.tensor(np.random.default_rng(0).uniform(0, 4, size=(B, N)).astype(np.float32))
    dur[0, :3] = torch.tensor([0.5, 1.5, 2.5])      # half-to-even: 0, 2, 2
    dur[2, 5:] = 0
    x = _rand(B * N, D, seed=1)
    cum = torch.empty(B, N, dtype=torch.int32, device=DEV)
    tot = torch.empty(B, dtype=torch.int32, device=DEV)
    ops.durations_cumsum(dur.to(DEV), cum, tot)
    reps = np.rint(dur.numpy()).astype(np.int64)
    assert np.array_equal(cum.cpu().numpy(), np.cumsum(reps, axis=1))
    assert np.array_equal(tot.cpu().numpy(), reps.sum(axis=1))
    T = int(reps.sum(axis=1).max())
    out = torch.empty(B * T, D, device=DEV)
    fm = torch.empty(B, T, device=DEV)
    ops.expand(x.to(DEV), cum, out, fm, B, N, T, D)
    ref = np.zeros((B, T, D), np.float32)
    for b in range(B):
        rows = np.repeat(np.arange(N), reps[b])
        ref[b, :len(rows)] = x.view(B, N, D)[b].numpy()[rows]
    assert np.array_equal(out.cpu().numpy().reshape(B, T, D), ref)
    assert np.array_equal(fm.cpu().numpy(), (np.arange(T)[None] >= reps.sum(axis=1)[:, None]).astype(np.float32))


def test_gemm_batched_products_in_one_launch():
    """batch = 2: two products of one shape (different inputs, weights, biases, outputs) from a single launch, as the two
    recurrent products of a WaveRNN step are issued."""
    rng = np.random.default_rng(5)
    M, N, K = 200, 192, 96
    x = rng.normal(size=(2, M, K)).astype(np.float32)
    w = rng.normal(size=(2 * N, K)).astype(np.float32) * 0.2
    b = rng.normal(size=2 * N).astype(np.float32)
    lin = PackedLinear(w, bias=b, device=DEV)
    lin.N = N
    xd = torch.from_numpy(x).to(DEV)
    out = torch.full((2, M, N), float('nan'), device=DEV)
    ops.gemm(xd[0], lin, out[0], split=False, batch=(2, M * K, N * lin.ldw, M * N, N))
    for z in range(2):
        ref = x[z].astype(np.float64) @ w[z * N:(z + 1) * N].astype(np.float64).T + b[z * N:(z + 1) * N]
        assert np.abs(out[z].cpu().numpy() - ref).max() < 3e-5 * np.abs(ref).max()
    with pytest.raises(Exception):
        ops.gemm(xd[0], lin, out[0], split=False, tune=4, batch=(2, M * K, N * lin.ldw, M * N, N))


@pytest.mark.parametrize('Lk,mode', [(100, 'sa'), (100, 'plain'), (37, 'ffn'), (256, 'single'), (1, 'sa')])
def test_cross_attention_row_is_projection_attention_projection(Lk, mode):
    # estts_cross_attention_row: one decoder position through q projection, attention over the encoder's keys / values and the
    # output projection over [x | ctx] in one launch (layers.py:258-274 for a single query row), against the three launches it
    # replaces -- gemm(ln=...) + attention + gemm -- with the pending normalisations an incremental decoder hands over (the
    # self-attention block's LN(LN(y) + x), a closing FFN norm with ReLU, a single norm, none), a key mask, and several rows
    d, H = 256, 4
    for M in (1, 3):
        g = 61000 + Lk + 7 * M
        wq = PackedLinear(_rand(d, d, seed=g, scale=1 / 16).numpy(), bias=_rand(d, seed=g + 1).numpy(), device=DEV)
        wo = PackedLinear(_rand(d, 2 * d, seed=g + 2, scale=1 / 23).numpy(), bias=_rand(d, seed=g + 3).numpy(), device=DEV)
        kv = _rand(Lk, 2 * d, seed=g + 4).to(DEV)
        mask = (torch.rand(M, Lk, generator=torch.Generator().manual_seed(g + 5)) < 0.2).float().to(DEV) if M == 1 else None
        if mask is not None and Lk > 1:
            mask[0, 0] = 0.0
        yin, xin = _rand(M, d, seed=g + 6).to(DEV), _rand(M, d, seed=g + 7).to(DEV)
        vec = lambda s: ((_rand(d, seed=s).abs() + 0.5).to(DEV), _rand(d, seed=s + 1).to(DEV))
        ln = {'sa': dict(ln_a=vec(g + 8), ln_b=vec(g + 10), x=xin), 'ffn': dict(ln_a=vec(g + 8), act='relu', ln_b=vec(g + 10), x=xin),
              'single': dict(ln_b=vec(g + 10), x=xin), 'plain': None}[mode]
        assert ops.can_cross_attention_row(wq, wo, d, H, Lk)
        # the three launches
        xc = torch.zeros(M, 2 * d, device=DEV)
        q, y_ref, attn_ref = torch.empty(M, d, device=DEV), torch.empty(M, d, device=DEV), torch.empty(M, H, 1, Lk, device=DEV)
        if ln is None:
            xc[:, :d] = yin
            ops.gemm(xc[:, :d], wq, q)
        else:
            ops.rownorm(yin, xc[:, :d], ln_a=ln.get('ln_a'), act=ln.get('act'), residual=ln['x'], ln_b=ln.get('ln_b'))
            ops.gemm(xc[:, :d], wq, q)
        kk, vv = (kv[:, :d], kv[:, d:])
        for m in range(M):      # the keys / values are shared by the rows: one launch per row of the reference form
            ops.attention(q[m:m + 1], kk, vv, xc[m:m + 1, d:], 1, H, 1, Lk, d // H, key_mask=None if mask is None else mask[m:m + 1],
                          attn=attn_ref[m:m + 1])
        ops.gemm(xc, wo, y_ref)
        # one launch
        y = torch.full((M + 1, d + 4), float('nan'), device=DEV)
        attn = torch.full((M, H * Lk + 3), float('nan'), device=DEV)
        lo = torch.full((M, d), float('nan'), device=DEV)
        if ln is None:
            ops.cross_attention_row(wq, wo, kk, vv, y[:M, :d], H, x=yin, key_mask=mask, attn=attn)
        else:
            ops.cross_attention_row(wq, wo, kk, vv, y[:M, :d], H, ln=dict(y=yin, out=lo, **ln), key_mask=mask, attn=attn)
            assert (lo - xc[:, :d]).abs().max().item() < 2e-6 * xc[:, :d].abs().max().item()
        assert torch.isnan(y[M:]).all() and torch.isnan(y[:, d:]).all() and torch.isnan(attn[:, H * Lk:]).all()
        assert torch.isfinite(y_ref).all()
        err = (y[:M, :d] - y_ref).abs().max().item() / y_ref.abs().max().item()
        aerr = (attn[:, :H * Lk].view(M, H, 1, Lk) - attn_ref).abs().max().item()
        assert err < 2e-5 and aerr < 2e-5, (Lk, mode, M, err, aerr)


@pytest.mark.parametrize('Lk', [1, 2, 60, 64, 65, 128, 129, 256])
def test_cross_attention_row_with_given_queries_is_attention_then_projection(Lk):
    # the same entry point with the query rows given (a self-attention block of an incremental decoder, layers.py:217-233, after
    # the product that wrote the position's q | k | v row): attention over the cache + output projection over [x | ctx], against
    # the two launches it replaces; the key counts straddle the kernel's register variants (64 / 128 / 256 keys)
    d, H = 256, 4
    g = 62000 + Lk
    wo = PackedLinear(_rand(d, 2 * d, seed=g + 2, scale=1 / 23).numpy(), bias=_rand(d, seed=g + 3).numpy(), device=DEV)
    qkv = _rand(Lk, 3 * d, seed=g + 4).to(DEV)
    mask = (torch.rand(1, Lk, generator=torch.Generator().manual_seed(g + 5)) < 0.2).float().to(DEV)
    mask[0, 0] = 0.0
    assert ops.can_cross_attention_row(None, wo, d, H, Lk) and not ops.can_cross_attention_row(None, wo, d, H, 257)
    xc = torch.zeros(1, 2 * d, device=DEV)
    xc[:, :d] = _rand(1, d, seed=g + 6).to(DEV)
    row = qkv[Lk - 1:Lk]
    y_ref = torch.empty(1, d, device=DEV)
    ops.attention(row[:, :d], qkv[:, d:2 * d], qkv[:, 2 * d:], xc[:, d:], 1, H, 1, Lk, d // H, key_mask=mask)
    ops.gemm(xc, wo, y_ref)
    for _ in range(2):      # (twice: the row's arrival counter must be back at zero)
        y = torch.full((2, d), float('nan'), device=DEV)
        ops.cross_attention_row(None, wo, qkv[:, d:2 * d], qkv[:, 2 * d:], y[:1], H, x=xc[:, :d], q=row[:, :d], key_mask=mask)
        assert torch.isnan(y[1:]).all()
        err = (y[:1] - y_ref).abs().max().item() / y_ref.abs().max().item()
        assert err < 2e-5, (Lk, err)
    with pytest.raises(Exception):
        ops.cross_attention_row(None, wo, qkv[:, d + 1:2 * d + 1], qkv[:, 2 * d:], y[:1], H, x=xc[:, :d], q=row[:, :d])   # misaligned keys


@pytest.mark.parametrize('Lk,positions,pending', [(100, (0, 1, 2, 7), 'ffn'), (100, (62, 63, 64, 65), None), (37, (127, 128, 129), 'single'),
                                                  (200, (0, 3), 'ffn'), (256, (254, 255), None)])
def test_decoder_block_row_is_the_block_of_the_reference_for_one_position(Lk, positions, pending):
    # estts_decoder_block_row: self-attention over the cache, cross-attention over the encoder and the folded feed-forward product
    # of a decoder block for one position in one launch (layers.py:277-297 with 217-233, 258-274, 104-124), against the same
    # block written out in float64 from the unpacked weights.  The positions straddle the kernel's register variants (64 / 128
    # cache rows, 128 / 256 encoder rows); successive launches on one cache also exercise the counters that the four workgroups
    # of a row leave for the next launch.
    d, H = 256, 4
    g = 63000 + Lk
    W = lambda n, k, s: _rand(n, k, seed=s, scale=1 / np.sqrt(k))
    w = dict(qkv=W(3 * d, d, g), os=W(d, 2 * d, g + 1), qc=W(d, d, g + 2), oc=W(d, 2 * d, g + 3), ff=W(d, d, g + 4))
    bias = {n: _rand(w[n].shape[0], seed=g + 10 + i) * 0.3 for i, n in enumerate(w)}
    lin = {n: PackedLinear(w[n].numpy(), bias=bias[n].numpy(), device=DEV) for n in w}
    vec = lambda s: (_rand(d, seed=s).abs() + 0.5, _rand(d, seed=s + 1) * 0.3)
    lns = dict(sa=vec(g + 20), sa_last=vec(g + 22), ca=vec(g + 24), pa=vec(g + 26), pb=vec(g + 28))
    dev = lambda t: tuple(x.to(DEV) for x in t)
    S = max(positions) + 1
    cache = _rand(S, 3 * d, seed=g + 30)
    enc = _rand(Lk, 2 * d, seed=g + 31)
    smask = (torch.rand(1, S, generator=torch.Generator().manual_seed(g + 32)) < 0.15).float()
    emask = (torch.rand(1, Lk, generator=torch.Generator().manual_seed(g + 33)) < 0.2).float()
    emask[0, 0] = 0.0
    assert ops.can_decoder_block_row(lin['qkv'], lin['os'], lin['qc'], lin['oc'], lin['ff'], d, H, S, Lk)
    cache_d, enc_d = cache.to(DEV).clone(), enc.to(DEV)
    f64 = lambda t: t.double()

    def LN(v, gb):
        m = v.mean(-1, keepdim=True)
        return (v - m) / torch.sqrt(((v - m) ** 2).mean(-1, keepdim=True) + 1e-6) * f64(gb[0]) + f64(gb[1])

    def attend(q, K, V, mask):
        out, ws = [], []
        for h in range(H):
            sl = slice(64 * h, 64 * h + 64)
            sc = (K[:, sl] @ q[0, sl]) / 8.0 + f64(mask[0, :K.shape[0]]) * -1e9
            p = torch.softmax(sc, 0)
            ws.append(p)
            out.append(p @ V[:, sl])
        return torch.cat(out)[None], torch.stack(ws)

    for pos in positions:
        yin, xin = _rand(1, d, seed=g + 40 + pos), _rand(1, d, seed=g + 41 + pos)
        # float64 block
        if pending == 'ffn':
            x = LN(torch.relu(LN(f64(yin), lns['pa'])) + f64(xin), lns['pb'])
            ln = dict(y=yin.to(DEV), x=xin.to(DEV), ln_a=dev(lns['pa']), act='relu', ln_b=dev(lns['pb']))
        elif pending == 'single':
            x = LN(f64(yin) + f64(xin), lns['pb'])
            ln = dict(y=yin.to(DEV), x=xin.to(DEV), ln_b=dev(lns['pb']))
        else:
            x, ln = f64(yin), None
        row = x @ f64(w['qkv']).T + f64(bias['qkv'])
        cache[pos] = row[0].float()
        C = f64(cache[:pos + 1])
        C[pos] = row[0]
        ctx, _ = attend(row[:, :d], C[:, d:2 * d], C[:, 2 * d:], smask)
        ys = torch.cat([x, ctx], 1) @ f64(w['os']).T + f64(bias['os'])
        x1 = LN(LN(ys, lns['sa']) + x, lns['sa_last'])
        qc = x1 @ f64(w['qc']).T + f64(bias['qc'])
        ctx2, aw = attend(qc, f64(enc[:, :d]), f64(enc[:, d:]), emask)
        yc = torch.cat([x1, ctx2], 1) @ f64(w['oc']).T + f64(bias['oc'])
        x2_ref = LN(yc + x1, lns['ca'])
        y2_ref = x2_ref @ f64(w['ff']).T + f64(bias['ff'])
        # one launch
        y2 = torch.full((2, d), float('nan'), device=DEV)
        x2 = torch.full((2, d), float('nan'), device=DEV)
        attn = torch.full((1, H * Lk + 4), float('nan'), device=DEV)
        ops.decoder_block_row(lin['qkv'], lin['os'], dev(lns['sa']), dev(lns['sa_last']), lin['qc'], lin['oc'], dev(lns['ca']), lin['ff'],
                              cache_d, pos, enc_d[:, :d], enc_d[:, d:], y2[:1], x2[:1], x=None if ln is not None else yin.to(DEV), ln=ln,
                              self_mask=smask.to(DEV), key_mask=emask.to(DEV), attn=attn[:, :H * Lk])
        assert torch.isnan(y2[1:]).all() and torch.isnan(x2[1:]).all() and torch.isnan(attn[:, H * Lk:]).all()
        rel = lambda a, b: (a.double().cpu() - b).abs().max().item() / b.abs().max().item()
        assert rel(cache_d[pos:pos + 1], row) < 2e-5, (pos, 'cache row')
        assert (attn[0, :H * Lk].view(H, Lk).double().cpu() - aw).abs().max().item() < 2e-5, (pos, 'weights')
        assert rel(x2[:1], x2_ref) < 5e-5, (pos, 'x2', rel(x2[:1], x2_ref))
        assert rel(y2[:1], y2_ref) < 5e-5, (pos, 'y2', rel(y2[:1], y2_ref))
        cache_d[pos] = cache[pos].to(DEV)      # both sides go on from the same cache
    assert not torch.equal(cache_d[:1], torch.zeros_like(cache_d[:1]))


@pytest.mark.parametrize('C,rate', [(80, 0.0), (80, 0.5), (128, 0.3), (4, 0.0)])
def test_prenet_row_is_mask_and_two_dense_layers(C, rate):
    # estts_prenet_row: the decoder pre-net of one position (layers.py:394-402: Dense + ReLU + Dropout twice, dropout on at
    # inference too) with the padding mask of the frame (transformer_utils.py:47-50), scaled and added to a positional row, in
    # one launch, against estts_row_zero_mask + two estts_gemm launches with the same dropout generator arguments
    d = 256
    g = 64000 + C
    l1 = PackedLinear(_rand(d, C, seed=g, scale=1 / np.sqrt(C)).numpy(), bias=_rand(d, seed=g + 1).numpy() * 0.2, device=DEV, act='relu')
    l2 = PackedLinear(_rand(d, d, seed=g + 2, scale=1 / 16).numpy(), bias=_rand(d, seed=g + 3).numpy() * 0.2, device=DEV, act='relu')
    assert ops.can_prenet_row(l1, l2, C) and not ops.can_prenet_row(l1, l2, C + 4)
    frames = _rand(5, C, seed=g + 4).to(DEV)
    frames[2] = 0.0
    pe = _rand(5, d, seed=g + 5).to(DEV)
    for i in range(5):
        x = frames[i:i + 1]
        mask_ref, p1, ref = torch.full((1,), -1.0, device=DEV), torch.empty(1, d, device=DEV), torch.empty(1, d, device=DEV)
        ops.row_zero_mask(x, mask_ref)
        ops.gemm(x, l1, p1, dropout=(rate, 1234, 0, i))
        ops.gemm(p1, l2, ref, alpha=16.0, residual=pe[i:i + 1], dropout=(rate, 1234, 1, i))
        mask, out = torch.full((2,), -1.0, device=DEV), torch.full((2, d + 4), float('nan'), device=DEV)
        ops.prenet_row(x, mask[:1], l1, l2, out[:1, :d], alpha=16.0, residual=pe[i:i + 1], dropout=(rate, 1234, 0, 1, i))
        assert mask[0].item() == mask_ref[0].item() == (1.0 if i == 2 else 0.0) and mask[1].item() == -1.0
        assert torch.isnan(out[1:]).all() and torch.isnan(out[:, d:]).all()
        err = (out[:1, :d] - ref).abs().max().item() / ref.abs().max().item()
        assert err < 3e-5, (C, rate, i, err)
        if rate > 0:      # the same units were dropped
            assert torch.equal(out[:1, :d] == pe[i:i + 1], ref == pe[i:i + 1])


def test_decoder_block_row_takes_several_utterances_at_once():
    # M rows = one position of M utterances, each with its own cache, encoder keys / values and masks: the same numbers as M
    # launches of one row (the workgroups of a row only ever meet each other), also when launched twice on the same counters
    d, H, Lk, S, M, pos = 256, 4, 70, 9, 3, 5
    g = 65000
    W = lambda n, k, s: _rand(n, k, seed=s, scale=1 / np.sqrt(k))
    lin = {n: PackedLinear(W(*sh, g + i).numpy(), bias=_rand(sh[0], seed=g + 10 + i).numpy() * 0.3, device=DEV)
           for i, (n, sh) in enumerate(dict(qkv=(3 * d, d), os=(d, 2 * d), qc=(d, d), oc=(d, 2 * d), ff=(d, d)).items())}
    vec = lambda s: ((_rand(d, seed=s).abs() + 0.5).to(DEV), (_rand(d, seed=s + 1) * 0.3).to(DEV))
    l1, l2, l3 = vec(g + 20), vec(g + 22), vec(g + 24)
    cache0 = _rand(M, S, 3 * d, seed=g + 30).to(DEV)
    enc = _rand(M, Lk, 2 * d, seed=g + 31).to(DEV)
    smask = (torch.rand(M, S, generator=torch.Generator().manual_seed(g + 32)) < 0.2).float().to(DEV)
    emask = (torch.rand(M, Lk, generator=torch.Generator().manual_seed(g + 33)) < 0.2).float().to(DEV)
    emask[:, 0] = 0.0
    x = _rand(M, d, seed=g + 34).to(DEV)
    run = lambda cache, k, v, y2, x2, xx, sm, em, at: ops.decoder_block_row(
        lin['qkv'], lin['os'], l1, l2, lin['qc'], lin['oc'], l3, lin['ff'], cache, pos, k, v, y2, x2, x=xx, self_mask=sm, key_mask=em, attn=at)
    ref = []
    for m in range(M):
        c, y2, x2, at = cache0[m].clone(), torch.empty(1, d, device=DEV), torch.empty(1, d, device=DEV), torch.empty(1, H * Lk, device=DEV)
        run(c, enc[m, :, :d], enc[m, :, d:], y2, x2, x[m:m + 1], smask[m:m + 1], emask[m:m + 1], at)
        ref.append((c, y2, x2, at))
    for _ in range(2):
        c, y2, x2, at = cache0.clone(), torch.empty(M, d, device=DEV), torch.empty(M, d, device=DEV), torch.empty(M, H * Lk, device=DEV)
        run(c, enc[:, :, :d], enc[:, :, d:], y2, x2, x, smask, emask, at)
        for m in range(M):
            assert torch.equal(c[m], ref[m][0]) and torch.equal(y2[m:m + 1], ref[m][1]) and torch.equal(x2[m:m + 1], ref[m][2])
            assert torch.equal(at[m:m + 1], ref[m][3])
