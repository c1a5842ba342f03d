"""estts_gemm_noalloc (split-K scratch in a static device array inside the library) against estts_gemm with a
caller-provided workspace, through the C ABI.  Both run the same kernels with the same split, and the last arriver
adds the slabs in a fixed order, so the outputs must be identical bit for bit.  Two calls in a row share the static
array; the second is smaller, so stale slabs or tickets of the first would show."""
import math

import pytest
import torch

import estts
from estts import _lib, ops
from estts.ops import PackedLinear

pytestmark = pytest.mark.gpu
DEV = 'cuda'


def _case(M, N, K, taps, seed):
    g = torch.Generator().manual_seed(seed)
    x = torch.randn(M, K // taps, generator=g)
    w = torch.randn(N, K, generator=g) / math.sqrt(K)
    b = torch.randn(N, generator=g)
    lin = PackedLinear(w.numpy(), bias=b.numpy(), device=DEV, taps=taps, lpad=(taps - 1) // 2, act='relu')
    xp = torch.nn.functional.pad(x.double(), (0, 0, (taps - 1) // 2, taps - 1 - (taps - 1) // 2))
    c = K // taps
    ref = torch.relu(sum(xp[j:j + M] @ w.double()[:, j * c:(j + 1) * c].t() for j in range(taps)) + b.double())
    return x.to(DEV), lin, ref


def _args(x, lin, out, tune):
    a = _lib.GemmArgs()
    a.A, a.lda, a.seg_len = x.data_ptr(), x.stride(0), x.shape[0]
    a.taps, a.c_per_tap, a.tap_lpad = lin.taps, lin.c_per_tap, lin.lpad
    a.W_hi, a.W_lo, a.ldw = lin.w_hi.data_ptr(), lin.w_lo.data_ptr(), lin.ldw
    a.M, a.N, a.K = x.shape[0], lin.N, lin.K
    a.b1 = lin.b1.data_ptr()
    a.act, a.alpha = lin.act, 1.0
    a.C, a.ldc = out.data_ptr(), out.stride(0)
    a.precision, a.tune = 3, tune
    return a


def _run_pair(M, N, K, taps, tune, seed):
    lib = _lib.load()
    x, lin, ref = _case(M, N, K, taps, seed)
    st = torch.cuda.current_stream().cuda_stream
    out_ws = torch.full((M, N), float('nan'), device=DEV)
    a = _args(x, lin, out_ws, tune)
    ws, cnt = ops.new_gemm_workspace(DEV, 4 << 20)
    a.ws, a.ws_bytes, a.counters, a.n_counters = ws.data_ptr(), ws.numel() * 4, cnt.data_ptr(), cnt.numel()
    _lib.check(lib.estts_gemm(a, st), 'estts_gemm')
    out_na = torch.full((M, N), float('nan'), device=DEV)
    b = _args(x, lin, out_na, tune)          # no workspace fields at all
    _lib.check(lib.estts_gemm_noalloc(b, st), 'estts_gemm_noalloc')
    torch.cuda.synchronize()
    assert torch.equal(out_na, out_ws), (out_na - out_ws).abs().max().item()
    assert (out_na.cpu().double() - ref).abs().max().item() < 5e-5


def test_noalloc_matches_workspace_entry_twice():
    # 32 x 32 tiles, K split 8 ways; then a smaller problem split 4 ways on the same static array
    _run_pair(200, 192, 1152, 3, 4 | (8 << 8), seed=1)
    _run_pair(37, 80, 512, 1, 4 | (4 << 8), seed=2)
    # the automatic choice (few tiles -> split) and an LDS-staged tile with split-K
    _run_pair(24, 64, 2048, 1, 0, seed=3)
    _run_pair(100, 128, 768, 3, 1 | (2 << 8), seed=4)


def test_noalloc_rejects_oversize_split():
    lib = _lib.load()
    x, lin, _ = _case(2048, 2048, 512, 1, seed=5)
    out = torch.empty(2048, 2048, device=DEV)
    a = _args(x, lin, out, 4 | (8 << 8))     # 4096 tiles x 8 slabs x 4 KiB = 128 MiB of slabs
    rc = lib.estts_gemm_noalloc(a, torch.cuda.current_stream().cuda_stream)
    msg = lib.estts_last_error().decode()
    assert rc != 0 and 'estts_gemm_noalloc' in msg and 'scratch bytes' in msg, msg
    torch.cuda.synchronize()


def test_noalloc_allocates_nothing_and_runs_under_stream_capture():
    """What the name promises: no device memory is allocated by the calls (free memory and torch's reservation unchanged),
    and a call can be recorded into a hipGraph once the scratch address has been looked up (the first call on a device
    does that, outside any capture); the replayed graph gives the same bits as the workspace entry."""
    lib = _lib.load()
    x, lin, _ = _case(24, 64, 2048, 1, seed=7)              # few tiles: the automatic choice splits K
    x2, lin2, _ = _case(300, 256, 512, 1, seed=8)           # many tiles: no split, so no ticket clear either
    outs = [torch.full((24, 64), float('nan'), device=DEV), torch.full((300, 256), float('nan'), device=DEV)]
    refs = [torch.empty_like(o) for o in outs]
    ws, cnt = ops.new_gemm_workspace(DEV, 4 << 20)
    st = torch.cuda.current_stream().cuda_stream
    for (xx, ll), ref in zip(((x, lin), (x2, lin2)), refs):
        a = _args(xx, ll, ref, 0)
        a.ws, a.ws_bytes, a.counters, a.n_counters = ws.data_ptr(), ws.numel() * 4, cnt.data_ptr(), cnt.numel()
        _lib.check(lib.estts_gemm(a, st), 'estts_gemm')
    _lib.check(lib.estts_gemm_noalloc(_args(x, lin, outs[0], 0), st), 'estts_gemm_noalloc')     # address looked up here
    torch.cuda.synchronize()
    outs[0].fill_(float('nan'))
    torch.cuda.synchronize()
    free0, reserved0 = torch.cuda.mem_get_info()[0], torch.cuda.memory_reserved()
    n0 = lib.estts_launch_count()
    for (xx, ll), out in zip(((x, lin), (x2, lin2)), outs):
        _lib.check(lib.estts_gemm_noalloc(_args(xx, ll, out, 0), st), 'estts_gemm_noalloc')
    torch.cuda.synchronize()
    assert lib.estts_launch_count() - n0 == 2
    assert torch.cuda.mem_get_info()[0] == free0 and torch.cuda.memory_reserved() == reserved0
    assert torch.equal(outs[0], refs[0]) and torch.equal(outs[1], refs[1])
    for o in outs:
        o.fill_(float('nan'))
    g = torch.cuda.CUDAGraph()
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.graph(g, stream=s):
        for (xx, ll), out in zip(((x, lin), (x2, lin2)), outs):
            _lib.check(lib.estts_gemm_noalloc(_args(xx, ll, out, 0), torch.cuda.current_stream().cuda_stream), 'estts_gemm_noalloc')
    g.replay()
    g.replay()
    torch.cuda.synchronize()
    assert torch.equal(outs[0], refs[0]) and torch.equal(outs[1], refs[1])
