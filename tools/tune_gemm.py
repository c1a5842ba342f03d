#!/usr/bin/env python3
"""Sweep estts_gemm tile / split-K choices over the ForwardTransformer's GEMM shapes (GPU box only)."""
import itertools
import math
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import estts  # noqa
from estts import ops
from estts.ops import PackedLinear

SHAPES = [  # (M, N, K, taps)
    (32, 512, 1536, 1), (32, 1536, 512, 1), (1, 512, 1536, 1), (1, 1536, 512, 1),
    (100, 1536, 512, 1), (100, 512, 1024, 1), (100, 512, 512, 1), (100, 512, 1536, 3),
    (600, 1024, 512, 1), (600, 256, 1024, 1), (600, 768, 256, 1), (600, 256, 512, 1), (600, 512, 768, 3),
    (600, 256, 1536, 3), (600, 80, 256, 1), (600, 256, 400, 5), (600, 256, 1280, 5), (600, 80, 1280, 5),
]


def timeit(fn, reps=50, rounds=5):
    """Device time per launch: `reps` launches captured in one hipGraph (no host launch cost), best of `rounds`."""
    fn()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(reps):
            fn()
    g.replay()
    best = 1e9
    for _ in range(rounds):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        g.replay()
        e1.record()
        e1.synchronize()
        best = min(best, e0.elapsed_time(e1) * 1e3 / reps)
    return best


def main():
    prec = sys.argv[1] if len(sys.argv) > 1 else 'bf16x3'
    g = torch.Generator().manual_seed(0)
    for M, N, K, taps in SHAPES:
        cin = K // taps
        x = torch.randn(M, cin, generator=g).cuda()
        w = torch.randn(N, K, generator=g) / math.sqrt(K)
        lin = PackedLinear(w.numpy(), device='cuda', taps=taps, lpad=(taps - 1) // 2)
        out = torch.empty(M, N, device='cuda')
        ref = None
        res = []
        cands = [(t, sk, 0) for t, sk in itertools.product((1, 3), (1, 4))] + \
                [(4, sk, wv) for sk, wv in itertools.product((1, 2, 3, 4, 6), (4, 8))] + [(6, 1, 0), (7, 1, 0), (8, 1, 0)] + [(t, 1, wv) for t in (9, 10) for wv in (4, 8)] + [(11, 1, 0), (12, 1, 0)]
        for tile, sk, wv in cands:
            tune = tile | (sk << 8) | (wv << 16)
            try:
                ops.gemm(x, lin, out, seg_len=M, precision=prec, tune=tune)
            except Exception as e:  # noqa
                if os.environ.get('TUNE_VERBOSE'):
                    print('   skipped', tile, sk, wv, str(e)[:100])
                continue
            if ref is None:
                ref = out.clone()
            err = (out - ref).abs().max().item()
            t = timeit(lambda: ops.gemm(x, lin, out, seg_len=M, precision=prec, tune=tune))
            res.append((t, tile * 100 + wv, sk, err))
        ops.gemm(x, lin, out, seg_len=M, precision=prec)
        t_auto = timeit(lambda: ops.gemm(x, lin, out, seg_len=M, precision=prec))
        res.sort()
        best = ' '.join(f'[t{tile} k{sk}: {t:5.1f}us e{err:.0e}]' for t, tile, sk, err in res[:(99 if os.environ.get('TUNE_VERBOSE') else 6)])
        print(f'M={M:4d} N={N:4d} K={K:4d} taps={taps} auto {t_auto:5.1f}us  {2e-6 * M * N * K / t_auto:6.1f} TF/s | best {best}', flush=True)


if __name__ == '__main__':
    main()
