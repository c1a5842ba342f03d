#!/usr/bin/env python3
"""Tile / split-K sweep of estts_gemm over the WaveRNN step's shapes (M = parallel sequences); GPU box only."""
import itertools
import math
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import estts  # noqa
from estts import ops
from estts.ops import PackedLinear
from tune_gemm import timeit


def main():
    M = int(sys.argv[1]) if len(sys.argv) > 1 else 832
    prec = sys.argv[2] if len(sys.argv) > 2 else 'bf16x3'
    flags = int(sys.argv[3]) << 24 if len(sys.argv) > 3 else 0
    g = torch.Generator().manual_seed(0)
    for N, K in ((1536, 512), (1536, 544), (512, 544), (512, 512), (512, 116)):
        x = torch.randn(M, K, generator=g).cuda()
        w = torch.randn(N, K, generator=g) / math.sqrt(K)
        lin = PackedLinear(w.numpy(), device='cuda')
        out = torch.empty(M, N, device='cuda')
        res = []
        cands = [(t, sk, 0) for t, sk in itertools.product((1, 2, 3), (1, 2, 4))] + \
                [(4, sk, wv) for sk, wv in itertools.product((1, 2), (4, 8))] + [(5, 1, 0), (6, 1, 0), (7, 1, 0), (8, 1, 0)]
        for tile, sk, wv in cands:
            tune = tile | (sk << 8) | (wv << 16) | flags
            try:
                ops.gemm(x, lin, out, precision=prec, tune=tune)
            except Exception:  # noqa
                continue
            t = timeit(lambda: ops.gemm(x, lin, out, precision=prec, tune=tune))
            res.append((t, tile, sk, wv))
        t_auto = timeit(lambda: ops.gemm(x, lin, out, precision=prec))
        res.sort()
        best = ' '.join(f'[t{tile} k{sk} w{wv}: {t:5.1f}us]' for t, tile, sk, wv in res[:6])
        print(f'M={M} N={N:4d} K={K:4d} auto {t_auto:5.1f}us {2e-6 * M * N * K / t_auto:6.1f} TF/s | {best}', flush=True)


if __name__ == '__main__':
    main()
