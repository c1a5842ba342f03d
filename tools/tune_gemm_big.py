#!/usr/bin/env python3
"""Compare the estts_gemm kernels (tile ids 1, 4, 5) on the large-batch shapes of configs 3 and 5 (GPU box only)."""
import math
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import estts  # noqa
from estts import ops
from estts.ops import PackedLinear
from tune_gemm import timeit

SHAPES = [  # (M, N, K, taps)
    (19200, 256, 1536, 3), (19200, 512, 768, 3), (19200, 256, 1280, 5), (19200, 768, 256, 1), (19200, 256, 512, 1),
    (19200, 256, 256, 1), (19200, 1024, 512, 1), (19200, 80, 1280, 5), (19200, 80, 256, 1), (19200, 256, 80, 1),
    (3200, 512, 1536, 3), (3200, 1536, 512, 1), (3200, 512, 1024, 1), (3200, 512, 512, 1), (3200, 768, 256, 1),
    (96000, 32, 288, 9), (4800, 256, 512, 1), (1200, 256, 512, 1), (600, 256, 1536, 3), (600, 512, 768, 3),
]


def main():
    prec = sys.argv[1] if len(sys.argv) > 1 else 'bf16x3'
    g = torch.Generator().manual_seed(0)
    for M, N, K, taps in SHAPES:
        cin = K // taps
        x = torch.randn(M, cin, generator=g).cuda()
        w = torch.randn(N, K, generator=g) / math.sqrt(K)
        lin = PackedLinear(w.numpy(), device='cuda', taps=taps, lpad=(taps - 1) // 2)
        out = torch.empty(M, N, device='cuda')
        res = []
        ref = None
        for tile in (4, 5, 6, 7, 8, 10 | (4 << 16), 10 | (8 << 16), 12):
            try:
                ops.gemm(x, lin, out, seg_len=M, precision=prec, tune=tile | (1 << 8))
            except RuntimeError:
                continue
            if ref is None:
                ref = out.clone()
            err = (out - ref).abs().max().item()
            t = timeit(lambda: ops.gemm(x, lin, out, seg_len=M, precision=prec, tune=tile | (1 << 8)), reps=10, rounds=3)
            res.append(f'tile{tile & 0xff}{"w%d" % (tile >> 16) if tile >> 16 else ""} {t:8.1f} us {2.0 * M * N * K / t / 1e6:7.1f} TF/s (d {err:.1e})')
        print(f'M={M:6d} N={N:5d} K={K:5d} taps={taps} | ' + ' | '.join(res), flush=True)


if __name__ == '__main__':
    main()
