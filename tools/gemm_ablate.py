#!/usr/bin/env python3
"""Ablation of the large-M product: which part of a K-step the time follows.  Needs a DIAGNOSTIC build of gemm_big_kernel in
which tune bit 26 skips the W stores to LDS, bit 27 the A stores and bit 28 the matrix instructions (fragments kept live
with empty asm statements); the shipped kernel ignores those bits, so on it every column prints the same time.  Measured
with such a build (MI355X, 19200 x 256 x 512, 64 x 128 tiles): full 34.0 us, no W stores 32.2, no A stores 32.1, no stores
30.3, no MFMA 26.6, neither 22.3 -- i.e. two thirds of the launch is operand delivery (global loads, LDS reads, epilogue),
not the LDS store path and not the matrix pipe."""
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
    g = torch.Generator().manual_seed(0)
    for M, N, K, taps in ((19200, 256, 512, 1), (19200, 512, 768, 3), (49152, 256, 512, 1)):
        w = torch.randn(N, K, generator=g) / math.sqrt(K)
        lin = PackedLinear(w.numpy(), device='cuda', taps=taps, lpad=(taps - 1) // 2)
        x = torch.randn(M, K // taps, generator=g).cuda()
        out = torch.empty(M, N, device='cuda')
        for tile in (8, 6):
            row = []
            for name, bits in (('full', 0), ('noWst', 1 << 26), ('noAst', 1 << 27), ('noSt', 3 << 26), ('noMFMA', 1 << 28),
                               ('noSt+noMFMA', 7 << 26)):
                t = timeit(lambda: ops.gemm(x, lin, out, seg_len=M, precision='bf16x3', tune=tile | (1 << 8) | bits), reps=10, rounds=3)
                row.append(f'{name} {t:6.1f}')
            print(f'M={M} N={N} K={K} taps={taps} tile{tile}: ' + ' | '.join(row), flush=True)


if __name__ == '__main__':
    main()
