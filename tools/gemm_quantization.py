#!/usr/bin/env python3
"""How much of the large-M product's time is workgroup quantization?  Times estts_gemm (tiles 6 and 8) at row counts that
give whole and fractional numbers of workgroups per CU, and prints microseconds per 1024 rows (GPU box only)."""
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
    for N, K, taps in ((256, 512, 1), (512, 768, 3), (256, 1536, 3), (768, 256, 1)):
        w = torch.randn(N, K, generator=g) / math.sqrt(K)
        lin = PackedLinear(w.numpy(), device='cuda', taps=taps, lpad=(taps - 1) // 2)
        for tile in (6, 8):
            row = []
            for M in (8192, 12288, 16384, 19200, 20480, 24576, 32768, 49152):
                x = torch.randn(M, K // taps, generator=g).cuda()
                out = torch.empty(M, N, device='cuda')
                t = timeit(lambda: ops.gemm(x, lin, out, seg_len=M, precision='bf16x3', tune=tile | (1 << 8)), reps=10, rounds=3)
                wgs = (M // 64) * (N // (128 if tile == 8 else 64))
                row.append(f'M={M} ({wgs / 256:.2f} wg/CU) {t:6.1f} us {t / M * 1024:5.2f}/Krow')
            print(f'N={N} K={K} taps={taps} tile{tile}: ' + ' | '.join(row), flush=True)


if __name__ == '__main__':
    main()
