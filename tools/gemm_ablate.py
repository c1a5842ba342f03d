#!/usr/bin/env python3
"""Ablation of the large-M product: which part of a K-step the time follows.  Needs a DIAGNOSTIC build of gemm_big_kernel in
which tune bits 26..29 switch parts of the kernel off at run time (fragments kept live with empty asm statements); the
shipped kernel ignores those bits, so on it every column prints the same time.  Two such builds were measured on MI355X at
19200 x 256 x 512 with 64 x 128 tiles (34.0 us as shipped):
  bits = no W stores to LDS / no A stores / no matrix instructions:  32.2 / 32.1 / 26.6 us; no stores at all 30.3; neither
         stores nor matrix instructions 22.3
  bits = no W loads / no A loads / no multiply phase (LDS reads + matrix instructions) / no C stores:  31.1 / 31.1 / 26.0 /
         29.3 us; no loads 28.1; no loads and no multiply 20.0; none of the four 15.9
No single part owns the launch: every one of them is worth 3 to 9 us and what remains without operand loads, multiply phase
and output stores (the split, the LDS stores, the barriers, the epilogue arithmetic) is still 16 us.  Eight waves on the same
64 x 128 tile (each link of a K-step's chain done by twice the waves) changed nothing (33.8 us), so the launch is not bound by
one workgroup's dependent chain either; 128 x 128 tiles with eight waves (a third less operand traffic) gain 7-10 % where they
still fill the chip (N >= 768) and lose to quantization at N = 256 (300 workgroups on 256 CUs)."""
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
            for name, bits in (('full', 0), ('noWld', 1 << 26), ('noAld', 1 << 27), ('noLd', 3 << 26), ('noMul', 1 << 28),
                               ('noCst', 1 << 29), ('noLd+noMul', 7 << 26), ('nothing', 15 << 26)):
                t = timeit(lambda: ops.gemm(x, lin, out, seg_len=M, precision='bf16x3', tune=tile | (1 << 8) | bits), reps=10, rounds=3)
                row.append(f'{name} {t:6.1f}')
            print(f'M={M} N={N} K={K} taps={taps} tile{tile}: ' + ' | '.join(row), flush=True)


if __name__ == '__main__':
    main()
