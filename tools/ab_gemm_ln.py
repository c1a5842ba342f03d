#!/usr/bin/env python3
"""The LayerNorm epilogue of estts_gemm (64 x 256 tiles, id 14) against the layer on tile 13 followed by rownorm(), and tile 14
against tile 13 on the plain layer, at the 256-output shapes of the batch-size forward passes (GPU box only)."""
import math
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import estts  # noqa
from estts import ops
from estts.ops import PackedLinear
from tune_gemm import timeit

SHAPES = [(19200, 512, 1), (19200, 256, 1), (20000, 512, 1), (24000, 512, 1), (17000, 256, 1), (9600, 512, 1), (9600, 256, 1), (4800, 512, 1), (4800, 256, 1), (3200, 512, 1), (3200, 256, 1),
          (19200, 1536, 3)]


def main():
    g = torch.Generator().manual_seed(0)
    N = 256
    for M, K, taps in SHAPES:
        x = torch.randn(M, K // taps, generator=g).cuda()
        lin = PackedLinear((torch.randn(N, K, generator=g) / math.sqrt(K)).numpy(), bias=torch.randn(N, generator=g).numpy(), device='cuda',
                           taps=taps, lpad=(taps - 1) // 2)
        ln_a = (torch.rand(N, generator=g).cuda() + 0.5, torch.randn(N, generator=g).cuda())
        ln_b = (torch.rand(N, generator=g).cuda() + 0.5, torch.randn(N, generator=g).cuda())
        res = torch.randn(M, N, generator=g).cuda()
        y, out, out2 = (torch.empty(M, N, device='cuda') for _ in range(3))
        norm = dict(ln_a=ln_a, residual=res, ln_b=ln_b)

        def two():
            ops.gemm(x, lin, y, seg_len=M, tune=13)
            ops.rownorm(y, out, ln_a=ln_a, residual=res, ln_b=ln_b)

        def fused():
            ops.gemm(x, lin, out2, seg_len=M, norm=norm)

        two(); fused()
        err = (out - out2).abs().max().item()
        t13 = timeit(lambda: ops.gemm(x, lin, y, seg_len=M, tune=13), reps=10, rounds=5)
        t14 = timeit(lambda: ops.gemm(x, lin, y, seg_len=M, tune=14), reps=10, rounds=5)
        tn = timeit(lambda: ops.rownorm(y, out, ln_a=ln_a, residual=res, ln_b=ln_b), reps=10, rounds=5)
        t2 = timeit(two, reps=10, rounds=5)
        tf = timeit(fused, reps=10, rounds=5)
        tf64 = timeit(lambda: ops.gemm(x, lin, out2, seg_len=M, norm=norm, tune=1 << 26), reps=10, rounds=5)   # 64-row tiles only
        print(f'M={M:6d} K={K:5d} taps={taps} | tile13 {t13:6.1f} us  tile14 {t14:6.1f} us  rownorm {tn:5.1f} us | layer + rownorm {t2:6.1f} us  '
              f'fused {tf:6.1f} us (64-row tiles only {tf64:6.1f})  ({"equal" if err == 0 else "max diff %.1e" % err})', flush=True)


if __name__ == '__main__':
    main()
