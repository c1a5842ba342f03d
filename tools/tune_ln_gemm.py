#!/usr/bin/env python3
"""Time the convolution with a LayerNorm prologue against rownorm + the plain kernels (GPU box only)."""
import math
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import estts  # noqa
from estts import ops
from estts.ops import PackedLinear
from tune_gemm import timeit

for M, N, C, taps in ((100, 512, 512, 3), (600, 512, 256, 3), (100, 512, 512, 5)):
    K = C * taps
    if K > 1536:
        continue
    g = torch.Generator().manual_seed(0)
    y, x = torch.randn(M, C, generator=g).cuda(), torch.randn(M, C, generator=g).cuda()
    ga, ba, gb, bb = (torch.randn(C, generator=g).cuda() for _ in range(4))
    lin = PackedLinear((torch.randn(N, K, generator=g) / math.sqrt(K)).numpy(), device='cuda', taps=taps, lpad=(taps - 1) // 2)
    rows, out = torch.empty(M, C, device='cuda'), torch.empty(M, N, device='cuda')
    t_norm = timeit(lambda: ops.rownorm(y, rows, ln_a=(ga, ba), residual=x, ln_b=(gb, bb)))
    res = [f'rownorm {t_norm:.1f}']
    for name, tune in (('auto', 0), ('t9w8', 9 | (8 << 16)), ('t10w8', 10 | (8 << 16)), ('t11', 11), ('t12', 12), ('ring', 4)):
        try:
            res.append(f'{name} {timeit(lambda: ops.gemm(rows, lin, out, seg_len=M, tune=tune)):.1f}')
        except RuntimeError:
            pass
    ln = dict(y=y, x=x, ln_a=(ga, ba), ln_b=(gb, bb), out=rows)
    for name, tune in (('fused16', 11), ('fused32', 12)):
        try:
            res.append(f'{name} {timeit(lambda: ops.gemm(rows, lin, out, seg_len=M, tune=tune, ln=ln)):.1f}')
        except RuntimeError as e:
            res.append(f'{name} n/a')
    ln1 = dict(y=y, ln_a=(ga, ba), act='relu', out=rows)
    res.append(f'fused-single {timeit(lambda: ops.gemm(rows, lin, out, seg_len=M, ln=ln1)):.1f}')
    print(f'M={M} N={N} C={C} taps={taps}: ' + ' | '.join(res), flush=True)
