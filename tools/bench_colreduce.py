#!/usr/bin/env python3
"""estts_colreduce on the training shapes (GPU box only)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import estts  # noqa
from estts import train_ops
from tune_gemm import timeit

for M, C in ((19200, 256), (19200, 512), (19200, 1024), (19200, 80), (3200, 512), (3200, 1536)):
    z, g = torch.randn(M, C).cuda(), torch.randn(M, C).cuda()
    o0, o1 = torch.zeros(C).cuda(), torch.zeros(C).cuda()
    mean, rstd = torch.zeros(C).cuda(), torch.ones(C).cuda()
    dy = torch.empty(M, C).cuda()
    t1 = timeit(lambda: train_ops.colreduce(1, o0, None, g=g), reps=20)
    t2 = timeit(lambda: train_ops.colreduce(2, o0, o1, z=z, g=g, mean=mean, rstd=rstd, dy_out=dy), reps=20)
    t0 = timeit(lambda: train_ops.colreduce(0, o0, o1, z=z), reps=20)
    b = M * C * 4
    print(f'M={M} C={C}: moments {t0:6.1f} us ({b / t0 / 1e6:5.2f} TB/s)  bias-grad {t1:6.1f} us ({b / t1 / 1e6:5.2f} TB/s)  '
          f'batchnorm {t2:6.1f} us ({3 * b / t2 / 1e6:5.2f} TB/s)', flush=True)
