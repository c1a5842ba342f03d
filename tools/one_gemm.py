#!/usr/bin/env python3
"""Run one estts_gemm shape repeatedly (for rocprofv3 --pmc / --kernel-trace runs)."""
import math
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import estts  # noqa
from estts import ops
from estts.ops import PackedLinear

M, N, K, taps, tune, reps = (int(a) for a in sys.argv[1:7])
prec = sys.argv[7] if len(sys.argv) > 7 else 'bf16x3'
g = torch.Generator().manual_seed(0)
x = torch.randn(M, K // taps, generator=g).cuda()
w = torch.randn(N, K, generator=g) / math.sqrt(K)
lin = PackedLinear(w.numpy(), device='cuda', taps=taps, lpad=(taps - 1) // 2)
out = torch.empty(M, N, device='cuda')
for _ in range(reps):
    ops.gemm(x, lin, out, seg_len=M, precision=prec, tune=tune)
torch.cuda.synchronize()
print('done')
