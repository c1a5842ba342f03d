#!/usr/bin/env python3
"""estts_attention: 16-query vs 64-query workgroups on the batched shapes (GPU box only)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import estts  # noqa
from estts import ops
from tune_gemm import timeit

for B, H, Lq, Lk, dh, causal, weights in ((32, 4, 600, 600, 64, True, False), (32, 4, 600, 600, 64, False, True),
                                          (32, 4, 600, 100, 64, False, True), (32, 4, 100, 100, 64, False, True),
                                          (32, 4, 100, 100, 128, False, True), (1, 4, 600, 600, 64, False, True),
                                          (8, 4, 600, 600, 64, False, True)):
    d = H * dh
    q, k, v = torch.randn(B * Lq, d).cuda(), torch.randn(B * Lk, d).cuda(), torch.randn(B * Lk, d).cuda()
    ctx = torch.empty(B * Lq, d).cuda()
    attn = torch.empty(B, H, Lq, Lk).cuda() if weights else None
    res = []
    for variant in (1, 2):
        f = lambda: ops.attention(q, k, v, ctx, B, H, Lq, Lk, dh, causal=causal, attn=attn, variant=variant)
        res.append(timeit(f, reps=10, rounds=3))
    fl = 4.0 * B * H * Lq * Lk * dh
    print(f'B={B} H={H} Lq={Lq} Lk={Lk} dh={dh} causal={causal} weights={weights}: 16-query {res[0]:7.1f} us ({fl / res[0] / 1e6:6.1f} TF/s)'
          f' | 64-query {res[1]:7.1f} us ({fl / res[1] / 1e6:6.1f} TF/s)', flush=True)
