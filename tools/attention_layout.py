"""Attention forward (with saved weights) and backward at the training shape (B = 32, H = 4, L = 600, depth 64): q / k / v as
column slices of one [M, 3 d] buffer and ctx as the right half of a [M, 2 d] buffer (the trainer's layout) against separate
contiguous tensors.  GPU box."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import estts  # noqa: E402,F401
from estts import ops, train_ops as T  # noqa: E402

B, H, d, L = 32, 4, 64, 600
M, D = B * L, H * d


def timed(fn, reps=200):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps * 1e6


attn = torch.empty(B, H, L, L, device='cuda')
delta = torch.empty(B * H * L, device='cuda')
for name in ('sliced', 'contiguous'):
    if name == 'sliced':
        qkv = torch.randn(M, 3 * D, device='cuda')
        cat = torch.empty(M, 2 * D, device='cuda')
        dqkv = torch.empty(M, 3 * D, device='cuda')
        q, k, v, ctx = qkv[:, :D], qkv[:, D:2 * D], qkv[:, 2 * D:], cat[:, D:]
        dq, dk, dv = dqkv[:, :D], dqkv[:, D:2 * D], dqkv[:, 2 * D:]
    else:
        q, k, v, ctx, dq, dk, dv = (torch.randn(M, D, device='cuda') for _ in range(7))
    dctx = torch.randn(M, D, device='cuda')
    f = timed(lambda: ops.attention(q, k, v, ctx, B, H, L, L, d, attn=attn))
    b = timed(lambda: T.attention_bwd(q, k, v, ctx, dctx, attn, delta, dq, dk, dv, B, H, L, L, d))
    print(f'{name}: forward with weights {f:.1f} us, backward {b:.1f} us', flush=True)
