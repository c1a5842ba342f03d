"""Does the row alignment of the saved attention weights matter?  Times estts_attention (with weights) and estts_attention_bwd at
B = 32, H = 4, depth 64 for sequence lengths around 600: rows of 600 floats start on 32-byte boundaries, rows of 608 / 640 floats
on 128-byte ones.  GPU box."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import estts  # noqa: E402,F401
from estts import ops, train_ops as T  # noqa: E402

B, H, d = 32, 4, 64


def timed(fn, reps=200):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps * 1e6


for L in (576, 592, 600, 608, 624, 640):
    M = B * L
    qkv = torch.randn(M, 3 * H * d, device='cuda')
    ctx = torch.empty(M, H * d, device='cuda')
    dctx = torch.randn(M, H * d, device='cuda')
    attn = torch.empty(B, H, L, L, device='cuda')
    dqkv = torch.empty(M, 3 * H * d, device='cuda')
    delta = torch.empty(B * H * L, device='cuda')
    D = H * d
    q, k, v = qkv[:, :D], qkv[:, D:2 * D], qkv[:, 2 * D:]
    f = timed(lambda: ops.attention(q, k, v, ctx, B, H, L, L, d, attn=attn))
    f0 = timed(lambda: ops.attention(q, k, v, ctx, B, H, L, L, d))
    b = timed(lambda: T.attention_bwd(q, k, v, ctx, dctx, attn, delta, dqkv[:, :D], dqkv[:, D:2 * D], dqkv[:, 2 * D:], B, H, L, L, d))
    print(f'L={L}: forward with weights {f:6.1f} us ({f / L / L * 1e3:.3f} ns per score), without {f0:6.1f} us, backward {b:6.1f} us '
          f'({b / L / L * 1e3:.3f} ns per score)', flush=True)
