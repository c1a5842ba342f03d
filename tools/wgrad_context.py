"""Does what runs before estts_gemm_wgrad change its duration?  Times the 19200 x 256 x 1536 three-tap weight gradient alone,
after an elementwise rewrite of dY, and after an elementwise rewrite of x (wall clock per pair minus the elementwise pass alone).
Run on the GPU box, once as is and once with ESTTS_WGRAD_PER_TAP=1."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import estts  # noqa: E402,F401
from estts import train_ops as T  # noqa: E402

M, N, K, taps, seg = 19200, 256, 1536, 3, 600
R = 16
xs = [torch.randn(M, K // taps, device='cuda') for _ in range(R)]
dys = [torch.randn(M, N, device='cuda') for _ in range(R)]
dw = torch.zeros(K, N, device='cuda')


def timed(fn, reps=960):
    for i in range(8):
        fn(i)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(reps):
        fn(i)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps * 1e6


def wg(i):
    T.wgrad(xs[i % R], dys[i % R], dw, taps=taps, lpad=1, seg_len=seg)


from estts import ops  # noqa: E402
lin = T.TrainLinear(torch.randn(K, N, device='cuda') * 0.02, torch.zeros(N, device='cuda'), taps=taps, lpad=1).fwd
outs = [torch.empty(M, N, device='cuda') for _ in range(R)]


def mm(i):     # the forward product of the same layer: the matrix-core load a training step surrounds the weight gradient with
    for _ in range(3):
        ops.gemm(xs[i % R], lin, outs[i % R], M=M, seg_len=seg)


t_w = timed(wg)
t_mm = timed(mm)
t_mw = timed(lambda i: (mm(i), wg(i)))
print(f'wgrad alone {t_w:.1f} us; between triples of the forward product {t_mw - t_mm:.1f} us (triple {t_mm:.1f})')
t_ey = timed(lambda i: dys[i % R].mul_(1.0001))
t_ex = timed(lambda i: xs[i % R].mul_(1.0001))
t_yw = timed(lambda i: (dys[i % R].mul_(1.0001), wg(i)))
t_xw = timed(lambda i: (xs[i % R].mul_(1.0001), wg(i)))
print(f'wgrad alone {t_w:.1f} us; after rewriting dY {t_yw - t_ey:.1f} us (pass {t_ey:.1f}); after rewriting x {t_xw - t_ex:.1f} us (pass {t_ex:.1f})')
