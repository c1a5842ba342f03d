"""Times estts_gemm_wgrad at the training shapes of the ForwardTransformer (B = 32): run on the GPU box, once as is and once
with ESTTS_WGRAD_PER_TAP=1 (the per-tap kernel) to compare.  python tools/tune_wgrad.py"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import estts  # noqa: E402,F401
from estts import train_ops as T  # noqa: E402

if os.environ.get('TUNE_SMALL'):      # the B = 8 step of the autoregressive trainer, the B = 32 encoder
    SHAPES = [(4800, 256, 1536, 3, 600), (4800, 512, 768, 3, 600), (4800, 256, 1280, 5, 600), (3200, 512, 1536, 3, 100),
              (800, 512, 1536, 3, 100)]
elif os.environ.get('TUNE_ONETAP'):   # dense layers of the B = 32 step (decoder 256 wide, encoder 512 wide)
    SHAPES = [(19200, 1024, 256, 1, 0), (19200, 256, 1024, 1, 0), (19200, 768, 256, 1, 0), (19200, 256, 256, 1, 0), (19200, 256, 512, 1, 0),
              (19200, 128, 256, 1, 0), (3200, 1536, 512, 1, 0), (3200, 512, 512, 1, 0), (3200, 2048, 512, 1, 0), (3200, 512, 2048, 1, 0),
              (19200, 200, 260, 1, 0)]
else:
  SHAPES = [(19200, 256, 1536, 3, 600), (19200, 512, 768, 3, 600), (3200, 512, 1536, 3, 100), (19200, 256, 1280, 5, 600),
            (19200, 80, 1280, 5, 600), (19200, 256, 768, 1, 0)]
SPLITS = [int(v) for v in os.environ.get('TUNE_SPLITS', '0').split(',')]      # 0 = the library's own choice
ROTATE = int(os.environ.get('TUNE_ROTATE', '1'))     # operand sets used in turn: 8 of them no longer fit the Infinity Cache
for M, N, K, taps, seg in SHAPES:
    xs = [torch.randn(M, K // taps, device='cuda') for _ in range(ROTATE)]
    dys = [torch.randn(M, N, device='cuda') for _ in range(ROTATE)]
    dw = torch.zeros(K, N, device='cuda')
    db = torch.zeros(N, device='cuda') if os.environ.get('TUNE_DB') else None
    for sp in SPLITS:
        if os.environ.get('TUNE_CHECK'):     # against a float64 product (one-tap shapes)
            dw.zero_()
            T.wgrad(xs[0], dys[0], dw, taps=taps, lpad=(taps - 1) // 2, seg_len=seg, splits=sp, db=db)
            if taps == 1:
                ref = xs[0].double().t() @ dys[0].double()
                print('   max err vs float64: %.2e (scale %.1f)' % ((dw.double() - ref).abs().max().item(), ref.abs().max().item()))
        for i in range(3):
            T.wgrad(xs[i % ROTATE], dys[i % ROTATE], dw, taps=taps, lpad=(taps - 1) // 2, seg_len=seg, splits=sp, db=db)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        reps = int(os.environ.get("TUNE_REPS", "48"))
        for i in range(reps):
            T.wgrad(xs[i % ROTATE], dys[i % ROTATE], dw, taps=taps, lpad=(taps - 1) // 2, seg_len=seg, splits=sp, db=db)
        torch.cuda.synchronize()
        us = (time.perf_counter() - t0) / reps * 1e6
        print(f'M={M} N={N} K={K} taps={taps} splits={sp}: {us:7.1f} us  {2.0 * M * N * K / us / 1e6:6.1f} TFLOP/s', flush=True)
