#!/usr/bin/env python3
"""Times the LDS-staged tiles of estts_gemm (ids 6, 8 and the W-from-memory tile 13) on the large-M shapes of configs 3 and 5 and checks them against a
float64 product of a row sample (GPU box only).  Run once per library build to compare two builds on one box:
  cp ab_libs/old.so <package>/libestts.so && python3 tools/ab_gemm_big.py && cp ab_libs/new.so <package>/libestts.so && ..."""
import math
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import estts  # noqa
from estts import ops
from estts.ops import PackedLinear
from tune_gemm import timeit

TILES = [int(t, 0) for t in os.environ.get('AB_TILES', '6,8,13').split(',')]   # e.g. AB_TILES=8,13
SHAPES = [  # (M, N, K, taps)
    (19200, 256, 512, 1), (19200, 256, 1536, 3), (19200, 512, 768, 3), (19200, 256, 1280, 5), (19200, 768, 256, 1),
    (19200, 1024, 512, 1), (19200, 256, 256, 1), (19200, 256, 96, 1), (19200, 256, 160, 1), (4800, 256, 512, 1), (3200, 1536, 512, 1),
]


SMALL = [(4800, 256, 512, 1), (4800, 256, 1536, 3), (4800, 512, 768, 3), (4800, 256, 1280, 5), (3200, 512, 1536, 3), (3200, 512, 1024, 1),
         (3200, 512, 512, 1), (2400, 256, 512, 1), (1200, 256, 512, 1), (2400, 512, 768, 3), (1600, 512, 1536, 3), (9600, 256, 512, 1)]


WIDE = [(19200, 512, 768, 3), (19200, 1024, 512, 1), (19200, 768, 256, 1), (19200, 512, 512, 1), (19200, 512, 1536, 3), (9600, 512, 768, 3),
        (9600, 1024, 512, 1), (9600, 768, 256, 1), (4800, 512, 768, 3), (4800, 1024, 256, 1), (4800, 768, 256, 1), (3200, 512, 1536, 3),
        (3200, 1536, 512, 1), (3200, 2048, 512, 1), (3200, 512, 2048, 1), (3200, 512, 512, 1), (6400, 512, 1536, 3), (6400, 1536, 512, 1)]


MID = [(3200, 512, 1536, 3), (3200, 512, 1024, 1), (3200, 512, 1536, 1), (3200, 512, 512, 1), (3200, 256, 512, 1), (3200, 768, 256, 1),
       (3200, 1536, 512, 1), (4800, 256, 512, 1), (4800, 256, 1536, 3), (4800, 512, 768, 3), (2400, 256, 512, 1), (2400, 512, 1536, 3),
       (1600, 512, 1536, 3), (6400, 512, 1536, 3), (6400, 256, 512, 1), (9600, 256, 512, 1), (800, 512, 1536, 3), (3200, 256, 256, 1)]


GRID = [(M, N, K, t) for M in (600, 832, 1200, 1600, 2400, 3200) for N in (256, 512, 768, 1536) for K, t in ((256, 1), (512, 1), (1536, 3))]


NARROW = [(19200, 80, 1280, 5), (19200, 80, 256, 1), (19200, 96, 512, 1), (19200, 112, 1280, 5), (4800, 80, 1280, 5), (19200, 64, 512, 1),
          (24000, 64, 288, 1), (96000, 32, 288, 1)]


def main():
    g = torch.Generator().manual_seed(0)
    for M, N, K, taps in (SMALL if os.environ.get('AB_SMALL') else WIDE if os.environ.get('AB_WIDE') else MID if os.environ.get('AB_MID') else GRID if os.environ.get('AB_GRID') else NARROW if os.environ.get('AB_NARROW') else SHAPES):
        cin = K // taps
        x = torch.randn(M, cin, generator=g).cuda()
        w = torch.randn(N, K, generator=g) / math.sqrt(K)
        lin = PackedLinear(w.numpy(), device='cuda', taps=taps, lpad=(taps - 1) // 2)
        out = torch.empty(M, N, device='cuda')
        # float64 reference on the rows around the sequence start and end and a few in the middle
        rows = torch.tensor([0, 1, 2, 63, 64, 65, M // 2, M - 66, M - 3, M - 2, M - 1])
        xp = torch.nn.functional.pad(x.double().cpu(), (0, 0, (taps - 1) // 2, taps - 1 - (taps - 1) // 2))
        cols = torch.cat([xp[rows + t] for t in range(taps)], dim=1)       # tap-major columns
        ref = cols @ w.double().t()
        res = []
        prev = None
        for tile in TILES:
            ops.gemm(x, lin, out, seg_len=M, precision='bf16x3', tune=tile | (1 << 8))
            err = (out[rows.cuda()].double().cpu() - ref).abs().max().item()
            same = '' if prev is None or tile < 13 else (' = prev' if torch.equal(out, prev) else ' != prev (max %.1e)' % (out - prev).abs().max().item())
            prev = out.clone()
            t = timeit(lambda: ops.gemm(x, lin, out, seg_len=M, precision='bf16x3', tune=tile | (1 << 8)), reps=10, rounds=5)
            res.append(f'tile{tile & 0xff}{"*" if tile >> 8 else ""} {t:7.1f} us {2.0 * M * N * K / t / 1e6:6.1f} TF/s (err {err:.1e}{same})')
        print(f'M={M:6d} N={N:5d} K={K:5d} taps={taps} | ' + ' | '.join(res), flush=True)


if __name__ == '__main__':
    main()
