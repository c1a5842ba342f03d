#!/usr/bin/env python3
"""Where a row step of wgrad_kernel spends its cycles: needs a library built with -DESTTS_WGRAD_STAMPS (see
csrc/gemm_wgrad.hip), in which lane 0 of every wave of one workgroup in the middle of the grid adds up the shader-clock cycles
between the marks of the row loop.  Prints, per wave, cycles per 32-row step of: waiting for the loads + split + LDS stores |
barrier | issuing the next loads | LDS fragment reads + matrix instructions; then the atomics and whole-kernel cycles.
(GPU box only.)"""
import ctypes
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import estts  # noqa
from estts import _lib
from estts import train_ops as T


def main():
    lib = _lib.load()
    fn = lib.estts_debug_wgrad_stamps   # only in a library built with -DESTTS_WGRAD_STAMPS
    fn.restype = ctypes.c_int
    fn.argtypes = [ctypes.c_void_p]
    for M, N, K in ((19200, 256, 512), (19200, 1024, 256), (19200, 256, 1024), (3200, 1536, 512)):
        x = torch.randn(M, K, device='cuda')
        dy = torch.randn(M, N, device='cuda')
        dw = torch.zeros(K, N, device='cuda')
        for sp in (0, 16):
            for _ in range(3):
                T.wgrad(x, dy, dw, taps=1, lpad=0, seg_len=0, splits=sp)
            torch.cuda.synchronize()
            buf = (ctypes.c_ulonglong * 32)()
            rc = fn(ctypes.cast(buf, ctypes.c_void_p))
            assert rc == 0, rc
            print(f'M={M} N={N} K={K} splits={sp}')
            for wv in range(4):
                v = [buf[wv * 8 + i] for i in range(8)]
                loop = sum(v[:4])
                print(f'  wave {wv}: wait+split+store {v[0]:7d} | barrier {v[1]:7d} | load requests {v[2]:6d} | reads+MFMA {v[3]:7d} | '
                      f'loop {loop:7d}  atomics {v[4]:6d}  kernel {v[7]:8d} cycles')


if __name__ == '__main__':
    main()
