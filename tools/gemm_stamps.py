#!/usr/bin/env python3
"""Where a K-step of the W-from-memory GEMM kernel (tile 13) spends its cycles: needs a library built with
-DESTTS_GEMM_STAMPS (see csrc/gemm.hip), in which lane 0 of every wave of one workgroup in the middle of the grid adds up the
shader-clock cycles between the marks of the steady-state loop.  Prints, per wave, cycles per K-step of: waiting for the
operands + split + LDS stores | barrier | issuing the A requests | LDS fragment reads + matrix instructions | issuing the W
requests; then epilogue and whole-kernel cycles.  (GPU box only.)"""
import ctypes
import math
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import estts  # noqa
from estts import _lib, ops
from estts.ops import PackedLinear


def main():
    lib = _lib.load()
    fn = lib.estts_debug_gemm_stamps   # only in a library built with -DESTTS_GEMM_STAMPS
    fn.restype = ctypes.c_int
    fn.argtypes = [ctypes.c_void_p]
    g = torch.Generator().manual_seed(0)
    for M, N, K, taps in ((19200, 256, 512, 1), (19200, 256, 1536, 3), (19200, 1024, 512, 1)):
        w = torch.randn(N, K, generator=g) / math.sqrt(K)
        lin = PackedLinear(w.numpy(), device='cuda', taps=taps, lpad=(taps - 1) // 2)
        x = torch.randn(M, K // taps, generator=g).cuda()
        out = torch.empty(M, N, device='cuda')
        for _ in range(3):
            ops.gemm(x, lin, out, seg_len=M, precision='bf16x3', tune=13 | (1 << 8))
        torch.cuda.synchronize()
        buf = (ctypes.c_ulonglong * 32)()
        rc = fn(ctypes.cast(buf, ctypes.c_void_p))
        assert rc == 0, rc
        nk = (K + 31) // 32
        steady = (nk - 2) // 2 * 2 if nk >= 4 else 0
        print(f'M={M} N={N} K={K} taps={taps}: {nk} K-steps, {steady} in the steady-state loop')
        for wv in range(4):
            v = [buf[wv * 8 + i] for i in range(8)]
            per = [c / max(steady, 1) for c in v[:5]]
            print(f'  wave {wv}: wait+split+store {per[0]:7.0f} | barrier {per[1]:7.0f} | A requests {per[2]:6.0f} | reads+MFMA {per[3]:7.0f} | '
                  f'W requests {per[4]:6.0f} | sum {sum(per):7.0f} per K-step;  tail {v[5]:7d}  epilogue {v[6]:7d}  kernel {v[7]:8d} cycles')


if __name__ == '__main__':
    main()
