#!/usr/bin/env python3
"""Where the one-shot convolution kernel with the LayerNorm prologue (the dominant launch of the batch-1 forward: 600 x 512 x 768,
three taps) spends its time: needs a library built with -DESTTS_ONESHOT_STAMPS (see csrc/gemm_oneshot.hip), in which lane 0 of
wave 0 of every workgroup leaves the 100 MHz clock at its phase boundaries.  (GPU box only.)"""
import ctypes
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import estts  # noqa
from estts import _lib, ops
from estts.ops import PackedLinear

NAMES = ['start', 'all requests issued', 'rows normalised, in LDS', 'barrier', 'matrix instructions', 'partial tiles exchanged', 'end']


def main():
    lib = _lib.load()
    fn = lib.estts_debug_oneshot_stamps
    fn.restype = ctypes.c_int
    fn.argtypes = [ctypes.c_void_p]
    g = torch.Generator().manual_seed(0)
    M, N, C, taps = 600, 512, 256, 3
    lin = PackedLinear((torch.randn(N, taps * C, generator=g) / 28).numpy(), bias=torch.randn(N, generator=g).numpy(), device='cuda', taps=taps,
                       lpad=1, act='relu')
    y, x = torch.randn(M, C, generator=g).cuda(), torch.randn(M, C, generator=g).cuda()
    vec = lambda: (torch.rand(C, generator=g).cuda() + .5, torch.randn(C, generator=g).cuda())
    la, lb = vec(), vec()
    out, lo = torch.empty(M, N, device='cuda'), torch.empty(M, C, device='cuda')
    big = torch.randn(32 << 20, device='cuda')
    acc = []
    for it in range(10):
        big.add_(1.0)                                  # something else through the caches in between, as in a forward pass
        ops.gemm(lo, lin, out, seg_len=M, ln=dict(y=y, x=x, ln_a=la, ln_b=lb, out=lo))
        torch.cuda.synchronize()
        buf = (ctypes.c_ulonglong * (1024 * 8))()
        assert fn(ctypes.cast(buf, ctypes.c_void_p)) == 0
        st = np.frombuffer(buf, dtype=np.uint64).reshape(1024, 8)[:304, :7].astype(np.int64)
        if it >= 2:
            acc.append((st - st[:, 0].min()) * 0.01)
    a = np.mean(acc, axis=0)                           # [workgroup, mark] in us since the first workgroup started
    print('us since the first workgroup started: mean over workgroups | earliest | latest')
    for i, n in enumerate(NAMES):
        print(f'{n:28s} {a[:, i].mean():6.2f} | {a[:, i].min():6.2f} | {a[:, i].max():6.2f}')
    d = np.diff(a, axis=1)
    print('phase durations per workgroup (us): mean | max')
    for i, n in enumerate(NAMES[1:]):
        print(f'-> {n:25s} {d[:, i].mean():6.2f} | {d[:, i].max():6.2f}')
    order = np.argsort(a[:, 0])
    print('start times of workgroups (sorted), every 32nd:', ' '.join(f'{v:.2f}' for v in a[order, 0][::32]))
    print('end   times of the same workgroups:            ', ' '.join(f'{v:.2f}' for v in a[order, 6][::32]))


if __name__ == '__main__':
    main()
