#!/usr/bin/env python3
"""What the first load of a kernel costs, by what happened to its target since it was last read (GPU box only).
Build first, here: hipcc --offload-arch=gfx950 -O3 -shared -fPIC tools/probe/latency_probe.hip -o tools/probe/latency_probe.so"""
import ctypes
import os

import torch

lib = ctypes.CDLL(os.path.join(os.path.dirname(os.path.abspath(__file__)), 'latency_probe.so'))
vp = ctypes.c_void_p


def main():
    dev = 'cuda'
    out = torch.zeros(8, dtype=torch.int64, device=dev)
    sink = torch.zeros(4, device=dev)
    st = torch.cuda.current_stream().cuda_stream
    small = [torch.randn(256 * 1024, device=dev) for _ in range(64)]        # 1 MiB each, separate allocations
    big = torch.randn(64 << 20, device=dev)                                  # 256 MiB, one allocation
    evict = {'32 MiB': torch.randn(8 << 20, device=dev), '512 MiB': torch.randn(128 << 20, device=dev)}

    def run(buf, far, pre):
        res = []
        for it in range(12):
            if pre is not None:
                lib.sweep(vp(pre.data_ptr()), ctypes.c_long(pre.numel() // 4), vp(sink.data_ptr()), vp(st))
            torch.cuda.synchronize()
            b = buf[it % len(buf)] if isinstance(buf, list) else buf[it * (1 << 20):]
            lib.probe(vp(b.data_ptr()), ctypes.c_long(far), vp(out.data_ptr()), vp(sink.data_ptr()), vp(st))
            torch.cuda.synchronize()
            if it >= 2:
                res.append(out[:4].cpu().numpy() * 0.01)
        import numpy as np
        return np.mean(res, axis=0)

    print('us: first load | another page (+512 KiB) | next line | same line again')
    for name, pre in (('nothing in between (same buffer region re-read)', None), ('32 MiB read in between', evict['32 MiB']),
                      ('512 MiB read in between', evict['512 MiB'])):
        print(f'{name:50s} separate 1 MiB allocations: ', ' '.join(f'{v:6.2f}' for v in run(small, 128 * 1024, pre)))
        print(f'{name:50s} inside one 256 MiB allocation:', ' '.join(f'{v:6.2f}' for v in run(big, 128 * 1024, pre)))


if __name__ == '__main__':
    main()
