#!/usr/bin/env python3
"""Where a launch of estts_decoder_block_row spends its time: the 100 MHz clock at the phase boundaries of each of the four
workgroups (needs a library built with -DESTTS_BLOCK_STAMPS; GPU box only)."""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import estts  # noqa
from estts import ops
from estts.ops import PackedLinear

NAMES = ['start', 'vectors in LDS', 'q|k|v done', 'self-attention', 'arrived 1', 'released 1', 'y_s summed', 'q\' done', 'cross-attention + share',
         'arrived 2', 'released 2', 'end']


def main():
    d, H, Lk, S = 256, 4, 100, 60
    g = torch.Generator().manual_seed(0)
    R = lambda *s: torch.randn(*s, generator=g)
    blocks = []
    for b in range(4):
        lin = {n: PackedLinear((R(*sh) / np.sqrt(sh[1])).numpy(), bias=R(sh[0]).numpy(), device='cuda')
               for n, sh in dict(qkv=(3 * d, d), os=(d, 2 * d), qc=(d, d), oc=(d, 2 * d), ff=(d, d)).items()}
        vec = lambda: (R(d).abs().cuda() + .5, R(d).cuda())
        blocks.append((lin, vec(), vec(), vec(), R(S, 3 * d).cuda(), R(Lk, 2 * d).cuda()))
    y2 = [torch.empty(1, d, device='cuda') for _ in range(2)]
    x2 = [torch.empty(1, d, device='cuda') for _ in range(2)]
    x = R(1, d).cuda()
    ws = ops.gemm_workspace(x.device)[0]
    pa, pb = (R(d).abs().cuda() + .5, R(d).cuda()), (R(d).abs().cuda() + .5, R(d).cuda())
    rows = []
    for it in range(12):
        pos = 10 + it
        for b, (lin, l1, l2, l3, cache, enc) in enumerate(blocks):
            ln = None if b == 0 else dict(y=y2[(b - 1) & 1], x=x2[(b - 1) & 1], ln_a=pa, act='relu', ln_b=pb)
            ops.decoder_block_row(lin['qkv'], lin['os'], l1, l2, lin['qc'], lin['oc'], l3, lin['ff'], cache, pos, enc[:, :d], enc[:, d:],
                                  y2[b & 1], x2[b & 1], x=x if b == 0 else None, ln=ln)
            torch.cuda.synchronize()
            st = ws[2 * H * d:2 * H * d + 2 * 16 * H].view(torch.int64).cpu().numpy().reshape(H, 16)[:, :12]
            if it >= 4:
                rows.append(st - st[:, :1].min())
    m = np.mean(rows, axis=0) * 0.01       # us
    print('us since the first workgroup started (mean of %d launches), per workgroup' % len(rows))
    for i, n in enumerate(NAMES):
        print(f'{n:26s} ' + ' '.join(f'{v:7.2f}' for v in m[:, i]))


if __name__ == '__main__':
    main()
