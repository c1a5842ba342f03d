#!/usr/bin/env python3
"""estts_cross_attention_row against the three launches it replaces (one decoder position, 100 encoder positions), each timed
inside a hipGraph so that the host is out of the picture (GPU box only)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import estts  # noqa
from estts import ops
from estts.ops import PackedLinear
from tune_gemm import timeit


def main():
    d, H = 256, 4
    gen = torch.Generator().manual_seed(0)
    for Lk in (100, 200):
        # several blocks' worth of weights in turn, as a decoder step sees them (nothing stays in L2 between uses)
        blocks = []
        for b in range(8):
            wq = PackedLinear((torch.randn(d, d, generator=gen) / 16).numpy(), bias=torch.randn(d, generator=gen).numpy(), device='cuda')
            wo = PackedLinear((torch.randn(d, 2 * d, generator=gen) / 23).numpy(), bias=torch.randn(d, generator=gen).numpy(), device='cuda')
            kv = torch.randn(Lk, 2 * d, generator=gen).cuda()
            blocks.append((wq, wo, kv))
        yin, xin = torch.randn(1, d, generator=gen).cuda(), torch.randn(1, d, generator=gen).cuda()
        la = (torch.rand(d, generator=gen).cuda() + .5, torch.randn(d, generator=gen).cuda())
        lb = (torch.rand(d, generator=gen).cuda() + .5, torch.randn(d, generator=gen).cuda())
        xc = torch.zeros(1, 2 * d, device='cuda'); q = torch.empty(1, d, device='cuda'); y = torch.empty(1, d, device='cuda')
        attn = torch.empty(1, H * Lk, device='cuda'); lo = torch.empty(1, d, device='cuda')
        state = {'i': 0}

        def three():
            wq, wo, kv = blocks[state['i'] % 8]; state['i'] += 1
            ops.gemm(xc[:, :d], wq, q, ln=dict(y=yin, x=xin, ln_a=la, ln_b=lb, out=xc[:, :d]))
            ops.attention(q, kv[:, :d], kv[:, d:], xc[:, d:], 1, H, 1, Lk, d // H, attn=attn.view(1, H, 1, Lk))
            ops.gemm(xc, wo, y)

        def one():
            wq, wo, kv = blocks[state['i'] % 8]; state['i'] += 1
            ops.cross_attention_row(wq, wo, kv[:, :d], kv[:, d:], y, H, ln=dict(y=yin, x=xin, ln_a=la, ln_b=lb, out=lo), attn=attn)

        for name, fn in (('three launches', three), ('one launch', one)):
            state['i'] = 0
            t = timeit(fn, reps=48, rounds=5)
            print(f'Lk={Lk} {name}: {t:6.2f} us per block', flush=True)


if __name__ == '__main__':
    main()
