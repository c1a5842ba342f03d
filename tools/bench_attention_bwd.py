#!/usr/bin/env python3
"""estts_attention_bwd at the training shapes: time per call (delta prepass + the two passes) and the largest deviation from
float64 autograd on one (batch, head) (GPU box only)."""
import math
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import estts  # noqa
from estts import ops, train_ops
from tune_gemm import timeit

SHAPES = ((32, 4, 600, 600, 64), (32, 4, 100, 100, 128), (32, 4, 600, 100, 64), (8, 4, 600, 600, 64), (32, 4, 608, 608, 64),
          (32, 4, 130, 70, 64))


def main():
    g = torch.Generator().manual_seed(0)
    for B, H, Lq, Lk, dh in SHAPES:
        d = H * dh
        q, k, v = (torch.randn(B * L, d, generator=g).cuda() for L in (Lq, Lk, Lk))
        dctx = torch.randn(B * Lq, d, generator=g).cuda()
        ctx = torch.empty(B * Lq, d).cuda()
        attn = torch.empty(B, H, Lq, Lk).cuda()
        ops.attention(q, k, v, ctx, B, H, Lq, Lk, dh, attn=attn)
        delta = torch.empty(B * H * Lq).cuda()
        dq, dk, dv = torch.empty_like(q), torch.empty_like(k), torch.empty_like(v)
        f = lambda: train_ops.attention_bwd(q, k, v, ctx, dctx, attn, delta, dq, dk, dv, B, H, Lq, Lk, dh)
        f()
        # float64 autograd on the last (batch, head)
        b, h = B - 1, H - 1
        sl = slice(h * dh, (h + 1) * dh)
        q64, k64, v64 = (t[b * L:(b + 1) * L, sl].double().cpu().requires_grad_() for t, L in ((q, Lq), (k, Lk), (v, Lk)))
        p = torch.softmax(q64 @ k64.t() / math.sqrt(dh), dim=-1)
        (p @ v64).backward(dctx[b * Lq:(b + 1) * Lq, sl].double().cpu())
        errs = []
        for got, ref, L in ((dq, q64.grad, Lq), (dk, k64.grad, Lk), (dv, v64.grad, Lk)):
            errs.append(((got[b * L:(b + 1) * L, sl].double().cpu() - ref).abs().max() / ref.abs().max()).item())
        t = timeit(f, reps=5, rounds=5)
        fl = 10.0 * B * H * Lq * Lk * dh
        print(f'B={B} H={H} Lq={Lq} Lk={Lk} dh={dh}: {t:7.1f} us ({fl / t / 1e6:6.1f} TF/s)  rel err dq {errs[0]:.1e} dk {errs[1]:.1e} dv {errs[2]:.1e}',
              flush=True)


if __name__ == '__main__':
    main()
