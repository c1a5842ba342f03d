import math, os, sys, torch
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/tools')
import estts
from estts import ops
from estts.ops import PackedLinear
from tune_gemm import timeit
g = torch.Generator().manual_seed(0)
for M, C, N, taps in ((600, 256, 512, 3), (100, 512, 512, 3)):
    K = taps * C
    lin = PackedLinear((torch.randn(N, K, generator=g) / math.sqrt(K)).numpy(), device='cuda', taps=taps, lpad=1, act='relu')
    y, x = torch.randn(M, C, generator=g).cuda(), torch.randn(M, C, generator=g).cuda()
    ga, ba = torch.ones(C).cuda(), torch.zeros(C).cuda()
    rows, out = torch.empty(M, C, device='cuda'), torch.empty(M, N, device='cuda')
    outs = []
    for bits in (0, 1 << 26, 0, 1 << 26):
        f = lambda: ops.gemm(rows, lin, out, seg_len=M, precision='bf16x3', tune=bits, ln=dict(y=y, x=x, ln_a=(ga, ba), ln_b=(ga, ba), out=rows))
        f(); outs.append(out.clone())
        print(M, C, N, 'panel' if bits == 0 else 'strip', f'{timeit(f, reps=20, rounds=5):.2f} us', flush=True)
    print('identical', torch.equal(outs[0], outs[1]))
