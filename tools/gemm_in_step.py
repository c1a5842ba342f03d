"""Replays the estts_gemm calls of one config-3 forward (GST + teacher-forced AR, B = 32; REPLAY_WORKLOAD=train: of one
ForwardTransformer training step) on their own operands and on fresh,
contiguous ones of the same shape: a gap between the two points at the operand layout (strides, alignment, epilogue operands)
rather than at the kernel.  GPU box.  python tools/gemm_in_step.py [min_us]"""
import ctypes
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import estts  # noqa: E402,F401
from estts import _lib, configs, ops, weights  # noqa: E402
from estts.autoregressive_transformer import AutoregressiveTransformer  # noqa: E402

B, N, T = 32, 100, 600
WORKLOAD = os.environ.get('REPLAY_WORKLOAD', 'gst')
calls = []
orig = ops._record


def spy(kind, info, fn, *args, keep=()):
    if kind == 'gemm':
        calls.append((info, _lib.GemmArgs.from_buffer_copy(args[0]), keep))
    return orig(kind, info, fn, *args, keep=keep)


if WORKLOAD == 'gst':
    cfg = configs.autoregressive_config(system_type='style_text')
    w = weights.init_autoregressive_transformer(cfg, seed=1234)
    rng = np.random.default_rng(1234)
    tokens = rng.integers(1, cfg['vocab_size'], size=(B, N)).astype(np.int32)
    targets = np.clip(rng.normal(0, 1.5, size=(B, T, 80)), -4, 4).astype(np.float32)
    model = AutoregressiveTransformer(w, cfg, precision='bf16x3')
    model.set_constants(reduction_factor=1)
    tok_d, tar_d = torch.from_numpy(tokens).cuda(), torch.from_numpy(targets).cuda()

    def step():
        return model.call(tok_d, tar_d, None, return_attention=False)
else:           # one ForwardTransformer training step
    import bench
    from estts.forward_training import ForwardTransformerTrainer
    cfg = configs.forward_config()
    w = weights.init_forward_transformer(cfg, seed=1234)
    tokens, durs = bench.synthetic_utterance(cfg, B, N, T, seed=1234)
    mel = np.clip(np.random.default_rng(99).normal(0, 1.5, size=(B, T, cfg['mel_channels'])), -4, 4).astype(np.float32)
    model = ForwardTransformerTrainer(w, cfg, precision='bf16x3', seed=1234)
    model.set_constants(learning_rate=1e-4, decoder_prenet_dropout=0.0, drop_n_heads=0)
    tok_d, dur_d, mel_d = torch.from_numpy(tokens).cuda(), torch.from_numpy(durs).cuda(), torch.from_numpy(mel).cuda()

    def step():
        return model.train_step(tok_d, mel_d, dur_d)
step()
ops._record = spy
ops.start_trace()
out = step()
ops.stop_trace()
ops._record = orig
torch.cuda.synchronize()
lib = _lib.load()
st = torch.cuda.current_stream().cuda_stream


def timed(a, reps=100):
    for _ in range(5):
        lib.estts_gemm(a, st)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        lib.estts_gemm(a, st)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps * 1e6


min_us = float(sys.argv[1]) if len(sys.argv) > 1 else 10.0
seen = set()
for info, a, keep in calls:
    key = (a.M, a.N, a.K, a.taps, a.lda, a.ldc, a.ldr, bool(a.R), bool(a.s2), a.act, a.A % 256, a.C % 256, a.seg_len)
    if key in seen:
        continue
    seen.add(key)
    # fresh operands are built for plain dense layers and 1-D convolutions only: a 2-D convolution reads image pixels beyond M
    # rows, a batched call reads the other products behind the first, a LayerNorm prologue has operands of its own
    if a.conv_kw > 0 or a.batch > 1 or a.ln_y or a.r_mod > 0 or a.stat_slots:
        continue
    own = timed(a)
    if own < min_us:
        continue
    b = _lib.GemmArgs.from_buffer_copy(a)
    cin = a.K // a.taps
    xa = torch.randn(a.M, cin, device='cuda')
    xc = torch.empty(a.M, a.N, device='cuda')
    b.A, b.lda, b.C, b.ldc = xa.data_ptr(), cin, xc.data_ptr(), a.N
    if a.R:
        xr = torch.randn(a.M, a.N, device='cuda')
        b.R, b.ldr = xr.data_ptr(), a.N
    fresh = timed(b)
    print(f'M={a.M} N={a.N} K={a.K} taps={a.taps} lda={a.lda} ldc={a.ldc} ldr={a.ldr if a.R else 0} act={a.act} s2={bool(a.s2)} '
          f'A%256={a.A % 256} C%256={a.C % 256}: own {own:6.1f} us, fresh contiguous {fresh:6.1f} us', flush=True)
