#!/usr/bin/env python3
"""Operand-precision sensitivity of the two inference workloads, site by site (run by hand on a GPU box; not a test).

For BASELINE config 2 (ForwardTransformer, B=1, N=100, T=600) and config 3 (GST + teacher-forced AutoregressiveTransformer,
B=32 by default) every call site of estts.ops.PrecisionPolicy is switched from split-bf16 (bf16x3) to plain bf16 operands on
its own, then sites are added greedily in order of least damage.  Each row reports the mel L1 distance to the float32 CPU
restatement (oracle/tts_torch.py) and the time of one pass replayed from a hipGraph.

  python tests/precision_sweep.py [--batch3 32] [--out gpurun_out/precision_sweep.json]
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import estts  # noqa: E402,F401
from estts import configs, ops, weights  # noqa: E402

FORWARD_SITES = ['encoder.qkv', 'encoder.attention', 'encoder.attn_out', 'encoder.ffn', 'encoder.conv', 'dur_pred', 'prenet',
                 'decoder.qkv', 'decoder.attention', 'decoder.attn_out', 'decoder.ffn', 'decoder.conv', 'out', 'postnet']
AR_SITES = ['gst', 'encoder.qkv', 'encoder.attention', 'encoder.attn_out', 'encoder.ffn', 'encoder.conv', 'prenet',
            'decoder.qkv', 'decoder.attention', 'decoder.attn_out', 'decoder.cross_q', 'decoder.cross_kv', 'decoder.ffn',
            'decoder.conv', 'out', 'stop', 'postnet']


def timed(step, reps):
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        step()
    torch.cuda.current_stream().wait_stream(s)
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        out = step()
    for _ in range(5):
        g.replay()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        g.replay()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps * 1e3, out


def sweep(name, model, step, ref, sites, reps):
    rows = []

    def measure(label, table):
        model.precision = ops.as_precision(table)
        ms, out = timed(step, reps)
        l1 = float(np.abs(out.cpu().numpy() - ref).mean())
        rows.append({'policy': label, 'bf16_sites': sorted(k for k, v in table.items() if v == 'bf16' and k != 'default'),
                     'mel_l1': l1, 'ms': ms})
        print(f'{name:8s} {label:44s} mel-L1 {l1:.3e}  {ms:8.4f} ms', flush=True)
        return l1, ms

    base_l1, base_ms = measure('all bf16x3', {'default': 'bf16x3'})
    measure('all bf16', {'default': 'bf16'})
    single = {}
    for s in sites:
        single[s] = measure(f'only {s} in bf16', {'default': 'bf16x3', s: 'bf16'})
    # greedy: cheapest damage first
    order = sorted(sites, key=lambda s: single[s][0])
    table = {'default': 'bf16x3'}
    for s in order:
        table[s] = 'bf16'
        measure(f'+ {s}', dict(table))
    return rows


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--batch3', type=int, default=32)
    ap.add_argument('--out', default='gpurun_out/precision_sweep.json')
    args = ap.parse_args()
    from oracle import tts_torch as OT
    torch.set_num_threads(16)
    N, T = 100, 600
    result = {}

    from bench import synthetic_utterance
    from estts.forward_transformer import ForwardTransformer
    cfg = configs.forward_config()
    w = weights.init_forward_transformer(cfg, seed=1234)
    tokens, durs = synthetic_utterance(cfg, 1, N, T, seed=1234)
    model = ForwardTransformer(w, cfg)
    tok_d, dur_d = torch.from_numpy(tokens).cuda(), torch.from_numpy(durs.astype(np.float32)).cuda()
    ref = OT.forward_transformer_call(w, cfg, tokens, durs)['mel']
    result['config2'] = sweep('config 2', model, lambda: model.call(tok_d, target_durations=dur_d, max_frames=T)['mel'], ref,
                              FORWARD_SITES, 200)
    del model

    from estts.autoregressive_transformer import AutoregressiveTransformer
    B = args.batch3
    cfg = configs.autoregressive_config(system_type='style_text')
    w = weights.init_autoregressive_transformer(cfg, seed=1234)
    rng = np.random.default_rng(1234)
    tokens = rng.integers(1, cfg['vocab_size'], size=(B, N)).astype(np.int32)
    targets = np.clip(rng.normal(0, 1.5, size=(B, T, 80)), -4, 4).astype(np.float32)
    model = AutoregressiveTransformer(w, cfg)
    model.set_constants(reduction_factor=1)
    tok_d, tar_d = torch.from_numpy(tokens).cuda(), torch.from_numpy(targets).cuda()
    ref = OT.autoregressive_call(w, cfg, tokens, targets, None, r=1)['final_output']
    result['config3'] = sweep('config 3', model, lambda: model.call(tok_d, tar_d, None)['final_output'], ref, AR_SITES, 20)
    os.makedirs(os.path.dirname(os.path.abspath(args.out)), exist_ok=True)
    with open(args.out, 'w') as f:
        json.dump(result, f, indent=1)


if __name__ == '__main__':
    main()
