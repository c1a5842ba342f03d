#!/usr/bin/env python3
"""Benchmark of the hot path on MI355X.

A "step" is one ForwardTransformer forward (non-autoregressive TransformerTTS, 80-bin mel) over one batch of
synthetic LJSpeech-shaped input; at the default workload (BASELINE config 2) that is B=1, N=100 phoneme
tokens, T=600 mel frames.  Inputs and weights are resident in HBM before the timed region; the whole forward
is replayed from a hipGraph (captured through torch.cuda.CUDAGraph).  With --gpus N>1 every rank runs an
independent replica on its own utterances (the path shards by utterance; there is no data-path collective),
so scaling is "weak".

Prints ONE JSON line (rank 0) with the contract fields plus
  roofline      the kernel with the largest share of the forward's device time, timed in isolation with HIP
                events on the launch stream, priced against the dense bf16 MFMA peak (or HBM for byte-bound ops)
  cpu_baseline  the op-for-op PyTorch CPU float32 restatement of the same graph (oracle/tts_torch.py) on this
                host's cores, on the same utterance (rank 0, N=1 only)
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import estts  # noqa: E402,F401
from estts import configs, ops, weights  # noqa: E402
from estts.forward_transformer import ForwardTransformer  # noqa: E402

MFMA_BF16_PEAK = 2.5e15   # dense bf16, FLOP/s (MI355X)
HBM_PEAK = 8.0e12         # B/s


def synthetic_utterance(cfg, B, N, T, seed):
    rng = np.random.default_rng(seed)
    tokens = rng.integers(1, cfg['vocab_size'], size=(B, N)).astype(np.int32)
    d = rng.integers(1, 12, size=(B, N)).astype(np.float64)
    d = np.floor(d * T / d.sum(axis=1, keepdims=True))
    d[:, 0] += T - d.sum(axis=1)
    return tokens, d.astype(np.int32)


def algorithmic_flops(N, T):
    """MACs per utterance of the reference graph (per-token, per-frame and attention terms), times 2."""
    macs = N * 12_583_424 + T * 6_451_200 + 4 * 2 * N * N * 512 + 4 * 2 * T * T * 256
    return 2.0 * macs


def time_kernels(trace, reps=20, rounds=3):
    """Device time of every traced launch in isolation: `reps` launches of the same kernel captured into one hipGraph
    (so that the host's launch cost is not what is measured), replayed between two HIP events on the launch stream;
    best of `rounds`.  The figure includes the ~1-2 us dependent-launch boundary between consecutive replays."""
    rows = []
    stream = torch.cuda.Stream()
    with torch.cuda.stream(stream):
        for kind, info, fn in trace:
            if ops._debug_sync:
                print(f'[bench] timing {kind} {info}', file=sys.stderr, flush=True)
            fn()
            stream.synchronize()
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g, stream=stream):
                for _ in range(reps):
                    fn()
            g.replay()
            best = float('inf')
            for _ in range(rounds):
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record(stream)
                g.replay()
                e1.record(stream)
                e1.synchronize()
                best = min(best, e0.elapsed_time(e1) * 1e-3 / reps)
            rows.append((kind, info, best))
            del g
    torch.cuda.synchronize()
    from estts import train_ops
    train_ops.SumSlots.reset_all()      # a replayed layer that sums its output's moments leaves them where no consumer clears them
    return rows


def launch_floor_us(n_launches=7, reps=50):
    """Device time per dependent launch of a kernel that does nothing to speak of (one 4-element estts_axpby), replayed from
    a hipGraph of reps x n_launches launches: what a chain of n_launches kernels costs before any of them does work."""
    from estts import _lib
    lib = _lib.load()
    x = torch.zeros(4, device='cuda')
    y = torch.zeros(4, device='cuda')
    stream = torch.cuda.Stream()
    with torch.cuda.stream(stream):
        def chain():
            for _ in range(n_launches):
                _lib.check(lib.estts_axpby(x.data_ptr(), 4, 1.0, y.data_ptr(), 4, 1.0, 1, 4, stream.cuda_stream), 'estts_axpby')
        chain()
        stream.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=stream):
            for _ in range(reps):
                chain()
        g.replay()
        best = float('inf')
        for _ in range(3):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(stream)
            g.replay()
            e1.record(stream)
            e1.synchronize()
            best = min(best, e0.elapsed_time(e1) * 1e3 / (reps * n_launches))
    torch.cuda.synchronize()
    return best


def wavernn_mol_pass(precision):
    """The same 64-utterance pass in the reference's shipped default output mode (voc_mode = 'MOL', WaveRNN/hparams.py:33:
    30 mixture-of-logistics parameters per sample instead of 512 classes, no mu-law)."""
    from estts.wavernn import WaveRNN
    hp = configs.wavernn_hparams_22k(voc_mode='MOL')
    m = WaveRNN.from_hparams(hp, precision=precision)
    m.load_state_dict(weights.init_wavernn(hp, seed=1234))
    rng = np.random.default_rng(1234)
    mels = [rng.uniform(0, 1, size=(hp['num_mels'], 560)).astype(np.float32) for _ in range(64)]
    m.generate_batch(mels[:1], True, 1000, 100, False, seed=0)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    outs = m.generate_batch(mels, True, hp['voc_target'], hp['voc_overlap'], False, seed=1)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    samples = sum(len(o) for o in outs)
    audio = samples / hp['sample_rate']
    return {'workload': 'WaveRNN MOL (10 logistics), 64 utterances x 560 frames, hop 256 / 22050 Hz', 'rtf': dt / audio, 'wall_s': dt,
            'audio_s': audio, 'samples_per_s': samples / dt, 'us_per_step': dt / (hp['voc_target'] + 2 * hp['voc_overlap']) * 1e6,
            'finite': bool(all(np.isfinite(o).all() for o in outs))}


def wavernn_pass(precision, cpu_leg=True, kernel_table=False):
    """BASELINE config 4, one pass: 64 utterances x 560 frames, hop 256 @ 22.05 kHz, RAW 9-bit mu-law, batched folds
    (target 11 000, overlap 550).  Reported beside the headline metric, not as `value`."""
    from estts.wavernn import WaveRNN, fold_starts
    hp = configs.wavernn_hparams_22k()
    m = WaveRNN.from_hparams(hp, precision=precision)
    w = weights.init_wavernn(hp, seed=1234)
    m.load_state_dict(w)
    rng = np.random.default_rng(1234)
    mels = [rng.uniform(0, 1, size=(hp['num_mels'], 560)).astype(np.float32) for _ in range(64)]
    m.generate_batch(mels[:1], True, 1000, 100, True, seed=0)          # warm-up (library load, allocator)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    outs = m.generate_batch(mels, True, hp['voc_target'], hp['voc_overlap'], True, seed=1)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    samples = sum(len(o) for o in outs)
    audio = samples / hp['sample_rate']
    steps = hp['voc_target'] + 2 * hp['voc_overlap']
    n_seq = int(sum(len(fold_starts(560 * hp['hop_length'], hp['voc_target'], hp['voc_overlap'])) for _ in range(64)))
    res = {'workload': 'WaveRNN RAW 9-bit mu-law, 64 utterances x 560 frames, hop 256 / 22050 Hz (BASELINE config 4)',
           'rtf': dt / audio, 'wall_s': dt, 'audio_s': audio, 'samples_per_s': samples / dt, 'us_per_step': dt / steps * 1e6,
           'sequences': n_seq, 'precision': precision,
           'includes': 'conditioning network, 12100 sampling steps (50 per hipGraph replay), cross-fade / mu-law expansion / '
                       'closing fade in float64 on the device, device-to-host copy of the waveforms'}

    # one step's launches, traced and timed in isolation: the dominant layer against the matrix-core peak, the step against
    # the launch-chain floor and the weight stream
    cond = m._prepare(mels)
    seq_utt, seq_start = [], []
    for ui in range(64):
        st_ = fold_starts(560 * hp['hop_length'], hp['voc_target'], hp['voc_overlap'])
        seq_utt += [ui] * len(st_)
        seq_start += st_
    i32 = dict(dtype=torch.int32, device='cuda')
    su, ss = torch.tensor(seq_utt, **i32), torch.tensor(seq_start, **i32)
    st = m._alloc_state(n_seq, 8)
    prep_args = m._prep_args(cond, st, su, ss, n_seq)
    if m._use_layers():
        st['cond_args'] = m._cond_args(cond, st, su, ss, n_seq)
    sample_args = m._sample_args(st, n_seq, 1, None, None, None)
    m._one_step(st, prep_args, sample_args)
    m._one_step(st, prep_args, sample_args)
    ops.start_trace()
    m._one_step(st, prep_args, sample_args)
    trace = ops.stop_trace()
    torch.cuda.synchronize()
    launches_per_step = len(trace) + (1 if m._use_layers() else 2)        # + the sampler (and the conditioning launch)
    roof, _ = roofline_from_trace(trace, kernel_table, 0)
    step_flops = sum(info['flops'] for _, info, _ in trace)
    iso_us = None
    if roof is not None:
        key = lambda kind, info: (kind,) + tuple(sorted((k, v) for k, v in info.items() if k not in ('flops', 'bytes')))
        groups = {}
        for kind, info, fn in trace:
            groups.setdefault(key(kind, info), (kind, info, fn))
        times = dict(zip(groups, (t for _, _, t in time_kernels(list(groups.values()), reps=20))))
        iso_us = sum(times[key(kind, info)] for kind, info, _ in trace) * 1e6
    floor = launch_floor_us(launches_per_step)
    n_params = sum(int(np.prod(v.shape)) for k, v in w.items() if k.startswith(('I.', 'rnn', 'fc')))
    us = res['us_per_step']
    res['roofline'] = {
        'dominant_layer': roof,
        'step_algorithmic_tflops': step_flops / (us * 1e-6) / 1e12,
        'step_mfma_frac': step_flops / (us * 1e-6) / MFMA_BF16_PEAK,
        'step_mfma_issue_frac': (3.0 if precision == 'bf16x3' else 1.0) * step_flops / (us * 1e-6) / MFMA_BF16_PEAK,
        'weight_stream_gb_s': (4.0 if precision == 'bf16x3' else 2.0) * n_params / (us * 1e-6) / 1e9,
        'weight_stream_note': 'bytes of the step\'s weights (bf16 hi + lo) per step: every step re-reads them from L2 / the Infinity '
                              'Cache; HBM peak 8000 GB/s',
        'launches_per_step': launches_per_step,
        'launch_chain_floor_us': floor * launches_per_step,
        'launch_chain_floor_note': f'{launches_per_step} dependent launches of an empty kernel replayed from a hipGraph, '
                                   f'{floor:.2f} us each',
        'sum_of_isolated_layer_kernels_us': iso_us,
    }
    if cpu_leg:
        # CPU leg: the float32 NumPy restatement's step loop on the folds of 4 of the utterances for a bounded number of steps
        from oracle import wavernn_numpy as W
        w32 = {k: np.asarray(v, np.float32) for k, v in w.items() if not k.endswith('num_batches_tracked') and k != 'step'}
        sub = np.stack(mels[:4])
        mp = np.pad(sub, ((0, 0), (0, 0), (hp['voc_pad'], hp['voc_pad'])))
        up, aux = W.upsample_network(mp, w32, hp['voc_upsample_factors'], hp['voc_res_blocks'], hp['voc_pad'])
        up = np.concatenate([W.fold_with_overlap(up[i:i + 1], hp['voc_target'], hp['voc_overlap']) for i in range(4)])
        aux = np.concatenate([W.fold_with_overlap(aux[i:i + 1], hp['voc_target'], hp['voc_overlap']) for i in range(4)])
        B = up.shape[0]
        h1 = np.zeros((B, hp['voc_rnn_dims']), np.float32)
        h2 = np.zeros_like(h1)
        x = np.zeros((B, 1), np.float32)
        u = np.random.default_rng(5).uniform(0, 1, size=(4000, B))
        t0 = time.perf_counter()
        n_steps = 0
        while n_steps < 4000 and time.perf_counter() - t0 < 12.0:
            lg, h1, h2 = W.step_logits(x, up[:, n_steps], aux[:, n_steps], h1, h2, w32)
            c = W.sample_raw(lg, u[n_steps])
            x = (2 * c.astype(np.float32) / 511. - 1.)[:, None]
            n_steps += 1
        t_cpu = time.perf_counter() - t0
        res['cpu_baseline'] = {'value': B * n_steps / t_cpu, 'unit': 'samples/s', 'cores': min(os.cpu_count() or 1, 16), 'kind': 'port',
                               'sample': f'{n_steps} steps of the {B} folds of 4 of the 64 utterances, float32 NumPy restatement '
                                         f'(oracle/wavernn_numpy.py step_logits + sample_raw; BLAS threads as the host gives)',
                               'rtf_at_this_rate': samples / (B * n_steps / t_cpu) / audio}
    return res


NO_ROOFLINE = False


def csrc_hash():
    """SHA-1 over the kernel sources (the same digest tools/update_pmc_traffic.py stores beside the PMC figures)."""
    import hashlib
    csrc = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'expressive-speech-synthesis-research_amd', 'csrc')
    h = hashlib.sha1()
    for name in sorted(os.listdir(csrc)):
        if name.endswith(('.hip', '.h')):
            h.update(name.encode())
            with open(os.path.join(csrc, name), 'rb') as f:
                h.update(f.read())
    return h.hexdigest()


def roofline_from_trace(trace, kernel_table, rank):
    """Groups the traced launches by kernel and shape, times one representative of every group in isolation and prices
    the dominant group against its roofline."""
    if NO_ROOFLINE:
        return None, len(trace)
    groups = {}
    for kind, info, fn in trace:
        key = (kind,) + tuple(sorted((k, v) for k, v in info.items() if k not in ('flops', 'bytes')))
        g = groups.setdefault(key, {'kind': kind, 'info': info, 'n': 0, 'fn': fn})
        g['n'] += 1
    reps = time_kernels([(g['kind'], g['info'], g['fn']) for g in groups.values()])
    for g, (_, _, t) in zip(groups.values(), reps):
        g['t'] = t * g['n']
        del g['fn']
    total_iso = sum(g['t'] for g in groups.values())
    rows = trace
    dom = max(groups.values(), key=lambda g: g['t'])
    if kernel_table and rank == 0:
        for g in sorted(groups.values(), key=lambda g: -g['t']):
            shape = {k: v for k, v in g['info'].items() if k not in ('flops', 'bytes')}
            print(f"{g['kind']:13s} x{g['n']:2d} avg {g['t'] / g['n'] * 1e6:8.2f} us  share {g['t'] / total_iso:5.1%} "
                  f"{g['info']['flops'] / (g['t'] / g['n']) / 1e12:8.2f} TF/s {g['info']['bytes'] / (g['t'] / g['n']) / 1e9:8.1f} GB/s {shape}",
                  file=sys.stderr)
        print(f'launches {len(rows)}, sum of isolated kernel times {total_iso * 1e6:.1f} us', file=sys.stderr)
    avg = dom['t'] / dom['n']
    if dom['kind'] in ('gemm', 'attention', 'wgrad', 'attention_bwd'):
        roofline = {'bound': 'mfma', 'achieved': dom['info']['flops'] / avg / 1e12, 'peak': MFMA_BF16_PEAK / 1e12,
                    'unit': 'TFLOP/s'}
    else:
        roofline = {'bound': 'hbm', 'achieved': dom['info']['bytes'] / avg / 1e9, 'peak': HBM_PEAK / 1e9, 'unit': 'GB/s'}
    roofline['frac'] = roofline['achieved'] / roofline['peak']
    roofline['kernel'] = dom['kind'] + ' ' + json.dumps({k: v for k, v in dom['info'].items() if k not in ('flops', 'bytes')})
    # HBM bytes per launch from the PMC passes committed under profiles/ (collected with rocprofv3 --pmc on this same
    # command); null when the dominant kernel of this run has no such measurement
    roofline['traffic'] = None
    try:
        with open(os.path.join(os.path.dirname(os.path.abspath(__file__)), 'profiles', 'pmc_traffic.json')) as f:
            table = json.load(f)
        pmc = table.get(roofline['kernel'])
        if pmc and table.get('csrc_sha1') != csrc_hash():
            # the kernels changed after the counter passes: an old measurement must not pass for the new code
            roofline['traffic_source'] = 'stale: kernel sources changed since the PMC passes (tools/update_pmc_traffic.py)'
            pmc = None
        if pmc:
            roofline['traffic'] = pmc['hbm_bytes']
            roofline['traffic_source'] = ('committed PMC pass (rocprofv3 --pmc on this command, same kernel sources by digest), '
                                          'not measured in this run: ' + pmc['source'])
            roofline['algorithmic_bytes'] = dom['info']['bytes']
    except OSError:
        pass
    roofline['avg_us'] = avg * 1e6
    roofline['launches_per_step'] = dom['n']
    # finish_roofline() turns this into share_of_step once the step itself has been timed
    roofline['share_of_isolated_kernel_sum'] = dom['t'] / total_iso
    return roofline, len(rows)


def finish_roofline(roofline, ms_per_step):
    """share_of_step: the dominant kernel's launches (isolated average x count) over the measured step time."""
    if roofline is not None:
        roofline['share_of_step'] = roofline['avg_us'] * roofline['launches_per_step'] / (ms_per_step * 1e3)
    return roofline


def launches_of(fn):
    """Kernels the library launches during fn() (estts_launch_count: every launch of every entry point, the training
    kernels included; torch's own fills and copies are not the library's and are not counted)."""
    from estts import _lib
    lib = _lib.load()
    n0 = lib.estts_launch_count()
    out = fn()
    return lib.estts_launch_count() - n0, out


def timed_steps(step, steps, warmup, dist):
    """`warmup` untimed and `steps` timed calls of step(), bracketed by barrier + device synchronisation on both sides;
    returns (seconds of the slowest rank, last result)."""
    out = None
    for _ in range(warmup):
        out = step()
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        out = step()
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    if dist is not None:
        dev = 'cuda' if dist.get_backend() == 'nccl' else 'cpu'
        t = torch.tensor([dt], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    return dt, out


def exchange_report(model, dist, dp_ms, steps, warmup, step):
    """Data-parallel runs: what the gradient exchange costs.  `allreduce_ms` is the bucketed all-reduce of the whole flat
    gradient buffer on its own (communication stream, HIP events), `exposed_ms` the difference between the step with and
    without the exchange (the same trainer, `exchange = False`: ranks then drift apart, which is fine for a timing),
    `hidden_ms` the rest, i.e. what ran beside the backward pass."""
    from estts import parallel
    world = dist.get_world_size()
    backend = dist.get_backend()
    grad = model.store.grad
    comm = torch.cuda.Stream()
    best = float('inf')
    for i in range(4):
        torch.cuda.synchronize()
        dist.barrier()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        parallel.allreduce_buckets(grad, model.bucket_elems, comm_stream=comm if backend == 'nccl' else None)
        e1.record()
        e1.synchronize()
        if i:
            best = min(best, e0.elapsed_time(e1))
    model.exchange = False
    dt, _ = timed_steps(step, steps, warmup, dist)
    model.exchange = True
    solo_ms = dt * 1e3 / steps
    exposed = max(dp_ms - solo_ms, 0.0)
    seen = [None] * world
    dist.all_gather_object(seen, (dist.get_rank(), torch.cuda.current_device(), torch.cuda.get_device_name()))
    nbytes = grad.numel() * 4
    return {'backend': 'rccl' if backend == 'nccl' else backend, 'ranks_seen': len({r for r, _, _ in seen}),
            'devices': sorted({f'{d}:{n}' for _, d, n in seen}), 'gradient_bytes': nbytes,
            'buckets': len(parallel.bucket_ranges(grad.numel(), model.bucket_elems)),
            'overlapped_with_backward': backend == 'nccl',
            'allreduce_ms': best, 'allreduce_bus_gb_s': 2.0 * (world - 1) / world * nbytes / (best * 1e-3) / 1e9,
            'step_ms_without_exchange': solo_ms, 'exposed_ms': exposed, 'hidden_ms': max(best - exposed, 0.0)}


def train_measure(args, rank, world, dist, B, steps, warmup, cpu_leg):
    """BASELINE config 5: ForwardTransformer training step (forward with dropout / batch statistics, masked-MAE losses,
    backward, bucketed gradient all-reduce over RCCL beside the backward pass, Adam), B utterances per GPU (weak scaling,
    global batch world * B).  Returns the bench line (rank 0; None elsewhere)."""
    from estts.forward_training import ForwardTransformerTrainer
    N, T = args.tokens, args.frames
    cfg = configs.forward_config()
    w = weights.init_forward_transformer(cfg, seed=1234)                 # same seed: identical replicas on every rank
    tokens, durs = synthetic_utterance(cfg, B, N, T, seed=1234 + rank)
    rng = np.random.default_rng(99 + rank)
    mel = np.clip(rng.normal(0, 1.5, size=(B, T, cfg['mel_channels'])), -4, 4).astype(np.float32)
    model = ForwardTransformerTrainer(w, cfg, precision=args.precision, seed=1234)
    if args.backward_precision:
        model.backward_precision = args.backward_precision
    if os.environ.get('ESTTS_BENCH_FOLD_SUMS') == '0':      # same-box comparison with the two-launch BatchNorm sums
        model.fold_sums = False
    model.set_constants(learning_rate=1e-4, decoder_prenet_dropout=0.0, drop_n_heads=0)
    tok_d = torch.from_numpy(tokens).cuda()
    mel_d = torch.from_numpy(mel).cuda()

    def step():
        return model.train_step(tok_d, mel_d, durs)

    out = step()
    torch.cuda.synchronize()
    loss0 = float(out['loss'])

    cpu = None
    if rank == 0 and world == 1 and cpu_leg:
        from oracle import tts_train_torch as OTT
        torch.set_num_threads(min(os.cpu_count() or 1, 16))
        bs = min(B, 2)
        t0 = time.perf_counter()
        OTT.forward_train_step(w, cfg, tokens[:bs], mel[:bs], durs[:bs], dtype=torch.float32)
        t_cpu = time.perf_counter() - t0
        cpu = {'value': bs * T / t_cpu, 'unit': 'mel-frames/s', 'cores': torch.get_num_threads(), 'kind': 'port',
               'sample': f'1 training step (forward + autograd backward, no optimizer) on {bs} of the {B} utterances '
                         f'(N={N}, T={T}), PyTorch CPU float32 restatement (oracle/tts_train_torch.py)'}

    # per-kernel timing (every rank goes through the same collectives, so every rank traces)
    ops.start_trace()
    launches, traced = launches_of(step)
    trace = ops.stop_trace()
    torch.cuda.synchronize()
    roofline, _ = roofline_from_trace(trace, args.kernel_table, rank)
    del traced, trace

    dt, out = timed_steps(step, steps, warmup, dist)
    ms = dt * 1e3 / steps
    exchange = exchange_report(model, dist, ms, max(steps // 2, 3), 1, step) if dist is not None else None
    if rank != 0:
        return None
    flops = 3.0 * B * algorithmic_flops(N, T)           # forward + data gradient + weight gradient
    line = {
        'metric': 'mel-frames/sec (ForwardTransformer training step)', 'value': world * B * T / (dt / steps),
        'unit': 'mel-frames/s', 'n_gpus': world, 'steps': steps, 'warmup': warmup, 'ms_per_step': ms,
        'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None, 'dtype': 'bf16', 'data': 'synthetic',
        'config': {'workload': f'ForwardTransformer train_step: forward (dropout 0.1, BatchNorm batch statistics) + backward + '
                               f'gradient all-reduce + Adam, batch={B} per GPU, N={N}, T={T} (BASELINE config 5); '
                               f'{args.precision} MFMA operands (backward products: {args.backward_precision or args.precision}), fp32 accumulate, fp32 master weights',
                   'global_batch': world * B, 'parallelism': f'dp{world}', 'hip_graph': False,
                   'launches_per_step': launches},
        'steps_per_s': steps / dt, 'algorithmic_tflops': world * flops / (dt / steps) / 1e12,
        'loss_first_step': loss0, 'loss_last_step': float(out['loss']),
        'roofline': finish_roofline(roofline, ms), 'cpu_baseline': cpu,
    }
    if exchange is not None:
        line['exchange'] = exchange
    return line


def train_bench(args, rank, world, dist):
    line = train_measure(args, rank, world, dist, args.batch, args.steps, args.warmup, not args.no_cpu_baseline)
    if rank == 0:
        print(json.dumps(line))
    if dist is not None:
        dist.destroy_process_group()


def exchange_bench(args, rank, world, dist):
    """The exchange step of data-parallel training on its own: the bucketed sum of one flat float32 gradient buffer over all
    ranks (parallel.allreduce_buckets; 19.18 M elements = the ForwardTransformer's 76.7 MB unless --elems says otherwise).
    With a GPU the buffer lives in HBM and the transport is RCCL (or gloo when ESTTS_DIST_BACKEND=gloo); without one it is
    host memory over gloo, which is how the CPU tests rehearse the N-rank launch."""
    from estts import parallel
    on_gpu = torch.cuda.is_available()
    n = args.elems
    flat = torch.full((n,), float(rank + 1), dtype=torch.float32, device='cuda' if on_gpu else 'cpu')
    bucket = int(25 * (1 << 20) // 4)
    comm = torch.cuda.Stream() if on_gpu and dist is not None and dist.get_backend() == 'nccl' else None

    def sync():
        if on_gpu:
            torch.cuda.synchronize()

    def step():
        parallel.allreduce_buckets(flat, bucket, comm_stream=comm)

    step()
    sync()
    expect = world * (world + 1) / 2.0
    ok = bool((flat == expect).all())
    flat.fill_(1.0 / world)           # stays 1 / world * world ... bounded whatever the step count
    for _ in range(args.warmup):
        step()
        flat.mul_(1.0 / world)
    sync()
    parallel.barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    sync()
    parallel.barrier()
    dt = parallel.max_over_ranks(time.perf_counter() - t0, device=flat.device if on_gpu and comm is not None else torch.device('cpu'))
    if rank == 0:
        nbytes = 4.0 * n
        print(json.dumps({
            'metric': 'gradient all-reduce bus bandwidth', 'value': 2.0 * (world - 1) / max(world, 1) * nbytes * args.steps / dt / 1e9,
            'unit': 'GB/s', 'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup, 'ms_per_step': dt * 1e3 / args.steps,
            'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None, 'dtype': 'f32', 'data': 'synthetic',
            'config': {'workload': f'bucketed all-reduce (sum) of one flat float32 gradient buffer of {n} elements over {world} ranks, '
                                   f'{len(parallel.bucket_ranges(n, bucket))} bucket(s) of <= 25 MB',
                       'backend': ('rccl' if dist.get_backend() == 'nccl' else dist.get_backend()) if dist is not None else None,
                       'memory': 'hbm' if on_gpu else 'host', 'parallelism': f'dp{world}'},
            'sum_correct': ok, 'roofline': None, 'cpu_baseline': None}))
    if dist is not None:
        dist.destroy_process_group()
    if not ok:
        raise SystemExit('all-reduce returned a wrong sum')


def train_ar_bench(args, rank, world, dist):
    """The TTS half of train_models_step (traning_steps.py:22-56) plus one MINE head: AutoregressiveTransformer training step,
    style_text system (reference encoder + style tokens), reduction factor 1, `--batch` utterances per GPU (the shipped
    tts_batch_size is 8), N tokens, T frames."""
    from estts.autoregressive_training import AutoregressiveTransformerTrainer
    from estts.mine import MINE
    from estts.training_steps import train_models_step
    B, N, T = args.batch, args.tokens, args.frames
    cfg = configs.autoregressive_config(system_type='style_text')
    w = weights.init_autoregressive_transformer(cfg, seed=1234)
    rng = np.random.default_rng(7 + rank)
    tokens = rng.integers(1, cfg['vocab_size'], size=(B, N)).astype(np.int32)
    mel = np.clip(rng.normal(0, 1.5, size=(B, T + 1, cfg['mel_channels'])), -3.9, 3.9).astype(np.float32)
    mel[:, 0], mel[:, -1] = cfg['mel_start_value'], cfg['mel_end_value']
    stop = np.ones((B, T + 1), np.int32)
    stop[:, -1] = 2
    model = AutoregressiveTransformerTrainer(w, cfg, precision=args.precision, seed=1234)
    model._compile(stop_scaling=8.0, optimizer={'learning_rate': 1e-4})
    model.set_constants(reduction_factor=1, decoder_prenet_dropout=0.0, drop_n_heads=0)
    heads = [MINE(dense_hidden_units=[512, 64], divergence_type='KL', pair_type='style_text', seed=1)]
    holder = {'use_mine': True, 'mine_sep_call': False, 'mi_loss': 0.0, 'exp_terms': np.zeros((3, 2), np.float32),
              'weight_factor': 0.1, 'smoothing_factor': 1.0}
    mel_d, tok_d, stop_d = torch.from_numpy(mel).cuda(), torch.from_numpy(tokens).cuda(), torch.from_numpy(stop).cuda()

    def step():
        return train_models_step(mel_d, tok_d, stop_d, None, 0, 0, 0, 0, model, heads, holder, True, True, True, False)

    out = step()
    torch.cuda.synchronize()
    loss0 = float(out['loss'])
    roofline, launches = None, None
    if rank == 0 or world > 1:
        ops.start_trace()
        traced = step()
        trace = ops.stop_trace()
        torch.cuda.synchronize()
        roofline, launches = roofline_from_trace(trace, args.kernel_table, rank)
        del traced, trace
    for _ in range(args.warmup):
        step()
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        out = step()
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    if dist is not None:
        t = torch.tensor([dt], device='cuda', dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    if rank == 0:
        print(json.dumps({
            'metric': 'mel-frames/sec (AutoregressiveTransformer training step)', 'value': world * B * T / (dt / args.steps),
            'unit': 'mel-frames/s', 'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup, 'ms_per_step': dt * 1e3 / args.steps,
            'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None, 'dtype': 'bf16', 'data': 'synthetic',
            'config': {'workload': f'train_models_step: AutoregressiveTransformer (style_text, r=1) forward + three losses + backward + '
                                   f'Adam, then one MINE head (KL, style_text) ascent step; batch={B} per GPU, N={N}, T={T}; '
                                   f'{args.precision} MFMA operands, fp32 accumulate, fp32 master weights',
                       'global_batch': world * B, 'parallelism': f'dp{world}', 'hip_graph': False, 'launches_per_step': launches},
            'steps_per_s': args.steps / dt, 'loss_first_step': loss0, 'loss_last_step': float(out['loss']),
            'roofline': roofline, 'cpu_baseline': None}))
    if dist is not None:
        dist.destroy_process_group()


def train_wavernn_bench(args, rank, world, dist):
    """WaveRNN training step (train_wavernn.py): `--batch` windows of 5 hops per GPU, RAW 9-bit, shipped hparams
    (hop 200, 16 kHz): teacher-forced forward, cross-entropy, BPTT, all-reduce, Adam."""
    from estts.wavernn_training import WaveRNNTrainer
    hp = configs.wavernn_hparams(voc_mode='RAW')
    B, hops = args.batch, 5
    L = hops * hp['hop_length']
    model = WaveRNNTrainer.from_hparams(hp, precision=args.precision)
    if os.environ.get('ESTTS_BENCH_WAVERNN_CHUNK'):      # steps between hand-overs of the two overlapped recurrences (measurements)
        model.CHUNK = int(os.environ['ESTTS_BENCH_WAVERNN_CHUNK'])
    w = weights.init_wavernn(hp, seed=1234)
    model.load_state_dict({k: torch.from_numpy(np.asarray(v)) for k, v in w.items()})
    rng = np.random.default_rng(7 + rank)
    n_cls = 2 ** hp['bits']
    labels = rng.integers(0, n_cls, size=(B, L + 1))
    x = (2 * labels[:, :L] / (n_cls - 1.) - 1.).astype(np.float32)
    y = labels[:, 1:].astype(np.int64)
    mels = rng.uniform(0, 1, size=(B, hp['num_mels'], hops + 2 * hp['voc_pad'])).astype(np.float32)
    xd, yd, md = torch.from_numpy(x).cuda(), torch.from_numpy(y).cuda(), torch.from_numpy(mels).cuda()

    def step():
        return model.train_step(xd, yd, md, lr=1e-4)

    ops.start_trace()
    loss0 = float(step())                 # eager pass: every launch is traced once (the loops are captured afterwards)
    trace = ops.stop_trace()
    torch.cuda.synchronize()
    roofline, launches = roofline_from_trace(trace, args.kernel_table, rank)
    del trace

    cpu = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        from oracle import wavernn_train_torch as OWT
        torch.set_num_threads(min(os.cpu_count() or 1, 16))
        bs = min(B, 4)
        t0 = time.perf_counter()
        OWT.train_step(w, hp, x[:bs], y[:bs], mels[:bs], dtype=torch.float32)
        t_cpu = time.perf_counter() - t0
        cpu = {'value': bs * L / t_cpu, 'unit': 'samples/s', 'cores': torch.get_num_threads(), 'kind': 'port',
               'sample': f'1 training step (forward + autograd backward, no optimizer) on {bs} of the {B} windows of {L} samples, '
                         f'PyTorch CPU float32 restatement (oracle/wavernn_train_torch.py)'}

    for _ in range(max(args.warmup, 1)):
        step()
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        loss = step()
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    if dist is not None:
        t = torch.tensor([dt], device='cuda', dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    if rank == 0:
        macs = 4_071_936                       # per sample per sequence, RAW-512 (fatchord_version.py:109-114 layer sizes)
        line = {
            'metric': 'samples/sec (WaveRNN training step)', 'value': world * B * L / (dt / args.steps), 'unit': 'samples/s',
            'n_gpus': world, 'steps': args.steps, 'warmup': max(args.warmup, 1), 'ms_per_step': dt * 1e3 / args.steps,
            'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None, 'dtype': 'bf16', 'data': 'synthetic',
            'config': {'workload': f'WaveRNN train step (RAW 9-bit, hop 200): teacher-forced forward over {L}-sample windows, '
                                   f'cross-entropy, BPTT through both GRUs, gradient all-reduce, Adam; batch={B} per GPU; '
                                   f'{args.precision} MFMA operands, fp32 accumulate, fp32 master weights',
                       'global_batch': world * B, 'parallelism': f'dp{world}', 'hip_graph': True, 'launches_per_step': launches},
            'steps_per_s': args.steps / dt, 'algorithmic_tflops': world * 3 * 2.0 * macs * B * L / (dt / args.steps) / 1e12,
            'loss_first_step': loss0, 'loss_last_step': float(loss), 'roofline': roofline, 'cpu_baseline': cpu,
        }
        print(json.dumps(line))
    if dist is not None:
        dist.destroy_process_group()


def predict_bench(args, rank, world, dist):
    """AutoregressiveTransformer.predict (the call synthesize_sentences_speaker.py makes): one utterance, N tokens,
    `--frames` mel frames (the stop class is disabled so that the length is fixed), incremental decoder vs the reference's
    re-decode-everything loop.  `--steps` = utterances timed."""
    from estts.autoregressive_transformer import AutoregressiveTransformer
    cfg = configs.autoregressive_config(system_type='style_text')
    w = weights.init_autoregressive_transformer(cfg, seed=1234)
    w['decoder_postnet.stop_linear.bias'] = np.array([5.0, 0.0, -5.0], np.float32)
    rng = np.random.default_rng(1234 + rank)
    N, T = args.tokens, args.frames
    tokens = rng.integers(1, cfg['vocab_size'], size=N)
    ref_mel = np.clip(rng.normal(0, 1.5, size=(400, 80)), -4, 4).astype(np.float32)
    model = AutoregressiveTransformer(w, cfg, precision=args.precision)
    res = {}
    for r in (10, 1):
        model.set_constants(reduction_factor=r)
        for cached in (True, False):
            if not cached and r == 1 and T > 300:
                continue                                   # the quadratic loop at r = 1 takes tens of seconds
            model.predict(tokens, ref_mel, None, max_length=min(T, 4 * r), encode=False, verbose=False, kv_cache=cached)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(args.steps):
                out = model.predict(tokens, ref_mel, None, max_length=T - r, encode=False, verbose=False, kv_cache=cached)
            torch.cuda.synchronize()
            dt = (time.perf_counter() - t0) / args.steps
            res[f'r{r}_{"incremental" if cached else "full_redecode"}'] = {'frames': int(out['mel'].shape[0]), 's_per_utterance': dt,
                                                                           'mel_frames_per_s': out['mel'].shape[0] / dt}
    # launches and weight bytes of one decoder step (r = 10): the step is a chain of one-row products, so what it can be priced
    # against is the stream of its weights (bf16 hi + lo = 4 bytes per weight: decoder blocks, prenet, projections, post-net)
    model.set_constants(reduction_factor=10)
    n_launch, out = launches_of(lambda: model.predict(tokens, ref_mel, None, max_length=T - 10, encode=False, verbose=False))
    steps_r10 = out['mel'].shape[0] // 10
    n_enc, _ = launches_of(lambda: model.predict(tokens, ref_mel, None, max_length=10, encode=False, verbose=False))   # encoder + 2 steps
    per_step = (n_launch - n_enc) / max(steps_r10 - 2, 1)
    wbytes = 4.0 * sum(int(np.prod(np.shape(v))) for k, v in w.items()
                       if k.startswith(('decoder.', 'decoder_prenet.', 'decoder_postnet.')) and k.endswith('.kernel'))
    wbytes += 4.0 * cfg['decoder_model_dimension'] * 10 * cfg['mel_channels']          # FinalProj restricted to r * 80 columns
    cpu = None
    if rank == 0 and not args.no_cpu_baseline:
        # the reference's own loop (the whole decoder again at every step) in float32 NumPy, bounded to the first 100 frames:
        # its cost per frame grows with the length, so the rate of a full 600-frame utterance is lower still
        from oracle import tts_numpy as O
        t0 = time.perf_counter()
        o = O.autoregressive_predict(w, cfg, tokens, ref_mel, None, r=10, max_length=90, dtype=np.float32)
        t_cpu = time.perf_counter() - t0
        cpu = {'value': o['mel'].shape[0] / t_cpu, 'unit': 'mel-frames/s', 'cores': min(os.cpu_count() or 1, 16), 'kind': 'port',
               'sample': f"the first {o['mel'].shape[0]} frames of the same utterance (N={N}, r=10), float32 NumPy restatement of the "
                         f"reference's re-decode-everything loop (oracle/tts_numpy.py autoregressive_predict; BLAS threads as the host gives)"}
    if rank == 0:
        best = res['r10_incremental']
        us_step = best['s_per_utterance'] * 1e6 / steps_r10
        roofline = {'bound': 'hbm', 'achieved': wbytes / (us_step * 1e-6) / 1e9, 'peak': HBM_PEAK / 1e9, 'unit': 'GB/s',
                    'kernel': 'decoder step (chain of one-row products; bytes = the step\'s weights, bf16 hi + lo)', 'traffic': None,
                    'algorithmic_bytes': wbytes, 'launches_per_step': per_step, 'us_per_step': us_step,
                    'note': 'latency-bound: every launch of the chain depends on the one before it'}
        roofline['frac'] = roofline['achieved'] / roofline['peak']
        print(json.dumps({'metric': 'mel-frames/sec (AutoregressiveTransformer.predict, one utterance)', 'value': best['mel_frames_per_s'],
                          'unit': 'mel-frames/s', 'n_gpus': 1, 'steps': args.steps, 'warmup': 1, 'ms_per_step': best['s_per_utterance'] * 1e3,
                          'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None, 'dtype': 'bf16', 'data': 'synthetic',
                          'config': {'workload': f'AR predict, style_text, N={N}, {T} frames, reduction factor 10 (value) and 1; '
                                                 f'{args.precision} MFMA operands'},
                          'variants': res, 'roofline': roofline, 'cpu_baseline': cpu}))
    if dist is not None:
        dist.destroy_process_group()


def secondary_pass(argv, precision):
    """Runs this script for another workload in a child process and returns the essentials of its JSON line."""
    import subprocess
    cmd = [sys.executable, os.path.abspath(__file__)] + argv + ['--precision', precision, '--no-cpu-baseline', '--no-wavernn']
    try:
        r = subprocess.run(cmd, capture_output=True, text=True, timeout=240)
        d = json.loads([l for l in r.stdout.splitlines() if l.startswith('{')][-1])
    except Exception as e:            # the headline line must survive a failing extra
        return {'error': f'{type(e).__name__}: {e}'}
    keep = ('metric', 'value', 'unit', 'ms_per_step', 'steps', 'warmup', 'algorithmic_tflops', 'steps_per_s', 'roofline')
    out = {k: d[k] for k in keep if k in d}
    out['workload'] = d['config']['workload']
    out['launches_per_step'] = d['config'].get('launches_per_step')
    return out


def config1_cpu_seconds(args):
    """BASELINE config 1: the DSP chain of gst_tacotron's eval path (one LJSpeech-length utterance -> spectrogram / 80-bin mel
    -> Griffin-Lim, 60 iterations -> end-point detection) on the host CPU.  Plumbing with no GPU part: only its seconds are
    reported, from the float64 NumPy restatement that the tests check (oracle/audio_numpy.py)."""
    from oracle import audio_numpy as A
    hp = A.GST_HPARAMS
    sr = hp['sample_rate']
    rng = np.random.default_rng(0)
    t = np.arange(int(6.5 * sr)) / sr                        # an LJSpeech sentence is about 6.5 s
    wav = 0.3 * np.sin(2 * np.pi * 180 * t) * (1 + 0.5 * np.sin(2 * np.pi * 3 * t)) + 0.02 * rng.normal(size=t.shape)
    times = []
    for _ in range(max(1, min(args.steps, 5))):
        t0 = time.perf_counter()
        spec = A.gst_spectrogram(wav, hp)
        mel = A.gst_melspectrogram(wav, hp)
        out = A.gst_inv_spectrogram(spec, hp, rng=np.random.default_rng(1))
        out = out[:A.find_endpoint(out, sr)]
        times.append(time.perf_counter() - t0)
    sec = float(np.median(times))
    print(json.dumps({
        'metric': 'seconds per utterance (gst_tacotron eval DSP chain on the CPU)', 'value': sec, 'unit': 's', 'n_gpus': 0,
        'steps': len(times), 'warmup': 0, 'ms_per_step': sec * 1e3, 'higher_is_better': False, 'scaling': 'weak', 'vs_baseline': None,
        'dtype': 'f64', 'data': 'synthetic',
        'config': {'workload': f'BASELINE config 1: {len(wav) / sr:.1f} s utterance at {sr} Hz -> linear + {mel.shape[0]}-bin mel '
                               f'spectrogram ({mel.shape[1]} frames) -> Griffin-Lim {hp["griffin_lim_iters"]} iterations -> end point; '
                               'the Tacotron network itself (TF 1.14 tf.contrib) is out of scope'},
        'roofline': None,
        'cpu_baseline': {'value': sec, 'unit': 's', 'cores': 1, 'kind': 'port',
                         'sample': 'the whole workload: one utterance, float64 NumPy (oracle/audio_numpy.py)'}}))


def in_flight_pass(w, cfg, tok_d, dur_d, T, want_attn, args, mel_ref):
    """Serving shape of the batch-1 workload: k independent requests in flight, each a hipGraph of its own (own model
    instance, own workspace) replayed on a HIP stream of its own.  A batch-1 forward is a chain of ~60 latency-bound launches
    that leave most of the chip idle, so requests overlap.  Reported beside the headline value, which stays one request at
    a time.  Every replica's mel must equal the one-at-a-time result bit for bit (same kernels, same inputs)."""
    res = {'note': 'k independent batch-1 forwards in flight (one hipGraph and HIP stream each); the headline value is k = 1',
           'points': []}
    B = tok_d.shape[0]
    replicas = []
    for k in (2, 4):
        while len(replicas) < k:
            m = ForwardTransformer(w, cfg, precision=args.precision)
            st = torch.cuda.Stream()
            st.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(st):
                m.call(tok_d, target_durations=dur_d, max_frames=T, return_attention=want_attn)
            torch.cuda.current_stream().wait_stream(st)
            torch.cuda.synchronize()
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g, stream=st):
                o = m.call(tok_d, target_durations=dur_d, max_frames=T, return_attention=want_attn)
            replicas.append((m, st, g, o))
        torch.cuda.synchronize()
        rounds = max(args.steps // k, 10)
        for timed in (False, True):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(rounds if timed else max(args.warmup // k, 3)):
                for m, st, g, o in replicas[:k]:
                    with torch.cuda.stream(st):
                        g.replay()
            torch.cuda.synchronize()
            dt = time.perf_counter() - t0
        same = all(bool(torch.equal(o['mel'], mel_ref)) for _, _, _, o in replicas[:k])
        res['points'].append({'in_flight': k, 'value': k * rounds * B * T / dt, 'unit': 'mel-frames/s',
                              'ms_per_forward': dt * 1e3 / (k * rounds), 'forwards_timed': k * rounds,
                              'identical_to_one_at_a_time': same})
    return res


def launch_ranks(n):
    """`python bench.py --gpus N` without a launcher's environment: starts N ranks of this same command as a child job
    (python -m torch.distributed.run --standalone: the agent's own TCP store on a port it binds itself, one process per GPU,
    RANK / LOCAL_RANK / WORLD_SIZE in the ranks' environment), relays rank 0's JSON line and fails unless that line reports
    n_gpus == N.  A child process, never an exec: this process has not touched the GPU and goes on waiting."""
    import subprocess
    cmd = [sys.executable, '-u', '-m', 'torch.distributed.run', '--standalone', '--nnodes=1', '--nproc-per-node', str(n),
           '--local-addr', '127.0.0.1', os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault('OMP_NUM_THREADS', str(max(1, min(16, (os.cpu_count() or 1)) // n)))
    r = subprocess.run(cmd, stdout=subprocess.PIPE, text=True, env=env)
    lines = [l for l in r.stdout.splitlines() if l.startswith('{')]
    for l in r.stdout.splitlines():
        if not l.startswith('{'):
            print(l, file=sys.stderr)
    if r.returncode != 0 or not lines:
        print(f'bench.py: the {n}-rank job failed (exit {r.returncode}, {len(lines)} JSON line(s))', file=sys.stderr)
        raise SystemExit(r.returncode or 1)
    line = json.loads(lines[-1])
    print(lines[-1], flush=True)
    if line.get('n_gpus') != n:
        print(f"bench.py: asked for {n} ranks, the line reports n_gpus = {line.get('n_gpus')}", file=sys.stderr)
        raise SystemExit(1)


def serving_pass(w, cfg, tokens, args):
    """The same lever through the product entry point: ForwardTransformer.predict_many (what scripts/synthesize_sentences.py
    --text-file calls) on 64 copies of the benchmark sentence, durations PREDICTED (the duration predictor's bias is raised so
    that they sum to about the benchmark's 600 frames; one small device-to-host read per sentence decides the frame count),
    lanes replaying the graphs of the shapes they have met.  Includes the copy of every result out of the graph buffers."""
    w2 = dict(w)
    w2['dur_pred.linear.bias'] = np.asarray(w['dur_pred.linear.bias'], np.float32) + 2.2
    model = ForwardTransformer(w2, cfg, precision=args.precision)
    sentences = [tokens[0]] * 64
    points = []
    for k in (1, 2, 4):
        model.predict_many(sentences[:3 * k], encode=False, in_flight=k)          # first visit, capture, replay
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        outs = model.predict_many(sentences, encode=False, in_flight=k)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        frames = sum(int(o['mel'].shape[0]) for o in outs)
        points.append({'in_flight': k, 'sentences': len(outs), 'frames_per_sentence': frames / len(outs),
                       'ms_per_sentence': dt * 1e3 / len(outs), 'value': frames / dt, 'unit': 'mel-frames/s'})
    return {'note': 'ForwardTransformer.predict_many, predicted durations, results copied out of the lanes\' graph buffers',
            'points': points}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=200)
    ap.add_argument('--warmup', type=int, default=20)
    ap.add_argument('--batch', type=int, default=1)
    ap.add_argument('--tokens', type=int, default=100)
    ap.add_argument('--frames', type=int, default=600)
    ap.add_argument('--precision', default='bf16x3', choices=['bf16', 'bf16x3'])
    ap.add_argument('--backward-precision', default=None, choices=['bf16', 'bf16x3'],
                    help='training workloads: operands of the backward products (default: the same as --precision)')
    ap.add_argument('--no-graph', action='store_true')
    ap.add_argument('--no-attention-weights', action='store_true',
                    help='skip materialising the attention maps the reference returns')
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--no-roofline', action='store_true',
                    help='skip the per-kernel timing pass (clean kernel statistics under rocprofv3); roofline is then null')
    ap.add_argument('--kernel-table', action='store_true', help='print the per-kernel timing table to stderr')
    ap.add_argument('--train-batch', type=int, default=32,
                    help='utterances per GPU of the data-parallel training block that a default run with --gpus N > 1 adds '
                         '(BASELINE config 5: 32 per GPU, global 256 at N = 8); 0 skips the block')
    ap.add_argument('--elems', type=int, default=19_178_529, help='--workload exchange: float32 elements of the gradient buffer')
    ap.add_argument('--workload', default='forward',
                    choices=['forward', 'gst', 'train', 'train_wavernn', 'train_ar', 'predict', 'config1', 'exchange'],
                    help="forward: ForwardTransformer (BASELINE config 2); gst: GST reference encoder + teacher-forced "
                         "AutoregressiveTransformer (config 3; use --batch 32); train: ForwardTransformer training step, "
                         "forward + backward + gradient all-reduce + Adam (config 5; use --batch 32)")
    ap.add_argument('--no-in-flight', action='store_true',
                    help='skip the block that runs several independent batch-1 forwards at once on HIP streams of their own')
    ap.add_argument('--no-wavernn', action='store_true',
                    help='skip the secondary passes of the default run: WaveRNN (config 4, reported under "wavernn") and the '
                         'child runs of configs 3 and 5 on one GPU ("config3", "config5_one_gpu")')
    args = ap.parse_args()
    global NO_ROOFLINE
    NO_ROOFLINE = args.no_roofline

    if args.workload == 'config1':
        return config1_cpu_seconds(args)
    if args.gpus > 1 and 'WORLD_SIZE' not in os.environ:
        return launch_ranks(args.gpus)            # before anything here has touched the GPU
    if os.environ.get('ESTTS_WATCHDOG_S'):      # a rank that stalls dumps every thread's stack and exits instead of hanging
        import faulthandler
        import threading
        limit = int(os.environ['ESTTS_WATCHDOG_S'])
        faulthandler.dump_traceback_later(limit, exit=True, file=sys.stderr)

        def native_threads():
            # shortly before the dump: what every native thread of this rank is doing (name, scheduler state, kernel wait
            # channel) and which of the pending collectives have completed
            time.sleep(max(limit - 5, 1))
            from estts import parallel as par
            rows = []
            for tid in sorted(os.listdir('/proc/self/task'), key=int):
                try:
                    comm = open(f'/proc/self/task/{tid}/comm').read().strip()
                    state = open(f'/proc/self/task/{tid}/stat').read().rsplit(')', 1)[1].split()[0]
                    wchan = open(f'/proc/self/task/{tid}/wchan').read().strip()
                    rows.append(f'{tid}:{comm}:{state}:{wchan}')
                except OSError:
                    pass
            ov = getattr(par.GradientOverlap, 'last', None)
            done = None if ov is None else [bool(w.is_completed()) for w in ov.works]
            print(f"[watchdog rank {os.environ.get('RANK')}] collectives issued {None if ov is None else ov.calls} completed {done}\n"
                  f"[watchdog rank {os.environ.get('RANK')}] threads " + ' '.join(rows), file=sys.stderr, flush=True)
        threading.Thread(target=native_threads, daemon=True).start()
    from estts import parallel
    rank, world, local_rank = parallel.env_ranks()
    if world != args.gpus:
        raise SystemExit(f'bench.py: --gpus {args.gpus} but the launcher started {world} rank(s) (WORLD_SIZE)')
    if args.workload == 'exchange':
        dist = None
        parallel.init_distributed(os.environ.get('ESTTS_DIST_BACKEND') or None)
        if world > 1:
            import torch.distributed as dist
        return exchange_bench(args, rank, world, dist)
    # (ESTTS_DIST_BACKEND=gloo with more ranks than GPUs rehearses the multi-rank paths on a one-GPU box; with gloo the
    # ForwardTransformer trainer exchanges its gradients after the backward pass instead of overlapped with it, see
    # parallel.GradientOverlap.  The driver's runs have one GPU per rank and use RCCL.)
    backend = os.environ.get('ESTTS_DIST_BACKEND', 'nccl')
    torch.cuda.set_device(local_rank % torch.cuda.device_count() if backend != 'nccl' else local_rank)
    parallel.init_distributed(backend)
    dist = None
    if world > 1:
        import torch.distributed as dist

    B, N, T = args.batch, args.tokens, args.frames
    want_attn = not args.no_attention_weights
    if args.workload == 'train':
        return train_bench(args, rank, world, dist)
    if args.workload == 'train_wavernn':
        return train_wavernn_bench(args, rank, world, dist)
    if args.workload == 'train_ar':
        return train_ar_bench(args, rank, world, dist)
    if args.workload == 'predict':
        return predict_bench(args, rank, world, dist)
    if args.workload == 'forward':
        cfg = configs.forward_config()
        w = weights.init_forward_transformer(cfg, seed=1234)
        tokens, durs = synthetic_utterance(cfg, B, N, T, seed=1234 + rank)
        model = ForwardTransformer(w, cfg, precision=args.precision)
        tok_d = torch.from_numpy(tokens).cuda()
        dur_d = torch.from_numpy(durs.astype(np.float32)).cuda()

        def step():
            return model.call(tok_d, target_durations=dur_d, max_frames=T, return_attention=want_attn)

        def cpu_step():
            from oracle import tts_torch as OT
            return OT.forward_transformer_call(w, cfg, tokens, durs)['mel']
        flops_per_step = B * algorithmic_flops(N, T)
        label = (f'ForwardTransformer forward (non-AR, 80-mel), batch={B}, N={N} tokens, T={T} frames per GPU '
                 f'(BASELINE config 2)')
    else:
        from estts.autoregressive_transformer import AutoregressiveTransformer
        cfg = configs.autoregressive_config(system_type='style_text')
        w = weights.init_autoregressive_transformer(cfg, seed=1234)
        rng = np.random.default_rng(1234 + rank)
        tokens = rng.integers(1, cfg['vocab_size'], size=(B, N)).astype(np.int32)
        targets = np.clip(rng.normal(0, 1.5, size=(B, T, 80)), -4, 4).astype(np.float32)
        model = AutoregressiveTransformer(w, cfg, precision=args.precision)
        model.set_constants(reduction_factor=1)
        tok_d, tar_d = torch.from_numpy(tokens).cuda(), torch.from_numpy(targets).cuda()

        def step():
            o = model.call(tok_d, tar_d, None, return_attention=want_attn)
            o['mel'] = o['final_output']
            return o

        def cpu_step():
            from oracle import tts_torch as OT
            return OT.autoregressive_call(w, cfg, tokens, targets, None, r=1)['final_output']
        flops_per_step = B * 9.6e9 * (T / 600.0)
        label = (f'GST reference encoder + 10 style tokens + teacher-forced AutoregressiveTransformer (style_text, r=1), '
                 f'batch={B}, N={N}, T={T} per GPU (BASELINE config 3)')

    out = step()
    torch.cuda.synchronize()

    # parity gate on the benchmarked utterance (rank 0): float32 restatement on the CPU
    cpu = None
    mel_l1 = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:      # the CPU leg belongs to the one-GPU run only
        torch.set_num_threads(min(os.cpu_count() or 1, 16))
        if B * T <= 2400:
            cpu_step()                                                    # warm-up
        runs, t_cpu = 0, 0.0
        while t_cpu < 20.0 and (runs < 3 or t_cpu < 10.0):      # about 10 s of CPU work, at least 3 passes within 20 s
            t0 = time.perf_counter()
            ref_mel = cpu_step()
            t_cpu += time.perf_counter() - t0
            runs += 1
        mel_l1 = float(np.abs(out['mel'].cpu().numpy() - ref_mel).mean())
        cpu = {'value': B * T * runs / t_cpu, 'unit': 'mel-frames/s', 'cores': torch.get_num_threads(), 'kind': 'port',
               'sample': f'{runs} pass(es) over the same batch (B={B}, N={N}, T={T}), op-for-op PyTorch CPU float32 '
                         f'restatement (oracle/tts_torch.py)'}

    # per-kernel timing for the roofline entry
    # (the traced step's outputs must stay referenced while its launches are replayed: torch.cuda.graph() empties
    # the allocator cache on entry, which would unmap any buffer that has been freed in the meantime)
    ops.start_trace()
    n_launches, traced_out = launches_of(step)
    trace = ops.stop_trace()
    torch.cuda.synchronize()
    roofline, _ = roofline_from_trace(trace, args.kernel_table, rank)
    del traced_out, trace

    # whole-forward timing
    graph = None
    if not args.no_graph:
        s = torch.cuda.Stream()
        s.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(s):
            step()
        torch.cuda.current_stream().wait_stream(s)
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph):
            out = step()
        run = graph.replay
    else:
        run = step
    for _ in range(args.warmup):
        run()
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        run()
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    if dist is not None:
        t = torch.tensor([dt], device='cuda', dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    ms = dt * 1e3 / args.steps
    value = world * B * T / (dt / args.steps)

    train_line = None
    if world > 1 and args.workload == 'forward' and args.train_batch > 0:
        # the data-parallel training step of BASELINE config 5 on the same ranks (its own model and buffers)
        try:
            train_line = train_measure(args, rank, world, dist, args.train_batch, 10, 3, False)
        except Exception as e:                # the headline line must survive a failing extra (every rank fails alike or not at all)
            import traceback
            traceback.print_exc()
            train_line = {'error': f'{type(e).__name__}: {e}'} if rank == 0 else None

    if rank == 0:
        finish_roofline(roofline, ms)
        line = {
            'metric': 'mel-frames/sec (fwd TransformerTTS)', 'value': value, 'unit': 'mel-frames/s', 'n_gpus': world,
            'steps': args.steps, 'warmup': args.warmup, 'ms_per_step': ms, 'higher_is_better': True, 'scaling': 'weak',
            'vs_baseline': None, 'dtype': 'bf16', 'data': 'synthetic',
            'config': {'workload': f'{label}; {args.precision} MFMA operands, fp32 accumulate',
                       'global_batch': world * B, 'parallelism': f'replicas x{world}', 'hip_graph': graph is not None,
                       'attention_weights_materialised': want_attn, 'launches_per_step': n_launches},
            'algorithmic_tflops': world * flops_per_step / (dt / args.steps) / 1e12,
            'mel_l1_vs_cpu_fp32': mel_l1,
            'roofline': roofline,
            'cpu_baseline': cpu,
        }
        if train_line is not None and 'error' in train_line:
            line['train'] = train_line
        elif train_line is not None:
            keep = ('metric', 'value', 'unit', 'ms_per_step', 'steps', 'warmup', 'steps_per_s', 'algorithmic_tflops', 'scaling',
                    'loss_first_step', 'loss_last_step', 'exchange', 'roofline')
            line['train'] = {k: train_line[k] for k in keep if k in train_line}
            line['train'].update(workload=train_line['config']['workload'], global_batch=train_line['config']['global_batch'],
                                 launches_per_step=train_line['config']['launches_per_step'])
        if world == 1 and args.workload == 'forward' and graph is not None and not args.no_in_flight:
            line['in_flight'] = in_flight_pass(w, cfg, tok_d, dur_d, T, want_attn, args, out['mel'])
            line['serving'] = serving_pass(w, cfg, tokens, args)
        if world == 1 and not args.no_wavernn:
            line['wavernn'] = wavernn_pass(args.precision, cpu_leg=not args.no_cpu_baseline, kernel_table=args.kernel_table)
            try:
                line['wavernn']['mol'] = wavernn_mol_pass(args.precision)
            except Exception as e:            # the headline line must survive a failing extra
                line['wavernn']['mol'] = {'error': f'{type(e).__name__}: {e}'}
            if args.workload == 'forward' and B == 1:
                # the other single-GPU configurations, each in a child process of its own (fresh allocator and graphs)
                line['config3'] = secondary_pass(['--workload', 'gst', '--batch', '32', '--steps', '20', '--warmup', '3'], args.precision)
                line['config5_one_gpu'] = secondary_pass(['--workload', 'train', '--batch', '32', '--steps', '10', '--warmup', '3'],
                                                         args.precision)
        print(json.dumps(line))
    if dist is not None:
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
