#!/usr/bin/env python3
"""Experiment (GPU box): the WaveRNN step chain of G independent groups of sequences on G streams, one hipGraph each,
replayed in turn, against the same sequences as one group.  Per-step time counts the steps of ONE group (all groups advance
together), so the figures compare directly."""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import estts  # noqa
from estts import configs, weights
from estts.wavernn import WaveRNN, fold_starts


def setup(m, mels, hp, L):
    cond = m._prepare(mels)
    seq_utt, seq_start = [], []
    for ui, T in enumerate(cond['frames']):
        s = fold_starts(T * hp['hop_length'], hp['voc_target'], hp['voc_overlap'])
        seq_utt += [ui] * len(s)
        seq_start += s
    n_seq = len(seq_utt)
    i32 = dict(dtype=torch.int32, device='cuda')
    su, ss = torch.tensor(seq_utt, **i32), torch.tensor(seq_start, **i32)
    st = m._alloc_state(n_seq, L)
    prep = m._prep_args(cond, st, su, ss, n_seq)
    st['cond_args'] = m._cond_args(cond, st, su, ss, n_seq)
    samp = m._sample_args(st, n_seq, 1, None, None, None)
    return dict(cond=cond, st=st, prep=prep, samp=samp, n_seq=n_seq, keep=(su, ss))


def main():
    hp = configs.wavernn_hparams_22k()
    m = WaveRNN.from_hparams(hp, precision='bf16x3')
    m.load_state_dict(weights.init_wavernn(hp, seed=1234))
    rng = np.random.default_rng(1234)
    mels = [rng.uniform(0, 1, size=(hp['num_mels'], 560)).astype(np.float32) for _ in range(64)]
    n = m.steps_per_graph
    reps = 40
    L = n * (reps + 2) + 2
    for G in (1, 2, 4):
        per = 64 // G
        groups = [setup(m, mels[i * per:(i + 1) * per], hp, L) for i in range(G)]
        streams = [torch.cuda.Stream() for _ in range(G)]
        graphs = []
        for gr, s in zip(groups, streams):
            s.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(s):
                m._one_step(gr['st'], gr['prep'], gr['samp'])
            torch.cuda.synchronize()
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g, stream=s):
                for _ in range(n):
                    m._one_step(gr['st'], gr['prep'], gr['samp'])
            graphs.append(g)
        torch.cuda.synchronize()
        for timed in (False, True):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(reps if timed else 2):
                for g, s in zip(graphs, streams):
                    with torch.cuda.stream(s):
                        g.replay()
            torch.cuda.synchronize()
            dt = time.perf_counter() - t0
        print(f'{G} group(s) of {groups[0]["n_seq"]} sequences: {dt / (reps * n) * 1e6:.1f} us per step of all {sum(g_["n_seq"] for g_ in groups)} sequences', flush=True)
        del graphs, groups


if __name__ == '__main__':
    main()
