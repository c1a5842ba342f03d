#!/usr/bin/env python3
"""A few hundred sampling steps of the config-4 WaveRNN state (64 utterances x 560 frames = 832 sequences, hop 256), issued
launch by launch: what the counter passes of tools/collect_profiles.sh run under rocprofv3 --pmc (the full 12 100-step pass
would write 70 000 counter rows)."""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import estts  # noqa
from estts import configs, ops, weights
from estts.wavernn import WaveRNN, fold_starts


def main():
    steps = int(sys.argv[1]) if len(sys.argv) > 1 else 200
    mode = sys.argv[2] if len(sys.argv) > 2 else 'RAW'
    hp = configs.wavernn_hparams_22k(voc_mode=mode)
    m = WaveRNN.from_hparams(hp)
    m.load_state_dict(weights.init_wavernn(hp, seed=1234))
    rng = np.random.default_rng(1234)
    mels = [rng.uniform(0, 1, size=(hp['num_mels'], 560)).astype(np.float32) for _ in range(64)]
    cond = m._prepare(mels)
    seq_utt, seq_start = [], []
    for ui in range(64):
        st_ = fold_starts(560 * hp['hop_length'], hp['voc_target'], hp['voc_overlap'])
        seq_utt += [ui] * len(st_)
        seq_start += st_
    n_seq = len(seq_utt)
    i32 = dict(dtype=torch.int32, device='cuda')
    su, ss = torch.tensor(seq_utt, **i32), torch.tensor(seq_start, **i32)
    st = m._alloc_state(n_seq, steps + 8)
    prep_args = m._prep_args(cond, st, su, ss, n_seq)
    st['cond_args'] = m._cond_args(cond, st, su, ss, n_seq)
    sample_args = m._sample_args(st, n_seq, 1, None, None, None)
    for _ in range(steps):
        m._one_step(st, prep_args, sample_args)
    torch.cuda.synchronize()
    print(f'{steps} steps of {n_seq} sequences ({mode})')


if __name__ == '__main__':
    main()
