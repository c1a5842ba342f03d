"""Writes tests/golden/wavernn_train_{raw,mol}_small.npz from the REFERENCE WaveRNN module: one training step
(model.train(); forward; F.cross_entropy or discretized_mix_logistic_loss; backward; torch.optim.Adam step) exactly as
WaveRNN/train_wavernn.py:27-42 runs it, on the small model of tests/wavernn_common.py.

Runs only in the build container (needs /root/reference); same three import-time shims as make_wavernn_goldens.py
(librosa stubs, np.cumproduct, Tensor.cuda = identity).  Stored: inputs, loss, every parameter gradient, the BatchNorm
running statistics and every parameter after the optimizer step.  Weights come from estts.weights.init_wavernn(hp, seed).

Run from the repository root:  python tests/golden/make_wavernn_train_goldens.py
"""
import os
import sys
import types

import numpy as np
import torch
import torch.nn.functional as F

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
REF = '/root/reference/WaveRNN'
sys.dont_write_bytecode = True
sys.path.insert(0, ROOT)
for name in ('librosa', 'librosa.filters', 'librosa.output'):
    sys.modules[name] = types.ModuleType(name)
np.cumproduct = np.cumprod
torch.Tensor.cuda = lambda self, *a, **k: self
os.environ.setdefault('MPLBACKEND', 'Agg')

import estts  # noqa: E402,F401
from estts import weights  # noqa: E402

sys.path.insert(0, REF)
_cwd = os.getcwd()
os.chdir(REF)
from models.fatchord_version import WaveRNN  # noqa: E402
from utility.distribution import discretized_mix_logistic_loss  # noqa: E402
os.chdir(_cwd)
sys.path.insert(0, os.path.join(ROOT, 'tests'))
from wavernn_common import small_hp  # noqa: E402

SEED, B, HOPS, LR = 21, 3, 4, 1e-4


def main():
    for mode in ('RAW', 'MOL'):
        hp = small_hp(mode)
        w = weights.init_wavernn(hp, seed=SEED)
        model = WaveRNN(rnn_dims=hp['voc_rnn_dims'], fc_dims=hp['voc_fc_dims'], bits=hp['bits'], pad=hp['voc_pad'],
                        upsample_factors=hp['voc_upsample_factors'], feat_dims=hp['num_mels'],
                        compute_dims=hp['voc_compute_dims'], res_out_dims=hp['voc_res_out_dims'],
                        res_blocks=hp['voc_res_blocks'], hop_length=hp['hop_length'], sample_rate=hp['sample_rate'], mode=mode)
        missing, unexpected = model.load_state_dict({k: torch.from_numpy(v) for k, v in w.items()}, strict=False)
        assert not unexpected
        model.train()
        rng = np.random.default_rng(SEED + 1)
        T = HOPS * hp['hop_length']
        n_cls = 2 ** hp['bits']
        labels = rng.integers(0, n_cls, size=(B, T + 1))
        wave = 2 * labels / (n_cls - 1.) - 1.                                  # label_2_float (utility/dsp.py)
        x = wave[:, :T].astype(np.float32)
        mels = rng.uniform(0, 1, size=(B, hp['num_mels'], HOPS + 2 * hp['voc_pad'])).astype(np.float32)
        if mode == 'RAW':
            y = labels[:, 1:].astype(np.int64)
        else:
            y = wave[:, 1:].astype(np.float32)
        opt = torch.optim.Adam(model.parameters())
        for g in opt.param_groups:
            g['lr'] = LR
        xt, mt, yt = torch.from_numpy(x), torch.from_numpy(mels), torch.from_numpy(y)
        y_hat = model(xt, mt)
        logits = y_hat.detach().numpy().copy()
        if mode == 'RAW':
            y_hat = y_hat.transpose(1, 2).unsqueeze(-1)
            loss = F.cross_entropy(y_hat, yt.unsqueeze(-1))
        else:
            loss = discretized_mix_logistic_loss(y_hat, yt.float().unsqueeze(-1))
        opt.zero_grad()
        loss.backward()
        out = {'x': x, 'y': y, 'mels': mels, 'loss': np.float64(loss.item()), 'logits': logits, 'seed': np.int64(SEED),
               'lr': np.float64(LR)}
        for k, p in model.named_parameters():
            if p.requires_grad:
                out['grad/' + k] = p.grad.numpy().copy()
        opt.step()
        for k, v in model.state_dict().items():
            if k != 'step' and not k.endswith('num_batches_tracked'):
                out['after/' + k] = v.numpy().copy()
        path = os.path.join(ROOT, 'tests', 'golden', f'wavernn_train_{mode.lower()}_small.npz')
        np.savez_compressed(path, **out)
        print(mode, 'loss', loss.item(), 'tensors', len(out), os.path.getsize(path), 'bytes')


if __name__ == '__main__':
    main()
