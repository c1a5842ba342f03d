"""Writes tests/golden/wavernn_*.npz from the REFERENCE WaveRNN module itself.

Runs only in the build container (needs /root/reference): imports
/root/reference/WaveRNN/models/fatchord_version.py on the CPU with three import-time shims
  1. stub modules `librosa`, `librosa.filters`, `librosa.output` (imported at module scope by utility/dsp.py;
     no librosa function is on the model path)
  2. np.cumproduct = np.cumprod (removed in NumPy 2; used at fatchord_version.py:65)
  3. torch.Tensor.cuda = identity (hard-coded .cuda() calls at fatchord_version.py:122-123,162,173-175,265,311)
and, so that sampled trajectories can be reproduced outside torch, patches the two random-draw sites while
generate() runs:
  * torch.distributions.Categorical.sample -> inverse CDF of `probs` at a recorded uniform number
  * torch.Tensor.uniform_                  -> fills from the same recorded stream (MOL sampler)
Weights come from estts.weights.init_wavernn(hp, seed) (loaded into the reference module with load_state_dict), so
the fixture stores only the seed, the inputs, the recorded uniforms and the reference's outputs.

Run from the repository root:  python tests/golden/make_wavernn_goldens.py
"""
import os
import sys
import types

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
REF = '/root/reference/WaveRNN'
sys.dont_write_bytecode = True
sys.path.insert(0, ROOT)

for name in ('librosa', 'librosa.filters', 'librosa.output'):
    sys.modules[name] = types.ModuleType(name)
np.cumproduct = np.cumprod
torch.Tensor.cuda = lambda self, *a, **k: self
os.environ.setdefault('MPLBACKEND', 'Agg')

import estts  # noqa: E402
from estts import configs, weights  # noqa: E402

sys.path.insert(0, REF)
_cwd = os.getcwd()
os.chdir(REF)
from models.fatchord_version import WaveRNN  # noqa: E402
os.chdir(_cwd)


sys.path.insert(0, os.path.join(ROOT, 'tests'))
from wavernn_common import small_hp, small_hp_hop256  # noqa: E402


def build(hp, seed):
    w = weights.init_wavernn(hp, seed=seed)
    model = WaveRNN(rnn_dims=hp['voc_rnn_dims'], fc_dims=hp['voc_fc_dims'], bits=hp['bits'], pad=hp['voc_pad'],
                    upsample_factors=hp['voc_upsample_factors'], feat_dims=hp['num_mels'],
                    compute_dims=hp['voc_compute_dims'], res_out_dims=hp['voc_res_out_dims'],
                    res_blocks=hp['voc_res_blocks'], hop_length=hp['hop_length'], sample_rate=hp['sample_rate'],
                    mode=hp['voc_mode'])
    sd = {k: torch.from_numpy(v) for k, v in w.items()}
    missing, unexpected = model.load_state_dict(sd, strict=False)
    assert not unexpected, unexpected
    assert all(k == 'step' or k.endswith('num_batches_tracked') for k in missing), missing
    return model, w


class Recorder:
    def __init__(self, seed):
        self.g = torch.Generator().manual_seed(seed)
        self.uniform_calls = []
        self.categorical_u = []
        self.logits = []

    def install(self, model):
        rec = self
        self._orig_sample = torch.distributions.Categorical.sample
        self._orig_uniform = torch.Tensor.uniform_

        def sample(dist, sample_shape=torch.Size()):
            probs = dist.probs
            u = torch.rand(probs.shape[0], generator=rec.g, dtype=torch.float64)
            rec.categorical_u.append(u.numpy().copy())
            cdf = torch.cumsum(probs.double(), dim=-1)
            c = (cdf <= u[:, None]).sum(dim=-1).clamp(max=probs.shape[-1] - 1)
            return c

        def uniform_(t, a=0.0, b=1.0, **kw):
            if 'generator' in kw:          # torch.nn.init (GRUCell construction inside generate): not a sampling draw
                return rec._orig_uniform(t, a, b, **kw)
            r = torch.rand(t.shape, generator=rec.g, dtype=torch.float64) * (b - a) + a
            rec.uniform_calls.append(r.numpy().copy())
            t.copy_(r.to(t.dtype))
            return t

        torch.distributions.Categorical.sample = sample
        torch.Tensor.uniform_ = uniform_
        self._hook = model.fc3.register_forward_hook(lambda m, i, o: rec.logits.append(o.detach().numpy().copy()))

    def remove(self):
        torch.distributions.Categorical.sample = self._orig_sample
        torch.Tensor.uniform_ = self._orig_uniform
        self._hook.remove()


def make(mode, seed, T, fname, hp=None, up_frames=None, logits_every=1):
    """up_frames: check the materialised upsampling on the first up_frames frames only (hop 256 writes 336 floats per
    sample; the generate trace below still covers every frame)."""
    hp = hp or small_hp(mode)
    model, w = build(hp, seed)
    model.eval()
    rng = np.random.default_rng(seed + 1)
    mel = rng.uniform(0, 1, size=(1, hp['num_mels'], T)).astype(np.float32)
    out = {'seed': seed, 'mel': mel}
    with torch.no_grad():
        mp = torch.from_numpy(np.pad(mel[:, :, :up_frames], ((0, 0), (0, 0), (hp['voc_pad'], hp['voc_pad']))))
        up, aux = model.upsample(mp)
        out['upsampled_mel'] = up.numpy()
        out['aux'] = aux.numpy()
        # teacher-forced forward on a short window (first 4 frames + padding)
        L = hp['hop_length'] * 4
        xw = torch.from_numpy(rng.uniform(-1, 1, size=(2, L)).astype(np.float32))
        mw = torch.from_numpy(rng.uniform(0, 1, size=(2, hp['num_mels'], 4 + 2 * hp['voc_pad'])).astype(np.float32))
        out['fwd_x'] = xw.numpy()
        out['fwd_mels'] = mw.numpy()
        out['fwd_logits'] = model.forward(xw, mw).numpy()
    rec = Recorder(seed + 2)
    rec.install(model)
    try:
        wav = model.generate(torch.from_numpy(mel), True, hp['voc_target'], hp['voc_overlap'], hp['mu_law'])
    finally:
        rec.remove()
    out['wav'] = wav
    out['gen_logits'] = np.stack(rec.logits)[::logits_every]      # (steps, folds, n_classes), every logits_every-th step
    if mode == 'RAW':
        out['u'] = np.stack(rec.categorical_u)                     # (steps, folds)
    else:
        out['u_mix'] = np.stack([c[0] for c in rec.uniform_calls[0::2]])        # (steps, folds, nr_mix)
        out['u_logistic'] = np.stack([c[0] for c in rec.uniform_calls[1::2]])   # (steps, folds)
    np.savez_compressed(os.path.join(ROOT, 'tests', 'golden', fname), **out)
    print(fname, 'wav', wav.shape, 'logits', out['gen_logits'].shape)


if __name__ == '__main__':
    make('RAW', 101, 23, 'wavernn_raw_small.npz')
    make('MOL', 202, 23, 'wavernn_mol_small.npz')
    make('RAW', 303, 23, 'wavernn_raw_hop256.npz', hp=small_hp_hop256(), up_frames=3, logits_every=8)
