"""The HIP path at the benchmark's own configurations (sizes of BASELINE.json configs 3, 4 and 5) against the CPU oracles,
plus the small fixtures under tests/golden that pin the same code paths.

  config 3  GST reference encoder + 10 style tokens + teacher-forced AutoregressiveTransformer, batch 32, N = 100, T = 600,
            style_text, r = 1 (model/AutoregressiveTransformer.py:243-263): mel mean-L1 <= 1e-3 against the float32
            restatement oracle/tts_torch.py (the gate of the benchmark), every output within the bounds stated below
  config 4  WaveRNN RAW 9-bit mu-law at hop 256 / upsample (4, 8, 8) / 22 050 Hz (fatchord_version.py:79-86,150-243):
            small-width fixture written by the reference module itself; the full-width step chain against the float64
            oracle on a forced trajectory; the 64-utterance batch through "a sequence in a batch = the sequence alone"
  config 5  one ForwardTransformer training step at batch 32, N = 100, T = 600 (models.py:389-404) against the float64
            autograd oracle: loss, outputs, gradient direction
"""
import os
import sys

import numpy as np
import pytest
import torch

import estts  # noqa: F401
from estts import configs, weights

pytestmark = pytest.mark.gpu
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden'))


def _ragged_lengths(rng, B, lo, hi):
    n = rng.integers(lo, hi + 1, size=B)
    n[0] = hi
    return n


# ------------------------------------------------------------------------------------------------ config 3
def test_config3_gst_autoregressive_b32_vs_oracle():
    from estts.autoregressive_transformer import AutoregressiveTransformer
    from oracle import tts_torch as OT
    cfg = configs.autoregressive_config(system_type='style_text')
    w = weights.init_autoregressive_transformer(cfg, seed=1234)
    rng = np.random.default_rng(1234)
    B, N, T = 32, 100, 600
    tokens = rng.integers(1, cfg['vocab_size'], size=(B, N)).astype(np.int32)
    targets = np.clip(rng.normal(0, 1.5, size=(B, T, 80)), -4, 4).astype(np.float32)
    for b, (n, t) in enumerate(zip(_ragged_lengths(rng, B, 60, N), _ragged_lengths(rng, B, 300, T))):
        tokens[b, n:] = 0
        targets[b, t:] = 0
    model = AutoregressiveTransformer(w, cfg, precision='bf16x3')
    model.set_constants(reduction_factor=1, decoder_prenet_dropout=0.0)
    out = model.call(tokens, targets, None)
    torch.set_num_threads(min(os.cpu_count() or 1, 16))
    ref = OT.autoregressive_call(w, cfg, tokens, targets, None, r=1)
    mel, rmel = out['final_output'].cpu().numpy(), np.asarray(ref['final_output'])
    assert mel.shape == (B, T, 80)
    l1 = float(np.abs(mel - rmel).mean())
    print(f'config 3: mel mean-L1 {l1:.2e}, max {np.abs(mel - rmel).max():.2e}')
    assert l1 <= 1e-3                                   # the benchmark's gate
    assert l1 < 1e-4 and np.abs(mel - rmel).max() < 5e-3   # what split-bf16 against float32 actually gives
    for key, tol in (('mel_linear', 5e-3), ('stop_prob', 5e-3), ('gst_output', 1e-3), ('text_enc_output', 2e-3)):
        assert np.abs(out[key].cpu().numpy() - np.asarray(ref[key])).max() < tol, key
    names = [f'Decoder_DenseBlock{i + 1}_CrossAttention' for i in range(4)]
    for name, ra in zip(names, ref['decoder_attention']):
        assert np.abs(out['decoder_attention'][name].cpu().numpy() - np.asarray(ra)).max() < 2e-3, name
    g = out['gst_encoder_attention']['RefEncoderGST_attention'].cpu().numpy()
    assert np.abs(g - np.asarray(ref['gst_encoder_attention'])).max() < 1e-3


def test_ar_gst_small_fixture(golden_dir):
    """The committed fixture tests/golden/ar_gst_small.npz (float64 oracle, ragged batch, speaker + style + text, r = 2)."""
    from estts.autoregressive_transformer import AutoregressiveTransformer
    from make_tts_goldens import ar_gst_inputs
    g = np.load(os.path.join(golden_dir, 'ar_gst_small.npz'))
    cfg, seed, *_ = ar_gst_inputs()
    assert seed == int(g['weight_seed'])
    w = weights.init_autoregressive_transformer(cfg, seed=seed)
    model = AutoregressiveTransformer(w, cfg)
    model.set_constants(reduction_factor=int(g['r']), decoder_prenet_dropout=0.0)
    out = model.call(g['tokens'], g['targets'], g['spk'])
    for key, tol in (('final_output', 3e-4), ('mel_linear', 3e-4), ('stop_prob', 3e-4), ('text_enc_output', 3e-4),
                     ('gst_output', 1e-4)):
        assert np.abs(out[key].cpu().numpy() - g[key]).max() < tol, key
    assert np.abs(out['gst_encoder_attention']['RefEncoderGST_attention'].cpu().numpy() - g['gst_encoder_attention']).max() < 1e-4
    assert np.abs(out['decoder_attention']['Decoder_ConvBlock1_CrossAttention'].cpu().numpy() - g['last_cross_attention']).max() < 1e-3


# ------------------------------------------------------------------------------------------------ config 4
def _wavernn(hp, seed, **kw):
    from estts.wavernn import WaveRNN
    m = WaveRNN.from_hparams(hp, **kw)
    w = weights.init_wavernn(hp, seed=seed)
    m.load_state_dict(w)
    return m, w


def test_config4_hop256_reference_fixture(golden_dir):
    """Upsampling (4, 8, 8), teacher-forced logits and a free-running RAW trajectory against the fixture the reference
    module wrote for this vocoder shape (tests/golden/make_wavernn_goldens.py)."""
    from wavernn_common import small_hp_hop256
    from oracle import wavernn_numpy as W
    g = np.load(os.path.join(golden_dir, 'wavernn_raw_hop256.npz'))
    hp = small_hp_hop256()
    m, w = _wavernn(hp, int(g['seed']))
    frames = g['upsampled_mel'].shape[1] // hp['hop_length']
    mp = np.pad(g['mel'][:, :, :frames], ((0, 0), (0, 0), (hp['voc_pad'], hp['voc_pad'])))
    up, aux = m.upsample(mp)
    assert np.abs(up.cpu().numpy() - g['upsampled_mel']).max() < 1e-4
    assert np.abs(aux.cpu().numpy() - g['aux']).max() < 1e-4
    logits = m.forward(g['fwd_x'], g['fwd_mels'])
    assert np.abs(logits.cpu().numpy() - g['fwd_logits']).max() < 2e-3
    m.steps_per_graph = 50
    wav, raw = m.generate(torch.from_numpy(g['mel']), True, hp['voc_target'], hp['voc_overlap'], hp['mu_law'],
                          uniforms=g['u'], return_trace=True)
    _, _, raw_ref = W.generate(g['mel'], w, hp, g['u'], return_trace=True)
    n = 2 ** hp['bits']
    k, k_ref = np.rint((raw + 1) * (n - 1) / 2).astype(int), np.rint((raw_ref + 1) * (n - 1) / 2).astype(int)
    assert np.array_equal(k, k_ref)                          # same classes at every step of every fold
    assert wav[0].shape == g['wav'].shape and np.abs(wav[0] - g['wav']).max() < 1e-6


def test_config4_full_width_step_chain_vs_oracle():
    """configs.wavernn_hparams_22k() as benchmarked (rnn 512, fc 512, 9-bit RAW, hop 256): 260 steps of the folds of one
    short utterance against the float64 oracle, fed the oracle's samples so that one moved draw cannot snowball."""
    from estts.wavernn import fold_starts
    from oracle import wavernn_numpy as W
    hp = configs.wavernn_hparams_22k()
    m, w = _wavernn(hp, 77)
    rng = np.random.default_rng(3)
    T = 22          # the reference's fade-out needs (T - 1) hop >= 20 hop samples
    mel = rng.uniform(0, 1, size=(1, hp['num_mels'], T)).astype(np.float32)
    target, overlap = 200, 30
    n_seq = len(fold_starts(T * hp['hop_length'], target, overlap))
    L = target + 2 * overlap
    u = rng.uniform(0, 1, size=(L, n_seq))
    _, _, raw_ref = W.generate(mel, w, hp, u, target=target, overlap=overlap, return_trace=True)
    _, raw = m.generate(torch.from_numpy(mel), True, target, overlap, True, uniforms=u, forced=raw_ref.T, return_trace=True)
    k = np.rint((raw + 1) * 511 / 2).astype(int)
    k_ref = np.rint((raw_ref + 1) * 511 / 2).astype(int)
    assert np.mean(k != k_ref) < 0.01       # a draw within ~1e-6 of a CDF step may land in the neighbouring class
    assert np.abs(k - k_ref).max() <= 1


def test_config4_64_utterances_equal_single_runs():
    """The benchmarked batch: 64 utterances x 560 frames, target 11 000 / overlap 550 (832 sequences x 12 100 steps).  Every
    fold is an independent sequence, so an utterance vocoded alone must give the batch's samples.  The batch runs free; the
    single runs are fed the batch's own samples (their products run through other tile shapes, so single draws that sit on a
    CDF step may differ, and must not snowball)."""
    from estts.wavernn import fold_starts
    hp = configs.wavernn_hparams_22k()
    m, w = _wavernn(hp, 1234)
    rng = np.random.default_rng(1234)
    mels = [rng.uniform(0, 1, size=(hp['num_mels'], 560)).astype(np.float32) for _ in range(64)]
    target, overlap = hp['voc_target'], hp['voc_overlap']
    per = len(fold_starts(560 * hp['hop_length'], target, overlap))
    L = target + 2 * overlap
    u_all = rng.uniform(0, 1, size=(L, 64 * per)).astype(np.float32)
    outs, raw = m.generate_batch(mels, True, target, overlap, True, uniforms=u_all, return_trace=True)
    assert raw.shape == (64 * per, L) and len(outs) == 64
    assert all(o.shape == ((560 - 1) * hp['hop_length'],) and np.isfinite(o).all() for o in outs)
    k_all = np.rint((raw + 1) * 511 / 2).astype(int)
    assert len(np.unique(k_all)) > 100                       # a real distribution, not a collapsed one
    for ui in (0, 41, 63):
        sl = slice(ui * per, (ui + 1) * per)
        alone, raw1 = m.generate_batch([mels[ui]], True, target, overlap, True, uniforms=u_all[:, sl], forced=raw[sl].T,
                                       return_trace=True)
        k1 = np.rint((raw1 + 1) * 511 / 2).astype(int)
        diff = k1 != k_all[sl]
        print(f'utterance {ui}: {int(diff.sum())} of {diff.size} draws differ')
        assert diff.mean() < 1e-4 and np.abs(k1 - k_all[sl]).max() <= 1


# ------------------------------------------------------------------------------------------------ config 5
def test_config5_train_step_b32_vs_oracle():
    from estts.forward_training import ForwardTransformerTrainer
    from oracle import tts_train_torch as OTT
    from test_gpu_training import grads_by_name
    cfg = configs.forward_config()
    w = weights.init_forward_transformer(cfg, seed=1234)
    rng = np.random.default_rng(99)
    B, N, T = 32, 100, 600
    tokens = rng.integers(1, cfg['vocab_size'], size=(B, N)).astype(np.int32)
    d = rng.integers(1, 12, size=(B, N)).astype(np.float64)
    lens = _ragged_lengths(rng, B, 60, N)
    for b, n in enumerate(lens):
        tokens[b, n:] = 0
        d[b, n:] = 0
    tot = np.floor(np.linspace(T, 0.55 * T, B))                  # frames per utterance, row 0 has all T
    d = np.floor(d * (tot / d.sum(axis=1))[:, None])
    d[:, 0] += tot - d.sum(axis=1)
    durs = d.astype(np.int32)
    assert durs.sum(1).max() == T and (durs >= 0).all()
    mel = np.clip(rng.normal(0, 1.5, size=(B, T, 80)), -4, 4).astype(np.float32)
    for b in range(B):
        mel[b, durs[b].sum():] = 0
    # dropout off: the oracle would need the device's 3.4 M-entry masks; the masked paths are covered at small size
    cfg['dropout_rate'] = 0.0
    model = ForwardTransformerTrainer(w, cfg, precision='bf16x3', seed=1)
    model.set_constants(learning_rate=1e-4, decoder_prenet_dropout=0.0, drop_n_heads=0)
    # the oracle differentiates the branches the device took (every ReLU's side, the sign of every MAE term), so that the
    # comparison below can be tight: without them a ReLU input within rounding of zero falls either way and moves everything
    # upstream of it by a percent
    model.gate_log = []
    out = model.train_step(tokens, mel, durs)
    gates = [g.cpu().numpy() if g.dtype == torch.bool else g.float().cpu().numpy() for g in model.gate_log]
    model.gate_log = None
    torch.set_num_threads(min(os.cpu_count() or 1, 16))
    ref = OTT.forward_train_step(w, cfg, tokens, mel, durs, gates=gates)
    del gates
    assert abs(float(out['loss']) - ref['loss']) < 1e-4 * max(1.0, abs(ref['loss']))
    assert abs(float(out['losses']['mel']) - ref['mel_loss']) < 1e-4
    assert abs(float(out['losses']['duration']) - ref['dur_loss']) < 1e-4
    err = np.abs(out['mel'].cpu().numpy() - ref['mel'])
    print(f'config 5: loss {float(out["loss"]):.6f} vs {ref["loss"]:.6f}, mel mean err {err.mean():.2e} max {err.max():.2e}')
    assert err.mean() < 1e-3 and err.max() < 2e-2
    mine = grads_by_name(model)
    dot = n1 = n2 = 0.0
    worst = []
    for k, r in ref['grads'].items():
        a = mine[k].reshape(-1).astype(np.float64)
        r = np.asarray(r).reshape(-1)
        dot += float(a @ r); n1 += float(a @ a); n2 += float(r @ r)
        if np.abs(r).max() > 1e-7:
            worst.append((float(np.abs(a - r).max() / np.abs(r).max()), k))
    worst.sort(reverse=True)
    cosine = dot / np.sqrt(n1 * n2)
    print(f'config 5: gradient cosine {cosine:.6f}, worst tensors {worst[:3]}, median {worst[len(worst) // 2][0]:.1e}')
    # first line: every tensor within 1e-3 of its own largest entry, as at small size (test_gpu_training.py); the query / key
    # projections of the self-attentions get 1e-2 (where a softmax saturates their true gradient vanishes and what remains
    # is the rounding of large cancelling terms).  Second line: the direction of the whole gradient.
    loose = [(e, k) for e, k in worst if '.mha.wq.' in k or '.mha.wk.' in k or '.mha.wqkv.' in k]
    tight = [(e, k) for e, k in worst if (e, k) not in loose]
    assert tight[0][0] < 1e-3, tight[:5]
    assert not loose or loose[0][0] < 1e-2, loose[:5]
    assert cosine > 0.99999
