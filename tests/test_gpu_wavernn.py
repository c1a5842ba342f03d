"""WaveRNN on the GPU (HIP kernels through the C ABI) against golden vectors written by the reference module
itself (tests/golden/wavernn_*.npz) and against the NumPy oracle at full size.

Tolerances: float32 activations with split-bf16 (bf16x3) matrix products:
  upsampled conditioning 1e-4, teacher-forced logits 2e-3 (they grow through two recurrences over 48 steps),
  RAW trajectories: identical classes for the fixture seed (the draw is an inverse-CDF threshold test),
  MOL trajectories: 2e-3 on sample values.
"""
import os

import numpy as np
import pytest
import torch

from wavernn_common import small_hp
from estts import configs, weights
from estts.wavernn import WaveRNN
from oracle import wavernn_numpy as W

pytestmark = pytest.mark.gpu


def _model(hp, seed, **kw):
    m = WaveRNN.from_hparams(hp, **kw)
    w = weights.init_wavernn(hp, seed=seed)
    m.load_state_dict(w)
    return m, w


def _golden(golden_dir, mode):
    g = np.load(os.path.join(golden_dir, f'wavernn_{mode.lower()}_small.npz'))
    hp = small_hp(mode)
    m, w = _model(hp, int(g['seed']))
    return g, hp, m, w


@pytest.mark.parametrize('mode', ['RAW', 'MOL'])
def test_upsample_and_forward_vs_reference_goldens(golden_dir, mode):
    g, hp, m, w = _golden(golden_dir, mode)
    mp = np.pad(g['mel'], ((0, 0), (0, 0), (hp['voc_pad'], hp['voc_pad'])))
    up, aux = m.upsample(mp)
    assert np.abs(up.cpu().numpy() - g['upsampled_mel']).max() < 1e-4
    assert np.abs(aux.cpu().numpy() - g['aux']).max() < 1e-4
    logits = m.forward(g['fwd_x'], g['fwd_mels'])
    assert tuple(logits.shape) == g['fwd_logits'].shape
    assert np.abs(logits.cpu().numpy() - g['fwd_logits']).max() < 2e-3


@pytest.mark.parametrize('use_graph', [False, True])
def test_generate_raw_vs_reference_golden(golden_dir, use_graph):
    g, hp, m, w = _golden(golden_dir, 'RAW')
    m.steps_per_graph = 20
    wav, raw = m.generate(torch.from_numpy(g['mel']), True, hp['voc_target'], hp['voc_overlap'], hp['mu_law'],
                          uniforms=g['u'], use_graph=use_graph, return_trace=True)
    _, _, raw_ref = W.generate(g['mel'], w, hp, g['u'], return_trace=True)
    assert raw.shape == raw_ref.shape
    n = 2 ** hp['bits']
    k, k_ref = np.rint((raw + 1) * (n - 1) / 2).astype(int), np.rint((raw_ref + 1) * (n - 1) / 2).astype(int)
    assert np.array_equal(k, k_ref)                          # same classes at every step of every fold
    assert np.abs(wav[0] - g['wav']).max() < 1e-6            # hence the reference's waveform


def test_generate_mol_vs_reference_golden(golden_dir):
    g, hp, m, w = _golden(golden_dir, 'MOL')
    uni = {'mix': g['u_mix'], 'logistic': g['u_logistic']}
    _, _, raw_ref = W.generate(g['mel'], w, hp, uni, return_trace=True)
    # replay the reference trajectory (forced feedback) so that one rounding difference cannot snowball
    wav, raw = m.generate(torch.from_numpy(g['mel']), True, hp['voc_target'], hp['voc_overlap'], hp['mu_law'],
                          uniforms=uni, forced=raw_ref.T, return_trace=True)
    assert np.abs(raw - raw_ref).max() < 2e-3
    assert np.abs(wav[0] - g['wav']).max() < 2e-3


def test_full_size_raw_step_trajectory_vs_oracle():
    """Shipped dimensions (rnn 512, 9-bit RAW, hop 200): 260 steps of two folds against the float64 oracle."""
    hp = configs.wavernn_hparams(voc_mode='RAW')
    m, w = _model(hp, 77)
    rng = np.random.default_rng(3)
    T = 22
    mel = rng.uniform(0, 1, size=(1, hp['num_mels'], T)).astype(np.float32)
    target, overlap = 200, 30
    n_seq = len(__import__('estts.wavernn', fromlist=['fold_starts']).fold_starts(T * hp['hop_length'], target, overlap))
    L = target + 2 * overlap
    u = rng.uniform(0, 1, size=(L, n_seq))
    wav_ref, logits_ref, raw_ref = W.generate(mel, w, hp, u, target=target, overlap=overlap, return_trace=True)
    wav, raw = m.generate(torch.from_numpy(mel), True, target, overlap, True, uniforms=u, forced=raw_ref.T,
                          return_trace=True)
    # classes are integers k with value 2k/511 - 1: count disagreements (a logit rounding can move a draw that sits
    # within ~1e-6 of a CDF step)
    k = np.rint((raw + 1) * 511 / 2).astype(int)
    k_ref = np.rint((raw_ref + 1) * 511 / 2).astype(int)
    assert np.mean(k != k_ref) < 0.01
    assert np.abs(k - k_ref).max() <= 1


def test_batch_of_utterances_matches_single_runs():
    """Several utterances of different lengths in one call: folds are independent sequences, so each waveform must
    equal the one produced alone (same seed-independent injected uniforms)."""
    hp = small_hp('RAW')
    m, w = _model(hp, 9)
    rng = np.random.default_rng(4)
    mels = [rng.uniform(0, 1, size=(hp['num_mels'], T)).astype(np.float32) for T in (23, 31, 26)]
    from estts.wavernn import fold_starts
    folds = [len(fold_starts(x.shape[1] * hp['hop_length'], hp['voc_target'], hp['voc_overlap'])) for x in mels]
    L = hp['voc_target'] + 2 * hp['voc_overlap']
    u_all = rng.uniform(0, 1, size=(L, sum(folds)))
    outs = m.generate_batch(mels, True, hp['voc_target'], hp['voc_overlap'], True, uniforms=u_all)
    o = 0
    for x, f, got in zip(mels, folds, outs):
        alone = m.generate_batch([x], True, hp['voc_target'], hp['voc_overlap'], True, uniforms=u_all[:, o:o + f])[0]
        o += f
        assert got.shape == ((x.shape[1] - 1) * hp['hop_length'],)
        assert np.array_equal(got, alone)


@pytest.mark.parametrize('mu_law', [True, False])
def test_unfold_on_device_matches_the_host_functions(mu_law):
    """estts_wavernn_unfold (cross-fade of the folds, mu-law expansion, cut, closing fade; float64 on the device) against the
    host path it replaces in batched generation -- xfade_and_unfold, decode_mu_law and the linspace fade of WaveRNN.generate --
    on random fold contents for utterances of different lengths: windows and sums are the same float64 operations, only pow()
    may round differently (2e-15 absolute on values in [-1, 1])."""
    from estts.wavernn import WaveRNN, decode_mu_law, fold_starts, xfade_and_unfold
    hp = configs.wavernn_hparams(voc_mode='RAW')
    hp.update(voc_rnn_dims=32, voc_fc_dims=32, voc_compute_dims=16, voc_res_out_dims=16, voc_res_blocks=1)
    m = WaveRNN.from_hparams(hp)
    hop, target, overlap = m.hop_length, 700, 90
    rng = np.random.default_rng(3)
    frames = [31, 47, 25, 60]
    folds = [len(fold_starts(T * hop, target, overlap)) for T in frames]
    L = target + 2 * overlap
    raw = rng.integers(0, m.n_classes, size=(sum(folds), L)).astype(np.float32) * (2.0 / (m.n_classes - 1)) - 1.0
    wave = torch.from_numpy(raw.astype(np.float32)).cuda()
    got = m._unfold_on_device(wave, frames, folds, target, overlap, mu_law)
    o = 0
    for T, F, g in zip(frames, folds, got):
        y = xfade_and_unfold(raw[o:o + F].astype(np.float64), target, overlap)
        o += F
        if mu_law:
            y = decode_mu_law(y, m.n_classes, False)
        y = y[:(T - 1) * hop]
        y[-20 * hop:] *= np.linspace(1, 0, 20 * hop)
        assert g.dtype == np.float64 and g.shape == y.shape
        assert np.abs(g - y).max() < 2e-15
        if not mu_law:
            assert np.array_equal(g, y)           # without pow(): bit for bit


def test_builtin_generator_is_deterministic_and_uniform():
    hp = small_hp('RAW')
    m, w = _model(hp, 9)
    mel = np.random.default_rng(5).uniform(0, 1, size=(1, hp['num_mels'], 40)).astype(np.float32)
    a = m.generate(torch.from_numpy(mel), True, hp['voc_target'], hp['voc_overlap'], True, seed=123)
    b = m.generate(torch.from_numpy(mel), True, hp['voc_target'], hp['voc_overlap'], True, seed=123)
    c = m.generate(torch.from_numpy(mel), True, hp['voc_target'], hp['voc_overlap'], True, seed=124)
    assert np.array_equal(a, b) and not np.array_equal(a, c)
    assert a.shape == (39 * hp['hop_length'],) and np.isfinite(a).all() and np.abs(a).max() <= 1.0


def test_checkpoint_round_trip(tmp_path):
    hp = small_hp('MOL')
    m, w = _model(hp, 11)
    path = str(tmp_path / 'latest_weights.pyt')
    m.step = 4321
    m.save(path)
    m2 = WaveRNN.from_hparams(hp)
    m2.restore(path)
    assert m2.get_step() == 4321
    sd = torch.load(path, weights_only=True)
    assert set(k for k in sd if k != 'step') == set(w)


def _layer_args(_lib, n_seq, n_units, mode, a0, ks0, w, b, **kw):
    a = _lib.WavernnLayerArgs()
    a.a0, a.ks0, a.kpitch0 = a0.data_ptr(), ks0, a0.shape[1]
    a.W, a.bias = w.data_ptr(), b.data_ptr()
    a.n_seq, a.n_units, a.mode = n_seq, n_units, mode
    for k, v in kw.items():
        setattr(a, k, v.data_ptr() if torch.is_tensor(v) else v)
    return a


@pytest.mark.parametrize('n_seq,H,kx,mode', [(832, 512, 113, 1), (832, 512, 544, 2), (5, 64, 89, 1), (131, 64, 72, 2)])
def test_gru_layer_vs_float64(n_seq, H, kx, mode):
    """estts_wavernn_layer modes 1 / 2 (both products, gates, state update, residual in one launch, activation panels in and
    out) against the float64 formula of torch.nn.GRUCell (fatchord_version.py:188-195) on the panels' own (hi + lo) values.
    Tolerance 5e-5: split-bf16 products, float32 gate math; the panel outputs round to ~2^-17 relative."""
    from estts import _lib, ops
    from estts.wavernn import pack_gru_stream, panel_alloc, panel_pack, panel_unpack
    rng = np.random.default_rng(n_seq + kx)
    d = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()
    xp = panel_pack(d(rng.normal(size=(n_seq, kx)).astype(np.float32)))
    h = rng.uniform(-1, 1, size=(n_seq, H)).astype(np.float32)
    hpan = panel_pack(d(h))
    sc = 1.0 / np.sqrt(H)
    wx = [(rng.normal(size=(H, kx)) * sc).astype(np.float32) for _ in range(4 if mode == 1 else 3)]
    wh = [(rng.normal(size=(H, H)) * sc).astype(np.float32) for _ in range(3)]
    b = rng.normal(size=(5 if mode == 1 else 4, H)).astype(np.float32) * 0.3
    res = rng.normal(size=(n_seq, H)).astype(np.float32)
    bd, wd = d(b), pack_gru_stream(wx, wh).cuda()
    # the layer's float32 side arrays are tile-native: element ((workgroup * 8 + wave) * 64 + lane) * 4 + j is the pair
    # (sequence 128 mt + 32 (wave & 3) + 16 (wave >> 2) + 4 (lane >> 4) + j, unit 16 ub + (lane & 15)), workgroup = mt * (H / 16) + ub
    n_mt, n_ub = (n_seq + 127) // 128, H // 16
    wg, wv, ln, jj = np.meshgrid(np.arange(n_mt * n_ub), np.arange(8), np.arange(64), np.arange(4), indexing='ij')
    rows = 128 * (wg // n_ub) + 32 * (wv & 3) + 16 * (wv >> 2) + 4 * (ln >> 4) + jj
    units = 16 * (wg % n_ub) + (ln & 15)
    ok = rows < n_seq
    to_native = lambda m: d(np.where(ok, m[np.minimum(rows, n_seq - 1), units], 0).astype(np.float32).reshape(-1))

    def from_native(t):
        out = np.zeros((n_seq, H), np.float32)
        out[rows[ok], units[ok]] = t.cpu().numpy().reshape(rows.shape)[ok]
        return out
    hd, resd = to_native(h), to_native(res)
    h_out = torch.full_like(hd, float('nan'))
    out = torch.full_like(hd, float('nan'))
    h_new_p, out_p = panel_alloc(n_seq, H // 32, 'cuda'), panel_alloc(n_seq, H // 32 + 1, 'cuda')
    a = _layer_args(_lib, n_seq, H, mode, xp, xp.shape[1], wd, bd, a1=hpan, ks1=H // 32, kpitch1=hpan.shape[1], h_prev=hd,
                    h_out=h_out, h_panel=h_new_p, h_kpitch=h_new_p.shape[1], out_f32=out, out_panel=out_p,
                    out_kpitch=out_p.shape[1])
    if mode == 2:
        a.x_res = resd.data_ptr()
    _lib.check(_lib.load().estts_wavernn_layer(a, ops._stream()), 'estts_wavernn_layer')
    h_out, out = torch.from_numpy(from_native(h_out)), torch.from_numpy(from_native(out))
    x64 = panel_unpack(xp, n_seq, kx).cpu().numpy().astype(np.float64)        # what the kernel multiplies: hi + lo
    hx64 = panel_unpack(hpan, n_seq, H).cpu().numpy().astype(np.float64)
    b64 = b.astype(np.float64)
    o = 1 if mode == 1 else 0
    sig = lambda v: 1 / (1 + np.exp(-v))
    r = sig(x64 @ wx[o].T.astype(np.float64) + hx64 @ wh[0].T.astype(np.float64) + b64[o])
    z = sig(x64 @ wx[o + 1].T.astype(np.float64) + hx64 @ wh[1].T.astype(np.float64) + b64[o + 1])
    n = np.tanh(x64 @ wx[o + 2].T.astype(np.float64) + b64[o + 2] + r * (hx64 @ wh[2].T.astype(np.float64) + b64[o + 3]))
    hn = (1 - z) * n + z * h.astype(np.float64)
    y = (x64 @ wx[0].T.astype(np.float64) + b64[0] if mode == 1 else res.astype(np.float64)) + hn
    assert np.abs(h_out.cpu().numpy() - hn).max() < 5e-5
    assert np.abs(out.cpu().numpy() - y).max() < 5e-5
    assert np.abs(panel_unpack(h_new_p, n_seq, H).cpu().numpy() - hn).max() < 6e-5
    assert np.abs(panel_unpack(out_p, n_seq, H).cpu().numpy() - y).max() < 1e-4
    assert float(panel_unpack(out_p, n_seq, 32 * out_p.shape[1])[:, H:].abs().max()) == 0.0     # columns past n_units untouched


@pytest.mark.parametrize('n_seq,N,K,nb,act,seq_tile', [
    (832, 512, 544, 1, 'relu', 0), (832, 512, 512, 4, None, 128), (70, 30, 64, 1, None, 0), (300, 512, 544, 2, 'relu', 0),
    (17, 32, 96, 2, 'relu', 0),
    # 64 sequences x 32 units per workgroup: full size, a ragged last sequence block and unit group, fewer rows than one tile
    (832, 512, 544, 2, 'relu', 64), (300, 500, 512, 2, None, 64), (70, 30, 64, 2, 'relu', 64), (17, 32, 96, 2, None, 64)])
def test_dense_layer_vs_float64(n_seq, N, K, nb, act, seq_tile):
    from estts import _lib, ops
    from estts.wavernn import pack_dense_stream, panel_alloc, panel_pack, panel_unpack
    rng = np.random.default_rng(n_seq + N + K)
    d = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()
    xp = panel_pack(d(rng.normal(size=(n_seq, K)).astype(np.float32)))
    w = (rng.normal(size=(N, K)) / np.sqrt(K)).astype(np.float32)
    stream, npad = pack_dense_stream(w, nb)
    b = np.zeros(npad, np.float32)
    b[:N] = rng.normal(size=N)
    out = torch.full((n_seq, N), float('nan'), device='cuda')
    out_p = panel_alloc(n_seq, (N + 31) // 32, 'cuda')
    a = _layer_args(_lib, n_seq, N, 0, xp, xp.shape[1], stream.cuda(), d(b), act=_lib.ACT[act], unit_blocks=nb, seq_tile=seq_tile, out_f32=out,
                    ldo=N, out_panel=out_p, out_kpitch=out_p.shape[1])
    _lib.check(_lib.load().estts_wavernn_layer(a, ops._stream()), 'estts_wavernn_layer')
    x64 = panel_unpack(xp, n_seq, K).cpu().numpy().astype(np.float64)
    y = x64 @ w.T.astype(np.float64) + b[:N]
    if act == 'relu':
        y = np.maximum(y, 0)
    assert np.abs(out.cpu().numpy() - y).max() < 5e-5
    assert np.abs(panel_unpack(out_p, n_seq, N).cpu().numpy() - y).max() < 1e-4
    if seq_tile == 64:     # the small tile exists for two unit blocks only
        a.unit_blocks = 1
        with pytest.raises(_lib.EsttsError, match='seq_tile'):
            _lib.check(_lib.load().estts_wavernn_layer(a, ops._stream()), 'estts_wavernn_layer')


def test_panel_pack_round_trip():
    from estts.wavernn import panel_pack, panel_unpack
    x = torch.randn(37, 75, device='cuda')
    p = panel_pack(x)
    assert tuple(p.shape) == (3, 3, 2, 512)
    assert ((panel_unpack(p, 37, 75) - x).abs() / x.abs().clamp(min=1.0)).max().item() < 2e-5      # hi + lo keeps ~16 mantissa bits
