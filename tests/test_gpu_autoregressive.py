"""GST reference encoder + teacher-forced AutoregressiveTransformer on the GPU against the float64 NumPy oracle
(oracle/tts_numpy.py) on identical seeded weights and inputs.  Tolerances (bf16x3): 3e-4 on mel outputs,
1e-3 on attention maps."""
import math

import numpy as np
import pytest
import torch

import estts
from estts import configs, ops, weights
from estts.autoregressive_transformer import AutoregressiveTransformer, ReferenceEncoderGST
from estts.ops import PackedLinear
from oracle import tts_numpy as O

pytestmark = pytest.mark.gpu
DEV = 'cuda'


def small_cfg(system_type, dense=1):
    return configs.autoregressive_config(
        system_type=system_type, encoder_model_dimension=64, decoder_model_dimension=64, encoder_prenet_dimension=64,
        decoder_prenet_dimension=48, encoder_num_heads=[2, 2], decoder_num_heads=[2, 2], encoder_dense_blocks=dense,
        decoder_dense_blocks=dense, encoder_feed_forward_dimension=96, decoder_feed_forward_dimension=96,
        encoder_attention_conv_filters=80, decoder_attention_conv_filters=80, ref_encoder_filters=[8, 8, 16],
        ref_encoder_gru_cell_units=32, gst_style_embed_dim=64, gst_multi_num_heads=2, gst_heads=5, speaker_embed_dim=16,
        mel_channels=80, max_r=3, postnet_conv_filters=32, postnet_conv_layers=3, vocab_size=30,
        encoder_max_position_encoding=64, decoder_max_position_encoding=256)


def test_conv2d_first_layer_and_implicit_gemm_layers():
    rng = np.random.default_rng(0)
    B, H, W = 2, 37, 80
    x = rng.normal(size=(B, H, W)).astype(np.float32)
    k0 = rng.normal(size=(3, 3, 1, 8)).astype(np.float32) * 0.3
    k1 = rng.normal(size=(3, 3, 8, 16)).astype(np.float32) * 0.1
    b0, b1 = rng.normal(size=8).astype(np.float32), rng.normal(size=16).astype(np.float32)
    ref0 = np.maximum(O.conv2d_same(x[..., None].astype(np.float64), k0.astype(np.float64), b0.astype(np.float64), 2), 0)
    ref1 = np.maximum(O.conv2d_same(ref0, k1.astype(np.float64), b1.astype(np.float64), 2), 0)
    Ho, pt = ops.same_pad(H, 3, 2); Wo, pl = ops.same_pad(W, 3, 2)
    y0 = torch.empty(B, Ho, Wo, 8, device=DEV)
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(DEV)
    ops.conv2d_c1(t(x), t(k0.reshape(9, 8)), torch.ones(8, device=DEV), t(b0), y0, 2, pt, pl)
    assert np.abs(y0.cpu().numpy() - ref0).max() < 1e-5
    H2, pt2 = ops.same_pad(Ho, 3, 2); W2, pl2 = ops.same_pad(Wo, 3, 2)
    lin = PackedLinear(k1.reshape(72, 16).T, bias=b1, device=DEV, taps=9, act='relu')
    y1 = torch.empty(B * H2 * W2, 16, device=DEV)
    ops.gemm(y0.view(B * Ho * Wo, 8), lin, y1, M=B * H2 * W2, conv2d=(Ho, Wo, H2, W2, 3, 2, pt2, pl2))
    assert np.abs(y1.cpu().numpy().reshape(B, H2, W2, 16) - ref1).max() < 5e-5


def test_row_zero_mask_and_masked_mae():
    rng = np.random.default_rng(1)
    x = rng.normal(size=(13, 80)).astype(np.float32)
    x[3] = 0; x[9] = 0
    m = torch.empty(13, device=DEV)
    ops.row_zero_mask(torch.from_numpy(x).to(DEV), m)
    assert m.cpu().numpy().tolist() == [1.0 if i in (3, 9) else 0.0 for i in range(13)]
    pred = rng.normal(size=(13, 80)).astype(np.float32)
    out = torch.zeros(1, device=DEV)
    ops.masked_mae(torch.from_numpy(x).to(DEV), torch.from_numpy(pred).to(DEV), out)
    ref = O.masked_mean_absolute_error(x[None], pred[None])
    assert abs(out.item() - ref) < 1e-5


def test_reference_encoder_gst_vs_oracle():
    cfg = small_cfg('style_text')
    w = weights.init_autoregressive_transformer(cfg, seed=5)
    rng = np.random.default_rng(2)
    x = rng.normal(size=(3, 45, 80)).astype(np.float32)
    enc = ReferenceEncoderGST(w, 'style_encoder', cfg, torch.device(DEV))
    out, attn = enc.run(torch.from_numpy(x).to(DEV), 'bf16x3')
    w64 = {k: np.asarray(v, np.float64) for k, v in w.items()}
    ref, ref_attn = O.reference_encoder_gst(x.astype(np.float64), w64, 'style_encoder', 3, 2, cfg['gst_multi_num_heads'])
    assert np.abs(out.cpu().numpy() - ref[:, 0]).max() < 1e-4
    assert np.abs(attn.cpu().numpy() - ref_attn).max() < 1e-4


@pytest.mark.parametrize('system_type,dense,r', [('style_text', 1, 1), ('speaker_style_text', 2, 3), ('speaker_text', 0, 2),
                                                 ('text', 1, 1)])
def test_teacher_forced_call_vs_oracle(system_type, dense, r):
    cfg = small_cfg(system_type, dense)
    w = weights.init_autoregressive_transformer(cfg, seed=7)
    rng = np.random.default_rng(3)
    B, N, Tq = 2, 11, 14
    tokens = rng.integers(1, cfg['vocab_size'], size=(B, N)).astype(np.int32); tokens[1, 8:] = 0
    targets = rng.normal(size=(B, Tq, 80)).astype(np.float32); targets[1, 10:] = 0
    spk = rng.normal(size=(B, 1, 16)).astype(np.float32)
    model = AutoregressiveTransformer(w, cfg)
    model.set_constants(reduction_factor=r, decoder_prenet_dropout=0.0)
    out = model.call(tokens, targets, spk)
    ref = O.autoregressive_call(w, cfg, tokens, targets, spk, r=r)
    assert tuple(out['final_output'].shape) == (B, Tq * r, 80)
    for key, tol in (('final_output', 3e-4), ('mel_linear', 3e-4), ('stop_prob', 3e-4), ('text_enc_output', 3e-4),
                     ('decoder_output', 3e-4)):
        assert np.abs(out[key].cpu().numpy() - ref[key]).max() < tol, key
    names = [f'Decoder_DenseBlock{i + 1}_CrossAttention' for i in range(dense)] + \
            [f'Decoder_ConvBlock{i + 1}_CrossAttention' for i in range(2 - dense)]
    for name, ra in zip(names, ref['decoder_attention']):
        assert np.abs(out['decoder_attention'][name].cpu().numpy() - ra).max() < 1e-3
    if 'style' in system_type:
        assert np.abs(out['gst_output'].cpu().numpy() - ref['gst_output']).max() < 1e-4
        assert np.abs(out['gst_encoder_attention']['RefEncoderGST_attention'].cpu().numpy() - ref['gst_encoder_attention']).max() < 1e-4


def test_shipped_config_shapes_and_batch_independence():
    """BASELINE config 3 dimensions (d 256, 4+4 dense blocks, GST filters 32..128) at a reduced batch: shapes,
    finiteness, and row 0 of a batch equals the same utterance alone."""
    cfg = configs.autoregressive_config(system_type='style_text')
    w = weights.init_autoregressive_transformer(cfg, seed=1234)
    rng = np.random.default_rng(4)
    B, N, Tq = 3, 40, 120
    tokens = rng.integers(1, cfg['vocab_size'], size=(B, N)).astype(np.int32)
    targets = np.clip(rng.normal(0, 1.5, size=(B, Tq, 80)), -4, 4).astype(np.float32)
    model = AutoregressiveTransformer(w, cfg)
    model.set_constants(reduction_factor=1)
    out = model.call(tokens, targets, None)
    assert tuple(out['final_output'].shape) == (B, Tq, 80) and torch.isfinite(out['final_output']).all()
    assert tuple(out['gst_output'].shape) == (B, 1, 256)
    alone = model.call(tokens[:1], targets[:1], None)
    assert (out['final_output'][0] - alone['final_output'][0]).abs().max().item() < 1e-5


def test_predict_loop_vs_oracle():
    """AR predict: same number of decoder steps and the same mel as the oracle loop (stop decided by argmax of the
    stop logits; the stop bias is raised so that the random-weight model stops within a few steps)."""
    cfg = small_cfg('style_text')
    w = weights.init_autoregressive_transformer(cfg, seed=9)
    w['decoder_postnet.stop_linear.bias'] = np.array([0.0, 0.3, 0.25], np.float32)
    rng = np.random.default_rng(5)
    tokens = rng.integers(1, cfg['vocab_size'], size=9)
    ref_mel = np.clip(rng.normal(0, 1.5, size=(30, 80)), -4, 4).astype(np.float32)
    model = AutoregressiveTransformer(w, cfg)
    model.set_constants(reduction_factor=2)
    ref = O.autoregressive_predict(w, cfg, tokens, ref_mel, None, r=2, max_length=24)
    outs = {}
    for cached in (False, True):      # the reference's re-decode-everything loop and the incremental (KV-cached) decoder
        out = outs[cached] = model.predict(tokens, ref_mel, None, max_length=24, encode=False, verbose=False, kv_cache=cached)
        assert out['mel'].shape == ref['mel'].shape
        assert np.abs(out['mel'].cpu().numpy() - ref['mel']).max() < 5e-4
    assert np.abs(outs[True]['mel'].cpu().numpy() - outs[False]['mel'].cpu().numpy()).max() < 2e-4
    for name, a in outs[False]['decoder_attention'].items():
        assert outs[True]['decoder_attention'][name].shape == a.shape
        assert np.abs(outs[True]['decoder_attention'][name].cpu().numpy() - a.cpu().numpy()).max() < 1e-3


@pytest.mark.parametrize('system_type,r,conv_blocks', [('speaker_text', 1, True), ('text', 3, False)])
def test_incremental_predict_matches_full_redecode(system_type, r, conv_blocks):
    """Same mel from the O(T) cached decoder and the O(T^2) loop, with causal-conv decoder blocks and with dense ones,
    run to max_length (the stop class never wins with this bias)."""
    cfg = small_cfg(system_type, dense=1 if conv_blocks else 2)
    w = weights.init_autoregressive_transformer(cfg, seed=21)
    w['decoder_postnet.stop_linear.bias'] = np.array([0.5, 0.0, -0.5], np.float32)
    rng = np.random.default_rng(8)
    tokens = rng.integers(1, cfg['vocab_size'], size=11)
    spk = rng.normal(size=(1, cfg['speaker_embed_dim'])).astype(np.float32) if 'speaker' in system_type else None
    model = AutoregressiveTransformer(w, cfg)
    model.set_constants(reduction_factor=r)
    a = model.predict(tokens, None, spk, max_length=6 * r, encode=False, verbose=False, kv_cache=False)
    b = model.predict(tokens, None, spk, max_length=6 * r, encode=False, verbose=False, kv_cache=True)
    assert a['mel'].shape == b['mel'].shape == (7 * r, 80)
    assert np.abs(a['mel'].cpu().numpy() - b['mel'].cpu().numpy()).max() < 2e-4


@pytest.mark.parametrize('system_type,r,dense,fold', [('style_text', 2, 2, True), ('speaker_text', 1, 2, False), ('text', 3, 1, True)])
def test_step_norms_folded_into_their_consumers_change_nothing(system_type, r, dense, fold):
    """Incremental decoding with the row norms computed inside the one-row products that consume them (the default) against
    the same steps with every norm as a launch of its own.  The two forms of a norm round differently in the last bit (other
    reduction order) and the loop feeds every frame back, so the mels agree to 1e-4 after nine steps (measured 1e-5 .. 3e-5;
    the cached decoder against the full re-decode is held to 2e-4); the folded form issues three launches per dense block fewer."""
    from estts import _lib
    cfg = small_cfg(system_type, dense=dense)
    w = weights.init_autoregressive_transformer(cfg, seed=23)
    w['decoder_postnet.stop_linear.bias'] = np.array([5.0, 0.0, -5.0], np.float32)        # the stop class never wins
    rng = np.random.default_rng(9)
    tokens = rng.integers(1, cfg['vocab_size'], size=12)
    spk = rng.normal(size=(1, cfg['speaker_embed_dim'])).astype(np.float32) if 'speaker' in system_type else None
    ref_mel = rng.normal(size=(40, 80)).astype(np.float32) if 'style' in system_type else None
    outs, launches = [], []
    lib = _lib.load()
    for fused in (True, False):
        model = AutoregressiveTransformer(w, cfg, fold_ffn=fold)
        model.set_constants(reduction_factor=r)
        model.fuse_step_norms = fused
        n0 = lib.estts_launch_count()
        outs.append(model.predict(tokens, ref_mel, spk, max_length=8 * r, encode=False, verbose=False))
        launches.append(lib.estts_launch_count() - n0)
    a, b = outs
    assert a['mel'].shape == b['mel'].shape == (9 * r, 80)
    assert np.abs(a['mel'].cpu().numpy() - b['mel'].cpu().numpy()).max() < 1e-4
    for name in a['decoder_attention']:
        assert np.abs(a['decoder_attention'][name].cpu().numpy() - b['decoder_attention'][name].cpu().numpy()).max() < 1e-4
    # 9 steps; three norms per dense block (the last block's third norm moves into the final projection), one per conv block
    assert launches[1] - launches[0] == 9 * (3 * dense + (2 - dense)), launches


def test_grouped_stop_checks_give_the_same_utterance():
    """predict(stop_check_every=k): the stop decision is read back once per k decoder steps; the utterance ends at the same step
    whatever k is, also when that step falls inside a group or is the very first one."""
    cfg = small_cfg('speaker_text', dense=1)
    w = weights.init_autoregressive_transformer(cfg, seed=21)
    rng = np.random.default_rng(3)
    tokens = rng.integers(1, cfg['vocab_size'], size=9)
    spk = rng.normal(size=(1, cfg['speaker_embed_dim'])).astype(np.float32)
    def run(delta, k):
        w['decoder_postnet.stop_linear.bias'] = np.array([0., 0., delta], np.float32)
        model = AutoregressiveTransformer(w, cfg)
        model.set_constants(reduction_factor=2)
        return model.predict(tokens, None, spk, max_length=40, encode=False, verbose=False, stop_check_every=k)

    # a stop head that ends the utterance somewhere in the middle: raising the bias of the stop class ends it at the step
    # whose stop margin is largest, so stop kernels are drawn until that step is neither the first nor the last, and the bias
    # is bisected between "never" and "at once"
    n = 42
    for seed in range(8):
        w['decoder_postnet.stop_linear.kernel'] = np.random.default_rng(seed).normal(0, 0.3, size=(80, 3)).astype(np.float32)
        lo, hi, mid = -200.0, 200.0, None
        assert run(lo, 1)['mel'].shape[0] == 42 and run(hi, 1)['mel'].shape[0] == 2
        for _ in range(30):
            mid = 0.5 * (lo + hi)
            n = run(mid, 1)['mel'].shape[0]
            if 2 < n < 42:
                break
            lo, hi = (mid, hi) if n == 42 else (lo, mid)
        if 2 < n < 42:
            break
    assert 2 < n < 42, n
    for delta in (-200.0, mid, 200.0):
        outs = [run(delta, k) for k in (1, 3, 8, 64)]
        for o in outs[1:]:
            assert o['mel'].shape == outs[0]['mel'].shape and torch.equal(o['mel'], outs[0]['mel'])
            for name in o['decoder_attention']:
                assert torch.equal(o['decoder_attention'][name], outs[0]['decoder_attention'][name])
    print(f'stop bias {mid:.4f}: utterance of {n} frames')


def test_step_program_equals_eager_launches():
    """The incremental decoder issued through the C step program (launch arguments patched per step) gives bit-identical
    output to the same launches issued one by one."""
    cfg = small_cfg('speaker_text', dense=1)
    w = weights.init_autoregressive_transformer(cfg, seed=4)
    w['decoder_postnet.stop_linear.bias'] = np.array([0.5, 0.0, -0.5], np.float32)
    rng = np.random.default_rng(2)
    tokens = rng.integers(1, cfg['vocab_size'], size=13)
    spk = rng.normal(size=(1, cfg['speaker_embed_dim'])).astype(np.float32)
    model = AutoregressiveTransformer(w, cfg)
    model.set_constants(reduction_factor=2)
    outs = [model._predict_incremental(tokens, None, spk, 18, False, False, program=p) for p in (False, True)]
    assert outs[0]['mel'].shape == outs[1]['mel'].shape == (20, 80)
    assert torch.equal(outs[0]['mel'], outs[1]['mel'])
    for name, a in outs[0]['decoder_attention'].items():
        assert torch.equal(a, outs[1]['decoder_attention'][name])


def wide_decoder_cfg(system_type):
    """A decoder of the shipped width (256, 4 heads of depth 64, 256-unit pre-net, dense blocks): the shapes that the one-launch
    kernels of the incremental decoder take (estts_prenet_row, estts_decoder_block_row, estts_cross_attention_row)."""
    return configs.autoregressive_config(
        system_type=system_type, encoder_model_dimension=64, decoder_model_dimension=256, encoder_prenet_dimension=64,
        decoder_prenet_dimension=256, encoder_num_heads=[2, 2], decoder_num_heads=[4, 4], encoder_dense_blocks=2,
        decoder_dense_blocks=2, encoder_feed_forward_dimension=96, decoder_feed_forward_dimension=160,
        encoder_attention_conv_filters=80, decoder_attention_conv_filters=80, ref_encoder_filters=[8, 8, 16],
        ref_encoder_gru_cell_units=32, gst_style_embed_dim=64, gst_multi_num_heads=2, gst_heads=5, speaker_embed_dim=16,
        mel_channels=80, max_r=3, postnet_conv_filters=32, postnet_conv_layers=3, vocab_size=30,
        encoder_max_position_encoding=64, decoder_max_position_encoding=640)


def _wide_model(w, cfg, r, rows):
    model = AutoregressiveTransformer(w, cfg)
    model.set_constants(reduction_factor=r)
    model.fuse_decoder_block = model.fuse_cross_attention = rows
    return model


@pytest.mark.parametrize('system_type,r,rate', [('speaker_text', 2, 0.0), ('style_text', 3, 0.5), ('text', 1, 0.0)])
def test_decoder_row_kernels_change_nothing_at_the_shipped_width(system_type, r, rate):
    """Incremental decoding at decoder width 256 with a position's pre-net and each decoder block as one launch (the default)
    against the same steps with a launch per layer, and against the full re-decode of the reference's loop
    (AutoregressiveTransformer.py:265-296); with the pre-net dropout on, the two incremental forms draw the same masks."""
    from estts import _lib
    cfg = wide_decoder_cfg(system_type)
    w = weights.init_autoregressive_transformer(cfg, seed=31)
    w['decoder_postnet.stop_linear.bias'] = np.array([5.0, 0.0, -5.0], np.float32)        # the stop class never wins
    rng = np.random.default_rng(12)
    tokens = rng.integers(1, cfg['vocab_size'], size=14)
    spk = rng.normal(size=(1, cfg['speaker_embed_dim'])).astype(np.float32) if 'speaker' in system_type else None
    ref_mel = rng.normal(size=(40, 80)).astype(np.float32) if 'style' in system_type else None
    lib = _lib.load()
    outs, launches = [], []
    for rows in (True, False):
        model = _wide_model(w, cfg, r, rows)
        model.decoder_prenet_dropout = rate
        n0 = lib.estts_launch_count()
        outs.append(model._predict_incremental(tokens, ref_mel, spk, 9 * r, False, False, dropout_seed=77))
        launches.append(lib.estts_launch_count() - n0)
    a, b = outs
    assert a['mel'].shape == b['mel'].shape == (10 * r, 80)
    assert torch.isfinite(a['mel']).all()
    # the row kernels add up a product in another order than the tiled kernels (and the loop feeds every frame back): measured
    # 2e-5 .. 1.2e-4 on frames of magnitude 2 after ten steps
    err = np.abs(a['mel'].cpu().numpy() - b['mel'].cpu().numpy()).max()
    print(f'{system_type} r={r}: max |mel difference| {err:.2e}')
    assert err < 3e-4
    for name in a['decoder_attention']:
        assert np.abs(a['decoder_attention'][name].cpu().numpy() - b['decoder_attention'][name].cpu().numpy()).max() < 1e-4
    # per step: the pre-net 3 -> 1 launches, each of the two dense blocks 7 -> 1 (q | k | v, attention, projection, query,
    # attention, projection, feed-forward product with their norms folded in)
    assert launches[1] - launches[0] == 10 * (2 + 2 * 6), launches
    if rate == 0.0:
        c = _wide_model(w, cfg, r, True).predict(tokens, ref_mel, spk, max_length=9 * r, encode=False, verbose=False, kv_cache=False)
        assert np.abs(a['mel'].cpu().numpy() - c['mel'].cpu().numpy()).max() < 3e-4


def test_decoder_row_kernels_hand_over_to_the_layer_launches_past_256_positions():
    """The row kernels hold a decoder cache of at most 256 rows: a longer utterance takes them for its first 256 positions and
    the launches per layer for the rest (two step programs over the same buffers).  Same utterance as with a launch per layer
    throughout, to the rounding of the different summation orders."""
    cfg = wide_decoder_cfg('text')
    w = weights.init_autoregressive_transformer(cfg, seed=33)
    w['decoder_postnet.stop_linear.bias'] = np.array([5.0, 0.0, -5.0], np.float32)
    tokens = np.random.default_rng(13).integers(1, cfg['vocab_size'], size=10)
    outs = [_wide_model(w, cfg, 1, rows).predict(tokens, None, None, max_length=270, encode=False, verbose=False) for rows in (True, False)]
    a, b = (o['mel'].cpu().numpy() for o in outs)
    assert a.shape == b.shape == (271, 80) and np.isfinite(a).all()
    err = np.abs(a - b).max(axis=1)
    print('max |difference| by position: first 32 %.2e, 224-256 %.2e, last 15 %.2e' % (err[:32].max(), err[224:256].max(), err[256:].max()))
    assert err.max() < 3e-4          # (measured 1.2e-4 early, 5e-5 .. 6e-5 around and past the hand-over)
    for name in outs[0]['decoder_attention']:
        x, y = outs[0]['decoder_attention'][name].cpu().numpy(), outs[1]['decoder_attention'][name].cpu().numpy()
        assert x.shape == y.shape and np.abs(x - y).max() < 3e-4


def _prenet_masks(rate, seed, rows, widths):
    """The scaled keep masks the device generator draws for sites 0 and 1 (rows = b * Tq + t)."""
    from estts import train_ops
    out = []
    for site, n in enumerate(widths):
        m = torch.empty(rows, n, device=DEV)
        train_ops.dropout(torch.ones(rows, n, device=DEV), m, rate, seed, site)
        out.append(m.cpu().numpy().astype(np.float64))
    return out


def test_decoder_prenet_dropout_matches_oracle_on_identical_masks():
    """DecoderPrenet's always-on dropout (layers.py:394-402) at rate 0.5: the teacher-forced call against the oracle fed
    the same masks; about half of the prenet activations are dropped."""
    cfg = small_cfg('style_text')
    w = weights.init_autoregressive_transformer(cfg, seed=13)
    rng = np.random.default_rng(3)
    B, N, Tq, r = 2, 9, 12, 2
    tokens = rng.integers(1, cfg['vocab_size'], size=(B, N)).astype(np.int32)
    targets = np.clip(rng.normal(0, 1.5, size=(B, Tq, 80)), -4, 4).astype(np.float32)
    model = AutoregressiveTransformer(w, cfg)
    model.set_constants(reduction_factor=r, decoder_prenet_dropout=0.5)
    enc, mask, *_ = model.call_encoder(tokens, targets, None)
    out = model.call_decoder(enc, targets, mask, dropout_seed=77)
    m1, m2 = _prenet_masks(0.5, 77, B * Tq, (cfg['decoder_prenet_dimension'], cfg['decoder_model_dimension']))
    assert 0.4 < (m1 == 0).mean() < 0.6 and set(np.unique(m1)) == {0.0, 2.0}
    ref = O.autoregressive_call(w, cfg, tokens, targets, None, r=r,
                                prenet_masks=(m1.reshape(B, Tq, -1), m2.reshape(B, Tq, -1)))
    assert np.abs(out['final_output'].cpu().numpy() - ref['final_output']).max() < 3e-4
    plain = O.autoregressive_call(w, cfg, tokens, targets, None, r=r)
    assert np.abs(plain['final_output'] - ref['final_output']).max() > 1e-2        # the masks do matter
    again = model.call_decoder(enc, targets, mask)                                  # next seed of the model's sequence
    assert (again['final_output'] - out['final_output']).abs().max().item() > 1e-3


def test_incremental_predict_with_prenet_dropout_is_teacher_forced_consistent():
    """Cached decoding with dropout draws position i's mask from generator row i: feeding its own output back through the
    teacher-forced decoder with the same seed reproduces it."""
    cfg = small_cfg('text')
    w = weights.init_autoregressive_transformer(cfg, seed=17)
    w['decoder_postnet.stop_linear.bias'] = np.array([0.5, 0.0, -0.5], np.float32)
    rng = np.random.default_rng(6)
    tokens = rng.integers(1, cfg['vocab_size'], size=10)
    r = 2
    model = AutoregressiveTransformer(w, cfg)
    model.set_constants(reduction_factor=r, decoder_prenet_dropout=0.5)
    outs = [model._predict_incremental(tokens, None, None, 16, False, False, program=p, dropout_seed=5) for p in (True, False)]
    assert torch.equal(outs[0]['mel'], outs[1]['mel'])
    mel = outs[0]['mel']                                                            # (9 r, 80)
    assert mel.shape == (9 * r, 80)
    dec_in = torch.cat([model.start_vec, mel[r - 1::r][:-1]])[None]                # start vector + last frame of each step
    tok = torch.as_tensor(tokens[None].astype(np.int32), device=DEV)
    enc, mask, *_ = model.call_encoder(tok, None, None)
    tf = model.call_decoder(enc, dec_in, mask, dropout_seed=5)
    assert (tf['final_output'][0] - mel).abs().max().item() < 2e-4
    other = model._predict_incremental(tokens, None, None, 16, False, False, dropout_seed=6)
    assert (other['mel'] - mel).abs().max().item() > 1e-3


def test_val_step_losses_and_duration_extraction():
    """The flow of the reference's extract_durations.py:125-185: val_step on (tokens, mel, stop labels), then the last decoder
    block's cross-attention into get_durations_from_alignment.  Losses against the float64 oracle (1e-4); durations, being
    integer work on the same maps, identical to the matrix-form oracle."""
    from estts.alignments import get_durations_from_alignment
    from oracle import alignments_numpy as OA
    cfg = small_cfg('speaker_style_text', dense=1)
    w = weights.init_autoregressive_transformer(cfg, seed=11)
    rng = np.random.default_rng(5)
    B, N, T = 3, 12, 31
    tokens = rng.integers(1, cfg['vocab_size'], size=(B, N)).astype(np.int32)
    tar = rng.normal(size=(B, T, 80)).astype(np.float32)
    stop = np.ones((B, T), np.int32)
    for b, (n, t) in enumerate(((12, 31), (9, 24), (7, 17))):
        tokens[b, n:] = 0
        tar[b, t:] = 0
        stop[b, t - 1] = 2
        stop[b, t:] = 0
    spk = rng.normal(size=(B, 1, 16)).astype(np.float32)
    model = AutoregressiveTransformer(w, cfg)
    model.set_constants(reduction_factor=1, decoder_prenet_dropout=0.0)
    model._compile(stop_scaling=8.0)
    out = model.val_step(tokens, tar, stop, spk)
    ref = O.autoregressive_call(w, cfg, tokens, tar[:, :-1], spk, r=1)
    total, vals = O.weighted_sum_losses((tar[:, 1:], stop[:, 1:], tar[:, 1:]),
                                        (ref['final_output'], ref['stop_prob'], ref['mel_linear']),
                                        [O.masked_mean_absolute_error, O.new_scaled_crossentropy(2, 8.0), O.masked_mean_absolute_error],
                                        [1., 1., 1.])
    assert abs(float(out['tts_loss']) - total) < 1e-4 * max(1.0, abs(total))
    for key, v in zip(('output', 'stop_prob', 'mel_linear'), vals):
        assert abs(float(out['losses'][key]) - v) < 1e-4 * max(1.0, abs(v)), key
    attn = out['decoder_attention']['Decoder_ConvBlock1_CrossAttention']
    assert tuple(attn.shape) == (B, 2, T - 1, N)
    kw = dict(binary=True, fix_jumps=True, fill_gaps=True, fill_mode='max')
    got = get_durations_from_alignment(attn, torch.from_numpy(tar).to(DEV), torch.from_numpy(tokens).to(DEV), **kw)
    want = OA.get_durations_from_alignment(attn.cpu().numpy(), tar, tokens, **kw)
    for b, (n, t) in enumerate(((12, 31), (9, 24), (7, 17))):
        assert got[0][b].sum() == t - 2 and len(got[0][b]) == n - 2
        assert np.array_equal(got[0][b], want[0][b])
        assert np.abs(got[3][b] - want[3][b]).max() < 1e-6
