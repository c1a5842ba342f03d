"""Oracle checks (CPU): the reference's own known-answer tests, and agreement of the two
independently written restatements (NumPy fp64 vs PyTorch ops)."""
import numpy as np
import pytest

import estts
from estts import configs, weights
from oracle import tts_numpy as O
from oracle import tts_torch as OT
import torch


def small_forward_cfg():
    return configs.forward_config(
        encoder_model_dimension=64, decoder_model_dimension=32, encoder_num_heads=[2, 2], decoder_num_heads=[2, 2],
        encoder_feed_forward_dimension=128, decoder_feed_forward_dimension=96,
        encoder_attention_conv_filters=48, decoder_attention_conv_filters=40,
        postnet_conv_filters=24, postnet_conv_layers=3, mel_channels=16, vocab_size=20,
        encoder_max_position_encoding=64, decoder_max_position_encoding=256)


def test_reference_crossentropy_known_answers():
    # TransformerTTS/tests/test_loss.py:10-24
    targets = np.array([[0, 1, 2]])
    logits = np.array([[[.3, .2, .1], [.3, .2, .1], [.3, .2, .1]]])
    assert O.new_scaled_crossentropy(index=2, scaling=5)(targets, logits) == pytest.approx(2.3705523014068604, abs=1e-6)
    assert O.new_scaled_crossentropy(index=2, scaling=1)(targets, logits) == pytest.approx(0.7679619193077087, abs=1e-6)
    assert O.masked_crossentropy(targets, logits) == pytest.approx(0.7679619193077087, abs=1e-6)


def test_reference_tokenizer_known_answers():
    # TransformerTTS/tests/test_char_tokenizer.py:11-23
    tok = O.Tokenizer(alphabet=list('ab c'))
    assert (tok.start_token_index, tok.end_token_index, tok.vocab_size) == (5, 6, 7)
    assert tok('a b d') == [5, 1, 3, 2, 3, 6]
    assert tok.decode(np.array([5, 1, 3, 2, 8, 6])) == '>a b<'


def test_masked_mae_counts_padding_in_denominator():
    t = np.zeros((1, 4, 2)); t[0, :2] = 1.0
    p = np.zeros((1, 4, 2)); p[0] = 3.0
    # two valid frames, each |1-3| = 2 ; divided by all 4 frames
    assert O.masked_mean_absolute_error(t, p) == pytest.approx(1.0)


def test_schedules():
    assert O.piecewise_linear_schedule(0, [[0, 0.], [25_000, 0.], [35_000, .5]]) == 0.0
    assert O.piecewise_linear_schedule(30_000, [[0, 0.], [25_000, 0.], [35_000, .5]]) == pytest.approx(0.25)
    assert O.piecewise_linear_schedule(50_000, [[0, 0.], [25_000, 0.], [35_000, .5]]) == 0.5
    assert O.reduction_schedule(10, [[0, 10], [80_000, 1]]) == 10
    assert O.reduction_schedule(80_000, [[0, 10], [80_000, 1]]) == 1


def test_expand_matches_docstring_example():
    # TransformerTTS/model/layers.py:650-660
    x = np.array([[[0.54710746, 0.8943467], [0.7140938, 0.97968304], [0.5347662, 0.15213418]]])
    out = O.expand(x, np.array([[[1.], [3.], [2.]]]))
    assert out.shape == (1, 6, 2)
    np.testing.assert_array_equal(out[0, :, 0], x[0, [0, 1, 1, 1, 2, 2], 0])
    # round-half-to-even, ragged batch padded with zeros
    x2 = np.arange(12, dtype=np.float64).reshape(2, 3, 2) + 1
    out2 = O.expand(x2, np.array([[[0.5], [1.5], [2.5]], [[1.0], [0.0], [1.0]]]))
    assert out2.shape == (2, 4, 2)          # rows: 0,2,2 -> 4 ; 1,0,1 -> 2
    assert (out2[1, 2:] == 0).all()


def test_positional_encoding_values():
    pe = O.positional_encoding(50, 16)
    assert pe.shape == (1, 50, 16) and pe.dtype == np.float32
    assert pe[0, 0, 0] == 0.0 and pe[0, 0, 1] == 1.0
    np.testing.assert_allclose(pe[0, 3, 4], np.sin(3 / 10000 ** (4 / 16)), rtol=1e-6)
    np.testing.assert_allclose(pe[0, 3, 5], np.cos(3 / 10000 ** (4 / 16)), rtol=1e-6)


@pytest.mark.parametrize('use_target', [True, False])
def test_forward_transformer_two_restatements_agree(use_target):
    cfg = small_forward_cfg()
    w = weights.init_forward_transformer(cfg, seed=7)
    rng = np.random.default_rng(0)
    tokens = rng.integers(1, cfg['vocab_size'], size=(2, 9))
    tokens[1, 6:] = 0
    durs = rng.integers(1, 4, size=(2, 9)); durs[1, 6:] = 0
    a = O.forward_transformer_call(w, cfg, tokens, durs if use_target else None)
    b = OT.forward_transformer_call(w, cfg, tokens, durs if use_target else None, dtype=torch.float64)
    assert a['mel'].shape == b['mel'].shape
    np.testing.assert_allclose(a['mel'], b['mel'], atol=1e-9)
    np.testing.assert_allclose(a['duration'], b['duration'], atol=1e-9)
    for x, y in zip(a['encoder_attention'] + a['decoder_attention'], b['encoder_attention'] + b['decoder_attention']):
        np.testing.assert_allclose(x, y, atol=1e-9)
    np.testing.assert_array_equal(a['expanded_mask'], b['expanded_mask'])


def test_autoregressive_gst_two_restatements_agree():
    cfg = configs.autoregressive_config(
        system_type='speaker_style_text', encoder_model_dimension=32, decoder_model_dimension=32,
        encoder_prenet_dimension=32, decoder_prenet_dimension=24, encoder_num_heads=[2, 2], decoder_num_heads=[2, 2],
        encoder_dense_blocks=1, decoder_dense_blocks=1, encoder_feed_forward_dimension=48,
        decoder_feed_forward_dimension=48, encoder_attention_conv_filters=40, decoder_attention_conv_filters=40,
        ref_encoder_filters=[4, 4, 8], ref_encoder_gru_cell_units=16, gst_style_embed_dim=32, gst_multi_num_heads=4,
        gst_heads=5, speaker_embed_dim=12, mel_channels=20, max_r=3, postnet_conv_filters=16, postnet_conv_layers=3,
        vocab_size=15, encoder_max_position_encoding=64, decoder_max_position_encoding=256)
    w = weights.init_autoregressive_transformer(cfg, seed=3)
    rng = np.random.default_rng(1)
    tokens = rng.integers(1, 15, size=(2, 7)); tokens[0, 5:] = 0
    targets = rng.normal(size=(2, 11, 20)); targets[1, 8:] = 0
    spk = rng.normal(size=(2, 1, 12))
    for r in (1, 3):
        a = O.autoregressive_call(w, cfg, tokens, targets, spk, r=r)
        b = OT.autoregressive_call(w, cfg, tokens, targets, spk, r=r, dtype=torch.float64)
        for key in ('final_output', 'mel_linear', 'stop_prob', 'gst_output', 'gst_encoder_attention'):
            np.testing.assert_allclose(a[key], b[key], atol=1e-9, err_msg=key)
        assert a['final_output'].shape == (2, 11 * r, 20)


def test_ar_gst_fixture_matches_the_second_restatement(golden_dir):
    """tests/golden/ar_gst_small.npz was written by oracle/tts_numpy.py; the independent PyTorch restatement reproduces it."""
    import os
    import sys
    sys.path.insert(0, os.path.join(golden_dir))
    from make_tts_goldens import ar_gst_inputs
    g = np.load(os.path.join(golden_dir, 'ar_gst_small.npz'))
    cfg, seed, tokens, targets, spk, r = ar_gst_inputs()
    assert seed == int(g['weight_seed']) and r == int(g['r'])
    np.testing.assert_array_equal(tokens, g['tokens'])
    np.testing.assert_array_equal(targets, g['targets'])
    w = weights.init_autoregressive_transformer(cfg, seed=seed)
    b = OT.autoregressive_call(w, cfg, tokens, targets, spk, r=r, dtype=torch.float64)
    for key in ('final_output', 'mel_linear', 'stop_prob', 'gst_output', 'gst_encoder_attention', 'text_enc_output'):
        np.testing.assert_allclose(b[key], g[key], atol=2e-6, err_msg=key)
    np.testing.assert_allclose(b['decoder_attention'][-1], g['last_cross_attention'], atol=2e-6)
