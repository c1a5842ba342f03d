"""ForwardTransformer on the GPU (HIP kernels through the C ABI) against the CPU oracle
(oracle/tts_numpy.py, float64) on the same seeded inputs and weights.

Tolerances (max abs, stated per precision mode):
  bf16x3  mel 2e-4, attention 1e-3, durations 1e-4   (split-bf16 MFMA, ~fp32 accuracy; attention maps
          are exponentials of accumulated logits, hence the wider bound)
  bf16    mel mean-L1 5e-2                             (plain bf16 operands; reported, not the parity mode)
"""
import numpy as np
import pytest
import torch

import estts
from estts import configs, weights
from estts.forward_transformer import ForwardTransformer
from oracle import tts_numpy as O

pytestmark = pytest.mark.gpu


def small_cfg():
    return configs.forward_config(
        encoder_model_dimension=128, decoder_model_dimension=64, encoder_num_heads=[2, 2, 2], decoder_num_heads=[2, 2],
        encoder_feed_forward_dimension=256, decoder_feed_forward_dimension=192,
        encoder_attention_conv_filters=96, decoder_attention_conv_filters=160,
        postnet_conv_filters=48, postnet_conv_layers=3, mel_channels=80, vocab_size=40,
        encoder_max_position_encoding=128, decoder_max_position_encoding=512)


def _inputs(cfg, B, N, seed, ragged=True):
    rng = np.random.default_rng(seed)
    tokens = rng.integers(1, cfg['vocab_size'], size=(B, N)).astype(np.int32)
    durs = rng.integers(0, 5, size=(B, N)).astype(np.int32)
    if ragged:
        for b in range(1, B):
            tokens[b, N - 2 * b:] = 0
            durs[b, N - 2 * b:] = 0
    return tokens, durs


@pytest.mark.parametrize('fold_ffn', [True, False])
def test_small_model_target_durations(fold_ffn):
    cfg = small_cfg()
    w = weights.init_forward_transformer(cfg, seed=11)
    tokens, durs = _inputs(cfg, 3, 21, seed=5)
    model = ForwardTransformer(w, cfg, precision='bf16x3', fold_ffn=fold_ffn)
    out = model.call(tokens, target_durations=durs, training=False)
    ref = O.forward_transformer_call(w, cfg, tokens, durs)
    assert tuple(out['mel'].shape) == ref['mel'].shape
    assert np.abs(out['mel'].cpu().numpy() - ref['mel']).max() < 2e-4
    assert np.abs(out['duration'].cpu().numpy() - ref['duration']).max() < 1e-4
    assert np.array_equal(out['expanded_mask'].cpu().numpy(), ref['expanded_mask'])
    enc = [out['encoder_attention'][k] for k in ('Encoder_DenseBlock1_SelfAttention', 'Encoder_ConvBlock1_SelfAttention',
                                                'Encoder_ConvBlock2_SelfAttention')]
    for a, r in zip(enc, ref['encoder_attention']):
        assert np.abs(a.cpu().numpy() - r).max() < 1e-3
    dec = [out['decoder_attention'][f'Decoder_ConvBlock{i}_SelfAttention'] for i in (1, 2)]
    for a, r in zip(dec, ref['decoder_attention']):
        assert np.abs(a.cpu().numpy() - r).max() < 1e-3


def test_small_model_predicted_durations_and_speed():
    cfg = small_cfg()
    w = weights.init_forward_transformer(cfg, seed=12)
    w['dur_pred.linear.bias'] = np.array([2.3], np.float32)     # longer, non-integer predicted durations
    tokens, _ = _inputs(cfg, 2, 17, seed=6)
    model = ForwardTransformer(w, cfg, precision='bf16x3')
    for scalar in (1.0, 1.0 / 0.8):
        out = model.forward(tokens, durations_scalar=scalar)
        ref = O.forward_transformer_call(w, cfg, tokens, None, durations_scalar=scalar)
        # frame counts are integers derived from rounding: they must agree exactly unless a predicted duration
        # sits within float32 rounding of x.5; the seeds here keep clear of that
        assert tuple(out['mel'].shape) == ref['mel'].shape
        assert np.abs(out['mel'].cpu().numpy() - ref['mel']).max() < 2e-4
        assert np.abs(out['duration'].cpu().numpy() - ref['duration']).max() < 1e-4


def test_graph_call_replays_the_eager_call():
    """graph_call (one hipGraph per shape) gives what call gives, bit for bit, for every new input of that shape: target
    durations, predicted durations with a fixed frame count, a second batch through the same graph, a second shape."""
    cfg = small_cfg()
    w = weights.init_forward_transformer(cfg, seed=11)
    model = ForwardTransformer(w, cfg, precision='bf16x3')
    eager = ForwardTransformer(w, cfg, precision='bf16x3')       # workspaces of its own: graph buffers stay untouched
    for B, N, seed in ((3, 21, 5), (3, 21, 6), (1, 17, 7)):
        tokens, durs = _inputs(cfg, B, N, seed=seed)
        T = 4 * N + 6                                             # fixed per shape: past the longest possible utterance
        ref = eager.call(tokens, target_durations=durs, max_frames=T)
        out = model.graph_call(tokens, target_durations=durs, max_frames=T)
        assert torch.equal(out['mel'], ref['mel']) and torch.equal(out['expanded_mask'], ref['expanded_mask'])
        for k, v in ref['decoder_attention'].items():
            assert torch.equal(out['decoder_attention'][k], v)
        ref = eager.call(tokens, max_frames=T, durations_scalar=1.3)
        out = model.graph_call(tokens, max_frames=T, durations_scalar=1.3)
        assert torch.equal(out['mel'], ref['mel']) and torch.equal(out['duration'], ref['duration'])
    assert len(model._graphs) == 4                                # (3, 21) twice through the same two graphs
    with pytest.raises(ValueError, match='max_frames'):
        model.graph_call(tokens)


def test_bf16_mode_error_is_bounded():
    cfg = small_cfg()
    w = weights.init_forward_transformer(cfg, seed=11)
    tokens, durs = _inputs(cfg, 2, 21, seed=5)
    out = ForwardTransformer(w, cfg, precision='bf16').call(tokens, target_durations=durs)
    ref = O.forward_transformer_call(w, cfg, tokens, durs)
    assert np.abs(out['mel'].cpu().numpy() - ref['mel']).mean() < 5e-2


def test_golden_fixture(golden_dir):
    """Committed golden vectors (tests/golden/forward_small.npz, written by tests/golden/make_tts_goldens.py
    from the float64 oracle): the GPU path must reproduce them without the oracle in the loop."""
    import os
    g = np.load(os.path.join(golden_dir, 'forward_small.npz'))
    cfg = small_cfg()
    w = weights.init_forward_transformer(cfg, seed=int(g['weight_seed']))
    out = ForwardTransformer(w, cfg, precision='bf16x3').call(g['tokens'], target_durations=g['durations'])
    assert np.abs(out['mel'].cpu().numpy() - g['mel']).max() < 2e-4
    assert np.abs(out['duration'].cpu().numpy() - g['duration']).max() < 1e-4


def test_full_size_model_properties():
    """BASELINE config 2 at full size (N=100, T=600): the float64 oracle takes too long to run in a unit test on
    every box, so check it once against the float32 PyTorch restatement (bounded: one forward) and otherwise
    through size-independent properties: batch independence and padding invariance."""
    from oracle import tts_torch as OT
    cfg = configs.forward_config()
    w = weights.init_forward_transformer(cfg, seed=1234)
    rng = np.random.default_rng(0)
    N, T = 100, 600
    tokens = rng.integers(1, cfg['vocab_size'], size=(1, N)).astype(np.int32)
    d = rng.integers(1, 12, size=(1, N)).astype(np.float64)
    d = np.floor(d * T / d.sum()); d[0, 0] += T - d.sum()
    model = ForwardTransformer(w, cfg, precision='bf16x3')
    out = model.call(tokens, target_durations=d)
    assert tuple(out['mel'].shape) == (1, T, 80)
    ref = OT.forward_transformer_call(w, cfg, tokens, d, dtype=torch.float32)
    l1 = np.abs(out['mel'].cpu().numpy() - ref['mel']).mean()
    assert l1 < 1e-4, l1                      # north-star gate is mel-L1 <= 1e-3
    # batch independence: the same utterance as row 0 of a batch of 2 (row 1 shorter, so it carries padded
    # frames) must give the same mel as alone
    tok2 = np.stack([tokens[0], rng.integers(1, 124, size=N).astype(np.int32)])
    d2 = np.stack([d[0], np.full(N, 4.0)])
    out2 = model.call(tok2, target_durations=d2)
    assert tuple(out2['mel'].shape) == (2, T, 80)
    # (not bit-identical: the GEMM picks its workgroup shape, and with it the summation order over K, from the row count)
    assert np.abs(out['mel'][0].cpu().numpy() - out2['mel'][0].cpu().numpy()).max() < 1e-4
    assert out2['expanded_mask'][1, 0, 0, 400:].min().item() == 1.0 and out2['expanded_mask'][0].max().item() == 0.0
    l1_bf16 = np.abs(ForwardTransformer(w, cfg, precision='bf16').call(tokens, target_durations=d)['mel'].cpu().numpy()
                     - ref['mel']).mean()
    print(f'full-size mel-L1 vs fp32 restatement: bf16x3 {l1:.2e}, bf16 {l1_bf16:.2e}')


@pytest.mark.parametrize('in_flight', [1, 2, 4])
def test_predict_many_is_bit_identical_to_one_at_a_time(in_flight):
    """ForwardTransformer.predict_many: sentences of several lengths, repeated so that every lane meets shapes for the first
    time (launch by launch), the second time (capture) and later (replay); every result must equal predict() on the same
    sentence bit for bit, durations and attention maps included."""
    cfg = small_cfg()
    w = weights.init_forward_transformer(cfg, seed=3)
    model = ForwardTransformer(w, cfg)
    rng = np.random.default_rng(5)
    base = [rng.integers(1, cfg['vocab_size'], size=n).astype(np.int32) for n in (23, 31, 23, 17, 40, 31, 23)]
    sentences = base * 3
    single = [model.predict(torch.from_numpy(t)[None], encode=False) for t in base]
    for attn in (False, True):
        many = model.predict_many(sentences, encode=False, in_flight=in_flight, return_attention=attn)
        assert len(many) == len(sentences)
        for i, out in enumerate(many):
            ref = single[i % len(base)]
            assert out['mel'].shape == ref['mel'].shape and torch.equal(out['mel'], ref['mel']), i
            assert torch.equal(out['duration'], ref['duration']) and torch.equal(out['expanded_mask'], ref['expanded_mask'])
            if attn:
                for name, a in ref['decoder_attention'].items():
                    assert torch.equal(out['decoder_attention'][name], a), (i, name)
                for name, a in ref['encoder_attention'].items():
                    assert torch.equal(out['encoder_attention'][name], a), (i, name)
            else:
                assert out['decoder_attention'] == {} and out['encoder_attention'] == {}
    lanes = model._lanes(in_flight)
    assert any(len(l.graphs) > 0 for l in lanes)            # shapes that came back were replayed from graphs


def test_synthesis_script_with_sentences_in_flight(tmp_path, monkeypatch):
    """scripts/synthesize_sentences.py --text-file: predict_many + one batched WaveRNN generation."""
    import importlib.util
    import os
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location('synth_script_many', os.path.join(root, 'scripts', 'synthesize_sentences.py'))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    text = 'D@ kwIk braUn fQks dZVmps oUv@ D@ leIzi dQg @nd kips rVnIN'
    tf = tmp_path / 'sentences.txt'
    tf.write_text('\n'.join([text, text[:40], text + ' ' + text[:20], text]) + '\n')
    out = tmp_path / 'wav.npy'
    monkeypatch.setattr(sys, 'argv', ['synthesize_sentences.py', '--config', os.path.join(root, 'config', 'wavernn'), '--model', 'forward',
                                      '--text-file', str(tf), '--in-flight', '2', '--out', str(out)])
    mod.main()
    wavs = [np.load(tmp_path / f'wav_{i}.npy') for i in range(4)]
    assert all(w.ndim == 1 and len(w) > 1000 and np.isfinite(w).all() for w in wavs)
    assert len(wavs[0]) == len(wavs[3]) and len(wavs[1]) < len(wavs[0]) < len(wavs[2])


def test_end_to_end_synthesis_script(tmp_path, monkeypatch):
    """scripts/synthesize_sentences.py: ConfigManager -> ForwardTransformer.predict -> (mel + 4) / 8 hand-off -> WaveRNN,
    on seeded random weights (checks the plumbing: shapes, finiteness, that nothing leaves the GPU path)."""
    import importlib.util
    import os
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location('synth_script', os.path.join(root, 'scripts', 'synthesize_sentences.py'))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    out = tmp_path / 'wav.npy'
    text = 'D@ kwIk braUn fQks dZVmps oUv@ D@ leIzi dQg @nd kips rVnIN'
    monkeypatch.setattr(sys, 'argv', ['synthesize_sentences.py', '--config', os.path.join(root, 'config', 'wavernn'), '--model', 'forward',
                                      '--text', text, '--out', str(out)])
    mod.main()
    wav = np.load(out)
    assert wav.ndim == 1 and len(wav) > 1000 and np.isfinite(wav).all() and np.abs(wav).max() <= 1.0 + 1e-6


def test_reference_keyword_constructors():
    """The reference builds its models from keywords (ConfigManager.get_model, utils/config_manager.py:119-162; ForwardTransformer,
    models.py:284-309): the same call works here and equals the (weights, config) form on the same seed."""
    from estts.autoregressive_transformer import AutoregressiveTransformer
    cfg = small_cfg()
    kw = dict(encoder_model_dimension=cfg['encoder_model_dimension'], decoder_model_dimension=cfg['decoder_model_dimension'],
              dropout_rate=0.1, decoder_num_heads=cfg['decoder_num_heads'], encoder_num_heads=cfg['encoder_num_heads'],
              encoder_maximum_position_encoding=cfg['encoder_max_position_encoding'],
              decoder_maximum_position_encoding=cfg['decoder_max_position_encoding'],
              postnet_conv_filters=cfg['postnet_conv_filters'], postnet_conv_layers=cfg['postnet_conv_layers'], postnet_kernel_size=5,
              encoder_dense_blocks=cfg['encoder_dense_blocks'], decoder_dense_blocks=cfg['decoder_dense_blocks'], mel_channels=80,
              phoneme_language='en', with_stress=False, encoder_attention_conv_filters=cfg['encoder_attention_conv_filters'],
              decoder_attention_conv_filters=cfg['decoder_attention_conv_filters'], encoder_attention_conv_kernel=3,
              decoder_attention_conv_kernel=3, encoder_feed_forward_dimension=cfg['encoder_feed_forward_dimension'],
              decoder_feed_forward_dimension=cfg['decoder_feed_forward_dimension'], debug=False, vocab_size=cfg['vocab_size'])
    a = ForwardTransformer(seed=5, **kw)
    b = ForwardTransformer(weights.init_forward_transformer(a.config, seed=5), a.config)
    tokens, durs = _inputs(cfg, 2, 11, seed=3)
    assert torch.equal(a.call(tokens, target_durations=durs)['mel'], b.call(tokens, target_durations=durs)['mel'])
    ar = AutoregressiveTransformer(system_type='text', mel_channels=80, encoder_model_dimension=64, decoder_model_dimension=64,
                                   encoder_num_heads=[2, 2], decoder_num_heads=[2, 2], encoder_feed_forward_dimension=96,
                                   decoder_feed_forward_dimension=96, encoder_maximum_position_encoding=64,
                                   decoder_maximum_position_encoding=256, encoder_dense_blocks=1, decoder_dense_blocks=1,
                                   decoder_prenet_dimension=48, encoder_prenet_dimension=64, encoder_attention_conv_kernel=3,
                                   decoder_attention_conv_kernel=3, encoder_attention_conv_filters=80,
                                   decoder_attention_conv_filters=80, postnet_conv_filters=32, postnet_conv_layers=3,
                                   postnet_kernel_size=5, dropout_rate=0.1, ref_encoder_filters=[8, 8, 16], ref_encoder_kernel_size=3,
                                   ref_encoder_strides=2, ref_encoder_gru_cell_units=32, gst_style_embed_dim=64, gst_multi_num_heads=2,
                                   gst_heads=5, max_r=3, mel_start_value=4, mel_end_value=-4, phoneme_language='en',
                                   with_stress=False, debug=False, vocab_size=30)
    out = ar.call(np.array([[3, 4, 5, 0]], np.int32), np.zeros((1, 6, 80), np.float32) + 0.5, None)
    assert tuple(out['final_output'].shape) == (1, 18, 80) and torch.isfinite(out['final_output']).all()
    with pytest.raises(TypeError):
        ForwardTransformer(None, cfg, bogus=1)
