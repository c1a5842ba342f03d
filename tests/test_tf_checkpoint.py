"""Reader of TensorFlow object-based checkpoints (estts/tf_checkpoint.py).  No real TensorFlow checkpoint exists here (none
ships with the reference, TensorFlow is not installable), so the container format is pinned only by published constants
(varint, CRC-32C check value, table magic) and by round trips through the independent writer in tests/tf_bundle_writer.py;
the attribute paths are pinned by known answers read off the reference's constructors."""
import numpy as np
import pytest

import estts  # noqa: F401
from estts import configs, weights
from estts import tf_checkpoint as T
from tests import tf_bundle_writer as W


def small_ar_cfg():
    return configs.autoregressive_config(
        system_type='speaker_style_text', encoder_model_dimension=32, decoder_model_dimension=32, encoder_prenet_dimension=32,
        decoder_prenet_dimension=24, encoder_num_heads=[2, 2, 2], decoder_num_heads=[2, 2, 2], encoder_dense_blocks=1,
        decoder_dense_blocks=2, encoder_feed_forward_dimension=48, decoder_feed_forward_dimension=48,
        encoder_attention_conv_filters=40, decoder_attention_conv_filters=40, ref_encoder_filters=[4, 8],
        ref_encoder_gru_cell_units=16, gst_style_embed_dim=32, gst_multi_num_heads=2, gst_heads=5, speaker_embed_dim=8,
        mel_channels=80, max_r=2, postnet_conv_filters=16, postnet_conv_layers=3, vocab_size=20,
        encoder_max_position_encoding=32, decoder_max_position_encoding=64)


def test_wire_format_constants():
    assert T.crc32c(b'123456789') == 0xe3069283                      # the CRC-32C check value
    assert T.mask_crc(0) == 0xa282ead8
    assert W.varint(300) == b'\xac\x02' and T._varint(b'\xac\x02', 0) == (300, 2)
    assert T.proto_fields(b'\x08\x96\x01') == [(1, 150)]             # the protobuf documentation's example
    assert T.proto_fields(b'\x12\x07testing') == [(2, b'testing')]
    assert T.TABLE_MAGIC == 0xdb4775248b80fb57


def test_attribute_paths_known_answers():
    fwd = {'encoder_dense_blocks': 2, 'decoder_dense_blocks': 0}
    ar = {'encoder_dense_blocks': 4, 'decoder_dense_blocks': 4}
    P = T.object_path
    # pretrained_models/for_text_encoder/models.py:315-351 + model/layers.py
    assert P('encoder_prenet.embeddings', 'forward', fwd) == ['encoder_prenet', 'embeddings']
    assert P('encoder.block1.sarn.mha.wq.kernel', 'forward', fwd) == ['encoder', 'encoder_SADB', '1', 'sarn', 'mha', 'wq', 'kernel']
    assert P('encoder.block3.conv.norm1.moving_mean', 'forward', fwd) == \
        ['encoder', 'encoder_SACB', '1', 'conv', 'normalization', '1', 'moving_mean']
    assert P('decoder.block0.conv.conv0.kernel', 'forward', fwd) == ['decoder', 'encoder_SACB', '0', 'conv', 'convolutions', '0', 'kernel']
    assert P('decoder.block2.conv.last_conv.bias', 'forward', fwd) == ['decoder', 'encoder_SACB', '2', 'conv', 'last_conv', 'bias']
    assert P('dur_pred.conv_blocks.norm0.gamma', 'forward', fwd) == ['dur_pred', 'conv_blocks', 'normalization', '0', 'gamma']
    assert P('dur_pred.linear.kernel', 'forward', fwd) == ['dur_pred', 'linear', 'kernel']
    assert P('decoder_postnet.conv2.kernel', 'forward', fwd) == ['decoder_postnet', 'convolutions', '2', 'kernel']
    assert P('out.bias', 'forward', fwd) == ['out', 'bias']
    # model/AutoregressiveTransformer.py:74-119 + model/layers.py
    assert P('text_encoder.block2.ffn.d2.kernel', 'autoregressive', ar) == ['text_encoder', 'encoder_SADB', '2', 'ffn', 'd2', 'kernel']
    assert P('decoder.block3.carn.layernorm.beta', 'autoregressive', ar) == ['decoder', 'CADB', '3', 'carn', 'layernorm', 'beta']
    assert P('decoder.block5.conv.conv0.kernel', 'autoregressive', ar) == ['decoder', 'CACB', '1', 'conv', 'convolutions', '0', 'kernel']
    assert P('style_encoder.gru.recurrent_kernel', 'autoregressive', ar) == ['style_encoder', 'rnn_net', 'cell', 'recurrent_kernel']
    assert P('style_encoder.rnn_proj.bias', 'autoregressive', ar) == ['style_encoder', 'rnn_proj_net', 'bias']
    assert P('style_encoder.conv1.kernel', 'autoregressive', ar) == ['style_encoder', 'convolutions', '1', 'kernel']
    assert P('style_encoder.gst_tokens', 'autoregressive', ar) == ['style_encoder', 'gst_tokens']
    assert P('decoder_postnet.conv_blocks.norm2.moving_variance', 'autoregressive', ar) == \
        ['decoder_postnet', 'conv_blocks', 'normalization', '2', 'moving_variance']
    assert P('decoder_postnet.stop_linear.kernel', 'autoregressive', ar) == ['decoder_postnet', 'stop_linear', 'kernel']
    assert P('final_proj_mel.kernel', 'autoregressive', ar) == ['final_proj_mel', 'kernel']
    assert P('decoder_prenet.d1.bias', 'autoregressive', ar) == ['decoder_prenet', 'd1', 'bias']


def checkpoint_variables(w, kind, cfg, rng):
    v = {tuple(['net'] + T.object_path(k, kind, cfg)): np.asarray(a) for k, a in w.items()}
    v[('step',)] = np.asarray(12345, np.int64)                                        # what the training scripts save beside net
    v[('net', 'decoder_prenet', 'rate')] = np.asarray(0.5, np.float32)
    v[('optimizer', 'iter')] = np.asarray(12344, np.int64)
    v[('optimizer', 'slots', 'm', '0')] = rng.normal(size=(3, 5)).astype(np.float32)
    return v


@pytest.mark.parametrize('kind,block_bytes', [('autoregressive', 4096), ('autoregressive', 300), ('forward', 1 << 20)])
def test_round_trip(tmp_path, kind, block_bytes):
    rng = np.random.default_rng(0)
    if kind == 'autoregressive':
        cfg = small_ar_cfg()
        w = weights.init_autoregressive_transformer(cfg, seed=3)
    else:
        cfg = configs.forward_config(encoder_model_dimension=32, decoder_model_dimension=24, encoder_num_heads=[2, 2, 2],
                                     decoder_num_heads=[2, 2], encoder_dense_blocks=1, decoder_dense_blocks=0,
                                     encoder_feed_forward_dimension=48, decoder_feed_forward_dimension=40,
                                     encoder_attention_conv_filters=40, decoder_attention_conv_filters=40,
                                     decoder_prenet_dimension=24, encoder_prenet_dimension=32, postnet_conv_filters=16,
                                     postnet_conv_layers=3, duration_conv_filters=[16, 32], vocab_size=20)
        w = weights.init_forward_transformer(cfg, seed=3)
    prefix = str(tmp_path / 'ckpt-7')
    W.write_checkpoint(prefix, checkpoint_variables(w, kind, cfg, rng), block_bytes)
    got, step = T.load_tf_checkpoint(prefix, kind, cfg, verify=True)
    assert step == 12345 and set(got) == set(w)
    for k in w:
        assert got[k].dtype == np.asarray(w[k]).dtype and np.array_equal(got[k], w[k]), k
    b = T.Bundle(prefix)
    assert 'net/decoder_prenet/rate/.ATTRIBUTES/VARIABLE_VALUE' in b.keys() and len(b.keys()) == len(w) + 5
    tree = T.ObjectGraph(b.tensor(T.OBJECT_GRAPH_KEY)).tree()
    assert 'step -> step/.ATTRIBUTES/VARIABLE_VALUE' in tree and len(tree) == len(w) + 4


def test_errors_are_specific(tmp_path):
    cfg = small_ar_cfg()
    w = weights.init_autoregressive_transformer(cfg, seed=1)
    rng = np.random.default_rng(1)
    prefix = str(tmp_path / 'ckpt-1')
    W.write_checkpoint(prefix, checkpoint_variables(w, 'autoregressive', cfg, rng))
    other = dict(cfg, decoder_dense_blocks=3)                                         # block 2 is now expected under CADB
    with pytest.raises(T.CheckpointError, match="has no child '2'"):
        T.load_tf_checkpoint(prefix, 'autoregressive', other)
    wider = dict(cfg, decoder_feed_forward_dimension=64)
    with pytest.raises(T.CheckpointError, match='this configuration expects'):
        T.load_tf_checkpoint(prefix, 'autoregressive', wider)
    with pytest.raises(T.CheckpointError, match="no child 'model'"):
        T.load_tf_checkpoint(prefix, 'autoregressive', cfg, root='model')
    raw = bytearray(open(prefix + '.index', 'rb').read())
    raw[20] ^= 0x40
    open(prefix + '.index', 'wb').write(raw)
    with pytest.raises(T.CheckpointError):
        T.Bundle(prefix)
    with pytest.raises(T.CheckpointError, match='not found'):
        T.Bundle(str(tmp_path / 'absent'))
    data = bytearray(open(prefix + '.data-00000-of-00001', 'rb').read())
    open(prefix + '.index', 'wb').write(bytes(raw[:20]) + bytes([raw[20] ^ 0x40]) + bytes(raw[21:]))
    data[5] ^= 1
    open(prefix + '.data-00000-of-00001', 'wb').write(data)
    with pytest.raises(T.CheckpointError, match='checksum'):
        T.load_tf_checkpoint(prefix, 'autoregressive', cfg, verify=True)


def test_config_manager_finds_and_converts_a_tensorflow_checkpoint(tmp_path):
    import yaml
    from estts.config_manager import ConfigManager
    from tests.test_config_manager import _write_configs
    _write_configs(tmp_path)
    ar = small_ar_cfg()
    ar.update({'learning_rate_tts_schedule': [[0, 1.0e-4]], 'reduction_factor_schedule': [[0, 2]],
               'decoder_prenet_dropout_schedule': [[0, 0.]], 'stop_loss_scaling': 8})
    ar.pop('vocab_size', None)
    ar.pop('max_r', None)
    (tmp_path / 'autoregressive_config.yaml').write_text(yaml.safe_dump(ar))
    cm = ConfigManager(tmp_path, 'autoregressive')
    cm.create_remove_dirs()
    assert cm.latest_checkpoint() is None
    cfg = cm.model_config_dict()
    w = weights.init_autoregressive_transformer(cfg, seed=5)
    W.write_checkpoint(str(cm.weights_dir / 'ckpt-3'), checkpoint_variables(w, 'autoregressive', cfg, np.random.default_rng(2)))
    (cm.weights_dir / 'checkpoint').write_text('model_checkpoint_path: "ckpt-3"\nall_model_checkpoint_paths: "ckpt-3"\n')
    assert cm.latest_checkpoint() == str(cm.weights_dir / 'ckpt-3')
    back, step = cm.read_any_weights(cm.latest_checkpoint())
    assert step == 12345 and all(np.array_equal(back[k], w[k]) for k in w)
    cm.save_weights(str(cm.weights_dir / 'ckpt-20.npz'), w, 20)
    assert cm.latest_checkpoint().endswith('ckpt-20.npz') and cm.read_any_weights(cm.latest_checkpoint())[1] == 20
