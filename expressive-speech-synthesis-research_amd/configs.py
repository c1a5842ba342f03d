"""Model configurations, with the key names and values the reference ships.

forward_config():        TransformerTTS/config/wavernn/forward_config.yaml:2-20
autoregressive_config(): TransformerTTS/config/wavernn/autoregressive_config.yaml:2-44
wavernn_hparams():       WaveRNN/hparams.py:15-54
"""
import copy

# number of phoneme symbols: TransformerTTS/preprocessing/text/symbols.py:1-12
# (len(set(phonemes)) + punctuation) = 123 symbols -> vocab 124 (+2 with start/end tokens)
VOCAB_SIZE_FORWARD = 124
VOCAB_SIZE_AUTOREGRESSIVE = 126

_FORWARD = {
    'decoder_model_dimension': 256,
    'encoder_model_dimension': 512,
    'decoder_num_heads': [4, 4, 4, 4],
    'encoder_num_heads': [4, 4, 4, 4],
    'encoder_feed_forward_dimension': 1024,
    'decoder_feed_forward_dimension': 1024,
    'decoder_prenet_dimension': 256,
    'encoder_prenet_dimension': 512,
    'encoder_attention_conv_filters': 512,
    'decoder_attention_conv_filters': 512,
    'encoder_attention_conv_kernel': 3,
    'decoder_attention_conv_kernel': 3,
    'encoder_max_position_encoding': 1000,
    'decoder_max_position_encoding': 10000,
    'postnet_conv_filters': 256,
    'postnet_conv_layers': 5,
    'postnet_kernel_size': 5,
    'encoder_dense_blocks': 1,
    'decoder_dense_blocks': 0,
    'mel_channels': 80,
    'vocab_size': VOCAB_SIZE_FORWARD,
    'dropout_rate': 0.1,
    'decoder_prenet_dropout_schedule': [[0, 0.]],
    'learning_rate_schedule': [[0, 1.0e-4]],
    'head_drop_schedule': [[0, 0]],
    'max_steps': 400_000,
    'batch_size': 16,
}

_AUTOREGRESSIVE = {
    'system_type': 'style_text',
    'decoder_model_dimension': 256,
    'encoder_model_dimension': 256,
    'decoder_num_heads': [4, 4, 4, 4],
    'encoder_num_heads': [4, 4, 4, 4],
    'encoder_feed_forward_dimension': 1024,
    'decoder_feed_forward_dimension': 1024,
    'decoder_prenet_dimension': 256,
    'encoder_prenet_dimension': 256,
    'encoder_attention_conv_filters': 256,
    'decoder_attention_conv_filters': 256,
    'encoder_attention_conv_kernel': 3,
    'decoder_attention_conv_kernel': 3,
    'encoder_max_position_encoding': 1000,
    'decoder_max_position_encoding': 10000,
    'postnet_conv_filters': 256,
    'postnet_conv_layers': 5,
    'postnet_kernel_size': 5,
    'encoder_dense_blocks': 4,
    'decoder_dense_blocks': 4,
    'ref_encoder_filters': [32, 32, 64, 64, 128, 128],
    'ref_encoder_kernel_size': 3,
    'ref_encoder_strides': 2,
    'ref_encoder_gru_cell_units': 128,
    'gst_style_embed_dim': 256,
    'gst_multi_num_heads': 4,
    'gst_heads': 10,
    'speaker_embed_dim': 256,
    'mel_channels': 80,
    'mel_start_value': 4,
    'mel_end_value': -4,
    'max_r': 10,
    'vocab_size': VOCAB_SIZE_AUTOREGRESSIVE,
    'stop_loss_scaling': 8,
    'dropout_rate': 0.1,
}

_WAVERNN = {
    'sample_rate': 16000,
    'num_mels': 80,
    'n_fft': 2048,                 # WaveRNN/hparams.py:16-23 (DSP)
    'hop_length': 200,
    'win_length': 800,
    'fmin': 40,
    'min_level_db': -100,
    'ref_level_db': 20,
    'bits': 9,
    'mu_law': True,
    'voc_mode': 'MOL',
    'voc_upsample_factors': (5, 5, 8),
    'voc_rnn_dims': 512,
    'voc_fc_dims': 512,
    'voc_compute_dims': 128,
    'voc_res_out_dims': 128,
    'voc_res_blocks': 10,
    'voc_pad': 2,
    'voc_batch_size': 64,
    'voc_lr': 1e-4,
    'voc_checkpoint_every': 25_000,    # WaveRNN/hparams.py:42-45 (training cadence)
    'voc_gen_at_checkpoint': 5,
    'voc_total_steps': 2_000_000,
    'voc_model_id': 'test',
    'voc_seq_len': 1000,
    'voc_gen_batched': True,
    'voc_target': 11_000,
    'voc_overlap': 550,
}


def forward_config(**overrides):
    c = copy.deepcopy(_FORWARD)
    c.update(overrides)
    return c


def autoregressive_config(**overrides):
    c = copy.deepcopy(_AUTOREGRESSIVE)
    c.update(overrides)
    return c


def wavernn_hparams(**overrides):
    c = copy.deepcopy(_WAVERNN)
    c.update(overrides)
    return c


def wavernn_hparams_22k(**overrides):
    """LJSpeech-shaped variant: hop 256 @ 22 050 Hz (TransformerTTS/config/melgan/data_config.yaml:16-19),
    RAW 9-bit mu-law output."""
    c = wavernn_hparams(sample_rate=22050, hop_length=256, n_fft=1024, win_length=1024, fmin=0, voc_upsample_factors=(4, 8, 8), voc_mode='RAW',
                        voc_seq_len=256 * 5)
    c.update(overrides)
    return c


def config_from_keywords(kind, **kw):
    """The reference constructs its models from keywords (utils/config_manager.py:119-162 for AutoregressiveTransformer,
    pretrained_models/for_text_encoder/models.py:284-309 for ForwardTransformer): the same keywords -> this package's config
    dict.  Keys the hot path does not use (phoneme_language, with_stress, debug) are accepted and kept."""
    base = autoregressive_config() if kind == 'autoregressive' else forward_config()
    rename = {'encoder_maximum_position_encoding': 'encoder_max_position_encoding',
              'decoder_maximum_position_encoding': 'decoder_max_position_encoding'}
    for k, v in kw.items():
        if v is not None:
            base[rename.get(k, k)] = v
    if 'decoder_prenet_dimension' not in kw and kind == 'forward':
        # ForwardTransformer's prenet is DecoderPrenet(model_dim=decoder_model_dimension, dense_hidden_units=decoder_feed_forward_dimension)
        base['decoder_prenet_dimension'] = base['decoder_model_dimension']
    return base
