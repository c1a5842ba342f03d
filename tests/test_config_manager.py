"""Host-side glue that mirrors the reference's ConfigManager / schedules (no GPU needed)."""
import numpy as np
import pytest
import yaml

import estts  # noqa: F401
from estts import configs, scheduling
from estts.config_manager import ConfigManager
from oracle import tts_numpy as O


def _write_configs(tmp_path):
    fwd = configs.forward_config()
    fwd.update({'dropout_rate': 0.1, 'decoder_prenet_dropout_schedule': [[0, 0.]], 'learning_rate_schedule': [[0, 1.0e-4]],
                'head_drop_schedule': [[0, 0]], 'max_steps': 10, 'batch_size': 4, 'weights_save_frequency': 5, 'debug': False})
    fwd.pop('vocab_size', None)
    ar = configs.autoregressive_config()
    ar.update({'learning_rate_tts_schedule': [[0, 1.0e-4]], 'reduction_factor_schedule': [[0, 10], [80_000, 5], [150_000, 3]],
               'decoder_prenet_dropout_schedule': [[0, 0.]], 'stop_loss_scaling': 8})
    ar.pop('vocab_size', None)
    ar.pop('max_r', None)
    data = {'log_directory': str(tmp_path / 'logs'), 'train_data_directory': None, 'data_directory': str(tmp_path), 'session_name': None,
            'mel_channels': 80, 'mel_start_value': .5, 'mel_end_value': -.5, 'phoneme_language': 'en', 'with_stress': False}
    (tmp_path / 'forward_config.yaml').write_text(yaml.safe_dump(fwd))
    (tmp_path / 'autoregressive_config.yaml').write_text(yaml.safe_dump(ar))
    (tmp_path / 'data_config.yaml').write_text(yaml.safe_dump(data))


def test_config_manager_surface(tmp_path):
    _write_configs(tmp_path)
    with pytest.raises(TypeError):
        ConfigManager(tmp_path, 'tacotron')
    cm = ConfigManager(tmp_path, 'forward', session_name='s1')
    assert cm.session_name.endswith('_s1') and cm.learning_rate == np.float32(1e-4)
    assert cm.model_config_dict()['vocab_size'] == configs.forward_config()['vocab_size']
    assert cm.new_adam(1e-4) == {'learning_rate': 1e-4, 'beta_1': 0.9, 'beta_2': 0.98, 'epsilon': 1e-9}
    cm.create_remove_dirs()
    cm.dump_config()
    assert (cm.base_dir / 'forward_config.yaml').exists() and cm.weights_dir.is_dir()
    w = {'a.kernel': np.arange(6, dtype=np.float32).reshape(2, 3), 'a.bias': np.zeros(3, np.float32)}
    for step in (5, 10, 100):
        cm.save_weights(str(cm.weights_dir / f'ckpt-{step}.npz'), w, step)
    assert cm.latest_checkpoint().endswith('ckpt-100.npz')
    back, step = cm.read_weights(cm.latest_checkpoint())
    assert step == 100 and all(np.array_equal(back[k], w[k]) for k in w)
    ar = ConfigManager(tmp_path, 'autoregressive')
    assert ar.max_r == 10 and ar.stop_scaling == 8
    assert ar.model_config_dict()['vocab_size'] == configs.autoregressive_config()['vocab_size']


def test_schedules_match_the_oracle_restatement():
    sched = [[0, 1.0e-4], [40_000, 5.0e-5], [100_000, 1.0e-5]]
    red = [[0, 10], [80_000, 5], [150_000, 3]]
    for step in (0, 1, 39_999, 40_000, 70_000, 100_000, 250_000):
        assert scheduling.piecewise_linear_schedule(step, sched) == np.float32(O.piecewise_linear_schedule(step, sched))
        assert scheduling.reduction_schedule(step, red) == O.reduction_schedule(step, red)


def test_shipped_autoregressive_config_has_the_reference_schedules():
    """config/wavernn/autoregressive_config.yaml carries the reference's schedule values
    (TransformerTTS/config/wavernn/autoregressive_config.yaml:51-64: decoder-prenet dropout 0 until 25 000 then to 0.5 at
    35 000; one head dropped from 15 000; reduction factor 10 until 80 000 then 1; system_type speaker_text), so that a
    reference checkpoint restored at step s decodes with the reference's constants for s."""
    import os
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    cm = ConfigManager(os.path.join(root, 'config', 'wavernn'), 'autoregressive')
    c = cm.config
    assert c['system_type'] == 'speaker_text' and c['use_mine'] is False and cm.max_r == 10
    for step, r, drop, heads in ((0, 10, 0.0, 0), (20_000, 10, 0.0, 1), (30_000, 10, 0.25, 1), (79_999, 10, 0.5, 1),
                                 (80_000, 1, 0.5, 1), (100_000, 1, 0.5, 1)):
        assert scheduling.reduction_schedule(step, c['reduction_factor_schedule']) == r
        assert abs(float(scheduling.piecewise_linear_schedule(step, c['decoder_prenet_dropout_schedule'])) - drop) < 1e-6
        assert scheduling.reduction_schedule(step, c['head_drop_schedule']) == heads
    f = ConfigManager(os.path.join(root, 'config', 'wavernn'), 'forward').config
    assert f['batch_size'] == 16 and f['weights_save_frequency'] == 5000 and f['decoder_prenet_dropout_schedule'] == [[0, 0.0]]
