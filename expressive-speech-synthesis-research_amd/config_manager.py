"""ConfigManager with the reference's surface (TransformerTTS/utils/config_manager.py:13-225): reads
`<config_path>/data_config.yaml` + `<config_path>/{autoregressive,forward}_config.yaml`, builds the model, compiles it,
restores weights.

Differences forced by the substrate: the models are this package's MI355X classes; weights are flat `name -> ndarray`
dicts stored as `.npz` (TensorFlow object-graph checkpoints cannot be read without TensorFlow); without a checkpoint the
model is initialised from `seed`; a forward model for training is a ForwardTransformerTrainer (`get_model(training=True)`),
whose `weights()` feed the inference class.
"""
import glob
import os
import subprocess
from pathlib import Path

import numpy as np
import yaml

from . import weights as _weights
from .scheduling import piecewise_linear_schedule, reduction_schedule


class ConfigManager:
    def __init__(self, config_path, model_kind, session_name=None):
        if model_kind not in ['autoregressive', 'forward']:
            raise TypeError(f"model_kind must be in {['autoregressive', 'forward']}")
        self.config_path = Path(config_path)
        self.model_kind = model_kind
        self.config, self.data_config, self.model_config = self._load_config()
        self.git_hash = self._get_git_hash()
        if session_name is None and self.config.get('session_name') is None:
            session_name = self.git_hash
        self.session_name = '_'.join(filter(None, [self.config_path.name, session_name]))
        self.base_dir = Path(self.config.get('log_directory') or '.') / self.session_name
        self.log_dir = self.base_dir / f'{model_kind}_logs'
        self.weights_dir = self.tts_weights_dir = self.base_dir / f'{model_kind}_weights'
        self.train_datadir = Path(self.config.get('train_data_directory') or self.config.get('data_directory') or '.')
        lr_key = 'learning_rate_tts_schedule' if 'learning_rate_tts_schedule' in self.config else 'learning_rate_schedule'
        self.learning_rate = np.float32(np.array(self.config[lr_key])[0, 1])
        if model_kind == 'autoregressive':
            self.max_r = int(np.array(self.config['reduction_factor_schedule'])[0, 1])
            self.stop_scaling = self.config.get('stop_loss_scaling', 1.)

    def _load_config(self):
        with open(self.config_path / 'data_config.yaml') as f:
            data_config = yaml.safe_load(f)
        with open(self.config_path / f'{self.model_kind}_config.yaml') as f:
            model_config = yaml.safe_load(f)
        all_config = {}
        all_config.update(model_config)
        all_config.update(data_config)
        return all_config, data_config, model_config

    @staticmethod
    def _get_git_hash():
        try:
            return subprocess.check_output(['git', 'describe', '--always'], stderr=subprocess.DEVNULL).strip().decode()
        except Exception:
            return None

    def print_config(self):
        print('\nCONFIGURATION', self.session_name)
        for k, v in self.config.items():
            print('-', k, ':', v)

    def update_config(self):
        for c in (self.config, self.model_config):
            c['git_hash'] = self.git_hash
            c['session_name'] = self.session_name
        self.data_config['session_name'] = self.session_name

    def dump_config(self):
        self.update_config()
        self.base_dir.mkdir(parents=True, exist_ok=True)
        with open(self.base_dir / f'{self.model_kind}_config.yaml', 'w') as f:
            yaml.safe_dump(self.model_config, f)
        with open(self.base_dir / 'data_config.yaml', 'w') as f:
            yaml.safe_dump(self.data_config, f)

    def create_remove_dirs(self, clear_dir=False, clear_logs=False, clear_weights=False):
        import shutil
        self.base_dir.mkdir(parents=True, exist_ok=True)
        if clear_dir or clear_logs:
            shutil.rmtree(self.log_dir, ignore_errors=True)
        if clear_dir or clear_weights:
            shutil.rmtree(self.weights_dir, ignore_errors=True)
        self.log_dir.mkdir(exist_ok=True)
        self.weights_dir.mkdir(exist_ok=True)

    # ------------------------------------------------------------------ models
    def model_config_dict(self):
        """The yaml keys the model classes read, plus what the reference derives (vocabulary size, max_r)."""
        cfg = dict(self.config)
        from .text import default_alphabet
        n_sym = len(default_alphabet())
        if self.model_kind == 'autoregressive':
            cfg.setdefault('vocab_size', n_sym + 3)          # pad + start / end tokens (tokenizer.py:4-27)
            cfg['max_r'] = self.max_r
            cfg.setdefault('speaker_embed_dim', 256)       # the reference's speaker embeddings are 256-d (synthesize_sentences_speaker.py:65)
        else:
            cfg.setdefault('vocab_size', n_sym + 1)
        return cfg

    def get_model(self, ignore_hash=False, weights=None, seed=1234, training=False, **kw):
        cfg = self.model_config_dict()
        if self.model_kind == 'autoregressive':
            w = weights if weights is not None else _weights.init_autoregressive_transformer(cfg, seed=seed)
            if training:
                from .autoregressive_training import AutoregressiveTransformerTrainer
                return AutoregressiveTransformerTrainer(w, cfg, seed=seed, **kw)
            from .autoregressive_transformer import AutoregressiveTransformer
            return AutoregressiveTransformer(w, cfg, **kw)
        w = weights if weights is not None else _weights.init_forward_transformer(cfg, seed=seed)
        if training:
            from .forward_training import ForwardTransformerTrainer
            return ForwardTransformerTrainer(w, cfg, seed=seed, **kw)
        from .forward_transformer import ForwardTransformer
        return ForwardTransformer(w, cfg, **kw)

    @staticmethod
    def new_adam(learning_rate):
        """The optimizer settings of the reference (config_manager.py:172-176) as the dict `_compile` accepts."""
        return {'learning_rate': float(learning_rate), 'beta_1': 0.9, 'beta_2': 0.98, 'epsilon': 1e-9}

    def compile_model(self, model):
        if self.model_kind == 'autoregressive':          # config_manager.py:165-166
            model._compile(stop_scaling=self.config.get('stop_loss_scaling', 1.), optimizer=self.new_adam(self.learning_rate))
        elif hasattr(model, '_compile'):
            model._compile(optimizer=self.new_adam(self.learning_rate))

    @staticmethod
    def save_weights(path, weights, step=0, optimizer_state=None):
        """optimizer_state: the trainer's optimizer_state() (Adam moments, step count, seed), stored beside the variables under
        keys of the form __name__ as the reference's tf.train.Checkpoint(step, optimizer, net) stores its optimizer."""
        os.makedirs(os.path.dirname(os.path.abspath(path)), exist_ok=True)
        np.savez(path, __step__=np.int64(step), **{k: np.asarray(v) for k, v in weights.items()}, **(optimizer_state or {}))

    @staticmethod
    def read_weights(path, with_optimizer=False):
        with np.load(path, allow_pickle=False) as z:
            w = {k: z[k] for k in z.files if not k.startswith('__')}
            step = int(z['__step__']) if '__step__' in z.files else 0
            opt = {k: z[k] for k in z.files if k.startswith('__') and k != '__step__'}
        return (w, step, opt) if with_optimizer else (w, step)

    def latest_checkpoint(self):
        """Newest ckpt-<step>.npz of this package, else the TensorFlow checkpoint prefix the reference's CheckpointManager
        recorded in <weights_dir>/checkpoint (config_manager.py:206-211)."""
        files = sorted(glob.glob(str(self.weights_dir / 'ckpt-*.npz')), key=lambda p: int(p.rsplit('-', 1)[1][:-4]))
        if files:
            return files[-1]
        state = self.weights_dir / 'checkpoint'
        if state.exists():
            for line in state.read_text().splitlines():
                if line.startswith('model_checkpoint_path:'):
                    prefix = line.split(':', 1)[1].strip().strip('"')
                    return prefix if os.path.isabs(prefix) else str(self.weights_dir / prefix)
        return None

    def read_any_weights(self, path):
        """(.npz of this package) or (TensorFlow object-based checkpoint prefix, converted on the fly) -> (weights, step)."""
        if not path.endswith('.npz') and os.path.exists(path + '.index'):
            from .tf_checkpoint import load_tf_checkpoint
            w, step = load_tf_checkpoint(path, self.model_kind, self.model_config_dict())
            if step is None:                                     # tf.train.Checkpoint(net=model) alone: the prefix carries it
                tail = path.rsplit('-', 1)[-1]
                step = int(tail) if tail.isdigit() else 0
            return w, step
        return self.read_weights(path)

    def load_model(self, checkpoint_path=None, verbose=True, training=False, **kw):
        path = checkpoint_path or self.latest_checkpoint()
        w, step = (None, 0)
        if path is None:
            print(f'WARNING: could not find weights file. Trying to load from \n {self.weights_dir}.')
        else:
            w, step = self.read_any_weights(path)
        model = self.get_model(weights=w, training=training, **kw)
        self.compile_model(model)
        if training:
            model.iterations = step
            if path and path.endswith('.npz'):
                opt = self.read_weights(path, with_optimizer=True)[2]
                if '__adam_m__' in opt:                 # resume Adam where it stopped (moments, step count, dropout seed)
                    model.load_optimizer_state(opt)
        if verbose and path:
            print(f'restored weights from {path} at step {step}')
        if training and self.model_kind == 'autoregressive':
            return model                     # the training loop sets the schedules' values before every step
        const = {}
        if 'decoder_prenet_dropout_schedule' in self.config:
            const['decoder_prenet_dropout'] = float(piecewise_linear_schedule(step, self.config['decoder_prenet_dropout_schedule']))
        if self.model_kind == 'autoregressive':
            const['reduction_factor'] = reduction_schedule(step, self.config['reduction_factor_schedule'])
        try:
            model.set_constants(**const)
        except NotImplementedError as e:            # e.g. a decoder-prenet dropout rate this build does not run
            print(f'WARNING: {e}')
        return model
