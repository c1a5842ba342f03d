"""Synthesize: the vocoder wrapper of the reference (WaveRNN/synthesizer_wavernn.py:8-33) over the MI355X WaveRNN."""
import os

import numpy as np
import torch

from . import configs, weights
from .wavernn import WaveRNN


class Synthesize:
    def __init__(self, model_path, hp=None, seed=1, **kw):
        """model_path: a reference `.pyt` state_dict; None = seeded random weights (exercises the pipeline, produces noise);
        hp: the hparams dict (defaults to the shipped WaveRNN/hparams.py values)."""
        self.hp = hp if hp is not None else configs.wavernn_hparams()
        print('\nInitialising WaveRNN Model...\n')
        self.voc_model = WaveRNN.from_hparams(self.hp, **kw)
        if model_path is None:
            print(f'WARNING: no vocoder weights given: random weights (seed {seed})')
            self.voc_model.load_state_dict({k: torch.from_numpy(np.asarray(v)) for k, v in weights.init_wavernn(self.hp, seed=seed).items()})
        elif os.path.exists(model_path):
            self.voc_model.restore(model_path)
        else:
            raise FileNotFoundError(f'{model_path}: no vocoder weights (the reference would start a new training session here)')

    def generate(self, mel, batch_pred=True):
        """mel: [1, feat, T] in the reference's [0, 1] normalisation (host or device) -> float64 waveform."""
        hp = self.hp
        return self.voc_model.generate(mel, batch_pred, hp['voc_target'], hp['voc_overlap'], hp['mu_law'])

    def generate_many(self, mels, batch_pred=True):
        """Several utterances in ONE batched generation (every fold of every utterance is one sequence of the step kernels;
        the reference vocodes them one after the other, synthesize_sentences_speaker.py:56-75).  mels: list of [feat, T] or
        [1, feat, T] mels in [0, 1] (host or device) -> list of float64 waveforms."""
        hp = self.hp
        flat = [m[0] if getattr(m, 'ndim', 2) == 3 else m for m in mels]
        return self.voc_model.generate_batch(flat, batch_pred, hp['voc_target'], hp['voc_overlap'], hp['mu_law'])
