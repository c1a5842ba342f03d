"""Shared by the WaveRNN golden generator and the tests: the small model the fixtures were made with."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import estts  # noqa: E402,F401
from estts import configs  # noqa: E402


def small_hp(mode):
    return configs.wavernn_hparams(voc_mode=mode, bits=5, hop_length=12, voc_upsample_factors=(2, 2, 3), voc_rnn_dims=64,
                                   voc_fc_dims=64, voc_compute_dims=32, voc_res_out_dims=32, voc_res_blocks=3,
                                   voc_target=60, voc_overlap=12, sample_rate=16000)


def small_hp_hop256():
    """The vocoder shape of BASELINE config 4 (hop 256 as upsample factors (4, 8, 8), 22 050 Hz, RAW mu-law) at small widths."""
    return configs.wavernn_hparams_22k(bits=5, voc_rnn_dims=64, voc_fc_dims=64, voc_compute_dims=32, voc_res_out_dims=32,
                                       voc_res_blocks=3, voc_target=600, voc_overlap=64)
