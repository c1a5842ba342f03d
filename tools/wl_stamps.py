#!/usr/bin/env python3
"""Where a WaveRNN layer launch spends its time: wall-clock stamps of the first and the last workgroup (library built with
-DESTTS_WL_STAMPS; diagnostic only)."""
import ctypes
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import estts  # noqa
from estts import configs, weights, _lib
from estts.wavernn import WaveRNN

if os.environ.get('WL_LIB'):
    _lib.LIB_PATH = os.path.abspath(os.environ['WL_LIB'])

hp = configs.wavernn_hparams_22k()
m = WaveRNN.from_hparams(hp)
m.load_state_dict(weights.init_wavernn(hp, seed=1234))
rng = np.random.default_rng(1234)
mels = [rng.uniform(0, 1, size=(hp['num_mels'], 560)).astype(np.float32) for _ in range(64)]
m.generate_batch(mels, True, 11000, 550, True, seed=1)
torch.cuda.synchronize()
lib = _lib.load()
buf = (ctypes.c_ulonglong * 48)()
lib.estts_debug_wl_stamps.argtypes = [ctypes.c_void_p]
print('rc', lib.estts_debug_wl_stamps(buf))
a = np.array(list(buf), dtype=np.int64).reshape(3, 2, 8)
for mode, name in ((1, 'rnn1'), (2, 'rnn2'), (0, 'dense')):
    for w, which in ((0, 'first wg'), (1, 'last wg')):
        t = a[mode, w]
        print(name, which, 'ns from start:', [int((t[i] - t[0]) * 10) for i in range(1, 7)])
    print(name, 'last wg start - first wg start (ns):', int((a[mode, 1, 0] - a[mode, 0, 0]) * 10), ' last end - first start:', int((a[mode, 1, 6] - a[mode, 0, 0]) * 10))
