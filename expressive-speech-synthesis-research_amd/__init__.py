"""MI355X-native acoustic + vocoder hot path for the expressive-TTS reference.

Host code is Python on PyTorch-ROCm (device memory, streams, torch.distributed);
all compute on the hot path runs in hand-written HIP kernels for gfx950 reached
through the C-ABI shared library `libestts.so` (see include/estts.h).
"""
from . import configs, weights  # noqa: F401

__all__ = ['configs', 'weights']
