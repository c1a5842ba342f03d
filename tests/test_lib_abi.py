"""CPU-side checks of the C ABI boundary: the shared library loads and exports every symbol that
include/estts.h declares (no compute is launched here)."""
import ctypes
import os
import re

import pytest

import estts
from estts import _lib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared_symbols():
    text = open(os.path.join(ROOT, 'include', 'estts.h')).read()
    text = re.sub(r'/\*.*?\*/', '', text, flags=re.S)
    return sorted(set(re.findall(r'\b(estts_[a-z0-9_]+)\s*\(', text)))


def test_header_symbols_are_bound_and_exported():
    declared = _declared_symbols()
    assert declared, 'no declarations parsed from estts.h'
    assert set(declared) == set(_lib.SIGNATURES), (declared, sorted(_lib.SIGNATURES))
    if not os.path.exists(_lib.LIB_PATH):
        pytest.fail(f'{_lib.LIB_PATH} missing: run __graft_entry__.build() first')
    lib = ctypes.CDLL(_lib.LIB_PATH)
    for name in declared:
        assert hasattr(lib, name), name


def test_abi_version_and_error_channel():
    lib = _lib.load()
    assert lib.estts_abi_version() == _lib.ABI_VERSION
    # argument validation happens before any HIP call, so it is safe without a GPU
    rc = lib.estts_gemm(None, None)
    assert rc != 0 and b'null args' in lib.estts_last_error()


def test_struct_sizes_match_header_layout(tmp_path):
    """sizeof of every ctypes mirror equals what the C compiler computes from include/estts.h."""
    import subprocess
    pairs = {'estts_gemm_args': _lib.GemmArgs, 'estts_attention_args': _lib.AttentionArgs,
             'estts_rownorm_args': _lib.RownormArgs, 'estts_wavernn_prep_args': _lib.WavernnPrepArgs,
             'estts_wavernn_sample_args': _lib.WavernnSampleArgs, 'estts_wavernn_unfold_args': _lib.WavernnUnfoldArgs, 'estts_wavernn_layer_args': _lib.WavernnLayerArgs, 'estts_wavernn_cond_args': _lib.WavernnCondArgs, 'estts_wgrad_args': _lib.WgradArgs,
             'estts_attention_bwd_args': _lib.AttentionBwdArgs, 'estts_rownorm_bwd_args': _lib.RownormBwdArgs,
             'estts_colreduce_args': _lib.ColreduceArgs, 'estts_wavernn_prep_bwd_args': _lib.WavernnPrepBwdArgs, 'estts_rowmask_args': _lib.RowmaskArgs,
             'estts_prog_op': _lib.ProgOp, 'estts_pack_job': _lib.PackJob, 'estts_cross_row_args': _lib.CrossRowArgs, 'estts_decoder_block_args': _lib.DecoderBlockArgs, 'estts_prenet_row_args': _lib.PrenetRowArgs}
    src = tmp_path / 'sizes.c'
    src.write_text('#include <stdio.h>\n#include "estts.h"\nint main(void) {\n' +
                   ''.join(f'  printf("{n} %zu\\n", sizeof({n}));\n' for n in pairs) + '  return 0;\n}\n')
    exe = tmp_path / 'sizes'
    subprocess.run(['gcc', '-I', os.path.join(ROOT, 'include'), str(src), '-o', str(exe)], check=True)
    out = subprocess.run([str(exe)], check=True, capture_output=True, text=True).stdout
    sizes = dict(line.split() for line in out.strip().splitlines())
    for name, cls in pairs.items():
        assert ctypes.sizeof(cls) == int(sizes[name]), name
    assert ctypes.sizeof(_lib.GemmArgs) == 360


def test_deferred_sum_arguments_are_validated_before_any_launch():
    """The argument checks of the BatchNorm sum entry points run before any HIP call (fake, never dereferenced pointers)."""
    lib = _lib.load()
    fake = 0x1000

    def err():
        return lib.estts_last_error().decode()

    c = _lib.ColreduceArgs()
    c.z, c.ldz, c.M, c.C, c.mode, c.defer = fake, 256, 64, 256, 0, 1            # defer without a workspace
    assert lib.estts_colreduce(c, None) != 0 and 'defer needs ws' in err()
    c.ws, c.ws_bytes = fake, 16 * 2 * 256 * 4 - 4                               # ... or with one that is too small
    assert lib.estts_colreduce(c, None) != 0 and 'defer needs ws' in err()
    c.ws_bytes, c.mode, c.g, c.ldg = 16 * 2 * 256 * 4, 1, fake, 256             # defer is for modes 0 and 2
    assert lib.estts_colreduce(c, None) != 0 and 'modes 0 and 2' in err()
    c.mode, c.clear, c.clear_floats = 0, fake + 64, 1024                        # the region to clear overlaps the sums
    assert lib.estts_colreduce(c, None) != 0 and 'overlap' in err()

    args = [fake, 512, fake, 64, 256, 1e-3, 0.99, 1.0] + [fake] * 9 + [None]     # stride 512 for 256 columns
    assert lib.estts_bn_finalize_slots(*args) != 0 and 'slot_stride' in err()
    args = [fake, 256, fake, 256, fake, fake, fake, fake, 512, fake, fake, fake, 256, 64, 256, 0, None, None, None]
    assert lib.estts_bn_bwd_slots(*args) != 0 and 'slot_stride' in err()
    args[8], args[16] = 256, fake                                                # scale without shift
    assert lib.estts_bn_bwd_slots(*args) != 0 and 'go together' in err()

    g = _lib.GemmArgs()
    g.A, g.lda, g.taps, g.c_per_tap, g.W_hi, g.W_lo, g.ldw = fake, 64, 1, 64, fake, fake, 64
    g.M, g.N, g.K, g.C, g.ldc, g.precision, g.alpha = 19200, 256, 64, fake, 256, 3, 1.0
    g.stat_slots, g.batch = fake, 2
    assert lib.estts_gemm(g, None) != 0 and 'stat_slots' in err()


@pytest.mark.parametrize('cname,cls', [('estts_cross_row_args', _lib.CrossRowArgs), ('estts_decoder_block_args', _lib.DecoderBlockArgs),
                                       ('estts_prenet_row_args', _lib.PrenetRowArgs)])
def test_row_kernel_field_offsets_match_header_layout(tmp_path, cname, cls):
    """Every field of the ctypes mirrors of the decoder-row argument blocks sits where the C compiler puts it (a step program
    patches these blocks by byte offset)."""
    import subprocess
    names = [n for n, _ in cls._fields_]
    src = tmp_path / 'offsets.c'
    src.write_text('#include <stdio.h>\n#include <stddef.h>\n#include "estts.h"\nint main(void) {\n' +
                   ''.join(f'  printf("{n} %zu\\n", offsetof({cname}, {n}));\n' for n in names) + '  return 0;\n}\n')
    exe = tmp_path / 'offsets'
    subprocess.run(['gcc', '-I', os.path.join(ROOT, 'include'), str(src), '-o', str(exe)], check=True)
    out = subprocess.run([str(exe)], check=True, capture_output=True, text=True).stdout
    offsets = dict(line.split() for line in out.strip().splitlines())
    for n in names:
        assert getattr(cls, n).offset == int(offsets[n]), n
