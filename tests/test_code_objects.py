"""Register audit of the built gfx950 code objects (CPU: reads the metadata notes of csrc/*.o, launches nothing): which kernels
use scratch memory or spill registers is a short, known list; a source or compiler change that makes another kernel spill --
in particular one of the latency-bound kernels, where a spill inside the dependent chain costs a memory round trip -- shows
up here before it shows up as a slower launch."""
import glob
import os
import re
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, 'expressive-speech-synthesis-research_amd', 'csrc')
LLVM = '/opt/rocm/lib/llvm/bin'
TOOLS = [os.path.join(LLVM, t) for t in ('llvm-objcopy', 'clang-offload-bundler', 'llvm-readelf')]

# kernel-name pattern (mangled) -> most registers it may spill; anything not listed must not spill or use scratch at all
ALLOWED = [
    (r'decoder_block_row_kernelILi1ELi4E', 24),      # 64 cache rows, 256 encoder rows
    (r'decoder_block_row_kernelILi2ELi[24]E', 32),   # 128 cache rows
    (r'decoder_block_row_kernelILi4ELi[24]E', 48),   # 256 cache rows
    (r'gemm_small_kernelILi[13]ELi16ELi0E', 140),    # the 16-wave register-ring variant (explicit selection only)
    (r'gemm_wdirect_kernelILi3ELi4ELi2ELi4ELb0E', 4),
    (r'gemm_kernelILi(64|32)ELi(64|32)E', 0),        # tile ids 1-3 (explicit selection only): index arrays in scratch, no spills
    (r'gemm_oneshot_conv_kernelILi3ELi2ELi8ELi6ELi2ELb1E', 16),
]
# kernels on the latency-bound paths that must stay clean
MUST_BE_CLEAN = [r'decoder_block_row_kernelILi1ELi2E', r'cross_attention_row_kernel', r'prenet_row_kernel',
                 r'gemm_oneshot_conv_kernelILi3ELi2ELi8ELi3ELi1ELb1E', r'attention_kernel', r'attention_wide_kernel',
                 r'attention_bwd', r'wavernn_layer_kernel', r'wavernn_sample_kernel', r'wgrad_kernel', r'gru_fused']


def kernels_of(obj, tmp):
    fat, co = os.path.join(tmp, 'fat.bin'), os.path.join(tmp, 'code.co')
    subprocess.run([TOOLS[0], '-O', 'binary', '--only-section=.hip_fatbin', obj, fat], check=True)
    subprocess.run([TOOLS[1], '--unbundle', '--type=o', '--targets=hipv4-amdgcn-amd-amdhsa--gfx950', f'--input={fat}', f'--output={co}'],
                   check=True, capture_output=True)
    notes = subprocess.run([TOOLS[2], '--notes', co], check=True, capture_output=True, text=True).stdout
    out = []
    for block in notes.split('- .agpr_count:')[1:]:
        get = lambda key: re.search(r'\.' + key + r':\s+(\S+)', block).group(1)
        out.append((get('name'), int(get('private_segment_fixed_size')), int(get('vgpr_count')), int(get('vgpr_spill_count'))))
    return out


@pytest.mark.skipif(not all(os.path.exists(t) for t in TOOLS), reason='LLVM binary tools not found')
def test_only_the_known_kernels_spill_or_use_scratch(tmp_path):
    objs = sorted(glob.glob(os.path.join(CSRC, '*.o')))
    if not objs:
        pytest.skip('no objects under csrc/ (they are built by __graft_entry__.build())')
    total, seen_clean = 0, set()
    for obj in objs:
        for name, scratch, vgprs, spills in kernels_of(obj, str(tmp_path)):
            total += 1
            assert vgprs <= 512, name          # (the unified file: vector + accumulation registers)
            for pat in MUST_BE_CLEAN:
                if re.search(pat, name):
                    seen_clean.add(pat)
                    assert scratch == 0 and spills == 0, (name, scratch, spills)
            if scratch or spills:
                limit = [lim for pat, lim in ALLOWED if re.search(pat, name)]
                assert limit, f'{os.path.basename(obj)}: {name} uses {scratch} B of scratch per lane and spills {spills} registers'
                assert spills <= limit[0], (name, spills, limit[0])
    assert total > 300 and seen_clean == set(MUST_BE_CLEAN), (total, sorted(set(MUST_BE_CLEAN) - seen_clean))
