#!/usr/bin/env python3
"""Rebuild profiles/pmc_traffic.json (HBM bytes per launch of the kernels bench.py quotes a roofline for) from a pair of
rocprofv3 --pmc passes, and stamp it with a hash of the kernel sources it was measured on: bench.py reports `traffic` only
while that hash matches the tree, so a changed kernel cannot inherit an old measurement.

usage: update_pmc_traffic.py FETCH_SIZE.csv WRITE_SIZE.csv [FETCH_SIZE.csv WRITE_SIZE.csv | summary.json ...]
(one pair of counter files per profiled command, or the per-kernel summary tools/pmc_summary.py made of such a pair; a kernel
is looked up in every set)"""
import hashlib
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, 'expressive-speech-synthesis-research_amd', 'csrc')

# bench.py roofline key -> (substring of the kernel name, note appended to the source line[, grid size in threads when one
# instantiation serves several shapes])
KERNELS = {
    'gemm {"M": 19200, "N": 256, "K": 512, "taps": 1}': (
        'gemm_wdirect_kernel<3, 4, 1, 4, false>',
        ' (the 19200 x 256 x 512 dense layers without a norm behind them -- config 3: the cross-attention key / value projections;'
        ' config 5: the output projections -- 64 x 128 tiles, W from memory, 512 workgroups of 64 rows + 176 of 32)', 176128),
    'gemm {"M": 19200, "N": 256, "K": 256, "taps": 1}': (
        'gemm_wdirect_kernel<3, 4, 1, 4, false>',
        ' (the 19200 x 256 x 256 dense layers run the same instantiation on the same grid as the 19200 x 256 x 512 ones; the'
        ' counters cannot tell the two apart, this is the mean over both: K = 256 reads 19.7 MB less of A)', 176128),
    'gemm {"M": 19200, "N": 256, "K": 512, "taps": 1, "ln_epilogue": true}': (
        'gemm_wdirect_kernel<3, 2, 1, 8, true>',
        ' (config 3: the attention output projections over [x | ctx] with the LayerNorm combination that follows them in the'
        ' epilogue; 64 x 256 tiles of eight waves, W from memory, 256 workgroups of 64 rows + 88 of 32; the counters cannot tell'
        ' K = 512 from the folded FFN product with K = 256 on the same grid, this is the mean over both)', 176128),
    'gemm {"M": 19200, "N": 256, "K": 256, "taps": 1, "ln_epilogue": true}': (
        'gemm_wdirect_kernel<3, 2, 1, 8, true>',
        ' (config 3: the folded FFN product with its LayerNorm combination in the epilogue; same instantiation and grid as the'
        ' K = 512 output projections, this is the mean over both)', 176128),
    'attention_bwd {"B": 32, "H": 4, "Lq": 600, "Lk": 600, "depth": 64}': (
        'attention_bwd_wide_kernel',
        ' (config 5: one of the two passes -- owning queries, owning keys -- of the decoder self-attention backward, 600 x 600,'
        ' batch 32; the roofline entry prices the pair, this is the mean of the two)', None),
    'wavernn_layer {"mode": 2, "n_seq": 832, "n_units": 512, "K": 544}': (
        'wavernn_layer_kernel<2, 4, 2>',
        ' (config 4: rnn2 of one sampling step, 832 sequences; 17.9 MB of weights per step are re-read from L2 / the Infinity'
        ' Cache, which these counters see only when they miss there)', None),
    'gemm {"M": 100, "N": 512, "K": 1536, "taps": 3}': (
        'gemm_oneshot_kernel<3, 2, 1, 8, 6>',
        ' (the second convolution of each encoder block: M = 100, N = 512, K = 3 x 512; 224 workgroups of 16 x 16 outputs)'),
    'gemm {"M": 100, "N": 512, "K": 1536, "taps": 3, "ln_prologue": true}': (
        'gemm_oneshot_conv_kernel<3, 1, 8, 6, 2,',
        ' (the first convolution of each encoder block with the attention block\'s LN(LN(y) + x) computed on the way in)'),
    'gemm {"M": 600, "N": 512, "K": 768, "taps": 3, "ln_prologue": true}': (
        'gemm_oneshot_conv_kernel<3, 2, 8, 3, 1,',
        ' (the first convolution of each decoder block with the attention block\'s LN(LN(y) + x) computed on the way in)'),
    'attention {"B": 1, "H": 4, "Lq": 600, "Lk": 600, "depth": 64, "weights": true}': ('attention_kernel<64, 2, 3, 8>', ''),
}

def csrc_hash():
    h = hashlib.sha1()
    for name in sorted(os.listdir(CSRC)):
        if name.endswith(('.hip', '.h')):
            h.update(name.encode())
            with open(os.path.join(CSRC, name), 'rb') as f:
                h.update(f.read())
    return h.hexdigest()


def main():
    sets, args = [], sys.argv[1:]
    while args:
        if args[0].endswith('.json'):           # a summary made earlier (the raw counter files of the larger workloads are not kept)
            with open(args[0]) as f:
                sets.append((args[0], None, json.load(f)))
            args = args[1:]
        else:
            f, w = args[0], args[1]
            sets.append((f, w, json.loads(subprocess.check_output([sys.executable, os.path.join(ROOT, 'tools', 'pmc_summary.py'), f, w]))))
            args = args[2:]
    out = {'csrc_sha1': csrc_hash()}
    for key, spec in KERNELS.items():
        pat, note = spec[0], spec[1]
        grid = spec[2] if len(spec) > 2 else None
        hit = [(r, f, w) for f, w, rows in sets for r in rows if pat in r['kernel'] and (grid is None or r['grid'] == grid)]
        if not hit:
            print(f'no dispatch of {pat} in the passes', file=sys.stderr)
            continue
        r, fetch, write = max(hit, key=lambda x: x[0]['dispatches'])
        out[key] = {
            'hbm_bytes': round(r['hbm_bytes']), 'read_bytes': round(r['read_bytes']), 'write_bytes': round(r['write_bytes']),
            'source': f"profiles/{os.path.basename(fetch)}" + (f" + {os.path.basename(write)}" if write else " (per-kernel summary of the two counter files)") +
                      f" (rocprofv3 --pmc, separate passes; read = 2 x "
                      f"FETCH_SIZE KiB per the gfx950 correction, write = WRITE_SIZE KiB), mean of {r['dispatches']} dispatches of "
                      f"{pat} with {r['grid']} threads{note}, summarised by tools/pmc_summary.py"}
    with open(os.path.join(ROOT, 'profiles', 'pmc_traffic.json'), 'w') as f:
        json.dump(out, f, indent=1)
    print(json.dumps(out, indent=1))


if __name__ == '__main__':
    main()
