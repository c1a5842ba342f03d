"""CPU models of the index arithmetic of the kernels that read packed weights block by block: every element a launch can touch
lies inside the operand the caller handed over, for sweeps of the shapes the entry points accept.  The formulas restate
csrc/decoder_row.hip (estts_cross_attention_row, estts_decoder_block_row, estts_prenet_row) and the W-fragment addressing of
csrc/gemm_oneshot.hip; a kernel change that moves an index has to move it here too.  (The 32 x 48 tile of the one-shot
convolution kernel that once read past its weight is kept as a known-bad case.)"""
import itertools

import pytest


def ceil_to(a, b):
    return (a + b - 1) // b * b


def packed_elems(N, K):
    """uint16 elements of one plane (hi or lo) of a packed [N, K] weight: N and K padded to 64."""
    return ceil_to(N, 64) * ceil_to(K, 64)


def block_span(nb, kb, ldw):
    """first and last element of fragment block (nb, kb): 64 lanes x 8 values = 512 elements"""
    base = (nb * (ldw // 32) + kb) * 512
    return base, base + 63 * 8 + 7


# ---------------------------------------------------------------------------------------------------------------------------
# csrc/decoder_row.hip
# ---------------------------------------------------------------------------------------------------------------------------
D, H = 256, 4


def cross_row_accesses(Lk, M, ldk, with_proj):
    """(operand, lowest index, highest index) of everything cross_attention_row_kernel reads or writes, over all workgroups."""
    acc = []
    NJ = 1 if (not with_proj and Lk <= 64) else (2 if Lk <= 128 else 4)
    for row, h, wave, lane in itertools.product(range(M), range(H), range(16), (0, 15, 16, 63)):
        qn, qk = wave & 3, wave >> 2
        if with_proj:
            for i in range(2):
                acc.append(('wq',) + block_span(4 * h + qn, 2 * qk + i, D))
        for i in range(4):
            acc.append(('wo',) + block_span(wave, (i >> 1) * (D // 32) + 2 * h + (i & 1), 2 * D))
        jq, col = 4 * wave + (lane >> 4), 64 * h + 4 * (lane & 15)
        for i in range(NJ):
            j = min(jq + 64 * i, Lk - 1)
            acc.append(('kv', j * ldk + col, j * ldk + col + 3))
            acc.append(('mask', row * Lk + j, row * Lk + j))
        acc.append(('slab', row * H * D + h * D + wave * 16, row * H * D + h * D + wave * 16 + 15))
        acc.append(('attn', row * H * Lk + h * Lk, row * H * Lk + h * Lk + Lk - 1))
    return acc


@pytest.mark.parametrize('Lk', [1, 2, 63, 64, 65, 128, 129, 255, 256])
@pytest.mark.parametrize('with_proj', [True, False])
def test_cross_attention_row_stays_inside_its_operands(Lk, with_proj):
    M, ldk = 3, 2 * D
    size = {'wq': packed_elems(D, D), 'wo': packed_elems(D, 2 * D), 'kv': Lk * ldk, 'mask': M * Lk, 'slab': M * H * D, 'attn': M * H * Lk}
    for name, lo, hi in cross_row_accesses(Lk, M, ldk, with_proj):
        assert 0 <= lo <= hi < size[name], (name, lo, hi, size[name])


def block_row_accesses(pos, Lk, M, S):
    acc = []
    for row, h, wave, lane in itertools.product(range(M), range(H), range(16), (0, 15, 16, 47, 63)):
        nbh, quarter, qn, qk = wave >> 2, wave & 3, wave & 3, wave >> 2
        for e, i in itertools.product(range(3), range(2)):
            acc.append(('wqkv',) + block_span(16 * e + 4 * h + nbh, 2 * quarter + i, D))
        for name, ldw in (('wos', 2 * D), ('woc', 2 * D)):
            for i in range(4):
                acc.append((name,) + block_span(wave, (i >> 1) * (D // 32) + 2 * h + (i & 1), ldw))
        for name in ('wqc', 'wff'):
            for i in range(2):
                acc.append((name,) + block_span(4 * h + qn, 2 * qk + i, D))
        jq, col = 4 * wave + (lane >> 4), 64 * h + 4 * (lane & 15)
        NJ1 = 1 if pos <= 64 else (2 if pos <= 128 else 4)
        for i in range(NJ1):
            j = min(jq + 64 * i, max(pos - 1, 0))
            for part in (1, 2):
                acc.append(('cache', (row * S + j) * 3 * D + part * D + col, (row * S + j) * 3 * D + part * D + col + 3))
            acc.append(('smask', row * S + j, row * S + j))
        acc.append(('smask', row * S + pos, row * S + pos))
        for t in (0, 191):                                        # the position's own row, written by threads 0 .. 191
            o = (row * S + pos) * 3 * D + (t >> 6) * D + 64 * h + (t & 63)
            acc.append(('cache', o, o))
        NJ2 = 2 if Lk <= 128 else 4
        for i in range(NJ2):
            j = min(jq + 64 * i, Lk - 1)
            acc.append(('enc', (row * Lk + j) * 2 * D + col, (row * Lk + j) * 2 * D + D + col + 3))
            acc.append(('emask', row * Lk + j, row * Lk + j))
        if lane < 48:
            o = (lane >> 4) * D + 64 * h + 4 * (lane & 15)
            acc.append(('bqkv', o, o + 3))
        for slab in (0, 1):
            o = row * 2 * H * D + slab * H * D + h * D + wave * 16
            acc.append(('ws', o, o + 15))
        acc.append(('counters', 2 * row, 2 * row + 1))
        acc.append(('rows', row * D + 64 * h, row * D + 64 * h + 63))          # y2 / x2
    return acc


@pytest.mark.parametrize('pos', [0, 1, 63, 64, 65, 127, 128, 129, 255])
@pytest.mark.parametrize('Lk', [1, 100, 128, 129, 256])
def test_decoder_block_row_stays_inside_its_operands(pos, Lk):
    M, S = 2, pos + 1                                             # the smallest cache the call may be given
    size = {'wqkv': packed_elems(3 * D, D), 'wos': packed_elems(D, 2 * D), 'woc': packed_elems(D, 2 * D), 'wqc': packed_elems(D, D),
            'wff': packed_elems(D, D), 'cache': M * S * 3 * D, 'smask': M * S, 'enc': M * Lk * 2 * D, 'emask': M * Lk, 'bqkv': 3 * D,
            'ws': M * 2 * H * D, 'counters': 2 * M, 'rows': M * D}
    for name, lo, hi in block_row_accesses(pos, Lk, M, S):
        assert 0 <= lo <= hi < size[name], (name, lo, hi, size[name])


@pytest.mark.parametrize('C', [4, 60, 64, 80, 124, 128])
def test_prenet_row_stays_inside_its_operands(C):
    ldw1 = ceil_to(C, 64)
    nkb1 = ldw1 // 32
    assert nkb1 <= 4
    for s, wave in itertools.product(range(4), range(16)):
        for kb in range(nkb1):
            lo, hi = block_span(wave, kb, ldw1)
            assert hi < packed_elems(256, C)
        for i in range(2):
            lo, hi = block_span(4 * s + (wave & 3), 2 * (wave >> 2) + i, 256)
            assert hi < packed_elems(256, 256)
    for lane in range(32):                                         # the frame: a float4 per lane while it starts inside the row
        if 4 * lane < C:
            assert 4 * lane + 3 < C


# ---------------------------------------------------------------------------------------------------------------------------
# csrc/gemm_oneshot.hip: W fragments of a tile of tn columns starting at n0 = tx * tn
# ---------------------------------------------------------------------------------------------------------------------------
def oneshot_w_blocks(N, tn, clamp):
    last = ceil_to(N, 64) // 16 - 1
    out = []
    for tx in range((N + tn - 1) // tn):
        for j in range(tn // 16):
            nb = tx * tn // 16 + j
            out.append(min(nb, last) if clamp else nb)
    return out, last


@pytest.mark.parametrize('N', [16, 40, 72, 80, 256, 500, 512, 768, 1000, 1024, 1104, 1536])
def test_oneshot_tiles_of_16_and_32_columns_stay_inside_the_padded_weight(N):
    for tn in (16, 32):
        blocks, last = oneshot_w_blocks(N, tn, clamp=False)
        assert max(blocks) <= last, (N, tn)


def test_oneshot_tile_of_48_columns_needs_its_block_row_clamped():
    # the case that faulted a GPU: 11 tiles of 48 columns cover 528 columns of a weight padded to 512
    blocks, last = oneshot_w_blocks(512, 48, clamp=False)
    assert max(blocks) == last + 1
    for N in (500, 512, 1000, 1104):
        blocks, last = oneshot_w_blocks(N, 48, clamp=True)
        assert max(blocks) <= last


# ---------------------------------------------------------------------------------------------------------------------------
# csrc/common.h xcd_tile_coord / csrc/gemm.hip region_tile_coord: every workgroup id of a grid names one tile, every tile once
# ---------------------------------------------------------------------------------------------------------------------------
def xcd_tile_coord(bx, by, bz, gx, gy, gz, y_first, remap=True, panel=0):
    nwg = gx * gy * gz
    i = bx + gx * (by + gy * bz)
    if remap and nwg >= 16:
        q, r, xcd = nwg >> 3, nwg & 7, i & 7
        i = (xcd * (q + 1) if xcd < r else r * (q + 1) + (xcd - r) * q) + (i >> 3)
    tz, tile = i % gz, i // gz
    if panel > 0 and gz == 1:
        per = panel * gy
        pn, rem = tile // per, tile % per
        w = min(panel, gx - pn * panel)
        ty = rem // w
        return pn * panel + (rem - ty * w), ty, tz
    return (tile // gy, tile % gy, tz) if y_first else (tile % gx, tile // gx, tz)


@pytest.mark.parametrize('gx,gy,gz', [(1, 1, 1), (16, 19, 1), (11, 19, 1), (3, 5, 1), (8, 2, 1), (24, 19, 1), (7, 7, 3), (16, 4, 8),
                                      (5, 1, 1), (1, 300, 1), (13, 10, 2)])
def test_xcd_tile_numbering_is_a_bijection(gx, gy, gz):
    for y_first, remap, panel in itertools.product((False, True), (False, True), (0, 4)):
        tiles = {xcd_tile_coord(bx, by, bz, gx, gy, gz, y_first, remap, panel)
                 for bx in range(gx) for by in range(gy) for bz in range(gz)}
        assert len(tiles) == gx * gy * gz, (y_first, remap, panel)
        assert all(0 <= tx < gx and 0 <= ty < gy and 0 <= tz < gz for tx, ty, tz in tiles), (y_first, remap, panel)
