// Fused dense / 1-D-convolution kernels on the gfx950 matrix cores (see estts_gemm in include/estts.h).
//
// Operands.  A = float32 activations, row-major; every kernel splits them into bf16 hi (+ lo) on the fly.
// W = bf16 weights, pre-packed by the host in FRAGMENT-MAJOR order: for each 16-row block nt of W[N][K]
// and each 32-wide K-step ks there is one contiguous 1 KiB block holding, for lane l = 0..63, the 8 values
// W[16 nt + (l & 15)][32 ks + 8 (l >> 4) .. + 7]  -- exactly the B operand of v_mfma_f32_16x16x32_bf16
// (lane l holds B[k 8(l>>4)..+7][col l&15]; A[row l&15][k 8(l>>4)..+7]; D[row 4(l>>4)+j][col l&15]).
// A wave therefore fetches a weight fragment with one fully coalesced 16-byte-per-lane load, and an LDS
// image of it is lane-linear (conflict-free).  Loading fragments row-wise instead (16 rows x 64 B per
// instruction) costs about 4x the L1 cycles, which is what bounds these kernels.
//
// A convolution is the same product with K = taps * c_per_tap: a K-chunk maps to (tap, channel), the A row
// is shifted by tap - tap_lpad, and rows that leave the sequence read as zero.
//
// Three kernels:
//  * gemm_big_kernel (tile id 5; batched synthesis and training, thousands of rows): see its own header below.
//  * gemm_small_kernel (tile id 4; batch-1 synthesis and everything that does not fill the chip with 128 x 128 tiles).  No barrier in the K loop: a
//    workgroup owns a 32 x 32 tile of C and its NW waves take interleaved K-steps.  Each wave loads its
//    A slice (32 rows x 32 k) with coalesced float4 loads, converts, and turns it into MFMA fragments
//    through a wave-private LDS image (80-byte rows: bank-conflict-free for ds_read_b128); the loads of the
//    next K-step are in flight during the MFMAs.  Partial tiles are summed through LDS at the end.
//  * gemm_kernel (tile ids 1-3; explicit selection only).  Classic LDS-staged BM x BN tile, K stages of 64, register
//    prefetch of the next stage, two LDS buffers, one barrier per stage.
// Both support split-K over gridDim.z: every workgroup writes its float32 partial to a slab, takes a ticket
// from a per-tile counter, and the last arriver sums the slabs and runs the epilogue.  Hand-off: every wave
// drains its stores, workgroup barrier, one lane issues an agent-scope release fence and then the relaxed
// agent-scope ticket add; the last arriver issues an agent-scope acquire fence before any wave reads the
// slabs.  The counter is reset by the last arriver, so it must be zero only before the first ever launch.
#include <mutex>

#include "common.h"
#include "a_operand.h"
#include "gemm_epilogue.h"

namespace {

constexpr int BK = 64;

// split-K hand-off; returns true in the workgroup that must finish the tile (all threads must call)
__device__ __forceinline__ bool splitk_last_arriver(const estts_gemm_args& p, int tile, int splitk, int* s_ticket) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (threadIdx.x == 0) {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        *s_ticket = __hip_atomic_fetch_add(p.counters + tile, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    __syncthreads();
    if (*s_ticket != splitk - 1) return false;
    if (threadIdx.x == 0) {
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __hip_atomic_store(p.counters + tile, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    __syncthreads();
    return true;
}

// ======================================================================================================
// small-M kernel
// ======================================================================================================
constexpr int SM_LD = 40;                      // bf16 per staged A row (32 + 8 pad = 80 B)
constexpr int SM_WAVE_ELEMS = 2 * 32 * SM_LD;  // hi + lo image per wave

template <int NS>
struct SmallStage {
    float4 a[4];
    bool ok[4];
    uint4 w[NS][2];
};

// KIND selects the A-operand addressing.  The K loop of these kernels is bound by vector instructions at small sizes
// (address arithmetic + operand split), so the common layer shapes get short forms of their own:
//   1 (dense layer: one tap, output row m reads input row m): the address is a per-row base pointer plus the K index and
//     only K itself is bounds-checked;
//   2 (1-D convolution over rows of seg_len pixels): base pointer + (tap - lpad) rows + channel, valid while the shifted
//     position stays inside the row's own sequence; (tap, channel) is carried by a cursor, not divided out per step;
//   0: the general 2-D geometry (a_operand.h).
template <int NSPLIT, int NW, int KIND>
__global__ __launch_bounds__(NW * 64) void gemm_small_kernel(const estts_gemm_args p, const int splitk) {
    constexpr int NS = (NSPLIT > 1) ? 2 : 1;
    constexpr int STAGE_BYTES = NW * SM_WAVE_ELEMS * 2;
    constexpr int RED_BYTES = NW * 16 * 64 * 4;
    __shared__ __attribute__((aligned(16))) unsigned char smem[STAGE_BYTES > RED_BYTES ? STAGE_BYTES : RED_BYTES];
    __shared__ int s_ticket;

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int i16 = lane & 15;
    const int g = lane >> 4;
    const int r8 = lane >> 3;
    const int c8 = lane & 7;
    const TileCoord tc = xcd_tile_coord((p.tune & (1 << 25)) ? false : gemm_y_first(p), !(p.tune & (1 << 24)));
    const int m0 = tc.ty * 32;
    const int n0 = tc.tx * 32;
    const int KS = p.ldw >> 5;  // K-steps per packed row block

    ARow a_row[4];
    const float* a_ptr[4];   // KIND 1 / 2: start of the row (rows past M re-read the last one; their results are never stored)
    int a_pos[4];            // KIND 2: position of the row inside its sequence
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int row = min(m0 + r8 + 8 * j, p.M - 1);
        if (KIND == 1) a_ptr[j] = p.A + (long)row * p.lda + c8 * 4;
        else if (KIND == 2) { a_ptr[j] = p.A + (long)row * p.lda; a_pos[j] = row % p.conv_w; }
        else a_row[j] = make_arow(p, m0 + r8 + 8 * j);
    }
    const uint16_t* whp[2];
    const uint16_t* wlp[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const long off = (long)((n0 >> 4) + i) * KS * 512 + lane * 8;
        whp[i] = p.W_hi + off;
        wlp[i] = p.W_lo + off;
    }

    const int nks_all = (p.K + 31) / 32;
    const int per = (nks_all + splitk - 1) / splitk;
    const int ks_end = min((tc.tz + 1) * per, nks_all);
    int ks = tc.tz * per + wave;

    auto load = [&](int k, SmallStage<NS>& st) {
        if (KIND == 1) {
            const bool ok = k * 32 + c8 * 4 < p.K;     // K % 4 == 0: a float4 is inside or outside as a whole
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                st.a[j] = *reinterpret_cast<const float4*>(a_ptr[j] + (ok ? k * 32 : -c8 * 4));
                st.ok[j] = ok;
            }
        } else if (KIND == 2) {
            const int kk = k * 32 + c8 * 4;
            const int tap = kk / p.c_per_tap;          // (the waves take interleaved K-steps: no cursor here)
            const int dw = tap - p.conv_pl;
            const long off = (long)dw * p.lda + (kk - tap * p.c_per_tap);
            const bool kok = kk < p.K;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const bool ok = kok && (unsigned)(a_pos[j] + dw) < (unsigned)p.conv_w;
                st.a[j] = *reinterpret_cast<const float4*>(a_ptr[j] + (ok ? off : 0L));
                st.ok[j] = ok;
            }
        } else {
            const ATap tp = make_atap(p, k * 32 + c8 * 4);
#pragma unroll
            for (int j = 0; j < 4; ++j) st.a[j] = load_a_raw(p, a_row[j], tp, st.ok[j]);
        }
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            st.w[0][i] = *reinterpret_cast<const uint4*>(whp[i] + (long)k * 512);
            if (NS == 2) st.w[NS - 1][i] = *reinterpret_cast<const uint4*>(wlp[i] + (long)k * 512);
        }
    };

    f32x4 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

    __bf16* stage = reinterpret_cast<__bf16*>(smem) + wave * SM_WAVE_ELEMS;  // [hi | lo][32][SM_LD]

    auto compute = [&](const SmallStage<NS>& st) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            // KIND 1 needs no zeroing: a K index past the end re-reads valid data of the row, and W is zero-padded there
            const float4 az = KIND == 1 ? st.a[j] : zero_unless(st.ok[j], st.a[j]);
            bf16x4 hi, lo;
            split_bf16x4(az, hi, lo);
            const int o = (r8 + 8 * j) * SM_LD + c8 * 4;
            *reinterpret_cast<bf16x4*>(stage + o) = hi;
            if (NS == 2) *reinterpret_cast<bf16x4*>(stage + 32 * SM_LD + o) = lo;
        }
        bf16x8 af[NS][2], wf[NS][2];
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int o = (i * 16 + i16) * SM_LD + g * 8;
            af[0][i] = *reinterpret_cast<const bf16x8*>(stage + o);
            if (NS == 2) af[NS - 1][i] = *reinterpret_cast<const bf16x8*>(stage + 32 * SM_LD + o);
#pragma unroll
            for (int s = 0; s < NS; ++s) wf[s][i] = __builtin_bit_cast(bf16x8, st.w[s][i]);
        }
        // small terms first; consecutive MFMAs go to different accumulators
        if (NS == 2) {
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[NS - 1][i], wf[0][j], acc[i][j], 0, 0, 0);
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[0][i], wf[NS - 1][j], acc[i][j], 0, 0, 0);
        }
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j)
                acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[0][i], wf[0][j], acc[i][j], 0, 0, 0);
    };

    // three register stages: the loads of the wave's next two K-steps are in flight while one is multiplied (a K-step costs
    // a full L2 round trip otherwise: measured ~0.9 us per step and wave at batch 1)
    SmallStage<NS> sa, sb, sc;
    if (ks < ks_end) {
        load(ks, sa);
        if (ks + NW < ks_end) load(ks + NW, sb);
        while (true) {
            if (ks + 2 * NW < ks_end) load(ks + 2 * NW, sc);
            compute(sa);
            ks += NW;
            if (ks >= ks_end) break;
            if (ks + 2 * NW < ks_end) load(ks + 2 * NW, sa);
            compute(sb);
            ks += NW;
            if (ks >= ks_end) break;
            if (ks + 2 * NW < ks_end) load(ks + 2 * NW, sb);
            compute(sc);
            ks += NW;
            if (ks >= ks_end) break;
        }
    }

    // sum the waves' partial tiles (the staging images are dead after the barrier); wave w < 4 finishes
    // sub-tile (w >> 1, w & 1), the other waves only keep the barriers company
    __syncthreads();
    float* red = reinterpret_cast<float*>(smem);  // [NW][16][64]
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 4; ++r) red[(wave * 16 + (i * 2 + j) * 4 + r) * 64 + lane] = acc[i][j][r];
    __syncthreads();
    const bool fin = wave < 4;
    f32x4 out = {0.f, 0.f, 0.f, 0.f};
    if (fin) {
#pragma unroll
        for (int w = 0; w < NW; ++w)
#pragma unroll
            for (int r = 0; r < 4; ++r) out[r] += red[(w * 16 + wave * 4 + r) * 64 + lane];
    }

    if (splitk > 1) {
        const int tile = tc.ty * gridDim.x + tc.tx;
        const long ntiles = (long)gridDim.x * gridDim.y;
        float* slab = p.ws + ((long)tc.tz * ntiles + tile) * 1024;
        if (fin) {
#pragma unroll
            for (int r = 0; r < 4; ++r) slab[r * 256 + tid] = out[r];
        }
        if (!splitk_last_arriver(p, tile, splitk, &s_ticket)) return;
        out = (f32x4){0.f, 0.f, 0.f, 0.f};
        if (fin) {
            for (int z = 0; z < splitk; ++z) {
                const float* sl = p.ws + ((long)z * ntiles + tile) * 1024;
#pragma unroll
                for (int r = 0; r < 4; ++r) out[r] += sl[r * 256 + tid];
            }
        }
    }
    if (!fin) return;

    const int col = n0 + (wave & 1) * 16 + i16;
    if (col >= p.N) return;
    const float s1 = p.s1 ? p.s1[col] : 1.0f;
    const float b1 = p.b1 ? p.b1[col] : 0.0f;
    const float s2 = p.s2 ? p.s2[col] : 1.0f;
    const float b2 = p.s2 ? p.b2[col] : 0.0f;
    float res[4];   // R first: it may be C itself, which keeps a load behind every store that precedes it
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int row = m0 + (wave >> 1) * 16 + g * 4 + r;
        res[r] = (p.R && row < p.M) ? epilogue_residual(p, row, col) : 0.0f;
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int row = m0 + (wave >> 1) * 16 + g * 4 + r;
        if (row < p.M) p.C[(long)row * p.ldc + col] = epilogue_value_r(p, out[r], row, col, s1, b1, s2, b2, res[r]);
    }
}

// ======================================================================================================
// large-M kernel
// ======================================================================================================
constexpr int A_LD = 72;  // bf16 per staged A row (64 + 8 pad = 144 B)

template <int BM, int BN, int WM, int WN, int NSPLIT>
__global__ __launch_bounds__(256) void gemm_kernel(const estts_gemm_args p, const int splitk) {
    constexpr int NS = (NSPLIT > 1) ? 2 : 1;
    constexpr int A_ITERS = BM * 16 / 256;
    constexpr int W_ITERS = BN * 8 / 256;
    constexpr int MI = BM / WM / 16;
    constexpr int NI = BN / WN / 16;
    static_assert(WM * WN == 4 && A_ITERS >= 1 && W_ITERS >= 1 && MI >= 1 && NI >= 1, "bad tile");

    __shared__ __attribute__((aligned(16))) __bf16 As[2][NS][BM * A_LD];
    __shared__ __attribute__((aligned(16))) __bf16 Ws[2][NS][BN * BK];  // (BN/16) x 2 fragment blocks of 512
    __shared__ int s_ticket;

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int wm = wave / WN;
    const int wn = wave % WN;
    const TileCoord tc = xcd_tile_coord((p.tune & (1 << 25)) ? false : gemm_y_first(p), !(p.tune & (1 << 24)));
    const int m0 = tc.ty * BM;
    const int n0 = tc.tx * BN;
    const int KS = p.ldw >> 5;

    ARow a_row[A_ITERS];
#pragma unroll
    for (int i = 0; i < A_ITERS; ++i) a_row[i] = make_arow(p, m0 + ((tid + i * 256) >> 4));
    const int a_ch = tid & 15;  // same 16-byte column chunk for every i

    float4 a_reg[A_ITERS];
    uint4 w_reg[NS][W_ITERS];

    auto load_tiles = [&](int kt) {
        const ATap tp = make_atap(p, kt * BK + a_ch * 4);
#pragma unroll
        for (int i = 0; i < A_ITERS; ++i) a_reg[i] = load_a(p, a_row[i], tp);
#pragma unroll
        for (int i = 0; i < W_ITERS; ++i) {
            const int idx = tid + i * 256;
            const int blk = idx >> 6;  // (n16 block j, k-step s) = (blk >> 1, blk & 1)
            const long off = ((long)((n0 >> 4) + (blk >> 1)) * KS + kt * 2 + (blk & 1)) * 512 + (idx & 63) * 8;
            w_reg[0][i] = *reinterpret_cast<const uint4*>(p.W_hi + off);
            if (NS == 2) w_reg[NS - 1][i] = *reinterpret_cast<const uint4*>(p.W_lo + off);
        }
    };

    auto store_tiles = [&](int buf) {
#pragma unroll
        for (int i = 0; i < A_ITERS; ++i) {
            const int o = ((tid + i * 256) >> 4) * A_LD + a_ch * 4;
            const float v[4] = {a_reg[i].x, a_reg[i].y, a_reg[i].z, a_reg[i].w};
            bf16x4 hi, lo;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                __bf16 h, l;
                split_bf16(v[j], h, l);
                hi[j] = h;
                lo[j] = l;
            }
            *reinterpret_cast<bf16x4*>(&As[buf][0][o]) = hi;
            if (NS == 2) *reinterpret_cast<bf16x4*>(&As[buf][NS - 1][o]) = lo;
        }
#pragma unroll
        for (int i = 0; i < W_ITERS; ++i) {
            const int o = (tid + i * 256) * 8;
            *reinterpret_cast<uint4*>(&Ws[buf][0][o]) = w_reg[0][i];
            if (NS == 2) *reinterpret_cast<uint4*>(&Ws[buf][NS - 1][o]) = w_reg[NS - 1][i];
        }
    };

    f32x4 acc[MI][NI];
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int j = 0; j < NI; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

    const int nk_all = (p.K + BK - 1) / BK;
    const int per = (nk_all + splitk - 1) / splitk;
    const int kt0 = tc.tz * per;
    const int kt1 = min(kt0 + per, nk_all);

    const int a_base = (wm * (BM / WM) + (lane & 15)) * A_LD + (lane >> 4) * 8;
    const int w_base = wn * NI * 2 * 512 + lane * 8;

    if (kt0 < kt1) load_tiles(kt0);
    int buf = 0;
    for (int kt = kt0; kt < kt1; ++kt) {
        store_tiles(buf);
        __syncthreads();
        if (kt + 1 < kt1) load_tiles(kt + 1);
#pragma unroll
        for (int ks = 0; ks < BK / 32; ++ks) {
            bf16x8 af[NS][MI], wf[NS][NI];
#pragma unroll
            for (int s = 0; s < NS; ++s) {
#pragma unroll
                for (int i = 0; i < MI; ++i)
                    af[s][i] = *reinterpret_cast<const bf16x8*>(&As[buf][s][a_base + i * 16 * A_LD + ks * 32]);
#pragma unroll
                for (int j = 0; j < NI; ++j)
                    wf[s][j] = *reinterpret_cast<const bf16x8*>(&Ws[buf][s][w_base + (j * 2 + ks) * 512]);
            }
            if (NS == 2) {
#pragma unroll
                for (int i = 0; i < MI; ++i)
#pragma unroll
                    for (int j = 0; j < NI; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[NS - 1][i], wf[0][j], acc[i][j], 0, 0, 0);
#pragma unroll
                for (int i = 0; i < MI; ++i)
#pragma unroll
                    for (int j = 0; j < NI; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[0][i], wf[NS - 1][j], acc[i][j], 0, 0, 0);
            }
#pragma unroll
            for (int i = 0; i < MI; ++i)
#pragma unroll
                for (int j = 0; j < NI; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[0][i], wf[0][j], acc[i][j], 0, 0, 0);
        }
        buf ^= 1;
    }

    if (splitk > 1) {
        const int tile = tc.ty * gridDim.x + tc.tx;
        const long ntiles = (long)gridDim.x * gridDim.y;
        float* slab = p.ws + ((long)tc.tz * ntiles + tile) * (BM * BN);
#pragma unroll
        for (int i = 0; i < MI; ++i)
#pragma unroll
            for (int j = 0; j < NI; ++j)
#pragma unroll
                for (int r = 0; r < 4; ++r) slab[((i * NI + j) * 4 + r) * 256 + tid] = acc[i][j][r];
        if (!splitk_last_arriver(p, tile, splitk, &s_ticket)) return;
#pragma unroll
        for (int i = 0; i < MI; ++i)
#pragma unroll
            for (int j = 0; j < NI; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
        for (int z = 0; z < splitk; ++z) {
            const float* sl = p.ws + ((long)z * ntiles + tile) * (BM * BN);
#pragma unroll
            for (int i = 0; i < MI; ++i)
#pragma unroll
                for (int j = 0; j < NI; ++j)
#pragma unroll
                    for (int r = 0; r < 4; ++r) acc[i][j][r] += sl[((i * NI + j) * 4 + r) * 256 + tid];
        }
    }

    const int col_l = lane & 15;
    const int row_l = (lane >> 4) * 4;
#pragma unroll
    for (int j = 0; j < NI; ++j) {
        const int col = n0 + wn * (BN / WN) + j * 16 + col_l;
        if (col >= p.N) continue;
        const float s1 = p.s1 ? p.s1[col] : 1.0f;
        const float b1 = p.b1 ? p.b1[col] : 0.0f;
        const float s2 = p.s2 ? p.s2[col] : 1.0f;
        const float b2 = p.s2 ? p.b2[col] : 0.0f;
        float res[MI][4];   // R first (see gemm_small_kernel)
#pragma unroll
        for (int i = 0; i < MI; ++i)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int row = m0 + wm * (BM / WM) + i * 16 + row_l + r;
                res[i][r] = (p.R && row < p.M) ? epilogue_residual(p, row, col) : 0.0f;
            }
#pragma unroll
        for (int i = 0; i < MI; ++i) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int row = m0 + wm * (BM / WM) + i * 16 + row_l + r;
                if (row < p.M) p.C[(long)row * p.ldc + col] = epilogue_value_r(p, acc[i][j][r], row, col, s1, b1, s2, b2, res[i][r]);
            }
        }
    }
}

// ======================================================================================================
// big-tile kernel (tile id 5; M in the thousands): 128 x 128 tile of C per workgroup, K stages of 32, the four waves
// each own 64 x 64 as 2 x 2 tiles of v_mfma_f32_32x32x16_bf16 (lane l: r = l & 31, h = l >> 5 holds A[row r][k 8h..+7],
// B[k 8h..+7][col r]; D[row (reg & 3) + 8 (reg >> 2) + 4 h][col r]).  Per MFMA this shape needs half the LDS fragment
// reads of the 16 x 16 x 32 one, and the 128-wide tile halves the LDS stores per MFMA, which is what bounds the 64 x 64
// kernel (the LDS store path moves ~75 B/clk/CU).  A is staged as in the other kernels (bf16 hi + lo, 80-byte rows);
// W blocks are copied to LDS in their packed order and a lane picks its 16 bytes out of the 16 x 16 x 32 fragment
// image: 16-column half (r >> 4), k-group 2 q + h of K-step q.
// ======================================================================================================
constexpr int BG_LD = 40;  // bf16 per staged A row (32 + 8 pad = 80 B)

typedef __attribute__((ext_vector_type(16))) float f32x16;

// Epilogue of the LDS-staged kernels: a wave holds TM x TN accumulator tiles of 32 x 32 (D[row (reg & 3) + 8 (reg >> 2) + 4 h][col r])
// whose first row / column inside the workgroup tile are 32 TM wm and 32 TN wn.
template <int TM, int TN>
__device__ __forceinline__ void big_epilogue(const estts_gemm_args& p, f32x16 (&acc)[TM][TN], const int m0, const int n0, const int wm,
                                             const int wn, const int r, const int h, const TileCoord& tc) {
    // column moments of the output (stat_slots): a lane sums its 16 TM rows of the column, the two half-waves are added with
    // one lane swap, and each wave adds its 32 TN columns to copy (row tile % 16) of the sums
    const bool stats = p.stat_slots != nullptr;
    const long stat_cp = (long)((p.N + 255) / 256) * 256;
    // (a wrapped residual -- r_mod: the positional-encoding table added by row modulo the sequence length -- stays in as long as
    // the 28 rows a lane spans wrap at most once)
    const bool plain = m0 + wm * 32 * TM + 32 * TM <= p.M && (p.r_mod <= 0 || p.r_mod >= 32) && !(p.drop_rate > 0.0f);   // uniform over the wave
#pragma unroll
    for (int j = 0; j < TN; ++j) {
        const int col = n0 + wn * 32 * TN + j * 32 + r;
        if (col >= p.N) continue;
        const float s1 = p.s1 ? p.s1[col] : 1.0f;
        const float b1 = p.b1 ? p.b1[col] : 0.0f;
        const float s2 = p.s2 ? p.s2[col] : 1.0f;
        const float b2 = p.s2 ? p.b2[col] : 0.0f;
        const float k0 = (stats && p.stat_shift) ? p.stat_shift[col] : 0.0f;
        float q0 = 0.0f, q1 = 0.0f;
        if (plain) {
            // Straight-line form for a wave whose rows all exist, without dropout: no test per
            // element, the activation chosen once per tile, row addresses by offset from the tile's first row.  (The general
            // form below costs a wave some 2000 instructions -- a row test, a division for r_mod, the activation's branches
            // and a 64-bit address product per element -- and all workgroups of a one-round launch run it at the same time:
            // a third of the launch at K = 512.)
            const int act = p.act;
            const float alpha = p.alpha;
            const bool has_r = p.R != nullptr, has2 = p.s2 != nullptr;
#pragma unroll
            for (int i = 0; i < TM; ++i) {
                const int row0 = m0 + wm * 32 * TM + i * 32 + 4 * h;
                float* cp = p.C + (long)row0 * p.ldc + col;
                float res[16];
                if (has_r) {   // all 16 first (see below)
                    if (p.r_mod > 0) {
                        const int rr0 = row0 % p.r_mod;   // one division per tile instead of one per element
#pragma unroll
                        for (int e = 0; e < 16; ++e) {
                            int rr = rr0 + (e & 3) + 8 * (e >> 2);
                            rr -= rr >= p.r_mod ? p.r_mod : 0;
                            res[e] = p.R[(long)rr * p.ldr + col];
                        }
                    } else {
                        const float* rp = p.R + (long)row0 * p.ldr + col;
#pragma unroll
                        for (int e = 0; e < 16; ++e) res[e] = rp[(long)((e & 3) + 8 * (e >> 2)) * p.ldr];
                    }
                }
                float v[16];
#pragma unroll
                for (int e = 0; e < 16; ++e) v[e] = acc[i][j][e] * s1 + b1;
                if (act == ESTTS_ACT_RELU) {
#pragma unroll
                    for (int e = 0; e < 16; ++e) v[e] = fmaxf(v[e], 0.0f);
                } else if (act == ESTTS_ACT_TANH) {
#pragma unroll
                    for (int e = 0; e < 16; ++e) v[e] = tanhf(v[e]);
                }
#pragma unroll
                for (int e = 0; e < 16; ++e) {
                    v[e] *= alpha;
                    if (has_r) v[e] += res[e];
                    if (has2) v[e] = v[e] * s2 + b2;
                }
#pragma unroll
                for (int e = 0; e < 16; ++e) {
                    cp[(long)((e & 3) + 8 * (e >> 2)) * p.ldc] = v[e];
                    const float d = v[e] - k0;
                    q0 += d;
                    q1 += d * d;
                }
            }
        } else
#pragma unroll
        for (int i = 0; i < TM; ++i) {
            // the tile's 16 values of R first: R may be C itself (accumulation), so the compiler has to keep every load behind
            // the stores before it, and loads left between the stores become 16 dependent round trips
            // (19200 x 256 x 256 with R: 28.3 us against 18.9 us without)
            float res[16];
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int row = m0 + wm * 32 * TM + i * 32 + (e & 3) + 8 * (e >> 2) + 4 * h;
                res[e] = (p.R && row < p.M) ? epilogue_residual(p, row, col) : 0.0f;
            }
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int row = m0 + wm * 32 * TM + i * 32 + (e & 3) + 8 * (e >> 2) + 4 * h;
                if (row < p.M) {
                    const float v = epilogue_value_r(p, acc[i][j][e], row, col, s1, b1, s2, b2, res[e]);
                    p.C[(long)row * p.ldc + col] = v;
                    const float d = v - k0;
                    q0 += d;
                    q1 += d * d;
                }
            }
        }
        if (stats) {
            q0 += lane_xor32(q0);
            q1 += lane_xor32(q1);
            if (h == 0) {
                float* dst = p.stat_slots + (long)(tc.ty % 16) * 2 * stat_cp + col;
                atomic_add_f32(dst, q0);
                atomic_add_f32(dst + stat_cp, q1);
            }
        }
    }
}


// TM x TN = 32 x 32 MFMA tiles per wave: the workgroup tile is (64 TM) x (64 TN).  2 x 2 is the kernel described above;
// 1 x 1, 2 x 1 and 1 x 2 keep its main loop for launches whose 128 x 128 tiles would leave most CUs idle.
template <int NSPLIT, int TM, int TN, int OCC, int KIND>
__global__ __launch_bounds__(256, OCC) void gemm_big_kernel(const estts_gemm_args p_in) {
    constexpr int NS = (NSPLIT > 1) ? 2 : 1;
    constexpr int BM = 64 * TM, BN = 64 * TN;
    constexpr int AI = 2 * TM;   // float4 loads of A per thread and K-step
    __shared__ __attribute__((aligned(16))) __bf16 As[2][NS][BM * BG_LD];
    __shared__ __attribute__((aligned(16))) __bf16 Ws[2][NS][4 * TN * 512];  // 16-column blocks of one 32-wide K-step

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int r = lane & 31, h = lane >> 5;
    const TileCoord tc = xcd_tile_coord((p_in.tune & (1 << 25)) ? false : gemm_y_first(p_in), !(p_in.tune & (1 << 24)));
    estts_gemm_args p = p_in;
    if (p.batch > 1) {   // gridDim.z = batch: product tc.tz of the launch
        const long z = tc.tz;
        p.A += z * p.batch_stride_a;
        p.W_hi += z * p.batch_stride_w;
        if (p.W_lo) p.W_lo += z * p.batch_stride_w;
        p.C += z * p.batch_stride_c;
        if (p.R) p.R += z * p.batch_stride_c;
        if (p.s1) p.s1 += z * p.batch_stride_v;
        if (p.b1) p.b1 += z * p.batch_stride_v;
        if (p.s2) { p.s2 += z * p.batch_stride_v; p.b2 += z * p.batch_stride_v; }
    }
    const int m0 = tc.ty * BM;
    const int n0 = tc.tx * BN;
    const int KS = p.ldw >> 5;

    const int nb_end = ((p.N + 63) / 64) * 4;  // 16-column blocks present in W
    const int a_ch = tid & 7;    // float4 chunk of the 32-wide K slice
    const int a_r0 = tid >> 3;   // rows a_r0 + 32 i
    ARow a_row[AI];
    const float* a_ptr[AI];   // KIND 1 / 2: start of the row (rows past M re-read the last one; their results are never stored)
    int a_pos[AI];            // KIND 2: position of the row inside its sequence
#pragma unroll
    for (int i = 0; i < AI; ++i) {
        const int row = min(m0 + a_r0 + 32 * i, p.M - 1);
        if (KIND == 1) a_ptr[i] = p.A + (long)row * p.lda + a_ch * 4;
        else if (KIND == 2) { a_ptr[i] = p.A + (long)row * p.lda; a_pos[i] = row % p.conv_w; }
        else a_row[i] = make_arow(p, m0 + a_r0 + 32 * i);
    }

    // two register sets: the loads of K-step kt + 2 are issued while step kt is being multiplied, so a set has two
    // multiply phases (of this workgroup and of the one sharing the CU) to arrive before it is split and stored to LDS;
    // with one set in flight the memory latency of a large launch (over 2000 clocks) showed in every K-step
    // (plain locals handed over by reference: gathered in a struct, or as an array of W registers, the compiler keeps them in
    // scratch memory)
    static_assert(TN == 1 || TN == 2, "one or two 16-byte W units per thread");
    KCursor kcur = make_kcursor(p, a_ch * 4);   // load_tiles is called for kt = 0, 1, 2, ... in order
    auto load_tiles = [&](float4 (&ga)[AI], bool (&gok)[AI], uint4& gwh0, uint4& gwh1, uint4& gwl0, uint4& gwl1, int kt) {
        if (KIND == 1) {
            const bool ok = kt * 32 + a_ch * 4 < p.K;
#pragma unroll
            for (int i = 0; i < AI; ++i) {
                ga[i] = *reinterpret_cast<const float4*>(a_ptr[i] + (ok ? kt * 32 : -a_ch * 4));
                gok[i] = ok;
            }
        } else if (KIND == 2) {
            const int dw = kcur.tap - p.conv_pl;
            const long off = (long)dw * p.lda + kcur.c;
            const bool kok = kcur.kk < p.K;
            advance(p, kcur, 32);
#pragma unroll
            for (int i = 0; i < AI; ++i) {
                const bool ok = kok && (unsigned)(a_pos[i] + dw) < (unsigned)p.conv_w;
                ga[i] = *reinterpret_cast<const float4*>(a_ptr[i] + (ok ? off : 0L));
                gok[i] = ok;
            }
        } else {
            const ATap tp = atap_from(p, kcur.kk, kcur.tap, kcur.c);
            advance(p, kcur, 32);
#pragma unroll
            for (int i = 0; i < AI; ++i) ga[i] = load_a_raw(p, a_row[i], tp, gok[i]);
        }
        auto w_off = [&](int i) {
            // W holds ceil(N / 64) * 64 rows, so the last tile may reach past it: read the last block again instead
            // (those columns are >= N and never stored); no load sits under a condition (see load_a_raw)
            const int idx = tid + i * 256;  // 16-byte unit: block idx >> 6, lane slot idx & 63
            const int nb = min((n0 >> 4) + (idx >> 6), nb_end - 1);
            return ((long)nb * KS + kt) * 512 + (idx & 63) * 8;
        };
        const long o0 = w_off(0);
        gwh0 = *reinterpret_cast<const uint4*>(p.W_hi + o0);
        if (NS == 2) gwl0 = *reinterpret_cast<const uint4*>(p.W_lo + o0);
        if (TN == 2) {
            const long o1 = w_off(1);
            gwh1 = *reinterpret_cast<const uint4*>(p.W_hi + o1);
            if (NS == 2) gwl1 = *reinterpret_cast<const uint4*>(p.W_lo + o1);
        }
    };
    auto store_tiles = [&](const float4 (&ga)[AI], const bool (&gok)[AI], const uint4& gwh0, const uint4& gwh1, const uint4& gwl0,
                           const uint4& gwl1, int buf) {
#pragma unroll
        for (int i = 0; i < AI; ++i) {
            const int o = (a_r0 + 32 * i) * BG_LD + a_ch * 4;
            // KIND 1 needs no zeroing: a K index past the end re-reads valid data of the row, and W is zero-padded there
            const float4 az = KIND == 1 ? ga[i] : zero_unless(gok[i], ga[i]);
            bf16x4 hi, lo;
            split_bf16x4(az, hi, lo);
            *reinterpret_cast<bf16x4*>(&As[buf][0][o]) = hi;
            if (NS == 2) *reinterpret_cast<bf16x4*>(&As[buf][NS - 1][o]) = lo;
        }
        *reinterpret_cast<uint4*>(&Ws[buf][0][tid * 8]) = gwh0;
        if (NS == 2) *reinterpret_cast<uint4*>(&Ws[buf][NS - 1][tid * 8]) = gwl0;
        if (TN == 2) {
            *reinterpret_cast<uint4*>(&Ws[buf][0][(tid + 256) * 8]) = gwh1;
            if (NS == 2) *reinterpret_cast<uint4*>(&Ws[buf][NS - 1][(tid + 256) * 8]) = gwl1;
        }
    };

    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.0f;

    const int nk = (p.K + 31) / 32;
    const int a_base = (wm * 32 * TM + r) * BG_LD + 8 * h;
    const int w_base = (wn * 2 * TN + (r >> 4)) * 512 + ((r & 15) + 16 * h) * 8;

    auto multiply = [&](int buf) {
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            bf16x8 af[NS][TM], wf[NS][TN];
#pragma unroll
            for (int s = 0; s < NS; ++s) {
#pragma unroll
                for (int i = 0; i < TM; ++i)
                    af[s][i] = *reinterpret_cast<const bf16x8*>(&As[buf][s][a_base + i * 32 * BG_LD + q * 16]);
#pragma unroll
                for (int j = 0; j < TN; ++j)
                    wf[s][j] = *reinterpret_cast<const bf16x8*>(&Ws[buf][s][w_base + j * 2 * 512 + q * 32 * 8]);
            }
            if (NS == 2) {
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int j = 0; j < TN; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[NS - 1][i], wf[0][j], acc[i][j], 0, 0, 0);
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int j = 0; j < TN; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[0][i], wf[NS - 1][j], acc[i][j], 0, 0, 0);
            }
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[0][i], wf[0][j], acc[i][j], 0, 0, 0);
        }
    };

    float4 a0[AI], a1[AI];
    bool ok0[AI], ok1[AI];
    uint4 u0, u1, u2, u3, v0, v1, v2, v3;
    // The steady-state loop holds NO load under a condition, and can only be entered behind two unconditional sets of loads.
    // The compiler's s_waitcnt counts are static: were a set's loads conditional (if (kt + 2 < nk) ...), the path that skips
    // them would make the older set the youngest loads in flight, so every iteration would wait for vmcnt(0) before its
    // first LDS store, i.e. for the set requested one multiply phase ago, and the second register set would buy nothing.
    int kt = 0;
    if (nk >= 4) {
        load_tiles(a0, ok0, u0, u1, u2, u3, 0);
        __builtin_amdgcn_sched_barrier(0);   // the same goes for the order on entry: all of set 0 before any of set 1
        load_tiles(a1, ok1, v0, v1, v2, v3, 1);
        __builtin_amdgcn_sched_barrier(0);
        for (; kt + 3 < nk; kt += 2) {
            store_tiles(a0, ok0, u0, u1, u2, u3, 0);
            __syncthreads();
            load_tiles(a0, ok0, u0, u1, u2, u3, kt + 2);
            multiply(0);
            store_tiles(a1, ok1, v0, v1, v2, v3, 1);
            __syncthreads();
            load_tiles(a1, ok1, v0, v1, v2, v3, kt + 3);
            multiply(1);
        }
    } else {
        load_tiles(a0, ok0, u0, u1, u2, u3, 0);
        if (nk > 1) load_tiles(a1, ok1, v0, v1, v2, v3, 1);
    }
    // tail: K-steps kt and kt + 1 are in the registers, kt + 2 (nk - kt = 3) is still to be requested
    store_tiles(a0, ok0, u0, u1, u2, u3, 0);
    __syncthreads();
    if (kt + 2 < nk) load_tiles(a0, ok0, u0, u1, u2, u3, kt + 2);
    multiply(0);
    if (kt + 1 < nk) {
        store_tiles(a1, ok1, v0, v1, v2, v3, 1);
        __syncthreads();
        multiply(1);
    }
    if (kt + 2 < nk) {
        store_tiles(a0, ok0, u0, u1, u2, u3, 0);   // every wave is past the barrier in front of multiply(1): buffer 0 is free
        __syncthreads();
        multiply(0);
    }

    big_epilogue<TM, TN>(p, acc, m0, n0, wm, wn, r, h, tc);
}

// ======================================================================================================
// W-from-memory kernel (tile id 13): the 64 x 128 tile with half the LDS traffic.  The four waves sit side by side over N:
// wave w owns all 64 rows x columns 32 w .. 32 w + 31 (two 32 x 32 accumulator tiles), so no W byte is needed by two waves
// and W never passes through LDS -- packed W is already the B-operand register image, a lane fetches its 16 bytes of
// K-step q with one coalesced load (four 256-byte runs per wave) straight into the registers the matrix instruction
// reads.  Only A is staged (split on its way in, as in the kernel above).  Per workgroup and 32-wide K-step: LDS stores
// 8 KB instead of 24, LDS fragment reads 32 KB instead of 48, the same 24 KB from memory and the same 48 matrix
// instructions.  Same arithmetic in the same order as the 64 x 128 tile of the kernel above: identical results.
// ======================================================================================================
#ifdef ESTTS_GEMM_STAMPS
// Diagnostic build (-DESTTS_GEMM_STAMPS, tools/gemm_stamps.py): lane 0 of every wave of ONE workgroup in the middle of the grid
// adds up the shader-clock cycles between the marks of the steady-state loop.  The marks constrain the instruction scheduler,
// so the kernel under test is not quite the shipped one.
__device__ unsigned long long gemm_stamps[4][8];
#define GS_DECL unsigned long long gs_acc[8] = {0, 0, 0, 0, 0, 0, 0, 0}, gs_prev = clock64(), gs_first = gs_prev; \
    const bool gs_on = (threadIdx.x & 63) == 0 && blockIdx.x == 0 && blockIdx.y == gridDim.y / 2
#define GS(i) do { const unsigned long long t_ = clock64(); gs_acc[i] += t_ - gs_prev; gs_prev = t_; } while (0)
#define GS_RESET do { gs_prev = clock64(); } while (0)
#define GS_WRITE do { if (gs_on) { for (int i_ = 0; i_ < 7; ++i_) gemm_stamps[threadIdx.x >> 6][i_] = gs_acc[i_]; \
    gemm_stamps[threadIdx.x >> 6][7] = clock64() - gs_first; } } while (0)
}  // namespace
extern "C" int estts_debug_gemm_stamps(unsigned long long* out) { return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(gemm_stamps), sizeof(gemm_stamps)); }
namespace {
#else
#define GS_DECL
#define GS(i)
#define GS_RESET
#define GS_WRITE
#endif

// NWV = 8: eight waves side by side, a workgroup owns 64 rows x 256 columns (id 14).  LNE (NWV = 8, N = 256: the workgroup
// owns whole rows): the epilogue is estts_rownorm, C = LNb(act_ln(LNa(acc s1 + b1)) + ln_x) -- the tile goes through LDS (in the
// place of the A images) so that a wave normalises whole rows exactly as rownorm_fast_kernel does, and writes them as float4.
template <int NSPLIT, int TM, int NWV, bool LNE, int KG = 1>
constexpr int wdirect_smem_bytes() {
    constexpr int a_bytes = KG * 2 * ((NSPLIT > 1) ? 2 : 1) * 32 * TM * BG_LD * 2;  // per K group: two buffers of the hi (+ lo) A images
    constexpr int t_bytes = LNE ? 32 * TM * (32 * NWV + 4) * 4 : 0;                // the epilogue's float32 tile
    return a_bytes > t_bytes ? a_bytes : t_bytes;
}

// the tile with first row m0 and first column n0; smem: wdirect_smem_bytes() of LDS
// KG = 2 (id 15; TM = 2, K a multiple of 64): two groups of NWV waves take the two halves of K for the same tile, each staging
// its own A images, and exchange halves of their accumulators through LDS at the end (group g finishes rows 32 g .. 32 g + 31).
// For launches with at most one tile per CU, where a K loop runs at the latency of its own steps: half as many steps.
template <int NSPLIT, int TM, int KIND, int NWV, bool LNE, int KG = 1>
__device__ __forceinline__ void wdirect_tile(const estts_gemm_args& p, unsigned char* smem, const int m0, const int n0, const TileCoord tc) {
    constexpr int NS = (NSPLIT > 1) ? 2 : 1;
    constexpr int BM = 32 * TM, BN = 32 * NWV;   // TM = 2 (or 4: 128 rows, half the W loads per matrix instruction)
    constexpr int RP = 8 * NWV;  // rows the workgroup's threads cover with one float4 each
    constexpr int AI = (BM + RP - 1) / RP;  // float4 loads of A per thread and K-step
    constexpr int T_LD = BN + 4;                           // floats per row of the epilogue's tile
    static_assert(KG == 1 || (KG == 2 && TM == 2 && !LNE), "two K groups: 64-row tiles, plain epilogue");
    const int kg = KG > 1 ? (int)threadIdx.x / (64 * NWV) : 0;     // K group of this wave (uniform per wave)
    __bf16 (*As)[NS][BM * BG_LD] = reinterpret_cast<__bf16 (*)[NS][BM * BG_LD]>(smem) + 2 * kg;

    const int tid = KG > 1 ? (int)threadIdx.x % (64 * NWV) : (int)threadIdx.x;   // thread within the K group
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int r = lane & 31, h = lane >> 5;
    const int KS = p.ldw >> 5;

    const int nb_end = ((p.N + 63) / 64) * 4;  // 16-column blocks present in W
    const int a_ch = tid & 7;    // float4 chunk of the 32-wide K slice
    const int a_r0 = tid >> 3;   // rows a_r0 + RP i
    // (32-row tiles of eight waves: the upper four waves stage nothing; uniform per wave)
    ARow a_row[AI];
    const float* a_ptr[AI];   // KIND 1 / 2: start of the row (rows past M re-read the last one; their results are never stored)
    int a_pos[AI];            // KIND 2: position of the row inside its sequence
#pragma unroll
    for (int i = 0; i < AI; ++i) {
        const int row = min(m0 + a_r0 + RP * i, p.M - 1);
        if (KIND == 1) a_ptr[i] = p.A + (long)row * p.lda + a_ch * 4;
        else if (KIND == 2) { a_ptr[i] = p.A + (long)row * p.lda; a_pos[i] = row % p.conv_w; }
        else a_row[i] = make_arow(p, m0 + a_r0 + RP * i);
    }
    const int nk = (p.K + 31) / 32 / KG;        // K-steps of this group: kt0 .. kt0 + nk - 1
    const int kt0 = kg * nk;
    KCursor kcur = make_kcursor(p, kt0 * 32 + a_ch * 4);   // load_a is called for kt = 0, 1, 2, ... in order
    auto load_a = [&](float4 (&ga)[AI], bool (&gok)[AI], int kt) {
        if constexpr (BM < RP) { if (a_r0 >= BM) return; }
        if (KIND == 1) {
            const bool ok = (kt0 + kt) * 32 + a_ch * 4 < p.K;
#pragma unroll
            for (int i = 0; i < AI; ++i) {
                ga[i] = *reinterpret_cast<const float4*>(a_ptr[i] + (ok ? (kt0 + kt) * 32 : -a_ch * 4));
                gok[i] = ok;
            }
        } else if (KIND == 2) {
            const int dw = kcur.tap - p.conv_pl;
            const long off = (long)dw * p.lda + kcur.c;
            const bool kok = kcur.kk < p.K;
            advance(p, kcur, 32);
#pragma unroll
            for (int i = 0; i < AI; ++i) {
                const bool ok = kok && (unsigned)(a_pos[i] + dw) < (unsigned)p.conv_w;
                ga[i] = *reinterpret_cast<const float4*>(a_ptr[i] + (ok ? off : 0L));
                gok[i] = ok;
            }
        } else {
            const ATap tp = atap_from(p, kcur.kk, kcur.tap, kcur.c);
            advance(p, kcur, 32);
#pragma unroll
            for (int i = 0; i < AI; ++i) ga[i] = load_a_raw(p, a_row[i], tp, gok[i]);
        }
    };
    auto store_a = [&](const float4 (&ga)[AI], const bool (&gok)[AI], int buf) {
        if constexpr (BM < RP) { if (a_r0 >= BM) return; }
#pragma unroll
        for (int i = 0; i < AI; ++i) {
            const int o = (a_r0 + RP * i) * BG_LD + a_ch * 4;
            // KIND 1 needs no zeroing: a K index past the end re-reads valid data of the row, and W is zero-padded there
            const float4 az = KIND == 1 ? ga[i] : zero_unless(gok[i], ga[i]);
            bf16x4 hi, lo;
            split_bf16x4(az, hi, lo);
            *reinterpret_cast<bf16x4*>(&As[buf][0][o]) = hi;
            if (NS == 2) *reinterpret_cast<bf16x4*>(&As[buf][NS - 1][o]) = lo;
        }
    };
    // this lane's 16 bytes of W for the two 16-wide halves q of K-step kt: 16-column block (n0 + 32 wave) / 16 + (r >> 4)
    // (clamped: W holds ceil(N / 64) * 64 rows, columns >= N are never stored), lane slot (r & 15) + 16 (2 q + h) of its 1 KiB
    const int w_nb = min((n0 >> 4) + wave * 2 + (r >> 4), nb_end - 1);
    const long w_lane = (long)w_nb * KS * 512 + ((r & 15) + 16 * h) * 8;
    auto load_w = [&](bf16x8 (&wf)[NS][2], int kt) {
        const long o = w_lane + (long)(kt0 + kt) * 512;
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            wf[0][q] = *reinterpret_cast<const bf16x8*>(p.W_hi + o + q * 32 * 8);
            if (NS == 2) wf[NS - 1][q] = *reinterpret_cast<const bf16x8*>(p.W_lo + o + q * 32 * 8);
        }
    };

    f32x16 acc[TM][1];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[i][0][e] = 0.0f;

    const int a_base = r * BG_LD + 8 * h;
    auto multiply = [&](int buf, const bf16x8 (&wf)[NS][2]) {
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            bf16x8 af[NS][TM];
#pragma unroll
            for (int s = 0; s < NS; ++s)
#pragma unroll
                for (int i = 0; i < TM; ++i)
                    af[s][i] = *reinterpret_cast<const bf16x8*>(&As[buf][s][a_base + i * 32 * BG_LD + q * 16]);
            auto mma = [&](const bf16x8& a, const bf16x8& w, f32x16 c) { return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, w, c, 0, 0, 0); };
            if (NS == 2) {
#pragma unroll
                for (int i = 0; i < TM; ++i) acc[i][0] = mma(af[NS - 1][i], wf[0][q], acc[i][0]);
#pragma unroll
                for (int i = 0; i < TM; ++i) acc[i][0] = mma(af[0][i], wf[NS - 1][q], acc[i][0]);
            }
#pragma unroll
            for (int i = 0; i < TM; ++i) acc[i][0] = mma(af[0][i], wf[0][q], acc[i][0]);
        }
    };

    // two register sets of each operand; steady state without a load under a condition (see the kernel above).  A set of W
    // is requested again as soon as its K-step has been multiplied, a set of A as soon as it has been staged.
    float4 a0[AI], a1[AI];
    bool ok0[AI], ok1[AI];
    bf16x8 w0[NS][2], w1[NS][2];
    int kt = 0;
    GS_DECL;
    if (nk >= 4) {
        load_a(a0, ok0, 0);
        load_w(w0, 0);
        __builtin_amdgcn_sched_barrier(0);
        load_a(a1, ok1, 1);
        load_w(w1, 1);
        __builtin_amdgcn_sched_barrier(0);
        GS_RESET;
        for (; kt + 3 < nk; kt += 2) {
            store_a(a0, ok0, 0);
            GS(0);
            __syncthreads();
            GS(1);
            load_a(a0, ok0, kt + 2);
            GS(2);
            multiply(0, w0);
            GS(3);
            load_w(w0, kt + 2);
            GS(4);
            store_a(a1, ok1, 1);
            GS(0);
            __syncthreads();
            GS(1);
            load_a(a1, ok1, kt + 3);
            GS(2);
            multiply(1, w1);
            GS(3);
            load_w(w1, kt + 3);
            GS(4);
        }
    } else {
        load_a(a0, ok0, 0);
        load_w(w0, 0);
        if (nk > 1) {
            load_a(a1, ok1, 1);
            load_w(w1, 1);
        }
    }
    // tail: K-steps kt and kt + 1 are in the registers, kt + 2 (nk - kt = 3) is still to be requested
    store_a(a0, ok0, 0);
    __syncthreads();
    if (kt + 2 < nk) load_a(a0, ok0, kt + 2);
    multiply(0, w0);
    if (kt + 2 < nk) load_w(w0, kt + 2);
    if (kt + 1 < nk) {
        store_a(a1, ok1, 1);
        __syncthreads();
        multiply(1, w1);
    }
    if (kt + 2 < nk) {
        store_a(a0, ok0, 0);   // every wave is past the barrier in front of multiply(1): buffer 0 is free
        __syncthreads();
        multiply(0, w0);
    }

    GS(5);
    if constexpr (LNE) {
        // the row operands of the norms first: their round trips run beside the trip of the tile through LDS
        constexpr int RW = BM / NWV;   // rows per wave; lane <-> columns 4 lane .. 4 lane + 3, as in rownorm_fast_kernel
        const int c = lane * 4;
        const bool lna = p.ln_ga != nullptr, lnb = p.ln_gb != nullptr, res = p.ln_x != nullptr;   // uniform
        const float4 zero4 = make_float4(0.f, 0.f, 0.f, 0.f);
        float4 v[RW], rx[RW];
        const int row0 = m0 + wave * RW;
        float4 ga = zero4, ba = zero4, gb = zero4, bb = zero4;
#pragma unroll
        for (int q = 0; q < RW; ++q) rx[q] = zero4;
        if (res) {   // (uniform branches: a load chosen by a condition becomes a load through a generic pointer)
#pragma unroll
            for (int q = 0; q < RW; ++q)   // a surplus slot repeats the last row and is not stored
                rx[q] = *reinterpret_cast<const float4*>(p.ln_x + (long)min(row0 + q, p.M - 1) * p.ln_ldx + c);
        }
        if (lna) {
            ga = *reinterpret_cast<const float4*>(p.ln_ga + c);
            ba = *reinterpret_cast<const float4*>(p.ln_ba + c);
        }
        if (lnb) {
            gb = *reinterpret_cast<const float4*>(p.ln_gb + c);
            bb = *reinterpret_cast<const float4*>(p.ln_bb + c);
        }
        __syncthreads();   // every wave is past its last fragment read: the tile takes the place of the A images
        float* T = reinterpret_cast<float*>(smem);
        {
            const int col = wave * 32 + r;   // n0 = 0
            const float s1 = p.s1 ? p.s1[col] : 1.0f, b1 = p.b1 ? p.b1[col] : 0.0f;
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int e = 0; e < 16; ++e) T[(i * 32 + (e & 3) + 8 * (e >> 2) + 4 * h) * T_LD + col] = acc[i][0][e] * s1 + b1;
        }
        __syncthreads();
#pragma unroll
        for (int q = 0; q < RW; ++q) v[q] = *reinterpret_cast<const float4*>(T + (wave * RW + q) * T_LD + c);
        auto layer_norm = [&](const float4& g, const float4& b) {
            float mean[RW], inv[RW];
#pragma unroll
            for (int q = 0; q < RW; ++q) mean[q] = v[q].x + v[q].y + v[q].z + v[q].w;
#pragma unroll
            for (int q = 0; q < RW; ++q) mean[q] = wave_sum(mean[q]) / (float)BN;
#pragma unroll
            for (int q = 0; q < RW; ++q) {
                const float a = v[q].x - mean[q], b2 = v[q].y - mean[q], c2 = v[q].z - mean[q], d = v[q].w - mean[q];
                inv[q] = a * a + b2 * b2 + c2 * c2 + d * d;
            }
#pragma unroll
            for (int q = 0; q < RW; ++q) inv[q] = rsqrtf(wave_sum(inv[q]) / (float)BN + p.ln_eps);
#pragma unroll
            for (int q = 0; q < RW; ++q) {
                v[q].x = (v[q].x - mean[q]) * inv[q] * g.x + b.x;
                v[q].y = (v[q].y - mean[q]) * inv[q] * g.y + b.y;
                v[q].z = (v[q].z - mean[q]) * inv[q] * g.z + b.z;
                v[q].w = (v[q].w - mean[q]) * inv[q] * g.w + b.w;
            }
        };
        if (lna) layer_norm(ga, ba);
#pragma unroll
        for (int q = 0; q < RW; ++q) {
            v[q].x = apply_act(v[q].x, p.ln_act) + rx[q].x;
            v[q].y = apply_act(v[q].y, p.ln_act) + rx[q].y;
            v[q].z = apply_act(v[q].z, p.ln_act) + rx[q].z;
            v[q].w = apply_act(v[q].w, p.ln_act) + rx[q].w;
        }
        if (lnb) layer_norm(gb, bb);
#pragma unroll
        for (int q = 0; q < RW; ++q)
            if (row0 + q < p.M) *reinterpret_cast<float4*>(p.C + (long)(row0 + q) * p.ldc + c) = v[q];
    } else if constexpr (KG == 2) {
        __syncthreads();   // every wave is past its last fragment read: the A images make room for the exchange
        float* X = reinterpret_cast<float*>(smem);   // [group][wave][16][64]: the half the other group finishes
        {
            float* mine = X + ((kg * NWV + wave) * 16) * 64 + lane;
#pragma unroll
            for (int e = 0; e < 16; ++e) mine[e * 64] = kg == 0 ? acc[1][0][e] : acc[0][0][e];
        }
        __syncthreads();
        f32x16 half[1][1];
        {
            const float* theirs = X + (((1 - kg) * NWV + wave) * 16) * 64 + lane;
#pragma unroll
            for (int e = 0; e < 16; ++e) half[0][0][e] = (kg == 0 ? acc[0][0][e] : acc[1][0][e]) + theirs[e * 64];
        }
        big_epilogue<1, 1>(p, half, m0 + 32 * kg, n0, 0, wave, r, h, tc);
    } else {
        big_epilogue<TM, 1>(p, acc, m0, n0, 0, wave, r, h, tc);
    }
    GS(6);
    GS_WRITE;
}

// Tile order of a launch region of gx x gy tiles whose workgroups the hardware hands to the XCDs round robin (id & 7): as
// xcd_tile_coord, for a region that is not the whole grid.
__device__ __forceinline__ TileCoord region_tile_coord(int id, const int gx, const int gy, const bool y_first, const bool remap) {
    const int nwg = gx * gy;
    if (remap && nwg >= 16) {
        const int q = nwg >> 3, r = nwg & 7, xcd = id & 7;
        id = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (id >> 3);
    }
    TileCoord t;
    t.tz = 0;
    if (y_first) { t.ty = id % gy; t.tx = id / gy; }
    else         { t.tx = id % gx; t.ty = id / gx; }
    return t;
}

// Rows in 64-row tiles and, behind them, 32-row tiles: workgroups 0 .. n_big - 1 (dispatched first) own 64 rows each, the rest
// 32 (rows gy_big * 64 and up).  Workgroups of one launch are all resident at these sizes, so a launch takes as long as its
// busiest CU: 19 200 x 256 as 600 equal tiles of 64 x 128 puts three tiles on 88 CUs and two on the others; as 512 + 176 the
// busiest CU has two and a half.
// id 15: 64 x 128 tiles, eight waves: two groups of four take the halves of K
template <int NSPLIT, int KIND>
__global__ __launch_bounds__(512, 2) void gemm_wdirect_k2_kernel(const estts_gemm_args p_in) {
    __shared__ __attribute__((aligned(16))) unsigned char smem[wdirect_smem_bytes<NSPLIT, 2, 4, false, 2>()];
    const TileCoord tc = xcd_tile_coord((p_in.tune & (1 << 25)) ? false : gemm_y_first(p_in), !(p_in.tune & (1 << 24)));
    wdirect_tile<NSPLIT, 2, KIND, 4, false, 2>(p_in, smem, tc.ty * 64, tc.tx * 128, tc);
}

template <int NSPLIT, int OCC, int KIND, int NWV = 4, bool LNE = false>
__global__ __launch_bounds__(64 * NWV, OCC) void gemm_wdirect_kernel(const estts_gemm_args p_in, const int n_big, const int gy_big, const int gy_small) {
    __shared__ __attribute__((aligned(16))) unsigned char smem[wdirect_smem_bytes<NSPLIT, 2, NWV, LNE>()];
    constexpr int BN = 32 * NWV;
    const int gx = (p_in.N + BN - 1) / BN;
    const bool y_first = (p_in.tune & (1 << 25)) ? false : gemm_y_first(p_in);
    const bool remap = !(p_in.tune & (1 << 24));
    const int id = blockIdx.x;
    if (id < n_big) {
        const TileCoord tc = region_tile_coord(id, gx, gy_big, y_first, remap);
        wdirect_tile<NSPLIT, 2, KIND, NWV, LNE>(p_in, smem, tc.ty * 64, tc.tx * BN, tc);
    } else {
        const TileCoord tc = region_tile_coord(id - n_big, gx, gy_small, y_first, remap && (n_big & 7) == 0);
        wdirect_tile<NSPLIT, 1, KIND, NWV, LNE>(p_in, smem, gy_big * 64 + tc.ty * 32, tc.tx * BN, tc);
    }
}

// ------------------------------------------------------------------------------------------------------
thread_local const char* g_entry = "estts_gemm";   // entry point named in the split-K size error
// estts_gemm_noalloc: tickets to clear on the launch stream in front of a launch that really splits (null otherwise)
thread_local void* g_clear_tickets = nullptr;
thread_local size_t g_clear_bytes = 0;

bool splitk_fits(const estts_gemm_args& a, dim3 grid, long tile_elems) {
    const long need = (long)grid.z * grid.x * grid.y * tile_elems * sizeof(float);
    if (!a.ws || !a.counters || need > a.ws_bytes || (long)grid.x * grid.y > a.n_counters) {
        estts_set_error("%s: split-K of this launch needs %ld scratch bytes and %ld tickets, the caller's buffer holds %ld and %d",
                        g_entry, need, (long)grid.x * grid.y, (long)a.ws_bytes, a.n_counters);
        return false;
    }
    return true;
}

// the static tickets must be zero when a splitting kernel starts; a finished launch leaves them zero, an aborted one may not
bool clear_static_tickets(hipStream_t st) {
    if (!g_clear_tickets) return true;
    const hipError_t e = hipMemsetAsync(g_clear_tickets, 0, g_clear_bytes, st);
    if (e != hipSuccess) {
        estts_set_error("%s: clearing the tickets failed: %s", g_entry, hipGetErrorString(e));
        return false;
    }
    return true;
}

// 1: dense layer, 2: 1-D convolution (the geometry estts_gemm fills in for both: rows of conv_w pixels, one kernel row,
// no stride), 0: anything else
int operand_kind(const estts_gemm_args& a) {
    const bool one_d = a.conv_h == 1 && a.conv_stride == 1 && a.conv_pt == 0 && a.conv_kw == a.taps && a.conv_wo == a.conv_w &&
                       a.conv_howo == a.conv_w;
    if (!one_d) return 0;
    return (a.taps == 1 && a.conv_pl == 0) ? 1 : 2;
}

template <int NW, int KIND>
int launch_small_nw(const estts_gemm_args& a, dim3 grid, hipStream_t st) {
    if (a.precision == 3)
        ESTTS_LAUNCH((gemm_small_kernel<3, NW, KIND>), grid, dim3(NW * 64), 0, st, a, (int)grid.z);
    else
        ESTTS_LAUNCH((gemm_small_kernel<1, NW, KIND>), grid, dim3(NW * 64), 0, st, a, (int)grid.z);
    ESTTS_CHECK_LAUNCH("estts_gemm");
    return 0;
}

int launch_small(const estts_gemm_args& a, int splitk, int nw, hipStream_t st) {
    dim3 grid((a.N + 31) / 32, (a.M + 31) / 32, splitk);
    if (splitk > 1 && !splitk_fits(a, grid, 1024)) return -1;
    if (splitk > 1 && !clear_static_tickets(st)) return -2;
    if (nw == 0) {
        // measured on MI355X: 8 waves pay from about 12 K-steps up, but only while the launch has fewer workgroups than
        // the chip has CUs (beyond that 4-wave workgroups fill the SIMDs better: 832 x 1536 x 512 runs 19 vs 22 us);
        // 16 waves never (register spills at 128 VGPRs)
        const int nks = ((a.K + 31) / 32 + splitk - 1) / splitk;
        const long wgs = (long)grid.x * grid.y * grid.z;
        nw = (nks >= 12 && wgs < 256) ? 8 : 4;
    }
    const int kind = operand_kind(a);
    switch (nw) {
        case 4: return kind == 1 ? launch_small_nw<4, 1>(a, grid, st) : kind == 2 ? launch_small_nw<4, 2>(a, grid, st) : launch_small_nw<4, 0>(a, grid, st);
        case 8: return kind == 1 ? launch_small_nw<8, 1>(a, grid, st) : kind == 2 ? launch_small_nw<8, 2>(a, grid, st) : launch_small_nw<8, 0>(a, grid, st);
        case 16: return launch_small_nw<16, 0>(a, grid, st);
        default: estts_set_error("estts_gemm: waves per workgroup must be 4, 8 or 16"); return -1;
    }
}

template <int BM, int BN, int WM, int WN>
int launch(const estts_gemm_args& a, int splitk, hipStream_t st) {
    dim3 grid((a.N + BN - 1) / BN, (a.M + BM - 1) / BM, splitk);
    if (splitk > 1 && !splitk_fits(a, grid, BM * BN)) return -1;
    if (splitk > 1 && !clear_static_tickets(st)) return -2;
    if (a.precision == 3)
        ESTTS_LAUNCH((gemm_kernel<BM, BN, WM, WN, 3>), grid, dim3(256), 0, st, a, splitk);
    else
        ESTTS_LAUNCH((gemm_kernel<BM, BN, WM, WN, 1>), grid, dim3(256), 0, st, a, splitk);
    ESTTS_CHECK_LAUNCH("estts_gemm");
    return 0;
}

template <int TM, int TN, int OCC, int KIND>
int launch_big_d(const estts_gemm_args& a, hipStream_t st) {
    dim3 grid((a.N + 64 * TN - 1) / (64 * TN), (a.M + 64 * TM - 1) / (64 * TM), a.batch > 1 ? a.batch : 1);
    if (a.precision == 3)
        ESTTS_LAUNCH((gemm_big_kernel<3, TM, TN, OCC, KIND>), grid, dim3(256), 0, st, a);
    else
        ESTTS_LAUNCH((gemm_big_kernel<1, TM, TN, OCC, KIND>), grid, dim3(256), 0, st, a);
    ESTTS_CHECK_LAUNCH("estts_gemm");
    return 0;
}

// Tile 13 (four waves, 64 x 128 tiles, four workgroups per CU) and tile 14 (eight waves, 64 x 256, two per CU; LNE: with the
// LayerNorm epilogue, N = 256).  When the launch's last round of tiles (a round = one tile per CU) would be at most half full,
// the row tiles of that round go out as 32-row tiles behind the others (tune bit 26: equal tiles only, for measurements).
template <int OCC, int KIND, int NWV, bool LNE>
int launch_wdirect_d(const estts_gemm_args& a, hipStream_t st) {
    constexpr int BN = 32 * NWV;
    const long cus = 256;   // MI355X in its single-partition mode, the one target of this library (a wrong count only costs balance)
    const long gx = (a.N + BN - 1) / BN, t64 = (a.M + 63) / 64;
    const long total = gx * t64;
    long gy_big = t64, gy_small = 0;
    if (!(a.tune & (1 << 26)) && total > cus) {
        const long rem = total % cus;
        if (rem != 0 && rem <= cus / 2) {
            gy_big = t64 - (rem + gx - 1) / gx;
            gy_small = (a.M - gy_big * 64 + 31) / 32;
        }
    }
    const long n_big = gy_big * gx, nwg = n_big + gy_small * gx;
    ESTTS_CHECK_ARG(nwg < (1L << 31), "estts_gemm: too many workgroups");
    dim3 grid((unsigned)nwg, 1, 1);
    if (a.precision == 3)
        ESTTS_LAUNCH((gemm_wdirect_kernel<3, OCC, KIND, NWV, LNE>), grid, dim3(64 * NWV), 0, st, a, (int)n_big, (int)gy_big, (int)gy_small);
    else
        ESTTS_LAUNCH((gemm_wdirect_kernel<1, OCC, KIND, NWV, LNE>), grid, dim3(64 * NWV), 0, st, a, (int)n_big, (int)gy_big, (int)gy_small);
    ESTTS_CHECK_LAUNCH("estts_gemm");
    return 0;
}

template <int KIND>
int launch_wdirect_k2_d(const estts_gemm_args& a, hipStream_t st) {
    dim3 grid((a.N + 127) / 128, (a.M + 63) / 64, 1);
    if (a.precision == 3)
        ESTTS_LAUNCH((gemm_wdirect_k2_kernel<3, KIND>), grid, dim3(512), 0, st, a);
    else
        ESTTS_LAUNCH((gemm_wdirect_k2_kernel<1, KIND>), grid, dim3(512), 0, st, a);
    ESTTS_CHECK_LAUNCH("estts_gemm");
    return 0;
}

int launch_wdirect_k2(const estts_gemm_args& a, hipStream_t st) {
    const int kind = operand_kind(a);
    return kind == 1 ? launch_wdirect_k2_d<1>(a, st) : kind == 2 ? launch_wdirect_k2_d<2>(a, st) : launch_wdirect_k2_d<0>(a, st);
}

template <int OCC, int NWV, bool LNE>
int launch_wdirect(const estts_gemm_args& a, hipStream_t st) {
    const int kind = operand_kind(a);
    return kind == 1 ? launch_wdirect_d<OCC, 1, NWV, LNE>(a, st) : kind == 2 ? launch_wdirect_d<OCC, 2, NWV, LNE>(a, st)
                                                                              : launch_wdirect_d<OCC, 0, NWV, LNE>(a, st);
}

template <int TM, int TN, int OCC>
int launch_big(const estts_gemm_args& a, hipStream_t st) {
    const int kind = operand_kind(a);
    return kind == 1 ? launch_big_d<TM, TN, OCC, 1>(a, st) : kind == 2 ? launch_big_d<TM, TN, OCC, 2>(a, st) : launch_big_d<TM, TN, OCC, 0>(a, st);
}

}  // namespace

extern "C" int estts_gemm(const estts_gemm_args* args, void* stream) {
    ESTTS_CHECK_ARG(args != nullptr, "estts_gemm: null args");
    estts_gemm_args a = *args;
    if (a.conv_kw <= 0) {  // dense layer or 1-D convolution over rows: one image row of seg_len pixels per sequence
        ESTTS_CHECK_ARG(a.seg_len == 0 || a.M % a.seg_len == 0, "estts_gemm: M not a multiple of seg_len");
        a.conv_h = 1;
        a.conv_w = a.seg_len > 0 ? a.seg_len : a.M;
        a.conv_wo = a.conv_w;
        a.conv_howo = a.conv_w;
        a.conv_kw = a.taps;
        a.conv_stride = 1;
        a.conv_pt = 0;
        a.conv_pl = a.tap_lpad;
    } else {
        ESTTS_CHECK_ARG(a.conv_h > 0 && a.conv_w > 0 && a.conv_wo > 0 && a.conv_howo > 0 && a.conv_stride > 0 &&
                        a.taps % a.conv_kw == 0 && a.M % a.conv_howo == 0 && a.conv_howo % a.conv_wo == 0,
                        "estts_gemm: inconsistent 2-D convolution geometry");
    }
    ESTTS_CHECK_ARG((a.A || a.ln_y) && a.W_hi && a.C, "estts_gemm: null A/W/C");
    ESTTS_CHECK_ARG(a.M > 0 && a.N > 0 && a.K > 0, "estts_gemm: bad shape M=%d N=%d K=%d", a.M, a.N, a.K);
    ESTTS_CHECK_ARG(a.precision == 1 || a.precision == 3, "estts_gemm: precision must be 1 or 3");
    ESTTS_CHECK_ARG(a.precision == 1 || a.W_lo, "estts_gemm: precision 3 needs W_lo");
    ESTTS_CHECK_ARG(a.taps >= 1 && a.c_per_tap > 0 && a.taps * a.c_per_tap == a.K,
                    "estts_gemm: taps*c_per_tap != K (%d*%d vs %d)", a.taps, a.c_per_tap, a.K);
    ESTTS_CHECK_ARG((a.K & 3) == 0 && (a.c_per_tap & 3) == 0 && (a.lda & 3) == 0,
                    "estts_gemm: K, c_per_tap and lda must be multiples of 4");
    ESTTS_CHECK_ARG(((uintptr_t)a.A & 15) == 0 && ((uintptr_t)a.W_hi & 15) == 0 && ((uintptr_t)a.W_lo & 15) == 0,
                    "estts_gemm: A / W must be 16-byte aligned");
    ESTTS_CHECK_ARG(a.ldw >= ((a.K + 63) / 64) * 64 && (a.ldw & 63) == 0, "estts_gemm: ldw must be K rounded up to 64");
    ESTTS_CHECK_ARG(a.taps == 1 || a.lda >= a.c_per_tap, "estts_gemm: lda < c_per_tap");
    ESTTS_CHECK_ARG(!a.R || a.ldr >= a.N, "estts_gemm: ldr < N");
    ESTTS_CHECK_ARG(a.ldc >= a.N, "estts_gemm: ldc < N");
    ESTTS_CHECK_ARG(a.drop_rate >= 0.0f && a.drop_rate < 1.0f, "estts_gemm: drop_rate must be in [0, 1)");
    hipStream_t st = (hipStream_t)stream;

    int tile = a.tune & 0xff;
    int splitk = (a.tune >> 8) & 0xff;
    if (tile == 0) {
        // measured on MI355X (tools/ab_gemm_big.py, tools/tune_gemm_big.py, tools/tune_wavernn_gemm.py): the LDS-staged kernels win
        // once their workgroups fill most of the chip -- as 64 x 128 tiles when there are enough of those, else as 64 x 64 tiles --
        // and below that the barrier-free 32 x 32 kernel does.  The 128 x 128, 128 x 64 and LDS-staged 64 x 128 shapes (ids 5, 7,
        // 8) and the older 64 x 64 kernels (ids 1-3) never win at these sizes and are kept for explicit selection only.
        const long mt = (long)((a.M + 63) / 64) * (a.batch > 1 ? a.batch : 1);
        const long n8 = mt * ((a.N + 127) / 128), n6 = mt * ((a.N + 63) / 64);
        const bool k2_ok = (a.K & 63) == 0 && a.K >= 256 && a.N >= 128;   // what id 15 needs
        if (a.batch > 1) {
            tile = 6;
        } else if (a.N > 64 && n8 >= 144) {   // (also with a partly empty second half of the 128 columns: 19200 x 80 x 1280, five
                                              // taps: 29.6 us against 38.6 as 64 x 64 tiles; 19200 x 112 x 1280: 29.6 / 40.2)
            // 64 x 128 tiles run on the W-from-memory kernel (id 13: 19200 x 256 x 512 21.7 us against 27.3 as 64 x 64 tiles;
            // 19200 x 256 x 1536, three taps: 52 / 75 us; 4800 x 256 x 512: 11.5 / 13.1 us; 3200 x 512 x 1024: 19.2 / 22.9 us)
            tile = 13;
            // at most one tile per CU: two groups of waves on the halves of K (id 15: 3200 x 512 x 1536, three taps: 24.5 against
            // 29.5 us; 2400 x 512 x 512: 10.9 / 12.5; 1600 x 768 x 256: 7.8 / 7.9; with more tiles id 13 wins: 3200 x 768 x 1536,
            // three taps, 300 tiles: 33.5 against 38.4 us)
            if (n8 <= 256 && k2_ok) tile = 15;
            // from 512 columns up, eight waves sharing the staged A rows (64 x 256 tiles, id 14) once there are two rounds of those:
            // 19200 x 512 x 768, three taps: 54.5 against 57.0 us; 19200 x 1024 x 512: 67.4 / 70.3; 19200 x 512 x 1536, three taps:
            // 99 / 104; with fewer tiles id 13 wins (9600 x 512 x 768, three taps: 31.4 against 33.6 us; 3200 x 512 x 2048: 32 / 42)
            else if (a.N >= 512 && mt * ((a.N + 255) / 256) >= 512) tile = 14;
        } else if (n6 >= 144) {
            // fewer 64 x 128 tiles than that: 64 x 64 tiles (id 6: 3200 x 256 x 512 9.8 us against 13.3 on 32 x 32 tiles and 10.6 as
            // id 13; 1600 x 512 x 256: 6.9 / 10.2 / 7.5; 832 x 768 x 256: 6.7 / 9.1 / 7.3), long K on id 15 (1600 x 512 x 1536,
            // three taps: 22.1 against 26.5 (id 6) and 27.6 us; 2400 x 256 x 1536: 21.6 / 25.5 / 22.8); below 144 of them the
            // 32 x 32 kernels win everywhere (1600 x 256 x 512: 8.1 against 9.1 us; 832 x 512 x 1536, three taps: 15.3 / 25.0)
            // (long K without what id 15 needs stays on the 32 x 32 kernels: 4800 x 80 x 1280, five taps: 13.3 against 21.6 us)
            tile = a.K >= 1024 ? (k2_ok ? 15 : 4) : 6;
        } else {
            tile = 4;
        }
        // (measurements: the rule these replaced -- ids 13 / 6 / 4 by tile counts alone)
        static const bool old_rule = [] { const char* e = getenv("ESTTS_GEMM_OLD_RULE"); return e && atoi(e) != 0; }();
        if (old_rule) tile = (a.N >= 128 && n8 >= 144 && a.batch <= 1) ? 13 : ((n6 >= 288 || a.batch > 1) ? 6 : 4);
    }
    static const bool equal_tiles = [] { const char* e = getenv("ESTTS_GEMM_EQUAL_TILES"); return e && atoi(e) != 0; }();   // (measurements)
    if (equal_tiles) a.tune |= 1 << 26;
    if (a.stat_slots) {
        ESTTS_CHECK_ARG(a.batch <= 1 && !a.ln_y, "estts_gemm: stat_slots not with batch > 1 or a LayerNorm prologue");
        if ((tile < 5 || tile > 8) && (tile < 13 || tile > 15)) {   // a kernel without the summing epilogue: the product, then a pass over C
            ESTTS_CHECK_ARG((a.N & 3) == 0 && (a.ldc & 3) == 0, "estts_gemm: stat_slots needs N and ldc multiples of 4 at this size");
            estts_gemm_args b = *args;
            b.stat_slots = nullptr;
            b.stat_shift = nullptr;
            const int r = estts_gemm(&b, stream);
            if (r != 0) return r;
            estts_colreduce_args c = {};
            c.z = a.C; c.ldz = a.ldc; c.M = a.M; c.C = a.N; c.mode = 0; c.defer = 1; c.shift = a.stat_shift;
            c.ws = a.stat_slots; c.ws_bytes = (int64_t)16 * 2 * ((a.N + 255) / 256) * 256 * 4;
            return estts_colreduce(&c, stream);
        }
    }
    if (a.ln_epilogue) {   // LayerNorm epilogue: the kernel whose workgroups own whole rows of 256 outputs
        ESTTS_CHECK_ARG(!a.ln_y && !a.ln_out, "estts_gemm: ln_epilogue not together with a LayerNorm prologue (ln_y / ln_out)");
        ESTTS_CHECK_ARG((!a.ln_ga == !a.ln_ba) && (!a.ln_gb == !a.ln_bb) && (a.ln_ga || a.ln_gb) && (a.ln_ga || a.ln_act == ESTTS_ACT_NONE),
                        "estts_gemm: LayerNorm epilogue needs at least one of (ln_ga, ln_ba) / (ln_gb, ln_bb); ln_act only with the first");
        ESTTS_CHECK_ARG(a.ln_act == ESTTS_ACT_NONE || a.ln_act == ESTTS_ACT_RELU, "estts_gemm: ln_act must be none or ReLU");
        ESTTS_CHECK_ARG(a.N == 256 && a.batch <= 1 && !a.R && !a.s2 && a.act == ESTTS_ACT_NONE && a.alpha == 1.0f && !(a.drop_rate > 0.0f) &&
                        !a.stat_slots && (tile == 13 || tile == 14 || (a.tune & 0xff) == 0),
                        "estts_gemm: this launch cannot take a LayerNorm epilogue (N = 256, one product, no residual / second affine / "
                        "activation / alpha / dropout / column moments of its own, tile 0 or 14): run estts_rownorm after it");
        ESTTS_CHECK_ARG(((uintptr_t)a.ln_x & 15) == 0 && ((uintptr_t)a.ln_ga & 15) == 0 && ((uintptr_t)a.ln_ba & 15) == 0 &&
                        ((uintptr_t)a.ln_gb & 15) == 0 && ((uintptr_t)a.ln_bb & 15) == 0 && ((uintptr_t)a.C & 15) == 0 && (a.ldc & 3) == 0 &&
                        (!a.ln_x || ((a.ln_ldx & 3) == 0 && a.ln_ldx >= a.N)),
                        "estts_gemm: LayerNorm epilogue operands and C must be 16-byte aligned with row strides >= N, multiples of 4");
        // C may be ln_x, and the A of a dense layer, row for row (a workgroup owns whole rows and writes them after its last read);
        // a convolution's neighbours read each other's rows
        ESTTS_CHECK_ARG(a.C != a.A || operand_kind(a) == 1, "estts_gemm: with a LayerNorm epilogue C may alias A only for a dense layer");
        return launch_wdirect<2, 8, true>(a, st);
    }
    if (a.ln_y) {   // LayerNorm prologue: only the kernel that stages a tile's source rows can compute them on the way in
        ESTTS_CHECK_ARG(a.ln_out && (!a.ln_ga == !a.ln_ba) && (!a.ln_gb == !a.ln_bb) && (a.ln_ga || a.ln_gb) &&
                        (a.ln_ga || a.ln_act == ESTTS_ACT_NONE),
                        "estts_gemm: LayerNorm prologue needs ln_out and at least one of (ln_ga, ln_ba) / (ln_gb, ln_bb); ln_act only with the first");
        ESTTS_CHECK_ARG(a.ln_out != a.ln_y && a.ln_out != a.ln_x, "estts_gemm: ln_out must not alias ln_y / ln_x (other workgroups still read those rows)");
        ESTTS_CHECK_ARG(a.C != a.ln_y && a.C != a.ln_x && a.C != a.ln_out && a.R != a.ln_out,
                        "estts_gemm: C must not alias an operand of the LayerNorm prologue, and R must not be ln_out (written by this launch)");
        ESTTS_CHECK_ARG(((uintptr_t)a.ln_y & 15) == 0 && ((uintptr_t)a.ln_x & 15) == 0 && ((uintptr_t)a.ln_out & 15) == 0 &&
                        ((uintptr_t)a.ln_ga & 15) == 0 && ((uintptr_t)a.ln_ba & 15) == 0 && ((uintptr_t)a.ln_gb & 15) == 0 &&
                        ((uintptr_t)a.ln_bb & 15) == 0 && (a.ln_ldy & 3) == 0 && (a.ln_ldx & 3) == 0 && (a.ln_ldo & 3) == 0 &&
                        a.ln_ldy >= a.c_per_tap && a.ln_ldo >= a.c_per_tap && (!a.ln_x || a.ln_ldx >= a.c_per_tap),
                        "estts_gemm: LayerNorm prologue operands must be 16-byte aligned with row strides >= c_per_tap, multiples of 4");
        const long t32 = (long)((a.M + 31) / 32) * ((a.N + 31) / 32);
        const int r = estts_launch_gemm_oneshot_conv(a, operand_kind(a), tile == 11 || (tile != 12 && t32 < 128), st);
        ESTTS_CHECK_ARG(r != 1, "estts_gemm: this launch cannot take a LayerNorm prologue (dense layer or 1-D convolution with up to 8 taps, "
                                "c_per_tap <= 512 and a multiple of 8, K <= 1536, precision 3): run estts_rownorm first");
        return r;
    }
    if (tile == 9 || tile == 10) {   // explicit one-shot kernel: 16 x 16 / 32 x 32 tiles
        const int r = estts_launch_gemm_oneshot(a, operand_kind(a), tile == 9, (a.tune >> 16) & 0xff, st);
        ESTTS_CHECK_ARG(r != 1, "estts_gemm: shape outside the one-shot kernel (dense / 1-D convolution, K and c_per_tap multiples of 8, K <= 192 x waves)");
        return r;
    }
    if (tile == 11 || tile == 12) {  // explicit one-shot convolution kernel (source rows staged in LDS): 16 x 16 / 32 x 32 tiles
        const int r = estts_launch_gemm_oneshot_conv(a, operand_kind(a), tile == 11, st);
        ESTTS_CHECK_ARG(r != 1, "estts_gemm: shape outside the one-shot convolution kernel (1-D convolution, K and c_per_tap multiples of 8, K <= 1536, staged rows within 96 KiB)");
        return r;
    }
    if (tile == 4 && (a.tune & 0xff) == 0 && splitk == 0) {
        // launches of a few hundred small tiles: every load issued up front (gemm_oneshot.hip); 16 x 16 tiles while 32 x 32
        // ones would leave half the chip idle
        // (measured, tools/tune_gemm.py: 16 x 16 tiles win at every K they can hold -- 100 x 512 x 1536: 8.4 -> 6.0 us,
        // 600 x 80 x 1280: 7.7 -> 5.3 us --; 32 x 32 tiles only up to 16 K-steps: beyond that a workgroup's 400 KB of operands
        // are bound by its CU's L1 fill rate and the register ring of gemm_small_kernel, with its coalesced A loads, is faster)
        const long t32 = (long)((a.M + 31) / 32) * ((a.N + 31) / 32);
        if (t32 < 128 || a.K <= 512) {
            const int r = estts_launch_gemm_oneshot(a, operand_kind(a), t32 < 128, 0, st);
            if (r != 1) return r;
        } else if (a.taps > 1) {
            // longer convolutions over 32 x 32 tiles: the tile's source rows once through LDS instead of once per tap
            // (600 x 256 x 1536, three taps: 8.5 -> 7.3 us; 600 x 256 x 1280, five taps: 7.8 -> 6.7 us)
            const int r = estts_launch_gemm_oneshot_conv(a, operand_kind(a), 0, st);
            if (r != 1) return r;
        }
    }
    ESTTS_CHECK_ARG(a.batch <= 1 || (tile >= 5 && tile <= 8), "estts_gemm: batch > 1 needs an LDS-staged tile (tune 0 or 5..8)");
    ESTTS_CHECK_ARG(a.batch <= 65535, "estts_gemm: batch exceeds the grid limit");
    long tiles;
    int nk;  // K units that one split must keep at least two of
    switch (tile) {
        case 1: tiles = (long)((a.M + 63) / 64) * ((a.N + 63) / 64); nk = (a.K + BK - 1) / BK; break;
        case 2: tiles = (long)((a.M + 31) / 32) * ((a.N + 63) / 64); nk = (a.K + BK - 1) / BK; break;
        case 3: tiles = (long)((a.M + 31) / 32) * ((a.N + 31) / 32); nk = (a.K + BK - 1) / BK; break;
        case 4: tiles = (long)((a.M + 31) / 32) * ((a.N + 31) / 32); nk = (a.K + 31) / 32 / 8; break;
        case 5: return launch_big<2, 2, 2>(a, st);
        case 6: return launch_big<1, 1, 4>(a, st);
        case 7: return launch_big<2, 1, 2>(a, st);
        case 8: return launch_big<1, 2, 3>(a, st);
        case 13:
            ESTTS_CHECK_ARG(a.batch <= 1, "estts_gemm: tile 13 takes no batch");
            return launch_wdirect<4, 4, false>(a, st);
        case 14:
            ESTTS_CHECK_ARG(a.batch <= 1, "estts_gemm: tile 14 takes no batch");
            return launch_wdirect<2, 8, false>(a, st);
        case 15:
            ESTTS_CHECK_ARG(a.batch <= 1 && (a.K & 63) == 0, "estts_gemm: tile 15 takes no batch and needs K a multiple of 64");
            return launch_wdirect_k2(a, st);

        default: estts_set_error("estts_gemm: unknown tile id %d", tile); return -1;
    }
    if (nk < 1) nk = 1;
    if (splitk == 0) {
        // the release / acquire fences of the hand-off cost more than the extra parallelism returns until a launch
        // has only a handful of tiles, so split automatically only then
        splitk = 1;
        if (a.ws && a.counters) {
            while (splitk < 8 && tiles * splitk * 2 <= 32 && nk / (splitk * 2) >= 2) splitk *= 2;
        }
    }
    if (splitk > nk) splitk = nk;
    switch (tile) {
        case 1: return launch<64, 64, 2, 2>(a, splitk, st);
        case 2: return launch<32, 64, 1, 4>(a, splitk, st);
        case 3: return launch<32, 32, 2, 2>(a, splitk, st);
        default: return launch_small(a, splitk, (a.tune >> 16) & 0xff, st);
    }
}

// ------------------------------------------------------------------------------------------------------
// estts_gemm without a caller workspace: the split-K slabs and tickets live in a file-scope device array.
// Layout: [NOALLOC_TICKETS int32 tickets | NOALLOC_SLAB_BYTES of float32 slabs].
namespace {
constexpr int NOALLOC_TICKETS = 4096;
constexpr long NOALLOC_SLAB_BYTES = 4l << 20;
constexpr int NOALLOC_MAX_DEVICES = 64;
}  // namespace

static __device__ __align__(256) unsigned char scratch[NOALLOC_TICKETS * 4 + NOALLOC_SLAB_BYTES];

namespace {
std::mutex g_scratch_mutex;
unsigned char* g_scratch_ptr[NOALLOC_MAX_DEVICES] = {};   // device address of `scratch` on each GPU (every GPU holds its own copy)

// Device address of this GPU's copy of `scratch`, looked up once per device with that device current.  The stream must
// belong to the current device: another GPU's stream would be handed this GPU's array.
int scratch_of_current_device(unsigned char** out, hipStream_t stream) {
    int dev = -1;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess || dev < 0 || dev >= NOALLOC_MAX_DEVICES) {
        estts_set_error("estts_gemm_noalloc: hipGetDevice failed or device index %d out of range: %s", dev, hipGetErrorString(e));
        return -2;
    }
    if (stream) {
        hipDevice_t sdev = -1;
        e = hipStreamGetDevice(stream, &sdev);
        if (e == hipSuccess && (int)sdev != dev) {
            estts_set_error("estts_gemm_noalloc: the stream belongs to device %d but device %d is current; make the stream's device "
                            "current before the call", (int)sdev, dev);
            return -1;
        }
    }
    std::lock_guard<std::mutex> lock(g_scratch_mutex);
    if (!g_scratch_ptr[dev]) {
        unsigned char* ptr = nullptr;
        e = hipGetSymbolAddress((void**)&ptr, HIP_SYMBOL(scratch));
        if (e != hipSuccess || !ptr) {
            estts_set_error("estts_gemm_noalloc: hipGetSymbolAddress(scratch) failed on device %d: %s", dev, hipGetErrorString(e));
            return -2;
        }
        g_scratch_ptr[dev] = ptr;
    }
    *out = g_scratch_ptr[dev];
    return 0;
}
}  // namespace

extern "C" int estts_gemm_noalloc(const estts_gemm_args* args, void* stream) {
    ESTTS_CHECK_ARG(args != nullptr, "estts_gemm_noalloc: null args");
    unsigned char* ptr = nullptr;
    const int rc = scratch_of_current_device(&ptr, (hipStream_t)stream);
    if (rc != 0) return rc;
    estts_gemm_args a = *args;
    a.counters = reinterpret_cast<int32_t*>(ptr);
    a.n_counters = NOALLOC_TICKETS;
    a.ws = reinterpret_cast<float*>(ptr + NOALLOC_TICKETS * 4);
    a.ws_bytes = NOALLOC_SLAB_BYTES;
    // (the tickets are cleared on the stream only in front of a launch that splits: see clear_static_tickets)
    g_entry = "estts_gemm_noalloc";
    g_clear_tickets = ptr;
    g_clear_bytes = NOALLOC_TICKETS * 4;
    const int r = estts_gemm(&a, stream);
    g_clear_tickets = nullptr;
    g_entry = "estts_gemm";
    return r;
}
