// Weight gradient of a dense / 1-D convolution layer on the matrix cores (estts_gemm_wgrad in include/estts.h):
//   dW[k, n] += sum_m A(m, k) * dY[m, n]          A(m, k) addressed as in estts_gemm (taps, segment borders)
//
// The reduction runs over the ROW index of both operands, so both MFMA operands are "transposed" views of row-major
// tiles.  A workgroup owns a 64 (k) x 64 (n) tile of dW and walks its row range in steps of 32 rows: the A tile
// [32][64] and the dY tile [32][64] are read with coalesced float4 loads (two steps ahead, two register sets), split into bf16 hi (+ lo)
// and staged in LDS with 144-byte rows; every operand fragment then comes from the transposing ds_read_b64_tr_b16
// (4 rows x 16 columns per 16-lane group, delivered column-per-lane).  Both operands use the same slot <-> row
// permutation, so the product needs no reordering.  Row ranges are spread over the grid and the partial tiles are
// added to dW with hardware float atomics.
#include "common.h"
#include "a_operand.h"

namespace {

typedef __attribute__((ext_vector_type(4))) short s16x4;
typedef __attribute__((ext_vector_type(8))) short s16x8;

constexpr int LD = 72;  // bf16 per staged row (64 + 8 pad = 144 B)

__device__ __forceinline__ bf16x8 tr_read_pair(const __bf16* p0, const __bf16* p1) {
    typedef s16x4 __attribute__((address_space(3))) * lds_ptr;
    const s16x4 a = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_ptr)p0);
    const s16x4 b = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_ptr)p1);
    const s16x8 r = {a[0], a[1], a[2], a[3], b[0], b[1], b[2], b[3]};
    return __builtin_bit_cast(bf16x8, r);
}

#ifdef ESTTS_WGRAD_STAMPS
// Diagnostic build (-DESTTS_WGRAD_STAMPS, tools/wgrad_stamps.py): lane 0 of every wave of ONE workgroup in the middle of the grid
// adds up the shader-clock cycles between the marks of the row loop.  The marks constrain the instruction scheduler, so the
// kernel under test is not quite the shipped one.
__device__ unsigned long long wgrad_stamps[4][8];
#define WS_DECL unsigned long long ws_acc[8] = {0, 0, 0, 0, 0, 0, 0, 0}, ws_prev = clock64(), ws_first = ws_prev; \
    const bool ws_on = (threadIdx.x & 63) == 0 && blockIdx.x == gridDim.x / 2
#define WS(i) do { const unsigned long long t_ = clock64(); ws_acc[i] += t_ - ws_prev; ws_prev = t_; } while (0)
#define WS_WRITE do { if (ws_on) { for (int i_ = 0; i_ < 7; ++i_) wgrad_stamps[threadIdx.x >> 6][i_] = ws_acc[i_]; \
    wgrad_stamps[threadIdx.x >> 6][7] = clock64() - ws_first; } } while (0)
}  // namespace
extern "C" int estts_debug_wgrad_stamps(unsigned long long* out) { return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(wgrad_stamps), sizeof(wgrad_stamps)); }
namespace {
#else
#define WS_DECL
#define WS(i)
#define WS_WRITE
#endif

// NB = 2: eight waves, 64 (k) x 128 (n) per workgroup (a wave still owns 32 x 32): the staged A rows serve twice the columns, a
// thread stages three float4 per step instead of four for the same matrix work per wave.
template <int NSPLIT, int NB = 1>
__global__ __launch_bounds__(256 * NB) void wgrad_kernel(const estts_gemm_args ga, const float* dY, const int lddy, float* dW,
                                                    const int lddw, const int rows_per_split, float* db, const int ktiles,
                                                    const int ntiles, const int nsplits) {
    constexpr int NS = (NSPLIT > 1) ? 2 : 1;
    constexpr int LDY = 64 * NB + 8;   // bf16 per staged dY row
    constexpr int AI = 2 / NB;         // float4 of A per thread and step (rows r0 + 16 NB i)
    __shared__ __attribute__((aligned(16))) __bf16 As[2][NS][32 * LD];
    __shared__ __attribute__((aligned(16))) __bf16 Ys[2][NS][32 * LDY];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int wk = wave / (2 * NB), wn = wave % (2 * NB);
    const int i16 = lane & 15, g = lane >> 4;
    // Workgroups go to the eight XCDs round robin.  Every XCD takes a contiguous eighth of the (row range, n tile, k tile)
    // order, k fastest: the tiles of one row range read the same A and dY rows, which are then fetched into one L2 instead
    // of all eight.
    const int per_xcd = gridDim.x >> 3;
    const int vid = (blockIdx.x & 7) * per_xcd + (blockIdx.x >> 3);
    if (vid >= ktiles * ntiles * nsplits) return;   // padding of the grid to a multiple of 8 (uniform over the workgroup)
    const int kx = vid % ktiles;
    const int k0 = kx * 64, n0 = (vid / ktiles) % ntiles * (64 * NB);
    const int m_begin = vid / (ktiles * ntiles) * rows_per_split;
    const int m_end = min(m_begin + rows_per_split, ga.M);

    const int chunk = tid & 15;                 // A: 16 float4 per row, rows r0 + 16 NB i
    const int r0 = tid >> 4;
    const int ychunk = tid % (16 * NB);         // dY: 16 NB float4 per row, rows yr0 + 16 i
    const int yr0 = tid / (16 * NB);
    const int ncol = n0 + ychunk * 4;
    const bool ycol_ok = ncol < ga.N;

    // A(m, k) of a dense layer or 1-D convolution: k = (tap, channel) is fixed per thread for the whole row loop, and the
    // source row of output row m is m + tap - lpad wherever that stays inside m's sequence, so the address is linear in m
    // and only the validity needs the position of m inside its sequence (carried along, rows only ever advance by 32).
    const int kk = k0 + chunk * 4;
    const bool k_ok = kk < ga.K;
    int tap = 0, ch = kk;
    if (ga.taps > 1) {
        tap = kk / ga.c_per_tap;
        ch = kk - tap * ga.c_per_tap;
    }
    const int seg = ga.conv_w;
    const int shift = tap - ga.conv_pl;
    const int pos_lo = -shift, pos_hi = seg - shift;   // valid positions of m inside its sequence
    const bool need_pos = ga.taps > 1 || ga.conv_pl != 0;   // uniform (a one-tap layer may still be shifted: h_{t-1} of a GRU)
    const bool want_db = db != nullptr && kx == 0;     // uniform

    // loads are unconditional and branch-free (an invalid element reads the nearest existing row) and zeroed where they are
    // used: a load or an address under a condition becomes a branch around it
    const float* a_base = ga.A + (k_ok ? ch : 0);
    const float* y_base = dY + (ycol_ok ? ncol : 0);
    const int last_row = ga.M - 1;
    float4 bsum = make_float4(0.f, 0.f, 0.f, 0.f);   // column sums of this thread's dY values (the bias gradient)
    int pos[AI];
#pragma unroll
    for (int i = 0; i < AI; ++i) pos[i] = need_pos ? (m_begin + r0 + 16 * NB * i) % seg : 0;
    auto load = [&](float4 (&a_reg)[AI], float4 (&y_reg)[2], bool (&a_ok)[AI], bool (&y_ok)[2], int m0) {   // m0 = m_begin, m_begin + 32, ... in order
#pragma unroll
        for (int i = 0; i < AI; ++i) {
            const int row = m0 + r0 + 16 * NB * i;
            bool ok = k_ok && row < ga.M;
            if (need_pos) {
                ok = ok && pos[i] >= pos_lo && pos[i] < pos_hi;
                pos[i] += 32;
                pos[i] -= pos[i] >= seg ? seg : 0;
                if (seg < 32) pos[i] %= seg;
            }
            a_ok[i] = ok;
            a_reg[i] = *reinterpret_cast<const float4*>(a_base + (long)min(max(row + shift, 0), last_row) * ga.lda);
        }
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int row = m0 + yr0 + 16 * i;
            y_ok[i] = row < ga.M && ycol_ok;
            y_reg[i] = *reinterpret_cast<const float4*>(y_base + (long)min(row, last_row) * lddy);
        }
    };
    auto stage = [&](const float4 (&a_reg)[AI], const float4 (&y_reg)[2], const bool (&a_ok)[AI], const bool (&y_ok)[2], int buf) {
#pragma unroll
        for (int i = 0; i < AI; ++i) {
            const int o = (r0 + 16 * NB * i) * LD + chunk * 4;
            bf16x4 ah, al;
            split_bf16x4(zero_unless(a_ok[i], a_reg[i]), ah, al);
            *reinterpret_cast<bf16x4*>(&As[buf][0][o]) = ah;
            if (NS == 2) *reinterpret_cast<bf16x4*>(&As[buf][NS - 1][o]) = al;
        }
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int o = (yr0 + 16 * i) * LDY + ychunk * 4;
            const float4 yz = zero_unless(y_ok[i], y_reg[i]);
            if (want_db) { bsum.x += yz.x; bsum.y += yz.y; bsum.z += yz.z; bsum.w += yz.w; }
            bf16x4 yh, yl;
            split_bf16x4(yz, yh, yl);
            *reinterpret_cast<bf16x4*>(&Ys[buf][0][o]) = yh;
            if (NS == 2) *reinterpret_cast<bf16x4*>(&Ys[buf][NS - 1][o]) = yl;
        }
    };

    f32x4 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

    const int frag = (4 * g + (i16 >> 2)) * LD + 4 * (i16 & 3);
    const int yfrag = (4 * g + (i16 >> 2)) * LDY + 4 * (i16 & 3);
    auto multiply = [&](int buf) {
        bf16x8 af[NS][2], yf[NS][2];
#pragma unroll
        for (int s = 0; s < NS; ++s)
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const __bf16* pa = &As[buf][s][frag + wk * 32 + i * 16];
                const __bf16* py = &Ys[buf][s][yfrag + wn * 32 + i * 16];
                af[s][i] = tr_read_pair(pa, pa + 16 * LD);
                yf[s][i] = tr_read_pair(py, py + 16 * LDY);
            }
        if (NS == 2) {
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[NS - 1][i], yf[0][j], acc[i][j], 0, 0, 0);
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[0][i], yf[NS - 1][j], acc[i][j], 0, 0, 0);
        }
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j)
                acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[0][i], yf[0][j], acc[i][j], 0, 0, 0);
    };

    // Two register sets, two steps in flight; the steady-state loop has no load under a condition (the compiler's wait counts
    // are static: a conditional prefetch makes every iteration wait for everything) and is entered behind loads pinned in the
    // loop's own order.
    float4 a0[AI], y0[2], a1[AI], y1[2];
    bool ao0[AI], yo0[2], ao1[AI], yo1[2];
    const int nk = (m_end - m_begin + 31) / 32;   // 32-row steps of this workgroup (>= 1)
    int kt = 0;
    WS_DECL;
    if (nk >= 4) {
        load(a0, y0, ao0, yo0, m_begin);
        __builtin_amdgcn_sched_barrier(0);
        load(a1, y1, ao1, yo1, m_begin + 32);
        __builtin_amdgcn_sched_barrier(0);
        for (; kt + 3 < nk; kt += 2) {
            stage(a0, y0, ao0, yo0, 0);
            WS(0);
            __syncthreads();
            WS(1);
            load(a0, y0, ao0, yo0, m_begin + (kt + 2) * 32);
            WS(2);
            multiply(0);
            WS(3);
            stage(a1, y1, ao1, yo1, 1);
            WS(0);
            __syncthreads();
            WS(1);
            load(a1, y1, ao1, yo1, m_begin + (kt + 3) * 32);
            WS(2);
            multiply(1);
            WS(3);
        }
    } else {
        load(a0, y0, ao0, yo0, m_begin);
        if (nk > 1) load(a1, y1, ao1, yo1, m_begin + 32);
    }
    // tail: steps kt and kt + 1 are in the registers, kt + 2 (nk - kt = 3) is still to be requested
    stage(a0, y0, ao0, yo0, 0);
    __syncthreads();
    if (kt + 2 < nk) load(a0, y0, ao0, yo0, m_begin + (kt + 2) * 32);
    multiply(0);
    if (kt + 1 < nk) {
        stage(a1, y1, ao1, yo1, 1);
        __syncthreads();
        multiply(1);
    }
    if (kt + 2 < nk) {
        stage(a0, y0, ao0, yo0, 0);   // every wave is past the barrier in front of multiply(1): buffer 0 is free
        __syncthreads();
        multiply(0);
    }

    // bias gradient: the workgroups of the first K tile add up the dY rows they staged anyway
    if (want_db) {   // uniform over the workgroup
        __syncthreads();                      // the staging buffers are dead
        float* red = reinterpret_cast<float*>(&Ys[0][0][0]);   // [16 row lanes][64 NB columns]
        *reinterpret_cast<float4*>(red + yr0 * (64 * NB) + ychunk * 4) = bsum;
        __syncthreads();
        if (tid < 64 * NB && n0 + tid < ga.N) {
            float t = 0.0f;
#pragma unroll
            for (int r = 0; r < 16; ++r) t += red[r * (64 * NB) + tid];
            atomic_add_f32(db + n0 + tid, t);
        }
    }

#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int col = n0 + wn * 32 + j * 16 + i16;
            if (col >= ga.N) continue;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int row = k0 + wk * 32 + i * 16 + 4 * g + r;
                if (row < ga.K) atomic_add_f32(dW + (long)row * lddw + col, acc[i][j][r]);
            }
        }
    WS(4);
    WS_WRITE;
}

// ------------------------------------------------------------------------------------------------------
// 1-D convolutions whose channel count is a multiple of 64: a workgroup owns 64 channels x 64 outputs for ALL taps.  The
// taps of a convolution read the same input rows one row apart, so one staged tile of 32 + TAPS - 1 input rows (and one dY
// tile) feeds TAPS times the matrix work of wgrad_kernel: the fragment of tap t is the same transposing LDS read t rows
// further down.  A pair (output row, tap) whose input row falls outside the row's sequence must not contribute: steps
// that touch a sequence border (uniform test) clear those rows in the A fragment registers; all other steps skip that.
// ------------------------------------------------------------------------------------------------------
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

template <int NSPLIT, int TAPS>
__global__ __launch_bounds__(256, TAPS == 3 ? 3 : 2) void wgrad_conv_kernel(const estts_gemm_args ga, const float* dY, const int lddy, float* dW,
                                                         const int lddw, const int rows_per_split, float* db, const int ctiles,
                                                         const int ntiles, const int nsplits) {
    constexpr int NS = (NSPLIT > 1) ? 2 : 1;
    constexpr int AR = 32 + TAPS - 1;   // staged input rows per step
    __shared__ __attribute__((aligned(16))) __bf16 As[2][NS][AR * LD];
    __shared__ __attribute__((aligned(16))) __bf16 Ys[2][NS][32 * LD];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int wk = wave >> 1, wn = wave & 1;
    const int i16 = lane & 15, g = lane >> 4;
    const int per_xcd = gridDim.x >> 3;   // same workgroup order as wgrad_kernel, channel tile fastest
    const int vid = (blockIdx.x & 7) * per_xcd + (blockIdx.x >> 3);
    if (vid >= ctiles * ntiles * nsplits) return;
    const int cx = vid % ctiles;
    const int c0 = cx * 64, n0 = (vid / ctiles) % ntiles * 64;
    const int m_begin = vid / (ctiles * ntiles) * rows_per_split;
    const int m_end = min(m_begin + rows_per_split, ga.M);

    const int chunk = tid & 15;
    const int r0 = tid >> 4;
    const int ncol = n0 + chunk * 4;
    const bool ycol_ok = ncol < ga.N;
    const int seg = ga.conv_w, pl = ga.conv_pl;
    const float* a_col = ga.A + c0 + chunk * 4;
    const bool want_db = db != nullptr && cx == 0;     // uniform

    float4 a_reg[3], y_reg[2];
    bool a_ok[3], y_ok[2];
    float4 bsum = make_float4(0.f, 0.f, 0.f, 0.f);
    auto load = [&](int m0) {
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            if (i == 2 && r0 >= TAPS - 1) break;          // rows 32 .. AR - 1 of the tile: the first TAPS - 1 row lanes
            const int row = m0 - pl + r0 + 16 * i;         // input row of tile row r0 + 16 i
            a_ok[i] = row >= 0 && row < ga.M;
            a_reg[i] = *reinterpret_cast<const float4*>(a_col + (a_ok[i] ? (long)row * ga.lda : 0L));
        }
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int row = m0 + r0 + 16 * i;
            y_ok[i] = row < ga.M && ycol_ok;
            y_reg[i] = *reinterpret_cast<const float4*>(dY + (y_ok[i] ? (long)row * lddy + ncol : 0L));
        }
    };
    auto stage = [&](int buf) {
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            if (i == 2 && r0 >= TAPS - 1) break;
            const int o = (r0 + 16 * i) * LD + chunk * 4;
            bf16x4 ah, al;
            split_bf16x4(zero_unless(a_ok[i], a_reg[i]), ah, al);
            *reinterpret_cast<bf16x4*>(&As[buf][0][o]) = ah;
            if (NS == 2) *reinterpret_cast<bf16x4*>(&As[buf][NS - 1][o]) = al;
        }
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int o = (r0 + 16 * i) * LD + chunk * 4;
            const float4 yz = zero_unless(y_ok[i], y_reg[i]);
            if (want_db) { bsum.x += yz.x; bsum.y += yz.y; bsum.z += yz.z; bsum.w += yz.w; }
            bf16x4 yh, yl;
            split_bf16x4(yz, yh, yl);
            *reinterpret_cast<bf16x4*>(&Ys[buf][0][o]) = yh;
            if (NS == 2) *reinterpret_cast<bf16x4*>(&Ys[buf][NS - 1][o]) = yl;
        }
    };

    f32x4 acc[TAPS][2][2];
#pragma unroll
    for (int t = 0; t < TAPS; ++t)
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j) acc[t][i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

    const int frag = (4 * g + (i16 >> 2)) * LD + 4 * (i16 & 3);
    if (m_begin < m_end) load(m_begin);
    int buf = 0;
    int pos0 = m_begin % seg;   // position of the step's first output row inside its sequence
    for (int m0 = m_begin; m0 < m_end; m0 += 32) {
        stage(buf);
        __syncthreads();
        if (m0 + 32 < m_end) load(m0 + 32);
        // rows pos .. pos + 31 (mod seg): every tap of every row stays inside the sequence when pos >= pl and the last row is
        // at most seg - (TAPS - pl)
        const bool border = pos0 < pl || pos0 + 31 > seg - (TAPS - pl);
        bf16x8 yf[NS][2];
#pragma unroll
        for (int s = 0; s < NS; ++s)
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const __bf16* py = &Ys[buf][s][frag + wn * 32 + i * 16];
                yf[s][i] = tr_read_pair(py, py + 16 * LD);
            }
#pragma unroll
        for (int t = 0; t < TAPS; ++t) {
            bf16x8 af[NS][2];
#pragma unroll
            for (int s = 0; s < NS; ++s)
#pragma unroll
                for (int i = 0; i < 2; ++i) {
                    const __bf16* pa = &As[buf][s][frag + t * LD + wk * 32 + i * 16];
                    af[s][i] = tr_read_pair(pa, pa + 16 * LD);
                }
            if (border) {   // uniform: this lane's eight rows are 4 g + e and 16 + 4 g + e, e = 0..3
                unsigned keep[4];
#pragma unroll
                for (int d = 0; d < 4; ++d) {
                    unsigned w = 0;
#pragma unroll
                    for (int hlf = 0; hlf < 2; ++hlf) {
                        const int e = 2 * d + hlf;
                        const int slot = (e < 4 ? 4 * g + e : 16 + 4 * g + (e - 4));
                        int q = pos0 + slot;                    // position of the row inside its sequence
                        if (seg >= 32) q -= q >= seg ? seg : 0;
                        else q %= seg;
                        const int p = q + t - pl;
                        if (p >= 0 && p < seg) w |= 0xffffu << (16 * hlf);
                    }
                    keep[d] = w;
                }
                const u32x4 kv = {keep[0], keep[1], keep[2], keep[3]};
#pragma unroll
                for (int s = 0; s < NS; ++s)
#pragma unroll
                    for (int i = 0; i < 2; ++i) af[s][i] = __builtin_bit_cast(bf16x8, __builtin_bit_cast(u32x4, af[s][i]) & kv);
            }
            if (NS == 2) {
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int j = 0; j < 2; ++j)
                        acc[t][i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[NS - 1][i], yf[0][j], acc[t][i][j], 0, 0, 0);
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int j = 0; j < 2; ++j)
                        acc[t][i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[0][i], yf[NS - 1][j], acc[t][i][j], 0, 0, 0);
            }
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j)
                    acc[t][i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[0][i], yf[0][j], acc[t][i][j], 0, 0, 0);
        }
        buf ^= 1;
        pos0 = (pos0 + 32) % seg;
    }

    if (want_db) {   // uniform over the workgroup
        __syncthreads();
        float* red = reinterpret_cast<float*>(&As[0][0][0]);   // [16 row lanes][64 columns]
        *reinterpret_cast<float4*>(red + r0 * 64 + chunk * 4) = bsum;
        __syncthreads();
        if (tid < 64 && n0 + tid < ga.N) {
            float t = 0.0f;
#pragma unroll
            for (int r = 0; r < 16; ++r) t += red[r * 64 + tid];
            atomic_add_f32(db + n0 + tid, t);
        }
    }

#pragma unroll
    for (int t = 0; t < TAPS; ++t)
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const int col = n0 + wn * 32 + j * 16 + i16;
                if (col >= ga.N) continue;
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int row = t * ga.c_per_tap + c0 + wk * 32 + i * 16 + 4 * g + r;
                    atomic_add_f32(dW + (long)row * lddw + col, acc[t][i][j][r]);
                }
            }
}

// Row splits of the all-taps kernel: one round of workgroups (measured on MI355X, tools/tune_wgrad.py: 19200 x 256 x 1536,
// three taps: 93 / 67 / 66 / 77 us with 8 / 16 / 24 / 32 splits of its 32 tiles; 19200 x 256 x 1280, five taps: 65 / 57 / 62 us
// with 16 / 24 / 32 splits of its 16 tiles).
int wgrad_conv_splits(const estts_wgrad_args& a) {
    const int tiles = (a.c_per_tap / 64) * ((a.N + 63) / 64);
    static const int target = [] { const char* e = getenv("ESTTS_WGRAD_CONV_WGS"); return e ? atoi(e) : 0; }();   // for measurements
    int splits = (target > 0 ? target : (a.taps == 3 ? 512 : 384)) / tiles;
    if (splits > 64) splits = 64;
    if (splits < 1) splits = 1;
    return splits;
}

template <int TAPS>
int launch_wgrad_conv(const estts_gemm_args& ga, const estts_wgrad_args& a, hipStream_t st) {
    const int ctiles = a.c_per_tap / 64, ntiles = (a.N + 63) / 64;
    const int tiles = ctiles * ntiles;
    int splits = a.splits;
    if (splits <= 0) splits = wgrad_conv_splits(a);
    const int steps = (a.M + 31) / 32;
    const int rows_per_split = ((steps + splits - 1) / splits) * 32;
    splits = (a.M + rows_per_split - 1) / rows_per_split;
    const long total = (long)tiles * splits;
    ESTTS_CHECK_ARG(total < (1L << 30), "estts_gemm_wgrad: too many workgroups");
    dim3 grid((unsigned)((total + 7) / 8 * 8));
    if (a.precision == 3)
        ESTTS_LAUNCH((wgrad_conv_kernel<3, TAPS>), grid, dim3(256), 0, st, ga, a.dY, a.lddy, a.dW, a.lddw, rows_per_split, a.db,
                           ctiles, ntiles, splits);
    else
        ESTTS_LAUNCH((wgrad_conv_kernel<1, TAPS>), grid, dim3(256), 0, st, ga, a.dY, a.lddy, a.dW, a.lddw, rows_per_split, a.db,
                           ctiles, ntiles, splits);
    ESTTS_CHECK_LAUNCH("estts_gemm_wgrad");
    return 0;
}

}  // namespace

extern "C" int estts_gemm_wgrad(const estts_wgrad_args* args, void* stream) {
    ESTTS_CHECK_ARG(args != nullptr, "estts_gemm_wgrad: null args");
    const estts_wgrad_args& a = *args;
    ESTTS_CHECK_ARG(a.A && a.dY && a.dW, "estts_gemm_wgrad: null A/dY/dW");
    ESTTS_CHECK_ARG(a.M > 0 && a.N > 0 && a.K > 0, "estts_gemm_wgrad: bad shape M=%d N=%d K=%d", a.M, a.N, a.K);
    ESTTS_CHECK_ARG(a.precision == 1 || a.precision == 3, "estts_gemm_wgrad: precision must be 1 or 3");
    ESTTS_CHECK_ARG(a.taps >= 1 && a.c_per_tap > 0 && a.taps * a.c_per_tap == a.K, "estts_gemm_wgrad: taps*c_per_tap != K");
    ESTTS_CHECK_ARG((a.K & 3) == 0 && (a.c_per_tap & 3) == 0 && (a.lda & 3) == 0 && (a.N & 3) == 0 && (a.lddy & 3) == 0,
                    "estts_gemm_wgrad: K, c_per_tap, N, lda and lddy must be multiples of 4");
    ESTTS_CHECK_ARG((((uintptr_t)a.A | (uintptr_t)a.dY) & 15) == 0, "estts_gemm_wgrad: A / dY must be 16-byte aligned");
    ESTTS_CHECK_ARG(a.seg_len == 0 || a.M % a.seg_len == 0, "estts_gemm_wgrad: M not a multiple of seg_len");
    ESTTS_CHECK_ARG(a.lddy >= a.N && a.lddw >= a.N && (a.taps == 1 ? a.lda >= a.K : a.lda >= a.c_per_tap), "estts_gemm_wgrad: ld too small");

    estts_gemm_args ga = {};
    ga.A = a.A;
    ga.lda = a.lda;
    ga.taps = a.taps;
    ga.c_per_tap = a.c_per_tap;
    ga.M = a.M;
    ga.N = a.N;
    ga.K = a.K;
    ga.conv_h = 1;
    ga.conv_w = a.seg_len > 0 ? a.seg_len : a.M;
    ga.conv_wo = ga.conv_w;
    ga.conv_howo = ga.conv_w;
    ga.conv_kw = a.taps;
    ga.conv_stride = 1;
    ga.conv_pt = 0;
    ga.conv_pl = a.tap_lpad;

    // convolutions over whole 64-channel tiles: all taps from one staged tile (ESTTS_WGRAD_PER_TAP=1: the per-tap kernel, for
    // measurements)
    static const bool per_tap = [] { const char* e = getenv("ESTTS_WGRAD_PER_TAP"); return e && atoi(e) != 0; }();
    // (a workgroup of that kernel adds TAPS tiles to dW at its end: it pays from about 16 steps per workgroup -- 3200 x 512 x
    // 1536, three taps: 37 us at best against 35 us per tap)
    if (!per_tap && (a.taps == 3 || a.taps == 5) && a.c_per_tap % 64 == 0 && a.tap_lpad >= 0 && a.tap_lpad < a.taps &&
        (a.splits > 0 || a.M / wgrad_conv_splits(a) >= 512))
        return a.taps == 3 ? launch_wgrad_conv<3>(ga, a, (hipStream_t)stream) : launch_wgrad_conv<5>(ga, a, (hipStream_t)stream);

    // large dW (48 tiles of 64 x 64 and more): the eight-wave 64 x 128 tile, two workgroups per CU -- 19200 x 1024 x 256: 59.8 ->
    // 57.5 us, 19200 x 256 x 1024: 58.3 -> 55.0; smaller outputs lose on it (19200 x 256 x 512: 33.0 -> 35.7 us, 19200 x 256 x 256:
    // 23.3 -> 26.0) and 3200-row launches do not care.  (ESTTS_WGRAD_WIDE=0 / 1: never / whenever N >= 128, for measurements)
    static const int wide_env = [] { const char* e = getenv("ESTTS_WGRAD_WIDE"); return e ? atoi(e) : -1; }();
    const bool wide = a.N >= 128 && (wide_env >= 0 ? wide_env != 0 : ((a.K + 63) / 64) * ((a.N + 63) / 64) >= 48);
    const int nt = wide ? 128 : 64;
    const int tiles = ((a.K + 63) / 64) * ((a.N + nt - 1) / nt);
    int splits = a.splits;
    if (splits <= 0) {
        splits = (wide ? 512 : 1024) / tiles;
        const int by_rows = (a.M + 255) / 256;
        if (splits > by_rows) splits = by_rows;
        if (splits > 64) splits = 64;
        if (splits < 1) splits = 1;
    }
    const int steps = (a.M + 31) / 32;
    const int rows_per_split = ((steps + splits - 1) / splits) * 32;
    splits = (a.M + rows_per_split - 1) / rows_per_split;
    const int ktiles = (a.K + 63) / 64, ntiles = (a.N + nt - 1) / nt;
    const long total = (long)ktiles * ntiles * splits;
    ESTTS_CHECK_ARG(total < (1L << 30), "estts_gemm_wgrad: too many workgroups");
    dim3 grid((unsigned)((total + 7) / 8 * 8));
#define ESTTS_WGRAD_GO(NSPLIT, NB)                                                                                                      \
    ESTTS_LAUNCH((wgrad_kernel<NSPLIT, NB>), grid, dim3(256 * NB), 0, (hipStream_t)stream, ga, a.dY, a.lddy, a.dW, a.lddw, rows_per_split, \
                 a.db, ktiles, ntiles, splits)
    if (a.precision == 3) {
        if (wide) ESTTS_WGRAD_GO(3, 2);
        else ESTTS_WGRAD_GO(3, 1);
    } else {
        if (wide) ESTTS_WGRAD_GO(1, 2);
        else ESTTS_WGRAD_GO(1, 1);
    }
#undef ESTTS_WGRAD_GO
    ESTTS_CHECK_LAUNCH("estts_gemm_wgrad");
    return 0;
}
