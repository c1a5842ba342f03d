// estts_gemm for launches whose whole K extent fits in the registers of one workgroup: batch-1 synthesis, incremental
// decoder steps, every dense layer / 1-D convolution with a few hundred output tiles and K <= 1536.
//
// Such launches are bound by dependent memory round trips, not by bandwidth or matrix work: a kernel that walks K in a
// loop pays one round trip per iteration it cannot cover, then one more for its epilogue operands.  Here every wave issues
// ALL its loads before it waits for any of them: first the epilogue operands of the outputs it will finish (scale, shift,
// residual), then for each of its S <= 6 K-steps the A rows and the W fragments.  A workgroup owns a 16 T x 16 T tile of C
// (T = 1 or 2); its NW waves take interleaved 32-wide K-steps.  Operands go from memory straight into the MFMA layout:
// lane l = (i16, g) loads the 8 consecutive floats A[row i16][32 ks + 8 g .. + 7] (two float4; a wave touches 16 rows x
// 128 B) and splits them into bf16 hi + lo in registers, and W is stored fragment-major (see gemm.hip), so no operand passes
// through LDS.  The waves' partial tiles are summed through LDS once, and every wave finishes a share of the outputs.
//
// K index -> (tap, channel) needs c_per_tap % 8 == 0 (an 8-wide lane group never straddles a tap); K % 8 == 0.
#include <mutex>

#include "common.h"
#include "a_operand.h"
#include "gemm_epilogue.h"

namespace {

__device__ __forceinline__ void split_bf16x8(const float4& x0, const float4& x1, bf16x8& hi, bf16x8& lo) {
    bf16x4 h0, l0, h1, l1;
    split_bf16x4(x0, h0, l0);
    split_bf16x4(x1, h1, l1);
    hi = (bf16x8){h0[0], h0[1], h0[2], h0[3], h1[0], h1[1], h1[2], h1[3]};
    lo = (bf16x8){l0[0], l0[1], l0[2], l0[3], l1[0], l1[1], l1[2], l1[3]};
}

template <int NSPLIT, int KIND, int T, int NW, int S>
__global__ __launch_bounds__(NW * 64) void gemm_oneshot_kernel(const estts_gemm_args p) {
    constexpr int NS = (NSPLIT > 1) ? 2 : 1;
    constexpr int NF = 4 * T * T;                // accumulator registers per lane = 64-output groups of the tile
    constexpr int OUTS = (NF + NW - 1) / NW;     // groups a wave finishes
    __shared__ float red[NW * NF * 64];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int i16 = lane & 15;
    const int g = lane >> 4;
    const TileCoord tc = xcd_tile_coord((p.tune & (1 << 25)) ? false : gemm_y_first(p), !(p.tune & (1 << 24)));
    const int m0 = tc.ty * 16 * T;
    const int n0 = tc.tx * 16 * T;
    const int KS = p.ldw >> 5;
    const int nks = (p.K + 31) >> 5;

    // ---- epilogue operands of the outputs this thread finishes: group f = (i T + j) 4 + r is element r of sub-tile (i, j)
    float e_s1[OUTS], e_b1[OUTS], e_s2[OUTS], e_b2[OUTS], e_r[OUTS];
    int e_row[OUTS], e_col[OUTS];
    bool e_ok[OUTS];
#pragma unroll
    for (int o = 0; o < OUTS; ++o) {
        const int f = wave + o * NW;
        const int i = f / (4 * T), j = (f >> 2) % T, r = f & 3;
        e_row[o] = m0 + i * 16 + g * 4 + r;
        e_col[o] = n0 + j * 16 + i16;
        e_ok[o] = f < NF && e_row[o] < p.M && e_col[o] < p.N;
        const int col = e_ok[o] ? e_col[o] : 0;
        const int row = e_ok[o] ? e_row[o] : 0;
        e_s1[o] = p.s1 ? p.s1[col] : 1.0f;
        e_b1[o] = p.b1 ? p.b1[col] : 0.0f;
        e_s2[o] = p.s2 ? p.s2[col] : 1.0f;
        e_b2[o] = p.s2 ? p.b2[col] : 0.0f;
        e_r[o] = p.R ? p.R[(long)(p.r_mod > 0 ? row % p.r_mod : row) * p.ldr + col] : 0.0f;
    }

    // ---- all operand loads of this wave's K-steps
    const float* a_ptr[T];
    int a_pos[T];
#pragma unroll
    for (int i = 0; i < T; ++i) {
        const int row = min(m0 + i * 16 + i16, p.M - 1);   // rows past M re-read the last one; their results are never stored
        a_ptr[i] = p.A + (long)row * p.lda;
        a_pos[i] = KIND == 2 ? row % p.conv_w : 0;
    }
    float4 av[S][T][2];
    bool aok[S][T];
    uint4 wh[S][T], wl[S][T];
#pragma unroll
    for (int s = 0; s < S; ++s) {
        const int ks = wave + s * NW;
        if (s < S - 1 || ks < nks) {                        // only a wave's last step can lie past K (uniform)
            const int kk = ks * 32 + g * 8;
            const bool kok = kk < p.K;
            long off = kok ? kk : 0;                        // past K: re-read valid data of the row (W is zero there)
            int dw = 0;
            if (KIND == 2) {
                const int tap = kk / p.c_per_tap;
                dw = tap - p.conv_pl;
                off = (long)dw * p.lda + (kk - tap * p.c_per_tap);
            }
#pragma unroll
            for (int i = 0; i < T; ++i) {
                bool ok = true;
                if (KIND == 2) ok = kok && (unsigned)(a_pos[i] + dw) < (unsigned)p.conv_w;
                const float* src = a_ptr[i] + ((KIND == 2 && !ok) ? 0L : off);
                av[s][i][0] = *reinterpret_cast<const float4*>(src);
                av[s][i][1] = *reinterpret_cast<const float4*>(src + 4);
                aok[s][i] = ok;
            }
#pragma unroll
            for (int j = 0; j < T; ++j) {
                const long woff = ((long)((n0 >> 4) + j) * KS + ks) * 512 + lane * 8;
                wh[s][j] = *reinterpret_cast<const uint4*>(p.W_hi + woff);
                if (NS == 2) wl[s][j] = *reinterpret_cast<const uint4*>(p.W_lo + woff);
            }
        }
    }

    // ---- matrix work, in the order the loads were issued
    f32x4 acc[T][T];
#pragma unroll
    for (int i = 0; i < T; ++i)
#pragma unroll
        for (int j = 0; j < T; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int s = 0; s < S; ++s) {
        const int ks = wave + s * NW;
        if (s < S - 1 || ks < nks) {
            bf16x8 ah[T], al[T];
#pragma unroll
            for (int i = 0; i < T; ++i) {
                float4 x0 = av[s][i][0], x1 = av[s][i][1];
                if (KIND == 2) { x0 = zero_unless(aok[s][i], x0); x1 = zero_unless(aok[s][i], x1); }
                split_bf16x8(x0, x1, ah[i], al[i]);
            }
            // small terms first; consecutive MFMAs go to different accumulators
            if (NS == 2) {
#pragma unroll
                for (int i = 0; i < T; ++i)
#pragma unroll
                    for (int j = 0; j < T; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(al[i], __builtin_bit_cast(bf16x8, wh[s][j]), acc[i][j], 0, 0, 0);
#pragma unroll
                for (int i = 0; i < T; ++i)
#pragma unroll
                    for (int j = 0; j < T; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah[i], __builtin_bit_cast(bf16x8, wl[s][j]), acc[i][j], 0, 0, 0);
            }
#pragma unroll
            for (int i = 0; i < T; ++i)
#pragma unroll
                for (int j = 0; j < T; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah[i], __builtin_bit_cast(bf16x8, wh[s][j]), acc[i][j], 0, 0, 0);
        }
    }

    // ---- sum the waves' partial tiles; wave w finishes groups w, w + NW, ...
#pragma unroll
    for (int i = 0; i < T; ++i)
#pragma unroll
        for (int j = 0; j < T; ++j)
#pragma unroll
            for (int r = 0; r < 4; ++r) red[(wave * NF + (i * T + j) * 4 + r) * 64 + lane] = acc[i][j][r];
    __syncthreads();
#pragma unroll
    for (int o = 0; o < OUTS; ++o) {
        const int f = wave + o * NW;
        if (f < NF) {
            float v = 0.0f;
#pragma unroll
            for (int w = 0; w < NW; ++w) v += red[(w * NF + f) * 64 + lane];
            if (e_ok[o]) {
                v = v * e_s1[o] + e_b1[o];
                v = apply_act(v, p.act);
                if (p.drop_rate > 0.0f)
                    v = dropout_keep(p.drop_seed, p.drop_site, (uint32_t)(p.drop_row0 + e_row[o]), (uint32_t)e_col[o], p.drop_rate)
                            ? v / (1.0f - p.drop_rate) : 0.0f;
                v = v * p.alpha + e_r[o];
                if (p.s2) v = v * e_s2[o] + e_b2[o];
                p.C[(long)e_row[o] * p.ldc + e_col[o]] = v;
            }
        }
    }
}

// 1-D convolutions: the taps of a convolution re-read the same rows, shifted (three taps: every row three times).  This
// variant fetches the 16 T + taps - 1 source rows of the tile ONCE (all waves, coalesced float4), splits them into bf16
// hi + lo planes in LDS, and a wave reads the A fragment of (row block, tap, channel group) from there with one
// ds_read_b128 per plane.  The W fragments are still requested first and land during the staging.  A staged row is raw
// data; whether output row m may use source row m + tap - lpad (same sequence) is decided where the fragment is used.
// LNCH > 0: the staged rows are not read from A but computed as LNb(act(LNa(ln_y)) + ln_x) (see estts_gemm_args.ln_y): one
// wave per source row, LNCH float4 per lane (c_per_tap <= 256 LNCH), every operand of the wave's rows requested before the
// first reduction; the workgroups of column tile 0 also write their own rows (not the halo) to ln_out.
#ifdef ESTTS_ONESHOT_STAMPS
// Diagnostic build (-DESTTS_ONESHOT_STAMPS, tools/oneshot_stamps.py): lane 0 of wave 0 of every workgroup of the convolution
// kernel leaves the 100 MHz clock at its phase boundaries.  The marks cost registers (122 -> 139 at 600 x 512 x 768 with the
// LayerNorm prologue, i.e. one workgroup per CU instead of two), so the kernel under test is not quite the shipped one.
__device__ unsigned long long oneshot_stamps[1024][8];
#define OS_STAMP(i) do { if (threadIdx.x == 0) oneshot_stamps[(blockIdx.x + gridDim.x * blockIdx.y) & 1023][i] = wall_clock64(); } while (0)
}  // namespace
extern "C" int estts_debug_oneshot_stamps(unsigned long long* out) { return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(oneshot_stamps), sizeof(oneshot_stamps)); }
namespace {
#else
#define OS_STAMP(i) do { } while (0)
#endif

template <int NSPLIT, int T, int NW, int S, int LNCH, bool LNA>
__global__ __launch_bounds__(NW * 64) void gemm_oneshot_conv_kernel(const estts_gemm_args p, const int rows_staged, const int lda_s) {
    constexpr int NS = (NSPLIT > 1) ? 2 : 1;
    constexpr int NF = 4 * T * T;
    constexpr int OUTS = (NF + NW - 1) / NW;
    extern __shared__ __attribute__((aligned(16))) unsigned char dyn_smem[];   // A planes, later the partial tiles
    __bf16* a_hi = reinterpret_cast<__bf16*>(dyn_smem);
    __bf16* a_lo = a_hi + rows_staged * lda_s;
    float* red = reinterpret_cast<float*>(dyn_smem);

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    OS_STAMP(0);
    const int i16 = lane & 15;
    const int g = lane >> 4;
    // (LayerNorm prologue: every column tile of a row tile re-reads that tile's y and x rows, so an L2 should hold a block of
    // rows x columns, not a strip: panels of four column tiles)
    const TileCoord tc = xcd_tile_coord((p.tune & (1 << 25)) ? false : gemm_y_first(p), !(p.tune & (1 << 24)),
                                        (LNCH > 0 && !(p.tune & (1 << 26))) ? 4 : 0);
    const int m0 = tc.ty * 16 * T;
    const int n0 = tc.tx * 16 * T;
    const int KS = p.ldw >> 5;
    const int nks = (p.K + 31) >> 5;
    const int C = p.c_per_tap;

    // ---- W fragments of this wave's K-steps
    uint4 wh[S][T], wl[S][T];
#pragma unroll
    for (int s = 0; s < S; ++s) {
        const int ks = wave + s * NW;
        if (s < S - 1 || ks < nks) {
#pragma unroll
            for (int j = 0; j < T; ++j) {
                const long woff = ((long)((n0 >> 4) + j) * KS + ks) * 512 + lane * 8;
                wh[s][j] = *reinterpret_cast<const uint4*>(p.W_hi + woff);
                if (NS == 2) wl[s][j] = *reinterpret_cast<const uint4*>(p.W_lo + woff);
            }
        }
    }

    // ---- epilogue operands
    float e_s1[OUTS], e_b1[OUTS], e_s2[OUTS], e_b2[OUTS], e_r[OUTS];
    int e_row[OUTS], e_col[OUTS];
    bool e_ok[OUTS];
#pragma unroll
    for (int o = 0; o < OUTS; ++o) {
        const int f = wave + o * NW;
        const int i = f / (4 * T), j = (f >> 2) % T, r = f & 3;
        e_row[o] = m0 + i * 16 + g * 4 + r;
        e_col[o] = n0 + j * 16 + i16;
        e_ok[o] = f < NF && e_row[o] < p.M && e_col[o] < p.N;
        const int col = e_ok[o] ? e_col[o] : 0;
        const int row = e_ok[o] ? e_row[o] : 0;
        e_s1[o] = p.s1 ? p.s1[col] : 1.0f;
        e_b1[o] = p.b1 ? p.b1[col] : 0.0f;
        e_s2[o] = p.s2 ? p.s2[col] : 1.0f;
        e_b2[o] = p.s2 ? p.b2[col] : 0.0f;
        e_r[o] = p.R ? p.R[(long)(p.r_mod > 0 ? row % p.r_mod : row) * p.ldr + col] : 0.0f;
    }

    // ---- stage source rows m0 - lpad .. + rows_staged - 1 (clamped into the buffer; validity is decided at use)
    if constexpr (LNCH == 0) {
        const int c4n = C >> 2;
        const int total = rows_staged * c4n;
        constexpr int MAXI = 10;      // float4 per thread the launcher guarantees to be enough
        float4 x[MAXI];
        int dst[MAXI];
        // no load under a condition (it would become a branch that waits for its data): surplus slots repeat the last item
#pragma unroll
        for (int it = 0; it < MAXI; ++it) {
            const int q = min(tid + it * (NW * 64), total - 1);
            const int rr = q / c4n, c4 = q - rr * c4n;
            const int src = min(max(m0 - p.conv_pl + rr, 0), p.M - 1);
            x[it] = *reinterpret_cast<const float4*>(p.A + (long)src * p.lda + c4 * 4);
            dst[it] = rr * lda_s + c4 * 4;
        }
#pragma unroll
        for (int it = 0; it < MAXI; ++it) {
            bf16x4 hi, lo;
            split_bf16x4(x[it], hi, lo);
            *reinterpret_cast<bf16x4*>(a_hi + dst[it]) = hi;
            if (NS == 2) *reinterpret_cast<bf16x4*>(a_lo + dst[it]) = lo;
        }
    }
    if constexpr (LNCH > 0) {
        constexpr int RPW = T == 1 ? 3 : 5;              // rows per wave: 16 T + taps - 1 <= 8 RPW (taps <= 8)
        float4 ga[LNCH], ba[LNCH], gb[LNCH], bb[LNCH], y[RPW][LNCH], x[RPW][LNCH];
        int cc[LNCH];
        float live[LNCH];                                 // 0 for the lanes past the row's end
#pragma unroll
        for (int i = 0; i < LNCH; ++i) {
            const int c = (lane + 64 * i) * 4;
            live[i] = c < C ? 1.0f : 0.0f;
            cc[i] = min(c, C - 4);
            // (no load under a condition -- it would become a branch with a wait of its own, one round trip each: an absent
            // operand is read from a present one and ignored)
            if constexpr (LNA) {
                ga[i] = *reinterpret_cast<const float4*>(p.ln_ga + cc[i]);
                ba[i] = *reinterpret_cast<const float4*>(p.ln_ba + cc[i]);
            } else {
                ga[i] = ba[i] = make_float4(0.f, 0.f, 0.f, 0.f);       // (never used)
            }
            gb[i] = *reinterpret_cast<const float4*>((p.ln_gb ? p.ln_gb : p.ln_ga) + cc[i]);
            bb[i] = *reinterpret_cast<const float4*>((p.ln_bb ? p.ln_bb : p.ln_ba) + cc[i]);
        }
        const float* xsrc = p.ln_x ? p.ln_x : p.ln_y;
        const int xld = p.ln_x ? p.ln_ldx : p.ln_ldy;
        const float xk = p.ln_x ? 1.0f : 0.0f;
        int src[RPW];
#pragma unroll
        for (int r = 0; r < RPW; ++r) {
            const int rr = min(wave + NW * r, rows_staged - 1);   // surplus slots repeat the last row (same values, same place)
            src[r] = min(max(m0 - p.conv_pl + rr, 0), p.M - 1);
#pragma unroll
            for (int i = 0; i < LNCH; ++i) {
                y[r][i] = *reinterpret_cast<const float4*>(p.ln_y + (long)src[r] * p.ln_ldy + cc[i]);
                x[r][i] = *reinterpret_cast<const float4*>(xsrc + (long)src[r] * xld + cc[i]);
            }
        }
        // every request above is issued before anything below is scheduled: left alone, the compiler sinks the rows' loads to
        // their uses to save registers, and the wave then pays one memory round trip per row instead of one in all
        __builtin_amdgcn_sched_barrier(0);
        OS_STAMP(1);
        const float inv_c = 1.0f / (float)C;
        const float floor_v = p.ln_act == ESTTS_ACT_RELU ? 0.0f : -INFINITY;
        // the rows of a wave are independent: each phase runs over all of them so that their reductions interleave
        auto layer_norm = [&](const float4 (&g)[LNCH], const float4 (&b)[LNCH]) {
            float mean[RPW], inv[RPW];
#pragma unroll
            for (int r = 0; r < RPW; ++r) {
                float t = 0.f;
#pragma unroll
                for (int i = 0; i < LNCH; ++i) t += (y[r][i].x + y[r][i].y + y[r][i].z + y[r][i].w) * live[i];
                mean[r] = t;
            }
#pragma unroll
            for (int r = 0; r < RPW; ++r) mean[r] = wave_sum(mean[r]) * inv_c;
#pragma unroll
            for (int r = 0; r < RPW; ++r) {
                float q = 0.f;
#pragma unroll
                for (int i = 0; i < LNCH; ++i) {
                    const float a = y[r][i].x - mean[r], b2 = y[r][i].y - mean[r], c = y[r][i].z - mean[r], d = y[r][i].w - mean[r];
                    q += (a * a + b2 * b2 + c * c + d * d) * live[i];
                }
                inv[r] = q;
            }
#pragma unroll
            for (int r = 0; r < RPW; ++r) inv[r] = rsqrtf(wave_sum(inv[r]) * inv_c + p.ln_eps);
#pragma unroll
            for (int r = 0; r < RPW; ++r)
#pragma unroll
                for (int i = 0; i < LNCH; ++i) {
                    y[r][i].x = (y[r][i].x - mean[r]) * inv[r] * g[i].x + b[i].x;
                    y[r][i].y = (y[r][i].y - mean[r]) * inv[r] * g[i].y + b[i].y;
                    y[r][i].z = (y[r][i].z - mean[r]) * inv[r] * g[i].z + b[i].z;
                    y[r][i].w = (y[r][i].w - mean[r]) * inv[r] * g[i].w + b[i].w;
                }
        };
        // (LNA false: the rows are LNb(ln_y + ln_x); a template parameter, not a test of the pointer: a branch around the first
        // norm cost the launches that have one 25 % -- 11 -> 14 us at 600 x 512 x 768 -- through the schedule it left)
        if constexpr (LNA) layer_norm(ga, ba);
#pragma unroll
        for (int r = 0; r < RPW; ++r)
#pragma unroll
            for (int i = 0; i < LNCH; ++i) {
                // (none or ReLU only, as a select: the library tanh is a branchy routine, and its blocks between the rows' loads and
                // their use kept the compiler from issuing the loads together)
                y[r][i].x = fmaxf(y[r][i].x, floor_v);
                y[r][i].y = fmaxf(y[r][i].y, floor_v);
                y[r][i].z = fmaxf(y[r][i].z, floor_v);
                y[r][i].w = fmaxf(y[r][i].w, floor_v);
                y[r][i].x += xk * x[r][i].x;
                y[r][i].y += xk * x[r][i].y;
                y[r][i].z += xk * x[r][i].z;
                y[r][i].w += xk * x[r][i].w;
            }
        if (p.ln_gb) layer_norm(gb, bb);
#pragma unroll
        for (int r = 0; r < RPW; ++r) {
            const int rr = min(wave + NW * r, rows_staged - 1);
            const int own = m0 - p.conv_pl + rr;          // unclamped source row
            const bool write_out = tc.tx == 0 && wave + NW * r < rows_staged && rr >= p.conv_pl && rr < p.conv_pl + 16 * T && own < p.M;
#pragma unroll
            for (int i = 0; i < LNCH; ++i) {
                if ((lane + 64 * i) * 4 < C) {
                    bf16x4 hi, lo;
                    split_bf16x4(y[r][i], hi, lo);
                    *reinterpret_cast<bf16x4*>(a_hi + rr * lda_s + cc[i]) = hi;
                    if (NS == 2) *reinterpret_cast<bf16x4*>(a_lo + rr * lda_s + cc[i]) = lo;
                    if (write_out) *reinterpret_cast<float4*>(p.ln_out + (long)own * p.ln_ldo + cc[i]) = y[r][i];
                }
            }
        }
    }
    OS_STAMP(2);
    __syncthreads();
    OS_STAMP(3);

    int a_pos[T];
#pragma unroll
    for (int i = 0; i < T; ++i) a_pos[i] = min(m0 + i * 16 + i16, p.M - 1) % p.conv_w;

    f32x4 acc[T][T];
#pragma unroll
    for (int i = 0; i < T; ++i)
#pragma unroll
        for (int j = 0; j < T; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int s = 0; s < S; ++s) {
        const int ks = wave + s * NW;
        if (s < S - 1 || ks < nks) {
            const int kk = ks * 32 + g * 8;
            const int tap = min(kk / C, p.taps - 1);
            const int c0 = kk - tap * C;          // (past K: c0 >= C is never read because kok is false)
            const bool kok = kk < p.K;
            bf16x8 ah[T], al[T];
#pragma unroll
            for (int i = 0; i < T; ++i) {
                const bool ok = kok && (unsigned)(a_pos[i] + tap - p.conv_pl) < (unsigned)p.conv_w;
                const int o = (i * 16 + i16 + tap) * lda_s + (kok ? c0 : 0);
                // (masked word by word: a select between two 16-byte values is compiled into a table in scratch memory)
                const unsigned keep = ok ? 0xffffffffu : 0u;
                uint4 h = *reinterpret_cast<const uint4*>(a_hi + o);
                h.x &= keep; h.y &= keep; h.z &= keep; h.w &= keep;
                ah[i] = __builtin_bit_cast(bf16x8, h);
                if (NS == 2) {
                    uint4 l = *reinterpret_cast<const uint4*>(a_lo + o);
                    l.x &= keep; l.y &= keep; l.z &= keep; l.w &= keep;
                    al[i] = __builtin_bit_cast(bf16x8, l);
                }
            }
            if (NS == 2) {
#pragma unroll
                for (int i = 0; i < T; ++i)
#pragma unroll
                    for (int j = 0; j < T; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(al[i], __builtin_bit_cast(bf16x8, wh[s][j]), acc[i][j], 0, 0, 0);
#pragma unroll
                for (int i = 0; i < T; ++i)
#pragma unroll
                    for (int j = 0; j < T; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah[i], __builtin_bit_cast(bf16x8, wl[s][j]), acc[i][j], 0, 0, 0);
            }
#pragma unroll
            for (int i = 0; i < T; ++i)
#pragma unroll
                for (int j = 0; j < T; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah[i], __builtin_bit_cast(bf16x8, wh[s][j]), acc[i][j], 0, 0, 0);
        }
    }

    OS_STAMP(4);
    __syncthreads();   // every wave is done with the A planes: the partial tiles take their place
#pragma unroll
    for (int i = 0; i < T; ++i)
#pragma unroll
        for (int j = 0; j < T; ++j)
#pragma unroll
            for (int r = 0; r < 4; ++r) red[(wave * NF + (i * T + j) * 4 + r) * 64 + lane] = acc[i][j][r];
    __syncthreads();
    OS_STAMP(5);
#pragma unroll
    for (int o = 0; o < OUTS; ++o) {
        const int f = wave + o * NW;
        if (f < NF) {
            float v = 0.0f;
#pragma unroll
            for (int w = 0; w < NW; ++w) v += red[(w * NF + f) * 64 + lane];
            if (e_ok[o]) {
                v = v * e_s1[o] + e_b1[o];
                v = apply_act(v, p.act);
                if (p.drop_rate > 0.0f)
                    v = dropout_keep(p.drop_seed, p.drop_site, (uint32_t)(p.drop_row0 + e_row[o]), (uint32_t)e_col[o], p.drop_rate)
                            ? v / (1.0f - p.drop_rate) : 0.0f;
                v = v * p.alpha + e_r[o];
                if (p.s2) v = v * e_s2[o] + e_b2[o];
                p.C[(long)e_row[o] * p.ldc + e_col[o]] = v;
            }
        }
    }
    OS_STAMP(6);
}

// Dynamic LDS beyond 64 KiB has to be allowed per kernel and device before the first launch.
std::mutex g_attr_mutex;
bool allow_large_lds(const void* kern, bool (&raised)[64]) {
    int dev = -1;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) {
        estts_set_error("estts_gemm: hipGetDevice failed or device index %d out of range", dev);
        return false;
    }
    std::lock_guard<std::mutex> lock(g_attr_mutex);
    if (!raised[dev]) {
        const hipError_t e = hipFuncSetAttribute(kern, hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024);
        if (e != hipSuccess) {
            estts_set_error("estts_gemm: hipFuncSetAttribute(MaxDynamicSharedMemorySize) failed: %s", hipGetErrorString(e));
            return false;
        }
        raised[dev] = true;
    }
    return true;
}

template <int NSPLIT, int T, int LNCH, bool LNA>
int launch_conv_s(const estts_gemm_args& a, dim3 grid, int S, int rows_staged, int lda_s, size_t smem, hipStream_t st) {
#define ESTTS_ONESHOT_CASE(s_)                                                                                              \
    case s_: {                                                                                                              \
        auto kern = gemm_oneshot_conv_kernel<NSPLIT, T, 8, s_, LNCH, LNA>;                                                    \
        static bool raised[64] = {};       /* per GPU: the attribute belongs to the device's copy of the kernel */         \
        if (!allow_large_lds(reinterpret_cast<const void*>(kern), raised)) return -2;                                      \
        ESTTS_LAUNCH(kern, grid, dim3(512), smem, st, a, rows_staged, lda_s);                                         \
    } break;
    switch (S) {
        ESTTS_ONESHOT_CASE(1) ESTTS_ONESHOT_CASE(2) ESTTS_ONESHOT_CASE(3)
        ESTTS_ONESHOT_CASE(4) ESTTS_ONESHOT_CASE(5) ESTTS_ONESHOT_CASE(6)
        default: return 1;
    }
#undef ESTTS_ONESHOT_CASE
    ESTTS_CHECK_LAUNCH("estts_gemm");
    return 0;
}

template <int NSPLIT, int KIND, int T, int NW>
int launch_s(const estts_gemm_args& a, dim3 grid, int S, hipStream_t st) {
#define ESTTS_ONESHOT_CASE(s_) \
    case s_: ESTTS_LAUNCH((gemm_oneshot_kernel<NSPLIT, KIND, T, NW, s_>), grid, dim3(NW * 64), 0, st, a); break;
    switch (S) {
        ESTTS_ONESHOT_CASE(1) ESTTS_ONESHOT_CASE(2) ESTTS_ONESHOT_CASE(3)
        ESTTS_ONESHOT_CASE(4) ESTTS_ONESHOT_CASE(5) ESTTS_ONESHOT_CASE(6)
        default: return 1;
    }
#undef ESTTS_ONESHOT_CASE
    ESTTS_CHECK_LAUNCH("estts_gemm");
    return 0;
}

template <int KIND, int T, int NW>
int launch_p(const estts_gemm_args& a, dim3 grid, int S, hipStream_t st) {
    return a.precision == 3 ? launch_s<3, KIND, T, NW>(a, grid, S, st) : launch_s<1, KIND, T, NW>(a, grid, S, st);
}

template <int T, int NW>
int launch_k(const estts_gemm_args& a, int kind, dim3 grid, int S, hipStream_t st) {
    return kind == 1 ? launch_p<1, T, NW>(a, grid, S, st) : launch_p<2, T, NW>(a, grid, S, st);
}

}  // namespace

int estts_launch_gemm_oneshot_conv(const estts_gemm_args& a, int kind, int tile16, hipStream_t st) {
    // (a dense layer is the one-tap case of the same kernel; it comes here only for the LayerNorm prologue, which needs the
    // staged rows -- without one, gemm_oneshot_kernel's direct fragment loads are the faster way to read A)
    if (!(kind == 2 || (kind == 1 && a.ln_y)) || (a.K & 7) || (a.c_per_tap & 7) || a.batch > 1) return 1;
    const int nks = (a.K + 31) / 32;
    const int S = (nks + 7) / 8;
    if (S > 6) return 1;
    const int tm = tile16 ? 16 : 32;
    const int rows_staged = tm + a.taps - 1;
    const int lda_s = a.c_per_tap + 8;                                   // bf16 per staged row (+16 B: conflict-free b128 reads)
    const size_t planes = (size_t)rows_staged * lda_s * 2 * (a.precision == 3 ? 2 : 1);
    const size_t red = (size_t)8 * (tile16 ? 4 : 16) * 64 * 4;
    const size_t smem = planes > red ? planes : red;
    if (smem > 96 * 1024 || (long)rows_staged * (a.c_per_tap >> 2) > 10 * 512) return 1;
    dim3 grid((a.N + tm - 1) / tm, (a.M + tm - 1) / tm, 1);
    if (grid.y > 65535) return 1;
    const int lnch = a.ln_y ? (a.c_per_tap <= 256 ? 1 : 2) : 0;
    if (lnch && (a.c_per_tap > 512 || a.taps > 8 || a.precision != 3 || (!tile16 && lnch == 2 && S > 3) ||
                 (a.ln_act != ESTTS_ACT_NONE && a.ln_act != ESTTS_ACT_RELU) ||
                 (!a.ln_ga && a.ln_act != ESTTS_ACT_NONE))) return 1;   // (S > 3 there: 256 registers and spills)
#define ESTTS_CONV_GO(t_)                                                                                               \
    (lnch == 0 ? (a.precision == 3 ? launch_conv_s<3, t_, 0, true>(a, grid, S, rows_staged, lda_s, smem, st)            \
                                   : launch_conv_s<1, t_, 0, true>(a, grid, S, rows_staged, lda_s, smem, st))           \
     : lnch == 1 ? launch_conv_s<3, t_, 1, true>(a, grid, S, rows_staged, lda_s, smem, st)                             \
                 : launch_conv_s<3, t_, 2, true>(a, grid, S, rows_staged, lda_s, smem, st))
    if (lnch && !a.ln_ga) {      // LNb(ln_y + ln_x): built for the 16-row tile only (the few-row products that use it)
        if (!tile16) return 1;
        return lnch == 1 ? launch_conv_s<3, 1, 1, false>(a, grid, S, rows_staged, lda_s, smem, st)
                         : launch_conv_s<3, 1, 2, false>(a, grid, S, rows_staged, lda_s, smem, st);
    }
    return tile16 ? ESTTS_CONV_GO(1) : ESTTS_CONV_GO(2);
#undef ESTTS_CONV_GO
}

int estts_launch_gemm_oneshot(const estts_gemm_args& a, int kind, int tile16, int waves, hipStream_t st) {
    if ((kind != 1 && kind != 2) || (a.K & 7) || (a.c_per_tap & 7) || a.batch > 1) return 1;
    const int nks = (a.K + 31) / 32;
    if (waves == 0) waves = nks >= 8 ? 8 : 4;
    if (waves != 4 && waves != 8) return 1;
    const int S = (nks + waves - 1) / waves;
    if (S > 6) return 1;
    const int tm = tile16 ? 16 : 32;
    dim3 grid((a.N + tm - 1) / tm, (a.M + tm - 1) / tm, 1);
    if (grid.y > 65535) return 1;
    if (tile16) return waves == 8 ? launch_k<1, 8>(a, kind, grid, S, st) : launch_k<1, 4>(a, kind, grid, S, st);
    return waves == 8 ? launch_k<2, 8>(a, kind, grid, S, st) : launch_k<2, 4>(a, kind, grid, S, st);
}
