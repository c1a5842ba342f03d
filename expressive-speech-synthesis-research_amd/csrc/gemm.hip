// Fused dense / 1-D-convolution kernel on the gfx950 matrix cores (see estts_gemm in include/estts.h).
//
// Tiling: one 256-thread workgroup (4 waves) owns a BM x BN tile of C and walks K in steps of 32.
// A (float32 activations) is read with 16-byte loads, split into bf16 hi (+ lo) in registers and
// staged in LDS; W (bf16, [N][K], K contiguous) is staged as is.  Both LDS images use 80-byte rows
// (32 bf16 + 8 pad) so that the 16 rows one ds_read_b128 lane group touches fall on disjoint banks.
// The next K-step's global loads are issued before the current step's MFMAs (register prefetch,
// two LDS buffers, one barrier per step).  v_mfma_f32_16x16x32_bf16: lane l holds
// A[row l&15][k 8(l>>4)..+7], B[k 8(l>>4)..+7][col l&15]; D[row 4(l>>4)+j][col l&15].
// A convolution is the same loop with K = taps * c_per_tap: K-chunk -> (tap, channel), the A row is
// shifted by tap - tap_lpad and rows that leave the sequence read as zero.
#include "common.h"

namespace {

constexpr int BK = 32;
constexpr int LDS_LD = 40;  // bf16 elements per LDS row

template <int BM, int BN, int WM, int WN, int NSPLIT>
__global__ __launch_bounds__(256) void gemm_kernel(const estts_gemm_args p) {
    constexpr int NS = (NSPLIT > 1) ? 2 : 1;
    constexpr int A_ITERS = BM * 8 / 256;
    constexpr int W_ITERS = BN * 4 / 256;
    constexpr int MI = BM / WM / 16;
    constexpr int NI = BN / WN / 16;
    static_assert(A_ITERS >= 1 && W_ITERS >= 1 && MI >= 1 && NI >= 1, "tile too small");

    __shared__ __attribute__((aligned(16))) __bf16 As[2][NS][BM * LDS_LD];
    __shared__ __attribute__((aligned(16))) __bf16 Ws[2][NS][BN * LDS_LD];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int wm = wave / WN;
    const int wn = wave % WN;
    const int m0 = blockIdx.y * BM;
    const int n0 = blockIdx.x * BN;
    const int seg = p.seg_len > 0 ? p.seg_len : p.M;

    // per-thread A load slots
    int a_row[A_ITERS], a_t[A_ITERS];
    bool a_ok[A_ITERS];
#pragma unroll
    for (int i = 0; i < A_ITERS; ++i) {
        const int idx = tid + i * 256;
        const int m = m0 + (idx >> 3);
        a_row[i] = m;
        a_ok[i] = m < p.M;
        a_t[i] = m % seg;
    }
    const int a_ch = tid & 7;  // (tid + i*256) & 7 is the same for every i

    float4 a_reg[A_ITERS];
    uint4 w_reg[NS][W_ITERS];

    auto load_tiles = [&](int k0) {
        const int kk = k0 + a_ch * 4;
        int tap = 0, c = kk;
        if (p.taps > 1) {
            tap = kk / p.c_per_tap;
            c = kk - tap * p.c_per_tap;
        }
        const int shift = tap - p.tap_lpad;
#pragma unroll
        for (int i = 0; i < A_ITERS; ++i) {
            const int t = a_t[i] + shift;
            const bool ok = a_ok[i] && kk < p.K && t >= 0 && t < seg;
            a_reg[i] = make_float4(0.f, 0.f, 0.f, 0.f);
            if (ok) a_reg[i] = *reinterpret_cast<const float4*>(p.A + (long)(a_row[i] + shift) * p.lda + c);
        }
#pragma unroll
        for (int i = 0; i < W_ITERS; ++i) {
            const int idx = tid + i * 256;
            const long off = (long)(n0 + (idx >> 2)) * p.ldw + k0 + (idx & 3) * 8;
            w_reg[0][i] = *reinterpret_cast<const uint4*>(p.W_hi + off);
            if (NS == 2) w_reg[NS - 1][i] = *reinterpret_cast<const uint4*>(p.W_lo + off);
        }
    };

    auto store_tiles = [&](int buf) {
#pragma unroll
        for (int i = 0; i < A_ITERS; ++i) {
            const int idx = tid + i * 256;
            const int o = (idx >> 3) * LDS_LD + a_ch * 4;
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
            const int idx = tid + i * 256;
            const int o = (idx >> 2) * LDS_LD + (idx & 3) * 8;
            *reinterpret_cast<uint4*>(&Ws[buf][0][o]) = w_reg[0][i];
            if (NS == 2) *reinterpret_cast<uint4*>(&Ws[buf][NS - 1][o]) = w_reg[NS - 1][i];
        }
    };

    f32x4 acc[MI][NI];
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int j = 0; j < NI; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

    const int nk = (p.K + BK - 1) / BK;
    const int frag_off = (lane & 15) * LDS_LD + (lane >> 4) * 8;
    const int a_base = wm * (BM / WM) * LDS_LD + frag_off;
    const int w_base = wn * (BN / WN) * LDS_LD + frag_off;

    load_tiles(0);
    int buf = 0;
    for (int kt = 0; kt < nk; ++kt) {
        store_tiles(buf);
        __syncthreads();
        if (kt + 1 < nk) load_tiles((kt + 1) * BK);

        bf16x8 af[NS][MI], wf[NS][NI];
#pragma unroll
        for (int s = 0; s < NS; ++s) {
#pragma unroll
            for (int i = 0; i < MI; ++i)
                af[s][i] = *reinterpret_cast<const bf16x8*>(&As[buf][s][a_base + i * 16 * LDS_LD]);
#pragma unroll
            for (int j = 0; j < NI; ++j)
                wf[s][j] = *reinterpret_cast<const bf16x8*>(&Ws[buf][s][w_base + j * 16 * LDS_LD]);
        }
#pragma unroll
        for (int i = 0; i < MI; ++i)
#pragma unroll
            for (int j = 0; j < NI; ++j) {
                if (NS == 2) {
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[NS - 1][i], wf[0][j], acc[i][j], 0, 0, 0);
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[0][i], wf[NS - 1][j], acc[i][j], 0, 0, 0);
                }
                acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[0][i], wf[0][j], acc[i][j], 0, 0, 0);
            }
        buf ^= 1;
    }

    // epilogue
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
#pragma unroll
        for (int i = 0; i < MI; ++i) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int row = m0 + wm * (BM / WM) + i * 16 + row_l + r;
                if (row >= p.M) continue;
                float v = acc[i][j][r] * s1 + b1;
                v = apply_act(v, p.act) * p.alpha;
                if (p.R) {
                    const int rr = p.r_mod > 0 ? row % p.r_mod : row;
                    v += p.R[(long)rr * p.ldr + col];
                }
                if (p.s2) v = v * s2 + b2;
                p.C[(long)row * p.ldc + col] = v;
            }
        }
    }
}

template <int BM, int BN, int WM, int WN>
int launch(const estts_gemm_args& a, hipStream_t st) {
    dim3 grid((a.N + BN - 1) / BN, (a.M + BM - 1) / BM);
    if (a.precision == 3)
        hipLaunchKernelGGL((gemm_kernel<BM, BN, WM, WN, 3>), grid, dim3(256), 0, st, a);
    else
        hipLaunchKernelGGL((gemm_kernel<BM, BN, WM, WN, 1>), grid, dim3(256), 0, st, a);
    ESTTS_CHECK_LAUNCH("estts_gemm");
    return 0;
}

}  // namespace

extern "C" int estts_gemm(const estts_gemm_args* args, void* stream) {
    ESTTS_CHECK_ARG(args != nullptr, "estts_gemm: null args");
    const estts_gemm_args& a = *args;
    ESTTS_CHECK_ARG(a.A && a.W_hi && a.C, "estts_gemm: null A/W/C");
    ESTTS_CHECK_ARG(a.M > 0 && a.N > 0 && a.K > 0, "estts_gemm: bad shape M=%d N=%d K=%d", a.M, a.N, a.K);
    ESTTS_CHECK_ARG(a.precision == 1 || a.precision == 3, "estts_gemm: precision must be 1 or 3");
    ESTTS_CHECK_ARG(a.precision == 1 || a.W_lo, "estts_gemm: precision 3 needs W_lo");
    ESTTS_CHECK_ARG(a.taps >= 1 && a.c_per_tap > 0 && a.taps * a.c_per_tap == a.K,
                    "estts_gemm: taps*c_per_tap != K (%d*%d vs %d)", a.taps, a.c_per_tap, a.K);
    ESTTS_CHECK_ARG((a.K & 3) == 0 && (a.c_per_tap & 3) == 0 && (a.lda & 3) == 0,
                    "estts_gemm: K, c_per_tap and lda must be multiples of 4");
    ESTTS_CHECK_ARG(((uintptr_t)a.A & 15) == 0 && ((uintptr_t)a.W_hi & 15) == 0 && ((uintptr_t)a.W_lo & 15) == 0,
                    "estts_gemm: A / W must be 16-byte aligned");
    ESTTS_CHECK_ARG(a.ldw >= ((a.K + 31) / 32) * 32 && (a.ldw & 7) == 0, "estts_gemm: ldw too small / unaligned");
    ESTTS_CHECK_ARG(a.seg_len == 0 || a.M % a.seg_len == 0, "estts_gemm: M not a multiple of seg_len");
    ESTTS_CHECK_ARG(a.taps == 1 || a.lda >= a.c_per_tap, "estts_gemm: lda < c_per_tap");
    ESTTS_CHECK_ARG(!a.R || a.ldr >= a.N, "estts_gemm: ldr < N");
    ESTTS_CHECK_ARG(a.ldc >= a.N, "estts_gemm: ldc < N");
    hipStream_t st = (hipStream_t)stream;
    const long tiles64 = (long)((a.M + 63) / 64) * ((a.N + 63) / 64);
    if (tiles64 >= 256) return launch<64, 64, 2, 2>(a, st);
    return launch<32, 64, 1, 4>(a, st);
}
