// One GRU time step of a training recurrence in ONE launch (estts_gru_fused_fwd / estts_gru_fused_bwd in include/estts.h).
//
// A recurrence over 1000 steps is a chain of dependent launches, a few microseconds each whatever their size, so the chain
// pays per launch: the recurrent product and the gate arithmetic as two launches cost 8.2 us (forward) / 10.9 us (backward)
// per step at 32 rows and 512 units.  Here the product keeps the shape of the one-shot GEMM kernel (gemm_oneshot.hip: every
// load of a wave issued before its first wait, operands straight into the MFMA layout, eight waves taking interleaved K-steps,
// partial tiles summed through LDS once) and the gate arithmetic is its epilogue:
//   forward   a workgroup owns 16 rows x 16 units and computes THREE 16 x 16 tiles of gh = h_prev W_hh^T + b_hh -- the r, z
//             and n gate of the same units, i.e. W column blocks j, H + j, 2 H + j -- so that its epilogue has everything the
//             unit needs: r, z, n, the new state (and the layer output with its residual);
//   backward  a workgroup owns 16 rows x 16 units of dh_rec = dgh(t + 1) W_hh + dhd(t + 1), the hidden-state gradient that
//             reaches step t from step t + 1, and its epilogue turns it (plus the gradient from above) into the gate gradients
//             of step t: what would otherwise be the next launch.
// The epilogue operands (input gates, previous state, gradients) are requested first, like every other load.
#include "common.h"

namespace {

__device__ __forceinline__ void split8(const float4& x0, const float4& x1, bf16x8& hi, bf16x8& lo) {
    bf16x4 h0, l0, h1, l1;
    split_bf16x4(x0, h0, l0);
    split_bf16x4(x1, h1, l1);
    hi = (bf16x8){h0[0], h0[1], h0[2], h0[3], h1[0], h1[1], h1[2], h1[3]};
    lo = (bf16x8){l0[0], l0[1], l0[2], l0[3], l1[0], l1[1], l1[2], l1[3]};
}

__device__ __forceinline__ float sigm(float x) { return 1.0f / (1.0f + expf(-x)); }   // as estts_gru_step_fwd / _bwd

constexpr int NW = 8;

// NT column tiles of the product per workgroup (3: forward, 1: backward); S K-steps per wave (K <= 256 S)
template <int NSPLIT, int NT, int S>
__device__ __forceinline__ void product(const float* A, int lda, int M, int K, const uint16_t* W_hi, const uint16_t* W_lo, int ldw,
                                        const int (&nb)[NT], int m0, int wave, int lane, f32x4 (&acc)[NT]) {
    constexpr int NS = (NSPLIT > 1) ? 2 : 1;
    const int i16 = lane & 15, g = lane >> 4;
    const int KS = ldw >> 5;
    const int nks = (K + 31) >> 5;
    const float* a_ptr = A + (long)min(m0 + i16, M - 1) * lda;   // rows past M re-read the last one; never stored
    float4 av[S][2];
    uint4 wh[S][NT], wl[S][NT];
#pragma unroll
    for (int s = 0; s < S; ++s) {
        const int ks = wave + s * NW;
        if (ks < nks) {                                          // uniform over the wave
            const int kk = ks * 32 + g * 8;
            const long off = kk < K ? kk : 0;                    // past K: valid data of the row (W is zero there)
            av[s][0] = *reinterpret_cast<const float4*>(a_ptr + off);
            av[s][1] = *reinterpret_cast<const float4*>(a_ptr + off + 4);
#pragma unroll
            for (int t = 0; t < NT; ++t) {
                const long woff = ((long)nb[t] * KS + ks) * 512 + lane * 8;
                wh[s][t] = *reinterpret_cast<const uint4*>(W_hi + woff);
                if (NS == 2) wl[s][t] = *reinterpret_cast<const uint4*>(W_lo + woff);
            }
        }
    }
#pragma unroll
    for (int t = 0; t < NT; ++t) acc[t] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int s = 0; s < S; ++s) {
        const int ks = wave + s * NW;
        if (ks < nks) {
            bf16x8 ah, al;
            split8(av[s][0], av[s][1], ah, al);
            if (NS == 2) {   // small terms first
#pragma unroll
                for (int t = 0; t < NT; ++t)
                    acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(al, __builtin_bit_cast(bf16x8, wh[s][t]), acc[t], 0, 0, 0);
#pragma unroll
                for (int t = 0; t < NT; ++t)
                    acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, __builtin_bit_cast(bf16x8, wl[s][t]), acc[t], 0, 0, 0);
            }
#pragma unroll
            for (int t = 0; t < NT; ++t)
                acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, __builtin_bit_cast(bf16x8, wh[s][t]), acc[t], 0, 0, 0);
        }
    }
}

// wave w < 4 finishes element r = w of the lane's accumulator column: rows m0 + 4 g + r, column i16 of every tile
template <int NT>
__device__ __forceinline__ void reduce_tiles(float* red, const f32x4 (&acc)[NT], int wave, int lane, float (&v)[NT]) {
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int r = 0; r < 4; ++r) red[((wave * NT + t) * 4 + r) * 64 + lane] = acc[t][r];
    __syncthreads();
    if (wave < 4) {
#pragma unroll
        for (int t = 0; t < NT; ++t) {
            float s = 0.0f;
#pragma unroll
            for (int w = 0; w < NW; ++w) s += red[((w * NT + t) * 4 + wave) * 64 + lane];
            v[t] = s;
        }
    }
}

// blockIdx.z selects the argument block: two independent steps of the same width (the two layers of a WaveRNN window, one
// running a chunk of steps behind the other) share a launch, so that two recurrences cost one chain of launches
template <int NSPLIT, int S>
__global__ __launch_bounds__(NW * 64) void gru_fused_fwd_kernel(const estts_gru_fused_args pa, const estts_gru_fused_args pb) {
    __shared__ float red[NW * 3 * 4 * 64];
    const estts_gru_fused_args& p = blockIdx.z == 0 ? pa : pb;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int i16 = lane & 15, g = lane >> 4;
    const int m0 = blockIdx.y * 16, j0 = blockIdx.x * 16;
    const int H = p.H;
    if (m0 >= p.M) return;   // (the other block has more rows; uniform over the workgroup)
    // epilogue operands of element (row, col) -- loaded by every wave (the last four discard them), before the product's loads
    const int row = min(m0 + 4 * g + (wave & 3), p.M - 1);
    const int col = j0 + i16;
    const float* gi = p.gi + (long)row * p.ldgi + col;
    const float ir = gi[0], iz = gi[H], in = gi[2 * H];
    const float hp = p.h_prev[(long)row * p.ldhp + col];
    const float br = p.bias ? p.bias[col] : 0.0f, bz = p.bias ? p.bias[H + col] : 0.0f, bn = p.bias ? p.bias[2 * H + col] : 0.0f;
    const float xr = (p.out && p.x_res) ? p.x_res[(long)row * p.ldx + col] : 0.0f;

    const int nb[3] = {j0 >> 4, (H + j0) >> 4, (2 * H + j0) >> 4};
    f32x4 acc[3];
    product<NSPLIT, 3, S>(p.h_prev, p.ldhp, p.M, H, p.W_hi, p.W_lo, p.ldw, nb, m0, wave, lane, acc);
    float v[3];
    reduce_tiles<3>(red, acc, wave, lane, v);
    if (wave >= 4 || m0 + 4 * g + wave >= p.M) return;
    const float hr = v[0] + br, hz = v[1] + bz, hn = v[2] + bn;
    float* gh = p.gh + (long)row * p.ldgh + col;
    gh[0] = hr;
    gh[H] = hz;
    gh[2 * H] = hn;
    const float rr = sigm(ir + hr);
    const float zz = sigm(iz + hz);
    const float nn = tanhf(in + rr * hn);
    const float h = (1.0f - zz) * nn + zz * hp;
    p.h_out[(long)row * p.ldho + col] = h;
    if (p.out) p.out[(long)row * p.ldo + col] = h + xr;
}

template <int NSPLIT, int S>
__global__ __launch_bounds__(NW * 64) void gru_fused_bwd_kernel(const estts_gru_fused_args pa, const estts_gru_fused_args pb) {
    __shared__ float red[NW * 4 * 64];
    const estts_gru_fused_args& p = blockIdx.z == 0 ? pa : pb;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int i16 = lane & 15, g = lane >> 4;
    const int m0 = blockIdx.y * 16, j0 = blockIdx.x * 16;
    const int H = p.H;
    if (m0 >= p.M) return;
    const int row = min(m0 + 4 * g + (wave & 3), p.M - 1);
    const int col = j0 + i16;
    const float* gi = p.gi + (long)row * p.ldgi + col;
    const float* ghp = p.gh + (long)row * p.ldgh + col;
    const float ir = gi[0], iz = gi[H], in = gi[2 * H];
    const float hr = ghp[0], hz = ghp[H], hn = ghp[2 * H];
    const float hp = p.h_prev[(long)row * p.ldhp + col];
    const float d_above = p.dh[(long)row * p.lddh + col];
    const float d_z = p.dhd[(long)row * p.lddhd + col];        // z-path gradient of step t + 1 (read before this step's is stored)

    const int nb[1] = {j0 >> 4};
    f32x4 acc[1];
    product<NSPLIT, 1, S>(p.dgh_next, p.lddghn, p.M, 3 * H, p.W_hi, p.W_lo, p.ldw, nb, m0, wave, lane, acc);
    float v[1];
    reduce_tiles<1>(red, acc, wave, lane, v);
    if (wave >= 4 || m0 + 4 * g + wave >= p.M) return;
    const float d = d_above + (v[0] + d_z);
    const float rr = sigm(ir + hr);
    const float zz = sigm(iz + hz);
    const float nn = tanhf(in + rr * hn);
    const float dn = d * (1.0f - zz);
    const float dz = d * (hp - nn);
    const float da_n = dn * (1.0f - nn * nn);
    const float ar = da_n * hn * rr * (1.0f - rr);
    const float az = dz * zz * (1.0f - zz);
    float* x = p.dgi + (long)row * p.lddgi + col;
    float* y = p.dgh + (long)row * p.lddgh + col;
    x[0] = ar;
    x[H] = az;
    x[2 * H] = da_n;
    y[0] = ar;
    y[H] = az;
    y[2 * H] = da_n * rr;
    p.dhd[(long)row * p.lddhd + col] = d * zz;
}

template <int NSPLIT, bool FWD>
int launch(const estts_gru_fused_args& a, const estts_gru_fused_args* second, hipStream_t st) {
    const estts_gru_fused_args& b = second ? *second : a;
    const int nks = ((FWD ? a.H : 3 * a.H) + 31) / 32;
    const int s = (nks + NW - 1) / NW;
    dim3 grid(a.H / 16, ((a.M > b.M ? a.M : b.M) + 15) / 16, second ? 2 : 1);
#define ESTTS_GRU_CASE(SV)                                                                                        \
    case SV:                                                                                                      \
        if (FWD) ESTTS_LAUNCH((gru_fused_fwd_kernel<NSPLIT, SV>), grid, dim3(NW * 64), 0, st, a, b);      \
        else ESTTS_LAUNCH((gru_fused_bwd_kernel<NSPLIT, SV>), grid, dim3(NW * 64), 0, st, a, b);          \
        break;
    switch (s) {
        ESTTS_GRU_CASE(1)
        ESTTS_GRU_CASE(2)
        ESTTS_GRU_CASE(3)
        ESTTS_GRU_CASE(4)
        ESTTS_GRU_CASE(6)
        default:
            if (s == 5) {
                if (FWD) ESTTS_LAUNCH((gru_fused_fwd_kernel<NSPLIT, 6>), grid, dim3(NW * 64), 0, st, a, b);
                else ESTTS_LAUNCH((gru_fused_bwd_kernel<NSPLIT, 6>), grid, dim3(NW * 64), 0, st, a, b);
                break;
            }
            estts_set_error("estts_gru_fused: H = %d is beyond one launch (H <= 1536 forward, 3 H <= 1536 backward)", a.H);
            return -1;
    }
#undef ESTTS_GRU_CASE
    return 0;
}

int check_common(const estts_gru_fused_args& a, const char* who) {
    if (!(a.gi && a.gh && a.h_prev && a.W_hi && a.M > 0 && a.H > 0 && (a.H & 15) == 0)) {
        estts_set_error("%s: null pointer or H not a positive multiple of 16", who);
        return -1;
    }
    if (!(a.precision == 1 || (a.precision == 3 && a.W_lo))) {
        estts_set_error("%s: precision must be 1, or 3 with W_lo", who);
        return -1;
    }
    if (a.ldgi < 3 * a.H || a.ldgh < 3 * a.H || a.ldhp < a.H) {
        estts_set_error("%s: ld too small", who);
        return -1;
    }
    return 0;
}

}  // namespace

namespace {

int check_fwd(const estts_gru_fused_args& a) {
    if (check_common(a, "estts_gru_fused_fwd") != 0) return -1;
    ESTTS_CHECK_ARG(a.h_out && a.ldho >= a.H && (!a.out || a.ldo >= a.H) && (!a.x_res || a.ldx >= a.H), "estts_gru_fused_fwd: h_out / out / x_res");
    ESTTS_CHECK_ARG((a.ldhp & 3) == 0 && ((uintptr_t)a.h_prev & 15) == 0, "estts_gru_fused_fwd: h_prev must be 16-byte aligned, ld a multiple of 4");
    ESTTS_CHECK_ARG(a.ldw >= ((a.H + 63) / 64) * 64 && (a.ldw & 63) == 0, "estts_gru_fused_fwd: ldw must be H rounded up to 64");
    ESTTS_CHECK_ARG(a.h_out != a.h_prev, "estts_gru_fused_fwd: h_out must not be h_prev (other workgroups still read it)");
    return 0;
}

int check_bwd(const estts_gru_fused_args& a) {
    if (check_common(a, "estts_gru_fused_bwd") != 0) return -1;
    ESTTS_CHECK_ARG(a.dgh_next && a.dh && a.dgi && a.dgh && a.dhd, "estts_gru_fused_bwd: null gradient pointer");
    ESTTS_CHECK_ARG(a.lddghn >= 3 * a.H && a.lddgi >= 3 * a.H && a.lddgh >= 3 * a.H && a.lddh >= a.H && a.lddhd >= a.H, "estts_gru_fused_bwd: ld too small");
    ESTTS_CHECK_ARG((a.lddghn & 3) == 0 && ((uintptr_t)a.dgh_next & 15) == 0, "estts_gru_fused_bwd: dgh_next must be 16-byte aligned, ld a multiple of 4");
    ESTTS_CHECK_ARG(a.ldw >= ((3 * a.H + 63) / 64) * 64 && (a.ldw & 63) == 0, "estts_gru_fused_bwd: ldw must be 3 H rounded up to 64");
    ESTTS_CHECK_ARG(a.dgh != a.dgh_next, "estts_gru_fused_bwd: dgh must not be dgh_next (other workgroups still read it)");
    return 0;
}

template <bool FWD>
int run(const estts_gru_fused_args* args, const estts_gru_fused_args* second, void* stream, const char* who) {
    ESTTS_CHECK_ARG(args != nullptr, "%s: null args", who);
    const estts_gru_fused_args& a = *args;
    if ((FWD ? check_fwd(a) : check_bwd(a)) != 0) return -1;
    if (second) {
        if ((FWD ? check_fwd(*second) : check_bwd(*second)) != 0) return -1;
        ESTTS_CHECK_ARG(second->H == a.H && second->precision == a.precision, "%s: the two steps of a launch must have the same H and precision", who);
    }
    const int r = a.precision == 3 ? launch<3, FWD>(a, second, (hipStream_t)stream) : launch<1, FWD>(a, second, (hipStream_t)stream);
    if (r != 0) return r;
    ESTTS_CHECK_LAUNCH(who);
    return 0;
}

}  // namespace

extern "C" int estts_gru_fused_fwd(const estts_gru_fused_args* args, void* stream) {
    return run<true>(args, nullptr, stream, "estts_gru_fused_fwd");
}

extern "C" int estts_gru_fused_bwd(const estts_gru_fused_args* args, void* stream) {
    return run<false>(args, nullptr, stream, "estts_gru_fused_bwd");
}

extern "C" int estts_gru_fused_fwd2(const estts_gru_fused_args* a, const estts_gru_fused_args* b, void* stream) {
    ESTTS_CHECK_ARG(b != nullptr, "estts_gru_fused_fwd2: null args");
    return run<true>(a, b, stream, "estts_gru_fused_fwd2");
}

extern "C" int estts_gru_fused_bwd2(const estts_gru_fused_args* a, const estts_gru_fused_args* b, void* stream) {
    ESTTS_CHECK_ARG(b != nullptr, "estts_gru_fused_bwd2: null args");
    return run<false>(a, b, stream, "estts_gru_fused_bwd2");
}
