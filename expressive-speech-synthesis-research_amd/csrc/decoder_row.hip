// One decoder position through a whole CrossAttentionResnorm block in one launch (estts_cross_attention_row in include/estts.h):
//   x  = LNb(act(LNa(ln_y)) + ln_x)          (the normalisation pending from the block before, as in estts_gemm's LayerNorm prologue;
//                                             or the row as given), written to ln_out for the residual uses that follow
//   q  = x Wq + bq;   p_h = softmax_j(q_h . k_jh / sqrt(depth) - 1e9 mask_j);   ctx_h = sum_j p_hj v_jh
//   y  = [x | ctx] Wo + bo
// (TransformerTTS/model/layers.py:258-274 with transformer_utils.py:24-44 for one query row.)  In an incremental decoder these are
// three launches of one row each -- projection, attention over the encoder's keys, output projection -- whose time is their
// launch and first-round-trip latency.  Here a row is four workgroups of 1024 threads, one per head.  The work of a head is
// independent of the other heads up to the last sum: its 64 outputs of q, its scores and context, and the part of the output
// projection that multiplies its 64 columns of x and of ctx.  That is 240 KB of weights, keys and values per workgroup, 60
// registers per thread: EVERY operand is requested in the first instructions of the kernel, so the launch pays one round trip to
// memory and the phases after it are arithmetic and LDS.  (One workgroup per row cannot hold 0.9 MB in registers; it paid a
// round trip per phase and took 19 us where the three launches take 12.)  The four partial outputs go through a slab; the last
// workgroup of the row to arrive adds them in head order, so the result does not depend on the order of arrival.
// One row is a matrix-vector product: the vector ALU does it in float32 on hi + lo, which is at least as exact as the split
// products of estts_gemm.
#include "common.h"

namespace {

constexpr int CR_D = 256;        // model width this kernel is written for
constexpr int CR_H = 4;          // heads, of depth 64
constexpr int CR_LK_MAX = 256;   // encoder positions it can hold scores for

__device__ __forceinline__ float4 ld4g(const float* p) { return *reinterpret_cast<const float4*>(p); }

// sum over the 16 lanes of a row of the wave (every lane of the row gets it)
__device__ __forceinline__ float row16_sum(float v) {
    const float ident = 0.0f;
#define CR_DPP(ctrl) __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(__builtin_bit_cast(int, ident), __builtin_bit_cast(int, v), ctrl, 0xf, 0xf, false))
    v += CR_DPP(0xB1);    // quad_perm [1, 0, 3, 2]
    v += CR_DPP(0x4E);    // quad_perm [2, 3, 0, 1]
    v += CR_DPP(0x141);   // row_half_mirror
    v += CR_DPP(0x140);   // row_mirror
#undef CR_DPP
    return v;
}

// A 1 KiB block of the packed weight image holds, at lane L's 16 bytes, the eight weights of output (block row) * 16 + (L & 15)
// for inputs (block column) * 32 + 8 (L >> 4) .. + 7 (the B-operand register image of the matrix instruction).
// dot16: sixteen outputs of x . W^T over NKB such blocks already in registers, xs = their NKB * 32 inputs in LDS; every lane
// gets the sum of its output (L & 15).
template <int NKB>
__device__ __forceinline__ float dot16(const uint4* wh, const uint4* wl, const float* xs, const int lane) {
    float acc = 0.0f;
    const int ko = 8 * (lane >> 4);
#pragma unroll
    for (int kb = 0; kb < NKB; ++kb) {
        const float4 x0 = *reinterpret_cast<const float4*>(xs + kb * 32 + ko), x1 = *reinterpret_cast<const float4*>(xs + kb * 32 + ko + 4);
        const float xv[8] = {x0.x, x0.y, x0.z, x0.w, x1.x, x1.y, x1.z, x1.w};
        const uint32_t hw[4] = {wh[kb].x, wh[kb].y, wh[kb].z, wh[kb].w}, lw[4] = {wl[kb].x, wl[kb].y, wl[kb].z, wl[kb].w};
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const float h0 = __builtin_bit_cast(float, hw[e] << 16), h1 = __builtin_bit_cast(float, hw[e] & 0xffff0000u);
            const float l0 = __builtin_bit_cast(float, lw[e] << 16), l1 = __builtin_bit_cast(float, lw[e] & 0xffff0000u);
            acc += xv[2 * e] * (h0 + l0);
            acc += xv[2 * e + 1] * (h1 + l1);
        }
    }
    acc += lane_xor16(acc);
    acc += lane_xor32(acc);
    return acc;
}

// NJ = keys per lane group: key 4 wave + (lane >> 4) + 64 i, i < NJ.  PROJ = false: the query row is given (p.q) and the kernel
// is attention + output projection, for a self-attention block whose q | k | v row one product has just written.
template <int NJ, bool PROJ>
__global__ __launch_bounds__(1024) void cross_attention_row_kernel(const estts_cross_row_args p) {
    __shared__ __attribute__((aligned(16))) float xs[CR_D];           // the row
    __shared__ float qp[4][64];                                        // q of this head from the four quarters of x
    __shared__ __attribute__((aligned(16))) float qs[64];
    __shared__ float sc[CR_LK_MAX];                                    // scores, then weights
    __shared__ __attribute__((aligned(16))) float part[16][64];       // partial contexts of the sixteen waves
    __shared__ __attribute__((aligned(16))) float cs[64];             // context of this head
    __shared__ int s_ticket;
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int h = blockIdx.x, row = blockIdx.y;
    const int Lk = p.Lk;

    // ---- every operand, requested now
    // q: wave w multiplies block row 4 h + (w & 3) of Wq by the (w >> 2)-th quarter of x
    const int qn = wave & 3, qk = wave >> 2;
    uint4 wqh[2], wql[2];
    if constexpr (PROJ) {
        const long base = ((long)(4 * h + qn) * (p.ldwq >> 5) + 2 * qk) * 512 + lane * 8;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            wqh[i] = *reinterpret_cast<const uint4*>(p.wq_hi + base + i * 512);
            wql[i] = *reinterpret_cast<const uint4*>(p.wq_lo + base + i * 512);
        }
    }
    // y: wave w multiplies block row w of Wo by this head's 64 columns of x (blocks 2 h, 2 h + 1) and of ctx (8 + 2 h, 9 + 2 h)
    uint4 woh[4], wol[4];
    {
        const long base = (long)wave * (p.ldwo >> 5) * 512 + lane * 8;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int kb = (i >> 1) * (CR_D / 32) + 2 * h + (i & 1);
            woh[i] = *reinterpret_cast<const uint4*>(p.wo_hi + base + kb * 512);
            wol[i] = *reinterpret_cast<const uint4*>(p.wo_lo + base + kb * 512);
        }
    }
    // keys and values: 16 lanes x float4 = the 64 columns of this head in one row
    const int jq = 4 * wave + (lane >> 4), col = 64 * h + 4 * (lane & 15);
    const float* kb = p.k + (long)row * p.kv_batch_stride + col;
    const float* vb = p.v + (long)row * p.kv_batch_stride + col;
    const float* mask = p.key_mask ? p.key_mask + (long)row * Lk : nullptr;
    float4 kr[NJ], vr[NJ];
    float mk[NJ];
#pragma unroll
    for (int i = 0; i < NJ; ++i) {
        const int j = min(jq + 64 * i, Lk - 1);
        kr[i] = ld4g(kb + (long)j * p.ldk);
        mk[i] = mask ? mask[j] : 0.0f;
    }
    if constexpr (NJ <= 2) {
#pragma unroll
        for (int i = 0; i < NJ; ++i) vr[i] = ld4g(vb + (long)min(jq + 64 * i, Lk - 1) * p.ldv);
    }

    // ---- the row itself: as given, or the pending normalisation (wave 0; lane <-> columns 4 lane .. 4 lane + 3)
    if (wave == 0) {
        const int c = lane * 4;
        float4 v;
        if (p.ln_y) {
            v = ld4g(p.ln_y + (long)row * p.ln_ldy + c);
            float4 xr = make_float4(0.f, 0.f, 0.f, 0.f), ga = xr, ba = xr, gb = xr, bb = xr;
            if (p.ln_x) xr = ld4g(p.ln_x + (long)row * p.ln_ldx + c);
            if (p.ln_ga) { ga = ld4g(p.ln_ga + c); ba = ld4g(p.ln_ba + c); }
            if (p.ln_gb) { gb = ld4g(p.ln_gb + c); bb = ld4g(p.ln_bb + c); }
            auto layer_norm = [&](const float4& g, const float4& b) {
                const float mean = wave_sum(v.x + v.y + v.z + v.w) / (float)CR_D;
                const float a0 = v.x - mean, a1 = v.y - mean, a2 = v.z - mean, a3 = v.w - mean;
                const float inv = rsqrtf(wave_sum(a0 * a0 + a1 * a1 + a2 * a2 + a3 * a3) / (float)CR_D + p.ln_eps);
                v.x = a0 * inv * g.x + b.x; v.y = a1 * inv * g.y + b.y; v.z = a2 * inv * g.z + b.z; v.w = a3 * inv * g.w + b.w;
            };
            if (p.ln_ga) layer_norm(ga, ba);
            v.x = apply_act(v.x, p.ln_act) + xr.x; v.y = apply_act(v.y, p.ln_act) + xr.y;
            v.z = apply_act(v.z, p.ln_act) + xr.z; v.w = apply_act(v.w, p.ln_act) + xr.w;
            if (p.ln_gb) layer_norm(gb, bb);
            if (p.ln_out && h == 0) *reinterpret_cast<float4*>(p.ln_out + (long)row * p.ln_ldo + c) = v;
        } else {
            v = ld4g(p.x + (long)row * p.ldx + c);
        }
        *reinterpret_cast<float4*>(xs + c) = v;
    }
    if constexpr (!PROJ) {
        if (wave == 1 && lane < 16) *reinterpret_cast<float4*>(qs + 4 * lane) = ld4g(p.q + (long)row * p.ldq + 64 * h + 4 * lane);
    }
    __syncthreads();
    if constexpr (NJ > 2) {   // (no registers for these next to the normalisation's operands: requested one phase later)
#pragma unroll
        for (int i = 0; i < NJ; ++i) vr[i] = ld4g(vb + (long)min(jq + 64 * i, Lk - 1) * p.ldv);
    }

    // ---- q_h = x Wq_h + bq_h, and the x half of this head's share of y
    if constexpr (PROJ) {
        const float a = dot16<2>(wqh, wql, xs + 64 * qk, lane);
        if (lane < 16) qp[qk][qn * 16 + lane] = a;
    }
    float yacc = dot16<2>(woh, wol, xs + 64 * h, lane);
    if constexpr (PROJ) {
        __syncthreads();
        if (tid < 64) qs[tid] = ((qp[0][tid] + qp[1][tid]) + (qp[2][tid] + qp[3][tid])) + (p.bq ? p.bq[64 * h + tid] : 0.0f);
        __syncthreads();
    }

    // ---- scores
    {
        const float4 qv = *reinterpret_cast<const float4*>(qs + 4 * (lane & 15));
#pragma unroll
        for (int i = 0; i < NJ; ++i) {
            const int j = jq + 64 * i;
            const float d = row16_sum(kr[i].x * qv.x + kr[i].y * qv.y + kr[i].z * qv.z + kr[i].w * qv.w);
            if ((lane & 15) == 0 && j < Lk) sc[j] = d * 0.125f + mk[i] * -1e9f;   // 1 / sqrt(64)
        }
    }
    __syncthreads();

    // ---- softmax (wave 0), weights out
    if (wave == 0) {
        float m = -INFINITY;
        for (int j = lane; j < Lk; j += 64) m = fmaxf(m, sc[j]);
        m = wave_max(m);
        float l = 0.0f;
        for (int j = lane; j < Lk; j += 64) {
            const float e = __builtin_amdgcn_exp2f((sc[j] - m) * 1.4426950408889634f);
            sc[j] = e;
            l += e;
        }
        const float inv_l = 1.0f / wave_sum(l);
        float* ap = p.attn ? p.attn + (long)row * p.attn_row_stride + (long)h * Lk : nullptr;
        for (int j = lane; j < Lk; j += 64) {
            const float w = sc[j] * inv_l;
            sc[j] = w;
            if (ap) ap[j] = w;
        }
    }
    __syncthreads();

    // ---- context: the lane groups' keys, then the four groups of a wave, then the sixteen waves
    {
        float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
        for (int i = 0; i < NJ; ++i) {
            const int j = jq + 64 * i;
            const float w = j < Lk ? sc[j] : 0.0f;
            acc.x += w * vr[i].x; acc.y += w * vr[i].y; acc.z += w * vr[i].z; acc.w += w * vr[i].w;
        }
        acc.x += lane_xor16(acc.x); acc.y += lane_xor16(acc.y); acc.z += lane_xor16(acc.z); acc.w += lane_xor16(acc.w);
        acc.x += lane_xor32(acc.x); acc.y += lane_xor32(acc.y); acc.z += lane_xor32(acc.z); acc.w += lane_xor32(acc.w);
        if (lane < 16) *reinterpret_cast<float4*>(&part[wave][4 * lane]) = acc;
    }
    __syncthreads();
    if (tid < 64) {
        float t = 0.0f;
#pragma unroll
        for (int w = 0; w < 16; ++w) t += part[w][tid];
        cs[tid] = t;
    }
    __syncthreads();

    // ---- this head's share of y = [x | ctx] Wo, to the slab
    yacc += dot16<2>(woh + 2, wol + 2, cs, lane);
    float* slab = p.ws + (long)row * CR_H * CR_D;
    if (lane < 16) slab[h * CR_D + wave * 16 + lane] = yacc;

    // ---- the last of the row's four workgroups adds the shares in head order (release / ticket / acquire as in estts_gemm's
    // split-K hand-off; the counter is left at zero for the next launch)
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (tid == 0) {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        s_ticket = __hip_atomic_fetch_add(p.counters + row, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    __syncthreads();
    if (s_ticket != CR_H - 1) return;
    if (tid == 0) {
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __hip_atomic_store(p.counters + row, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    __syncthreads();
    if (tid < CR_D) {
        const float s = ((slab[tid] + slab[CR_D + tid]) + slab[2 * CR_D + tid]) + slab[3 * CR_D + tid];
        p.y[(long)row * p.ldy + tid] = s + (p.bo ? p.bo[tid] : 0.0f);
    }
}

}  // namespace

extern "C" int estts_cross_attention_row(const estts_cross_row_args* args, void* stream) {
    ESTTS_CHECK_ARG(args != nullptr, "estts_cross_attention_row: null args");
    const estts_cross_row_args& a = *args;
    ESTTS_CHECK_ARG(a.M > 0 && a.M <= 65535 && a.d == CR_D && a.H == CR_H,
                    "estts_cross_attention_row: this kernel takes width 256 with heads of depth 64 (got width %d, %d heads)", a.d, a.H);
    ESTTS_CHECK_ARG(a.Lk > 0 && a.Lk <= CR_LK_MAX, "estts_cross_attention_row: 1 .. %d keys (got %d)", CR_LK_MAX, a.Lk);
    ESTTS_CHECK_ARG((a.ln_y || a.x) && ((a.wq_hi && a.wq_lo) || a.q) && a.wo_hi && a.wo_lo && a.k && a.v && a.y, "estts_cross_attention_row: null operand");
    ESTTS_CHECK_ARG(!a.q || (((uintptr_t)a.q & 15) == 0 && (a.ldq & 3) == 0), "estts_cross_attention_row: q must be 16-byte aligned with a row stride that is a multiple of 4");
    ESTTS_CHECK_ARG((a.q || (a.ldwq >= CR_D && (a.ldwq & 63) == 0)) && a.ldwo >= 2 * CR_D && (a.ldwo & 63) == 0, "estts_cross_attention_row: ldw must be K rounded up to 64");
    ESTTS_CHECK_ARG(!a.ln_y || ((!a.ln_ga == !a.ln_ba) && (!a.ln_gb == !a.ln_bb) && (a.ln_ga || a.ln_gb) && (a.ln_ga || a.ln_act == ESTTS_ACT_NONE)),
                    "estts_cross_attention_row: the pending normalisation needs at least one of (ln_ga, ln_ba) / (ln_gb, ln_bb); ln_act only with the first");
    ESTTS_CHECK_ARG(a.ldk >= CR_D && a.ldv >= CR_D && (a.ldk & 3) == 0 && (a.ldv & 3) == 0 && (a.kv_batch_stride & 3) == 0 &&
                    (((uintptr_t)a.k | (uintptr_t)a.v | (uintptr_t)a.x | (uintptr_t)a.ln_y | (uintptr_t)a.ln_x | (uintptr_t)a.ln_out |
                      (uintptr_t)a.ln_ga | (uintptr_t)a.ln_ba | (uintptr_t)a.ln_gb | (uintptr_t)a.ln_bb | (uintptr_t)a.wq_hi | (uintptr_t)a.wq_lo |
                      (uintptr_t)a.wo_hi | (uintptr_t)a.wo_lo) & 15) == 0 &&
                    (a.ldx & 3) == 0 && (a.ln_ldy & 3) == 0 && (a.ln_ldx & 3) == 0 && (a.ln_ldo & 3) == 0,
                    "estts_cross_attention_row: operands must be 16-byte aligned with row strides that are multiples of 4");
    ESTTS_CHECK_ARG(!a.ln_out || (a.ln_out != a.ln_y && a.ln_out != a.ln_x), "estts_cross_attention_row: ln_out must not alias ln_y / ln_x");
    ESTTS_CHECK_ARG(a.ws && a.counters && a.ws_bytes >= (int64_t)a.M * CR_H * CR_D * 4 && a.n_counters >= a.M && ((uintptr_t)a.ws & 15) == 0,
                    "estts_cross_attention_row: needs a slab of M x 4 x 256 floats and M counters (zero before the first launch)");
    const dim3 grid(CR_H, a.M);
    hipStream_t st = (hipStream_t)stream;
    if (a.q) {
        if (a.Lk <= 64) ESTTS_LAUNCH((cross_attention_row_kernel<1, false>), grid, dim3(1024), 0, st, a);
        else if (a.Lk <= 128) ESTTS_LAUNCH((cross_attention_row_kernel<2, false>), grid, dim3(1024), 0, st, a);
        else ESTTS_LAUNCH((cross_attention_row_kernel<4, false>), grid, dim3(1024), 0, st, a);
    } else {
        if (a.Lk <= 128) ESTTS_LAUNCH((cross_attention_row_kernel<2, true>), grid, dim3(1024), 0, st, a);
        else ESTTS_LAUNCH((cross_attention_row_kernel<4, true>), grid, dim3(1024), 0, st, a);
    }
    ESTTS_CHECK_LAUNCH("estts_cross_attention_row");
    return 0;
}
