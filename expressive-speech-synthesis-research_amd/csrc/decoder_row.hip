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
// One row is a matrix-vector product: the vector ALU does it, on the split operands of estts_gemm (see dot16).
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
// dot16: sixteen outputs of x . W^T over NKB such blocks already in registers; xh / xl = their NKB * 32 inputs in LDS, split into
// bf16 hi + lo like the weights; every lane gets the sum of its output (L & 15).  The products are those of estts_gemm's split
// operands (hi hi + hi lo + lo hi, accumulated in float32), two per v_dot2c_f32_bf16 straight from the packed pairs: 1.5 vector
// instructions per weight where unpacking to float32 takes 4, and at one row per workgroup these kernels are bound by exactly
// that once their operands have arrived.
typedef __bf16 bf16p __attribute__((ext_vector_type(2)));
__device__ __forceinline__ float dot2(uint32_t a, uint32_t b, float c) {
    return __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(bf16p, a), __builtin_bit_cast(bf16p, b), c, false);
}
template <int NKB>
__device__ __forceinline__ float dot16(const uint4* wh, const uint4* wl, const __bf16* xh, const __bf16* xl, const int lane) {
    float acc0 = 0.0f, acc1 = 0.0f, acc2 = 0.0f;
    const int ko = 8 * (lane >> 4);
#pragma unroll
    for (int kb = 0; kb < NKB; ++kb) {
        const uint4 a = *reinterpret_cast<const uint4*>(xh + kb * 32 + ko), b = *reinterpret_cast<const uint4*>(xl + kb * 32 + ko);
        const uint32_t ah[4] = {a.x, a.y, a.z, a.w}, al[4] = {b.x, b.y, b.z, b.w};
        const uint32_t hw[4] = {wh[kb].x, wh[kb].y, wh[kb].z, wh[kb].w}, lw[4] = {wl[kb].x, wl[kb].y, wl[kb].z, wl[kb].w};
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            acc0 = dot2(hw[e], ah[e], acc0);
            acc1 = dot2(hw[e], al[e], acc1);
            acc2 = dot2(lw[e], ah[e], acc2);
        }
    }
    float acc = acc0 + (acc1 + acc2);
    acc += lane_xor16(acc);
    acc += lane_xor32(acc);
    return acc;
}

// a float4 of a row / one value of it, split into the bf16 hi and lo images that dot16 reads
__device__ __forceinline__ void put_split4(const float4 v, __bf16* xh, __bf16* xl, const int i4) {
    bf16x4 hi, lo;
    split_bf16x4(v, hi, lo);
    *reinterpret_cast<bf16x4*>(xh + 4 * i4) = hi;
    *reinterpret_cast<bf16x4*>(xl + 4 * i4) = lo;
}
__device__ __forceinline__ void put_split(const float v, __bf16* xh, __bf16* xl, const int i) {
    __bf16 hi, lo;
    split_bf16(v, hi, lo);
    xh[i] = hi;
    xl[i] = lo;
}

// NJ = keys per lane group: key 4 wave + (lane >> 4) + 64 i, i < NJ.  PROJ = false: the query row is given (p.q) and the kernel
// is attention + output projection, for a self-attention block whose q | k | v row one product has just written.
template <int NJ, bool PROJ>
__global__ __launch_bounds__(1024) void cross_attention_row_kernel(const estts_cross_row_args p) {
    __shared__ __attribute__((aligned(16))) __bf16 xh[CR_D], xl[CR_D];   // the row, split
    __shared__ float qp[4][64];                                        // q of this head from the four quarters of x
    __shared__ __attribute__((aligned(16))) float qs[64];
    __shared__ float sc[CR_LK_MAX];                                    // scores, then weights
    __shared__ __attribute__((aligned(16))) float part[16][64];       // partial contexts of the sixteen waves
    __shared__ __attribute__((aligned(16))) __bf16 ch[64], cl[64];     // context of this head, split
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
        put_split4(v, xh, xl, lane);
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
        const float a = dot16<2>(wqh, wql, xh + 64 * qk, xl + 64 * qk, lane);
        if (lane < 16) qp[qk][qn * 16 + lane] = a;
    }
    float yacc = dot16<2>(woh, wol, xh + 64 * h, xl + 64 * h, lane);
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
        put_split(t, ch, cl, tid);
    }
    __syncthreads();

    // ---- this head's share of y = [x | ctx] Wo, to the slab
    yacc += dot16<2>(woh + 2, wol + 2, ch, cl, lane);
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


// ======================================================================================================
// A whole decoder block for one position in one launch (estts_decoder_block_row)
// ======================================================================================================
// Self-attention, cross-attention and the folded feed-forward product of a block are a chain in which every link needs the
// whole row of the link before it, and in an incremental decoder every link is a launch of its own (or, above, one per
// attention): 4 launches of ~6 us each, none of which does more than a microsecond of work.  Here the four workgroups of a row
// (one per head, as above) stay for the whole block and meet twice at a barrier of their own:
//   phase 1  x = the pending normalisation of the block's input (every workgroup computes it), this head's q | k | v (written
//            to the cache row), attention over the cache, this head's share of y_s = [x | ctx] Wo_s            -> slab 0
//   phase 2  x' = LNb(LNa(y_s) + x) (every workgroup), this head's cross-attention, share of y_c = [x' | ctx'] Wo_c -> slab 1
//   phase 3  x2 = LN(y_c + x') (every workgroup), this head's 64 columns of y2 = x2 W_ff + b
// Memory is asked for twice: phase 1's operands in the first instructions of the kernel, the operands of phases 2 and 3 while
// the first barrier is being waited for (phase 1's registers are free by then), and the small vectors (normalisation gains,
// biases, the input row) through LDS at the start.  After that only the slabs are read.
//
// The barrier: counters X (this barrier) and O (the row's other barrier).  Every workgroup adds 1 to X; the one that gets
// ticket 3 sets O to 0 and then X to 5 (release); the others wait for X == 5.  O is not in use at that moment: everyone has left
// it (they are here) and nobody is at it yet (they need X == 5 first).  So the two counters of a row clear each other, launch
// after launch, from zero at allocation.  The wait is bounded: a workgroup that sees no release within ~2^20 polls goes on and
// the row's outputs are set to NaN (four workgroups of an otherwise idle stream are always resident together; the bound is
// there so that no wave can wait for ever).
constexpr int DB_POLLS = 1 << 20;

// build with -DESTTS_BLOCK_STAMPS to have every workgroup leave the 100 MHz clock at its phase boundaries behind the slabs
// (tools/decoder_block_stamps.py reads them)
#ifdef ESTTS_BLOCK_STAMPS
#define DB_STAMP(i) do { if (threadIdx.x == 0) stamps[i] = wall_clock64(); } while (0)
#else
#define DB_STAMP(i) do { } while (0)
#endif

__device__ __forceinline__ void row_barrier_arrive(int* x, int* s_ticket) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (threadIdx.x == 0) {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        *s_ticket = __hip_atomic_fetch_add(x, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
}
__device__ __forceinline__ void row_barrier_wait(int* x, int* other, int* s_ticket, int* s_fail) {
    if (threadIdx.x == 0) {
        if (*s_ticket == CR_H - 1) {
            __hip_atomic_store(other, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(x, CR_H + 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
        } else {
            int n = 0;
            while (__hip_atomic_load(x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != CR_H + 1 && ++n < DB_POLLS) __builtin_amdgcn_s_sleep(2);
            if (n >= DB_POLLS) *s_fail = 1;
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    __syncthreads();
}

// LayerNorm of a 256-wide row held as a float4 per lane of one wave; gains / offsets from LDS
__device__ __forceinline__ void ln256(float4& v, const float* g, const float* b, const int lane, const float eps) {
    const float mean = wave_sum(v.x + v.y + v.z + v.w) / (float)CR_D;
    const float a0 = v.x - mean, a1 = v.y - mean, a2 = v.z - mean, a3 = v.w - mean;
    const float inv = rsqrtf(wave_sum(a0 * a0 + a1 * a1 + a2 * a2 + a3 * a3) / (float)CR_D + eps);
    const float4 gg = *reinterpret_cast<const float4*>(g + 4 * lane), bb = *reinterpret_cast<const float4*>(b + 4 * lane);
    v.x = a0 * inv * gg.x + bb.x; v.y = a1 * inv * gg.y + bb.y; v.z = a2 * inv * gg.z + bb.z; v.w = a3 * inv * gg.w + bb.w;
}

// One head's attention of one query (qs: 64 floats in LDS) over n_reg keys / values held in registers (key 4 wave + (lane >> 4)
// + 64 i at [i]; 16 lanes x float4 = the head's 64 columns) plus, when knew is given, one more key / value pair in LDS (the
// position's own); weights to sc (and attn_out), context to ch / cl (split).  All 1024 threads.
template <int NJ>
__device__ __forceinline__ void head_attention(const float4 (&kr)[NJ], const float4 (&vr)[NJ], const float (&mk)[NJ], const int n_reg,
                                               const float* qs, const float* knew, const float* vnew, const float mknew,
                                               float* sc, float (*part)[64], __bf16* ch, __bf16* cl, float* attn_out) {
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int jq = 4 * wave + (lane >> 4);
    const int L = n_reg + (knew ? 1 : 0);
    {
        const float4 qv = *reinterpret_cast<const float4*>(qs + 4 * (lane & 15));
#pragma unroll
        for (int i = 0; i < NJ; ++i) {
            const int j = jq + 64 * i;
            const float d = row16_sum(kr[i].x * qv.x + kr[i].y * qv.y + kr[i].z * qv.z + kr[i].w * qv.w);
            if ((lane & 15) == 0 && j < n_reg) sc[j] = d * 0.125f + mk[i] * -1e9f;   // 1 / sqrt(64)
        }
        if (knew && wave == 0) {
            const float4 kv = *reinterpret_cast<const float4*>(knew + 4 * (lane & 15));
            const float d = row16_sum(kv.x * qv.x + kv.y * qv.y + kv.z * qv.z + kv.w * qv.w);
            if (lane == 0) sc[n_reg] = d * 0.125f + mknew * -1e9f;
        }
    }
    __syncthreads();
    if (wave == 0) {
        float m = -INFINITY;
        for (int j = lane; j < L; j += 64) m = fmaxf(m, sc[j]);
        m = wave_max(m);
        float l = 0.0f;
        for (int j = lane; j < L; j += 64) {
            const float e = __builtin_amdgcn_exp2f((sc[j] - m) * 1.4426950408889634f);
            sc[j] = e;
            l += e;
        }
        const float inv_l = 1.0f / wave_sum(l);
        for (int j = lane; j < L; j += 64) {
            const float w = sc[j] * inv_l;
            sc[j] = w;
            if (attn_out) attn_out[j] = w;
        }
    }
    __syncthreads();
    {
        float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
        for (int i = 0; i < NJ; ++i) {
            const int j = jq + 64 * i;
            if (j < n_reg) {      // (a row past the end is whatever the clamped address held: never multiplied)
                const float w = sc[j];
                acc.x += w * vr[i].x; acc.y += w * vr[i].y; acc.z += w * vr[i].z; acc.w += w * vr[i].w;
            }
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
        if (vnew) t += sc[n_reg] * vnew[tid];
        put_split(t, ch, cl, tid);
    }
    __syncthreads();
}

// rows of the LDS image of the small vectors (256 floats each)
enum { V_LNY, V_LNX, V_GA, V_BA, V_GB, V_BB, V_SGA, V_SBA, V_SGB, V_SBB, V_CG, V_CB, V_BOS, V_BOC, V_BQKV, V_BQF, V_ROWS };

// NJ1: cache rows per lane group (pos <= 64 NJ1), NJ2: encoder rows per lane group (Lk <= 64 NJ2)
template <int NJ1, int NJ2>
__global__ __launch_bounds__(1024) void decoder_block_row_kernel(const estts_decoder_block_args p) {
    __shared__ __attribute__((aligned(16))) float vec[V_ROWS][CR_D];
    __shared__ __attribute__((aligned(16))) float xs[CR_D], xs2[CR_D], ys[CR_D];      // x, x', a summed y
    __shared__ __attribute__((aligned(16))) __bf16 xh[CR_D], xl[CR_D];                // the row that the phase's products read, split
    __shared__ float qp[4][192];
    __shared__ __attribute__((aligned(16))) float qkvs[192];           // q | k | v of this head
    __shared__ float sc[CR_LK_MAX + 1];
    __shared__ __attribute__((aligned(16))) float part[16][64];
    __shared__ __attribute__((aligned(16))) __bf16 ch[64], cl[64];
    __shared__ int s_ticket, s_fail;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int h = blockIdx.x, row = blockIdx.y;
    const int pos = p.pos;
    const int qn = wave & 3, qk = wave >> 2;      // a product of 64 outputs: block row qn of the head, quarter qk of the inputs
    const int jq = 4 * wave + (lane >> 4), col = 64 * h + 4 * (lane & 15);
    if (tid == 0) s_fail = 0;
#ifdef ESTTS_BLOCK_STAMPS
    unsigned long long* stamps = reinterpret_cast<unsigned long long*>(p.ws + (long)gridDim.y * 2 * CR_H * CR_D) + (blockIdx.y * CR_H + blockIdx.x) * 16;
#endif
    DB_STAMP(0);

    // ---- phase 1's operands, and the small vectors of all phases
    // q | k | v of this head: wave w multiplies block rows (w >> 2) of the head's q, k and v panels by quarter (w & 3) of x
    uint4 wah[6], wal[6];
    {
        const int nbh = wave >> 2, quarter = wave & 3;
#pragma unroll
        for (int e = 0; e < 3; ++e) {
            const long base = ((long)(16 * e + 4 * h + nbh) * (p.ldwqkv >> 5) + 2 * quarter) * 512 + lane * 8;
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                wah[2 * e + i] = *reinterpret_cast<const uint4*>(p.wqkv_hi + base + i * 512);
                wal[2 * e + i] = *reinterpret_cast<const uint4*>(p.wqkv_lo + base + i * 512);
            }
        }
    }
    uint4 woh[4], wol[4];      // block row `wave` of Wo_s: this head's 64 columns of x and of ctx
    {
        const long base = (long)wave * (p.ldwos >> 5) * 512 + lane * 8;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int kb = (i >> 1) * (CR_D / 32) + 2 * h + (i & 1);
            woh[i] = *reinterpret_cast<const uint4*>(p.wos_hi + base + kb * 512);
            wol[i] = *reinterpret_cast<const uint4*>(p.wos_lo + base + kb * 512);
        }
    }
    float* cache = p.qkv + (long)row * p.qkv_batch_stride;
    const float* smask = p.self_mask ? p.self_mask + (long)row * p.self_mask_stride : nullptr;
    float4 kr[NJ1], vr[NJ1];
    float mk[NJ1];
#pragma unroll
    for (int i = 0; i < NJ1; ++i) {
        const int j = min(jq + 64 * i, max(pos - 1, 0));
        kr[i] = ld4g(cache + (long)j * p.ldqkv + CR_D + col);
        vr[i] = ld4g(cache + (long)j * p.ldqkv + 2 * CR_D + col);
        mk[i] = smask ? smask[j] : 0.0f;
    }
    const float mknew = smask ? smask[pos] : 0.0f;
    {
        // wave w brings row w of `vec`: a float4 per lane
        const float* src = nullptr;
        int n4 = 64;
        switch (wave) {
            case V_LNY: src = p.ln_y ? p.ln_y + (long)row * p.ln_ldy : p.x + (long)row * p.ldx; break;
            case V_LNX: src = p.ln_x ? p.ln_x + (long)row * p.ln_ldx : nullptr; break;
            case V_GA: src = p.ln_ga; break;
            case V_BA: src = p.ln_ba; break;
            case V_GB: src = p.ln_gb; break;
            case V_BB: src = p.ln_bb; break;
            case V_SGA: src = p.sa_ga; break;
            case V_SBA: src = p.sa_ba; break;
            case V_SGB: src = p.sa_gb; break;
            case V_SBB: src = p.sa_bb; break;
            case V_CG: src = p.ca_g; break;
            case V_CB: src = p.ca_b; break;
            case V_BOS: src = p.bos; break;
            case V_BOC: src = p.boc; break;
            default: break;
        }
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (wave == V_BQKV) {            // this head's 3 x 64 biases of q | k | v
            if (p.bqkv && lane < 48) v = ld4g(p.bqkv + (lane >> 4) * CR_D + 64 * h + 4 * (lane & 15));
        } else if (wave == V_BQF) {      // ... its 64 of the cross-attention query and its 64 of the feed-forward product
            if (lane < 16) { if (p.bqc) v = ld4g(p.bqc + 64 * h + 4 * lane); }
            else if (lane < 32) { if (p.bff) v = ld4g(p.bff + 64 * h + 4 * (lane - 16)); }
        } else if (src) {
            v = ld4g(src + 4 * lane);
        }
        (void)n4;
        *reinterpret_cast<float4*>(&vec[wave][4 * lane]) = v;
    }
    __syncthreads();
    DB_STAMP(1);

    // ---- phase 1: the block's input row
    if (wave == 0) {
        float4 v = *reinterpret_cast<const float4*>(&vec[V_LNY][4 * lane]);
        if (p.ln_y) {
            const float4 xr = *reinterpret_cast<const float4*>(&vec[V_LNX][4 * lane]);
            if (p.ln_ga) ln256(v, vec[V_GA], vec[V_BA], lane, p.ln_eps);
            v.x = apply_act(v.x, p.ln_act) + xr.x; v.y = apply_act(v.y, p.ln_act) + xr.y;
            v.z = apply_act(v.z, p.ln_act) + xr.z; v.w = apply_act(v.w, p.ln_act) + xr.w;
            if (p.ln_gb) ln256(v, vec[V_GB], vec[V_BB], lane, p.ln_eps);
        }
        *reinterpret_cast<float4*>(xs + 4 * lane) = v;
        put_split4(v, xh, xl, lane);
    }
    __syncthreads();
    // q | k | v of this head and the x half of its share of y_s
#pragma unroll
    for (int e = 0; e < 3; ++e) {
        const float a = dot16<2>(wah + 2 * e, wal + 2 * e, xh + 64 * (wave & 3), xl + 64 * (wave & 3), lane);
        if (lane < 16) qp[wave & 3][64 * e + 16 * (wave >> 2) + lane] = a;
    }
    float yacc = dot16<2>(woh, wol, xh + 64 * h, xl + 64 * h, lane);
    __syncthreads();
    if (tid < 192) {
        const float v = ((qp[0][tid] + qp[1][tid]) + (qp[2][tid] + qp[3][tid])) + vec[V_BQKV][tid];
        qkvs[tid] = v;
        cache[(long)pos * p.ldqkv + (tid >> 6) * CR_D + 64 * h + (tid & 63)] = v;     // the position's row of the cache
    }
    __syncthreads();
    DB_STAMP(2);
    head_attention<NJ1>(kr, vr, mk, pos, qkvs, qkvs + 64, qkvs + 128, mknew, sc, part, ch, cl, nullptr);
    DB_STAMP(3);
    yacc += dot16<2>(woh + 2, wol + 2, ch, cl, lane);
    float* slab = p.ws + (long)row * 2 * CR_H * CR_D;
    if (lane < 16) slab[h * CR_D + wave * 16 + lane] = yacc;
    int* bar = p.counters + 2 * row;
    row_barrier_arrive(bar, &s_ticket);
    DB_STAMP(4);

    // ---- the operands of phases 2 and 3, on their way while the other heads arrive
    uint4 wqh[2], wql[2];      // cross-attention query: block row qn of the head, quarter qk of x'
    {
        const long base = ((long)(4 * h + qn) * (p.ldwqc >> 5) + 2 * qk) * 512 + lane * 8;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            wqh[i] = *reinterpret_cast<const uint4*>(p.wqc_hi + base + i * 512);
            wql[i] = *reinterpret_cast<const uint4*>(p.wqc_lo + base + i * 512);
        }
    }
    uint4 wch[4], wcl[4];      // block row `wave` of Wo_c
    {
        const long base = (long)wave * (p.ldwoc >> 5) * 512 + lane * 8;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int kb = (i >> 1) * (CR_D / 32) + 2 * h + (i & 1);
            wch[i] = *reinterpret_cast<const uint4*>(p.woc_hi + base + kb * 512);
            wcl[i] = *reinterpret_cast<const uint4*>(p.woc_lo + base + kb * 512);
        }
    }
    uint4 wfh[2], wfl[2];      // feed-forward product: block row qn of the head's 64 outputs, quarter qk of x2
    {
        const long base = ((long)(4 * h + qn) * (p.ldwff >> 5) + 2 * qk) * 512 + lane * 8;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            wfh[i] = *reinterpret_cast<const uint4*>(p.wff_hi + base + i * 512);
            wfl[i] = *reinterpret_cast<const uint4*>(p.wff_lo + base + i * 512);
        }
    }
    const int Lk = p.Lk;
    const float* ek = p.k + (long)row * p.kv_batch_stride + col;
    const float* ev = p.v + (long)row * p.kv_batch_stride + col;
    const float* emask = p.key_mask ? p.key_mask + (long)row * Lk : nullptr;
    float4 er[NJ2], fr[NJ2];
    float mk2[NJ2];
#pragma unroll
    for (int i = 0; i < NJ2; ++i) {
        const int j = min(jq + 64 * i, Lk - 1);
        er[i] = ld4g(ek + (long)j * p.ldk);
        fr[i] = ld4g(ev + (long)j * p.ldv);
        mk2[i] = emask ? emask[j] : 0.0f;
    }
    row_barrier_wait(bar, bar + 1, &s_ticket, &s_fail);
    DB_STAMP(5);

    // ---- phase 2: y_s, x' = LNb(LNa(y_s) + x), cross-attention of this head, its share of y_c
    if (tid < CR_D) ys[tid] = (((slab[tid] + slab[CR_D + tid]) + slab[2 * CR_D + tid]) + slab[3 * CR_D + tid]) + vec[V_BOS][tid];
    __syncthreads();
    DB_STAMP(6);
    if (wave == 0) {
        float4 v = *reinterpret_cast<const float4*>(ys + 4 * lane);
        const float4 xr = *reinterpret_cast<const float4*>(xs + 4 * lane);
        ln256(v, vec[V_SGA], vec[V_SBA], lane, p.ln_eps);
        v.x += xr.x; v.y += xr.y; v.z += xr.z; v.w += xr.w;
        ln256(v, vec[V_SGB], vec[V_SBB], lane, p.ln_eps);
        *reinterpret_cast<float4*>(xs2 + 4 * lane) = v;
        put_split4(v, xh, xl, lane);
    }
    __syncthreads();
    {
        const float a = dot16<2>(wqh, wql, xh + 64 * qk, xl + 64 * qk, lane);
        if (lane < 16) qp[qk][qn * 16 + lane] = a;
    }
    yacc = dot16<2>(wch, wcl, xh + 64 * h, xl + 64 * h, lane);
    __syncthreads();
    if (tid < 64) qkvs[tid] = ((qp[0][tid] + qp[1][tid]) + (qp[2][tid] + qp[3][tid])) + vec[V_BQF][tid];
    __syncthreads();
    DB_STAMP(7);
    head_attention<NJ2>(er, fr, mk2, Lk, qkvs, nullptr, nullptr, 0.0f, sc, part, ch, cl,
                        p.attn ? p.attn + (long)row * p.attn_row_stride + (long)h * Lk : nullptr);
    yacc += dot16<2>(wch + 2, wcl + 2, ch, cl, lane);
    slab += CR_H * CR_D;
    if (lane < 16) slab[h * CR_D + wave * 16 + lane] = yacc;
    DB_STAMP(8);
    row_barrier_arrive(bar + 1, &s_ticket);
    DB_STAMP(9);
    row_barrier_wait(bar + 1, bar, &s_ticket, &s_fail);
    DB_STAMP(10);

    // ---- phase 3: y_c, x2 = LN(y_c + x'), this head's 64 columns of y2 = x2 W_ff + b
    if (tid < CR_D) ys[tid] = (((slab[tid] + slab[CR_D + tid]) + slab[2 * CR_D + tid]) + slab[3 * CR_D + tid]) + vec[V_BOC][tid];
    __syncthreads();
    if (wave == 0) {
        float4 v = *reinterpret_cast<const float4*>(ys + 4 * lane);
        const float4 xr = *reinterpret_cast<const float4*>(xs2 + 4 * lane);
        v.x += xr.x; v.y += xr.y; v.z += xr.z; v.w += xr.w;
        ln256(v, vec[V_CG], vec[V_CB], lane, p.ln_eps);
        put_split4(v, xh, xl, lane);
        if (h == 0) *reinterpret_cast<float4*>(p.x2 + (long)row * p.ldx2 + 4 * lane) = v;
    }
    __syncthreads();
    {
        const float a = dot16<2>(wfh, wfl, xh + 64 * qk, xl + 64 * qk, lane);
        if (lane < 16) qp[qk][qn * 16 + lane] = a;
    }
    __syncthreads();
    if (tid < 64) {
        const float v = ((qp[0][tid] + qp[1][tid]) + (qp[2][tid] + qp[3][tid])) + vec[V_BQF][64 + tid];
        p.y2[(long)row * p.ldy2 + 64 * h + tid] = s_fail ? __builtin_nanf("") : v;
    }
    DB_STAMP(11);
}


// ======================================================================================================
// The decoder's pre-net for one position in one launch (estts_prenet_row)
// ======================================================================================================
// mask = (the frame is all zeros);  p1 = drop(relu(frame W1 + b1));  out = alpha drop(relu(p1 W2 + b2)) + R
// (TransformerTTS/model/layers.py:394-402, the padding mask of model/transformer_utils.py:47-50, and the positional encoding
// added by the caller in AutoregressiveTransformer._call_decoder).  Three launches of one row each in an incremental decoder.
// Four workgroups per row; each computes all 256 units of the first layer (82 KB of weights at 80 mel channels) and its own 64
// columns of the second (64 KB), so they share nothing and need no barrier.  Operands requested up front, as above.
__global__ __launch_bounds__(1024) void prenet_row_kernel(const estts_prenet_row_args p) {
    __shared__ __attribute__((aligned(16))) float fr[128], b1s[CR_D], b2s[64], rs[64];
    __shared__ __attribute__((aligned(16))) __bf16 xh[128], xl[128], ph[CR_D], pl[CR_D];
    __shared__ float qp[4][64];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int s = blockIdx.x, row = blockIdx.y;
    const int nkb1 = p.ldw1 >> 5;      // <= 4
    uint4 w1h[4], w1l[4];              // block row `wave` of W1
#pragma unroll
    for (int kb = 0; kb < 4; ++kb) {
        w1h[kb] = w1l[kb] = make_uint4(0u, 0u, 0u, 0u);
        if (kb < nkb1) {
            const long base = ((long)wave * nkb1 + kb) * 512 + lane * 8;
            w1h[kb] = *reinterpret_cast<const uint4*>(p.w1_hi + base);
            w1l[kb] = *reinterpret_cast<const uint4*>(p.w1_lo + base);
        }
    }
    const int qn = wave & 3, qk = wave >> 2;
    uint4 w2h[2], w2l[2];              // W2: block row qn of this workgroup's 64 columns, quarter qk of p1
    {
        const long base = ((long)(4 * s + qn) * (p.ldw2 >> 5) + 2 * qk) * 512 + lane * 8;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            w2h[i] = *reinterpret_cast<const uint4*>(p.w2_hi + base + i * 512);
            w2l[i] = *reinterpret_cast<const uint4*>(p.w2_lo + base + i * 512);
        }
    }
    if (wave == 0) {
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (lane < 32 && 4 * lane < p.C) v = ld4g(p.x + (long)row * p.ldx + 4 * lane);
        if (lane < 32) *reinterpret_cast<float4*>(fr + 4 * lane) = v;
    } else if (wave == 1) {
        *reinterpret_cast<float4*>(b1s + 4 * lane) = p.b1 ? ld4g(p.b1 + 4 * lane) : make_float4(0.f, 0.f, 0.f, 0.f);
    } else if (wave == 2) {
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (lane < 16) { if (p.b2) v = ld4g(p.b2 + 64 * s + 4 * lane); *reinterpret_cast<float4*>(b2s + 4 * lane) = v; }
        else if (lane < 32) { if (p.R) v = ld4g(p.R + (long)row * p.ldr + 64 * s + 4 * (lane - 16)); *reinterpret_cast<float4*>(rs + 4 * (lane - 16)) = v; }
    }
    __syncthreads();
    if (wave == 0) {
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (lane < 32) v = *reinterpret_cast<const float4*>(fr + 4 * lane);
        const float a = wave_sum(fabsf(v.x) + fabsf(v.y) + fabsf(v.z) + fabsf(v.w));
        if (p.mask && s == 0 && lane == 0) p.mask[row] = a == 0.0f ? 1.0f : 0.0f;
        if (lane < 32) put_split4(v, xh, xl, lane);
    }
    __syncthreads();
    {
        const float a = dot16<4>(w1h, w1l, xh, xl, lane);
        if (lane < 16) {
            const int col = 16 * wave + lane;
            float v = fmaxf(a + b1s[col], 0.0f);
            if (p.drop_rate > 0.0f) v = dropout_keep(p.drop_seed, p.drop_site1, (uint32_t)(p.drop_row0 + row), (uint32_t)col, p.drop_rate) ? v / (1.0f - p.drop_rate) : 0.0f;
            put_split(v, ph, pl, col);
        }
    }
    __syncthreads();
    {
        const float a = dot16<2>(w2h, w2l, ph + 64 * qk, pl + 64 * qk, lane);
        if (lane < 16) qp[qk][16 * qn + lane] = a;
    }
    __syncthreads();
    if (tid < 64) {
        const int col = 64 * s + tid;
        float v = fmaxf(((qp[0][tid] + qp[1][tid]) + (qp[2][tid] + qp[3][tid])) + b2s[tid], 0.0f);
        if (p.drop_rate > 0.0f) v = dropout_keep(p.drop_seed, p.drop_site2, (uint32_t)(p.drop_row0 + row), (uint32_t)col, p.drop_rate) ? v / (1.0f - p.drop_rate) : 0.0f;
        p.out[(long)row * p.ldo + col] = v * p.alpha + rs[tid];
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

extern "C" int estts_decoder_block_row(const estts_decoder_block_args* args, void* stream) {
    ESTTS_CHECK_ARG(args != nullptr, "estts_decoder_block_row: null args");
    const estts_decoder_block_args& a = *args;
    ESTTS_CHECK_ARG(a.M > 0 && a.M <= 64 && a.d == CR_D && a.H == CR_H,
                    "estts_decoder_block_row: this kernel takes 1 .. 64 rows of width 256 with 4 heads of depth 64 (got %d rows, width %d, %d heads)", a.M, a.d, a.H);
    ESTTS_CHECK_ARG(a.Lk > 0 && a.Lk <= CR_LK_MAX && a.pos >= 0 && a.pos < CR_LK_MAX,
                    "estts_decoder_block_row: 1 .. %d encoder positions and a decoder position below %d (got %d, %d)", CR_LK_MAX, CR_LK_MAX, a.Lk, a.pos);
    ESTTS_CHECK_ARG((a.ln_y || a.x) && a.wqkv_hi && a.wqkv_lo && a.qkv && a.wos_hi && a.wos_lo && a.sa_ga && a.sa_ba && a.sa_gb && a.sa_bb &&
                    a.wqc_hi && a.wqc_lo && a.k && a.v && a.woc_hi && a.woc_lo && a.ca_g && a.ca_b && a.wff_hi && a.wff_lo && a.y2 && a.x2,
                    "estts_decoder_block_row: null operand");
    ESTTS_CHECK_ARG(!a.ln_y || ((!a.ln_ga == !a.ln_ba) && (!a.ln_gb == !a.ln_bb) && (a.ln_ga || a.ln_gb) && (a.ln_ga || a.ln_act == ESTTS_ACT_NONE)),
                    "estts_decoder_block_row: the pending normalisation needs at least one of (ln_ga, ln_ba) / (ln_gb, ln_bb); ln_act only with the first");
    ESTTS_CHECK_ARG(a.ldwqkv >= CR_D && (a.ldwqkv & 63) == 0 && a.ldwqc >= CR_D && (a.ldwqc & 63) == 0 && a.ldwff >= CR_D && (a.ldwff & 63) == 0 &&
                    a.ldwos >= 2 * CR_D && (a.ldwos & 63) == 0 && a.ldwoc >= 2 * CR_D && (a.ldwoc & 63) == 0,
                    "estts_decoder_block_row: ldw must be K rounded up to 64");
    const uintptr_t ptrs = (uintptr_t)a.x | (uintptr_t)a.ln_y | (uintptr_t)a.ln_x | (uintptr_t)a.ln_ga | (uintptr_t)a.ln_ba | (uintptr_t)a.ln_gb |
                           (uintptr_t)a.ln_bb | (uintptr_t)a.wqkv_hi | (uintptr_t)a.wqkv_lo | (uintptr_t)a.bqkv | (uintptr_t)a.qkv | (uintptr_t)a.wos_hi |
                           (uintptr_t)a.wos_lo | (uintptr_t)a.bos | (uintptr_t)a.sa_ga | (uintptr_t)a.sa_ba | (uintptr_t)a.sa_gb | (uintptr_t)a.sa_bb |
                           (uintptr_t)a.wqc_hi | (uintptr_t)a.wqc_lo | (uintptr_t)a.bqc | (uintptr_t)a.k | (uintptr_t)a.v | (uintptr_t)a.woc_hi |
                           (uintptr_t)a.woc_lo | (uintptr_t)a.boc | (uintptr_t)a.ca_g | (uintptr_t)a.ca_b | (uintptr_t)a.wff_hi | (uintptr_t)a.wff_lo |
                           (uintptr_t)a.bff | (uintptr_t)a.y2 | (uintptr_t)a.x2 | (uintptr_t)a.ws;
    ESTTS_CHECK_ARG((ptrs & 15) == 0 && ((a.ldx | a.ln_ldy | a.ln_ldx | a.ldqkv | a.ldk | a.ldv | a.ldy2 | a.ldx2) & 3) == 0 && (a.qkv_batch_stride & 3) == 0 &&
                    (a.kv_batch_stride & 3) == 0 && a.ldqkv >= 3 * CR_D && a.ldk >= CR_D && a.ldv >= CR_D,
                    "estts_decoder_block_row: operands must be 16-byte aligned with row strides that are multiples of 4");
    ESTTS_CHECK_ARG(a.ws && a.counters && a.ws_bytes >= (int64_t)a.M * 2 * CR_H * CR_D * 4 && a.n_counters >= 2 * a.M,
                    "estts_decoder_block_row: needs a slab of M x 2 x 4 x 256 floats and 2 M counters of its own (zero at allocation)");
    ESTTS_CHECK_ARG(a.y2 != a.ln_y && a.y2 != a.ln_x && a.x2 != a.ln_y && a.x2 != a.ln_x && a.y2 != a.x && a.x2 != a.x,
                    "estts_decoder_block_row: y2 / x2 must not alias the input rows (the four workgroups of a row read them at different times)");
    const dim3 grid(CR_H, a.M);
    hipStream_t st = (hipStream_t)stream;
    const bool one = a.pos <= 64, enc2 = a.Lk <= 128;
    if (one && enc2) ESTTS_LAUNCH((decoder_block_row_kernel<1, 2>), grid, dim3(1024), 0, st, a);
    else if (one) ESTTS_LAUNCH((decoder_block_row_kernel<1, 4>), grid, dim3(1024), 0, st, a);
    else if (a.pos <= 128 && enc2) ESTTS_LAUNCH((decoder_block_row_kernel<2, 2>), grid, dim3(1024), 0, st, a);
    else if (a.pos <= 128) ESTTS_LAUNCH((decoder_block_row_kernel<2, 4>), grid, dim3(1024), 0, st, a);
    else if (enc2) ESTTS_LAUNCH((decoder_block_row_kernel<4, 2>), grid, dim3(1024), 0, st, a);
    else ESTTS_LAUNCH((decoder_block_row_kernel<4, 4>), grid, dim3(1024), 0, st, a);
    ESTTS_CHECK_LAUNCH("estts_decoder_block_row");
    return 0;
}

extern "C" int estts_prenet_row(const estts_prenet_row_args* args, void* stream) {
    ESTTS_CHECK_ARG(args != nullptr, "estts_prenet_row: null args");
    const estts_prenet_row_args& a = *args;
    ESTTS_CHECK_ARG(a.M > 0 && a.M <= 65535 && a.C > 0 && a.C <= 128 && (a.C & 3) == 0 && a.N1 == CR_D && a.N2 == CR_D,
                    "estts_prenet_row: this kernel takes frames of at most 128 channels (a multiple of 4) and two layers of 256 units (got %d, %d, %d)", a.C, a.N1, a.N2);
    ESTTS_CHECK_ARG(a.x && a.w1_hi && a.w1_lo && a.w2_hi && a.w2_lo && a.out, "estts_prenet_row: null operand");
    ESTTS_CHECK_ARG(a.ldw1 >= a.C && (a.ldw1 & 63) == 0 && a.ldw1 <= 128 && a.ldw2 >= CR_D && (a.ldw2 & 63) == 0, "estts_prenet_row: ldw must be K rounded up to 64");
    ESTTS_CHECK_ARG((((uintptr_t)a.x | (uintptr_t)a.w1_hi | (uintptr_t)a.w1_lo | (uintptr_t)a.w2_hi | (uintptr_t)a.w2_lo | (uintptr_t)a.b1 | (uintptr_t)a.b2 |
                      (uintptr_t)a.R) & 15) == 0 && ((a.ldx | a.ldr) & 3) == 0 && a.ldx >= a.C && a.ldo >= CR_D && (!a.R || a.ldr >= CR_D),
                    "estts_prenet_row: operands must be 16-byte aligned with row strides that are multiples of 4");
    ESTTS_CHECK_ARG(a.drop_rate >= 0.0f && a.drop_rate < 1.0f, "estts_prenet_row: dropout rate must be in [0, 1)");
    ESTTS_LAUNCH(prenet_row_kernel, dim3(4, a.M), dim3(1024), 0, (hipStream_t)stream, a);
    ESTTS_CHECK_LAUNCH("estts_prenet_row");
    return 0;
}
