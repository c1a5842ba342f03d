// Multi-head scaled-dot-product attention (see estts_attention in include/estts.h).
//
// A 256-thread workgroup owns 16 query rows of one (batch, head); its four waves take interleaved 32-key
// tiles and keep private online-softmax state (running max m, sum l, unnormalised output), which is merged
// through LDS at the end.  Both products are issued "swapped" so that a query stays on one lane column:
//   S^T (16 keys x 16 queries) = K_tile . Q^T     A = K rows (key on lane & 15),      B = Q^T
//   O^T (16 feats x 16 queries) = V^T . P^T       A = V^T (feature on lane & 15),     B = P^T
// With v_mfma_f32_16x16x32_bf16 the D layout is D[row 4(l>>4)+j][col l&15], so after S^T every lane holds,
// for its query (l & 15), the scores of keys 4g+j (g = l>>4) of each 16-key sub-tile; two sub-tiles give the
// 8 k-slots (8g+j') of the P^T operand without data movement: slot j' <-> key 16 (j'>>2) + 4g + (j'&3).
// K and V tiles are read from global memory with coalesced float4 loads (one tile ahead of their use), split
// into bf16 hi (+ lo) and staged in a wave-private LDS image [32 keys][depth + 8]: K fragments come back with
// ds_read_b128, V^T fragments with the transposing ds_read_b64_tr_b16 (4 keys x 16 features per 16-lane group,
// delivered feature-per-lane), in the key order of the P^T slots above.
// The optional attention-weight output keeps the scaled scores of up to 8 key tiles per wave (Lk <= 1024) in registers
// and writes exp(s - m) / l after the merge; longer key ranges recompute S in a second sweep.
#include "common.h"

namespace {

typedef __attribute__((ext_vector_type(4))) short s16x4;
typedef __attribute__((ext_vector_type(8))) short s16x8;

// exp() of softmax terms (arguments <= 0, -inf included): v_exp_f32 of x log2(e).  Rounding the product costs |x| 2^-24
// relative (1e-6 at a weight of 1e-9); the range reduction and fix-ups of expf() were a third of the vector
// instructions of a key tile.
__device__ __forceinline__ float exp_neg(float x) { return __builtin_amdgcn_exp2f(x * 1.4426950408889634f); }

template <int NS>
struct Frag {
    bf16x8 v[NS];
};

template <int NS>
__device__ __forceinline__ Frag<NS> make_frag(const float* x) {
    // conversions in pairs (v_cvt_pk_bf16_f32); element by element they were a quarter of a tile's vector instructions
    bf16x4 h0, l0, h1, l1;
    split_bf16x4(make_float4(x[0], x[1], x[2], x[3]), h0, l0);
    split_bf16x4(make_float4(x[4], x[5], x[6], x[7]), h1, l1);
    Frag<NS> f;
    f.v[0] = __builtin_shufflevector(h0, h1, 0, 1, 2, 3, 4, 5, 6, 7);
    if (NS == 2) f.v[NS - 1] = __builtin_shufflevector(l0, l1, 0, 1, 2, 3, 4, 5, 6, 7);
    return f;
}

template <int NS>
__device__ __forceinline__ f32x4 mfma_split(const Frag<NS>& a, const Frag<NS>& b, f32x4 c) {
    if (NS == 2) {
        c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a.v[NS - 1], b.v[0], c, 0, 0, 0);
        c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a.v[0], b.v[NS - 1], c, 0, 0, 0);
    }
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a.v[0], b.v[0], c, 0, 0, 0);
}

__device__ __forceinline__ bf16x8 tr_read_pair(const __bf16* p0, const __bf16* p1) {
    typedef s16x4 __attribute__((address_space(3))) * lds_ptr;
    const s16x4 a = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_ptr)p0);
    const s16x4 b = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_ptr)p1);
    const s16x8 r = {a[0], a[1], a[2], a[3], b[0], b[1], b[2], b[3]};
    return __builtin_bit_cast(bf16x8, r);
}

// MAXT > 0: the attention weights are wanted and every wave sweeps at most MAXT key tiles (Lk <= 128 MAXT): the scaled
// scores stay in registers and the weights are written after the merge, without a second sweep over K.
template <int DH, int NS, int MAXT, int NW>
__global__ __launch_bounds__(NW * 64) void attention_kernel(const estts_attention_args p) {
    constexpr int KS = DH / 32;        // k-steps of the score product
    constexpr int NT = DH / 16;        // 16-feature tiles of the output
    constexpr int LD = DH + 8;         // staged row length in bf16 (row stride = 2 DH + 16 bytes)
    constexpr int LPR = DH / 4;        // lanes per tile row (one float4 each)
    constexpr int RPI = 64 / LPR;      // tile rows per load instruction
    constexpr int NLD = 32 / RPI;      // load instructions per 32-key tile
    constexpr int IMG = 32 * LD;       // one staged image (hi or lo)
    constexpr int STAGE_BYTES = NW * NS * IMG * 2;
    constexpr int MERGE_FLOATS = NT * 256 + 128;  // per wave: O^T fragments + m + l
    constexpr int MERGE_BYTES = NW * MERGE_FLOATS * 4;
    __shared__ __attribute__((aligned(16))) unsigned char smem[STAGE_BYTES > MERGE_BYTES ? STAGE_BYTES : MERGE_BYTES];

    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    const int i = lane & 15;
    const int g = lane >> 4;
    const int h = blockIdx.y;
    const int b = blockIdx.z;
    const int q0 = blockIdx.x * 16;
    const int qi = q0 + i;
    const int qrow = min(qi, p.Lq - 1);
    const float* qp = p.q + ((long)b * p.Lq + qrow) * p.ldq + h * DH + 8 * g;
    const long kvb = p.kv_shared ? 0 : b;
    const float* kbase = p.k + kvb * p.Lk * p.ldk + h * DH;
    const float* vbase = p.v + kvb * p.Lk * p.ldv + h * DH;
    const float* mask = p.key_mask ? p.key_mask + (long)b * p.Lk : nullptr;
    const bool mask_quads = (p.Lk & 3) == 0 && (reinterpret_cast<uintptr_t>(p.key_mask) & 15) == 0;   // mask rows as float4
    const float scale = 1.0f / sqrtf((float)DH);
    __bf16* stage = reinterpret_cast<__bf16*>(smem) + wave * (NS * IMG);

    Frag<NS> qf[KS];
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
        float t[8];
        *reinterpret_cast<float4*>(t) = *reinterpret_cast<const float4*>(qp + ks * 32);
        *reinterpret_cast<float4*>(t + 4) = *reinterpret_cast<const float4*>(qp + ks * 32 + 4);
        qf[ks] = make_frag<NS>(t);
    }

    const int ld_row = lane / LPR;
    const int ld_col = (lane % LPR) * 4;

    auto load_tile = [&](const float* base, int ld, int kt, float4 (&x)[NLD]) {
#pragma unroll
        for (int j = 0; j < NLD; ++j) {
            const int key = min(kt * 32 + j * RPI + ld_row, p.Lk - 1);
            x[j] = *reinterpret_cast<const float4*>(base + (long)key * ld + ld_col);
        }
    };

    auto stage_tile = [&](const float4 (&x)[NLD]) {
#pragma unroll
        for (int j = 0; j < NLD; ++j) {
            bf16x4 hi, lo;
            split_bf16x4(x[j], hi, lo);
            const int o = (j * RPI + ld_row) * LD + ld_col;
            *reinterpret_cast<bf16x4*>(stage + o) = hi;
            if (NS == 2) *reinterpret_cast<bf16x4*>(stage + IMG + o) = lo;
        }
    };

    // scores of the staged K tile for this lane's query: s[sub][j] <-> key kt*32 + sub*16 + 4g + j
    auto scores = [&](int kt, float (&s)[2][4]) {
        // key-mask values of this lane's eight keys, loaded up front and without conditions (a load under a per-element
        // condition becomes a branch with its own wait: eight serialised round trips per tile)
        // A tile inside the key range and, under a causal mask, before the workgroup's first query needs neither the range
        // check nor the comparison with the query, and its mask values come as two float4 (uniform over the workgroup).
        const bool inside = kt * 32 + 32 <= p.Lk && mask_quads;
        const bool plain = inside && !(p.causal && kt * 32 + 31 > (int)blockIdx.x * 16);
        float mk[2][4];
        if (inside) {
            float4 m0 = make_float4(0.f, 0.f, 0.f, 0.f), m1 = m0;
            if (mask) {
                m0 = *reinterpret_cast<const float4*>(mask + kt * 32 + 4 * g);
                m1 = *reinterpret_cast<const float4*>(mask + kt * 32 + 16 + 4 * g);
            }
            mk[0][0] = m0.x, mk[0][1] = m0.y, mk[0][2] = m0.z, mk[0][3] = m0.w;
            mk[1][0] = m1.x, mk[1][1] = m1.y, mk[1][2] = m1.z, mk[1][3] = m1.w;
        } else {
#pragma unroll
            for (int sub = 0; sub < 2; ++sub)
#pragma unroll
                for (int j = 0; j < 4; ++j) mk[sub][j] = mask ? mask[min(kt * 32 + sub * 16 + 4 * g + j, p.Lk - 1)] : 0.0f;
        }
#pragma unroll
        for (int sub = 0; sub < 2; ++sub) {
            f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) {
                Frag<NS> kf;
                const int o = (sub * 16 + i) * LD + ks * 32 + g * 8;
                kf.v[0] = *reinterpret_cast<const bf16x8*>(stage + o);
                if (NS == 2) kf.v[NS - 1] = *reinterpret_cast<const bf16x8*>(stage + IMG + o);
                acc = mfma_split<NS>(kf, qf[ks], acc);
            }
            if (plain) {
#pragma unroll
                for (int j = 0; j < 4; ++j) s[sub][j] = acc[j] * scale + mk[sub][j] * -1e9f;
            } else {
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int key = kt * 32 + sub * 16 + 4 * g + j;
                    const float m = (p.causal && key > qi) ? 1.0f : mk[sub][j];
                    const float v = acc[j] * scale + m * -1e9f;
                    s[sub][j] = key >= p.Lk ? -INFINITY : v;
                }
            }
        }
    };

    f32x4 o[NT];
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) o[nt] = (f32x4){0.f, 0.f, 0.f, 0.f};
    float m_run = -INFINITY, l_run = 0.0f;
    const int nkt = (p.Lk + 31) / 32;

    float4 kx[NLD], vx[NLD];
    if (wave < nkt) {
        load_tile(kbase, p.ldk, wave, kx);
        load_tile(vbase, p.ldv, wave, vx);
    }
    auto tile_step = [&](int kt, float (&s)[2][4]) {
        stage_tile(kx);
        if (kt + NW < nkt) load_tile(kbase, p.ldk, kt + NW, kx);
        scores(kt, s);
        float tmax = -INFINITY;
#pragma unroll
        for (int sub = 0; sub < 2; ++sub)
#pragma unroll
            for (int j = 0; j < 4; ++j) tmax = fmaxf(tmax, s[sub][j]);
        tmax = fmaxf(tmax, lane_xor16(tmax));
        tmax = fmaxf(tmax, lane_xor32(tmax));
        const float m_new = fmaxf(m_run, tmax);
        const float alpha = exp_neg(m_run - m_new);  // exp(-inf) = 0 on the first tile
        float pv[8], psum = 0.0f;
#pragma unroll
        for (int sub = 0; sub < 2; ++sub)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const float e = exp_neg(s[sub][j] - m_new);
                pv[sub * 4 + j] = e;
                psum += e;
            }
        psum += lane_xor16(psum);
        psum += lane_xor32(psum);
        l_run = l_run * alpha + psum;
        m_run = m_new;
        const Frag<NS> pf = make_frag<NS>(pv);

        stage_tile(vx);  // same image: this wave's K fragment reads were issued before these writes (in-order LDS)
        if (kt + NW < nkt) load_tile(vbase, p.ldv, kt + NW, vx);
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
            Frag<NS> vf;
            const int o0 = (4 * g + (i >> 2)) * LD + nt * 16 + 4 * (i & 3);
            vf.v[0] = tr_read_pair(stage + o0, stage + o0 + 16 * LD);
            if (NS == 2) vf.v[NS - 1] = tr_read_pair(stage + IMG + o0, stage + IMG + o0 + 16 * LD);
            o[nt] = o[nt] * alpha;
            o[nt] = mfma_split<NS>(vf, pf, o[nt]);
        }
    };
    constexpr int KEEP = MAXT > 0 ? MAXT : 1;
    float sk[KEEP][8];
    if constexpr (MAXT > 0) {
#pragma unroll
        for (int it = 0; it < MAXT; ++it) {
            const int kt = wave + NW * it;
            if (kt < nkt) {
                float s[2][4];
                tile_step(kt, s);
#pragma unroll
                for (int e = 0; e < 8; ++e) sk[it][e] = s[e >> 2][e & 3];
            }
        }
    } else {
        for (int kt = wave; kt < nkt; kt += NW) {
            float s[2][4];
            tile_step(kt, s);
        }
    }

    // merge the four waves' states
    __syncthreads();  // every staging image is dead
    float* mg = reinterpret_cast<float*>(smem);
    {
        float* mine = mg + wave * MERGE_FLOATS;
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
#pragma unroll
            for (int r = 0; r < 4; ++r) mine[(nt * 4 + r) * 64 + lane] = o[nt][r];
        mine[NT * 256 + lane] = m_run;
        mine[NT * 256 + 64 + lane] = l_run;
    }
    __syncthreads();
    float m_all = -INFINITY;
#pragma unroll
    for (int w = 0; w < NW; ++w) m_all = fmaxf(m_all, mg[w * MERGE_FLOATS + NT * 256 + lane]);
    float sc[NW], l_all = 0.0f;
#pragma unroll
    for (int w = 0; w < NW; ++w) {
        sc[w] = exp_neg(mg[w * MERGE_FLOATS + NT * 256 + lane] - m_all);
        l_all += mg[w * MERGE_FLOATS + NT * 256 + 64 + lane] * sc[w];
    }
    const float inv_l = 1.0f / l_all;
    if (qi < p.Lq) {
        float* op = p.ctx + ((long)b * p.Lq + qi) * p.ldo + h * DH + 4 * g;
        for (int nt = wave; nt < NT; nt += NW) {
            float r4[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                float acc = 0.0f;
#pragma unroll
                for (int w = 0; w < NW; ++w) acc += mg[w * MERGE_FLOATS + (nt * 4 + r) * 64 + lane] * sc[w];
                r4[r] = acc * inv_l;
            }
            *reinterpret_cast<float4*>(op + nt * 16) = make_float4(r4[0], r4[1], r4[2], r4[3]);
        }
    }

    if constexpr (MAXT > 0) {
        if (qi < p.Lq) {
            float* ap = p.attn + (((long)b * p.H + h) * p.Lq + qrow) * p.Lk;
            const bool vec = (p.Lk & 3) == 0;
#pragma unroll
            for (int it = 0; it < MAXT; ++it) {
                const int kt = wave + NW * it;
                if (kt < nkt) {
#pragma unroll
                    for (int sub = 0; sub < 2; ++sub) {
                        const int key0 = kt * 32 + sub * 16 + 4 * g;
                        float w4[4];
#pragma unroll
                        for (int j = 0; j < 4; ++j) w4[j] = exp_neg(sk[it][sub * 4 + j] - m_all) * inv_l;
                        if (vec && key0 + 3 < p.Lk) {
                            *reinterpret_cast<float4*>(ap + key0) = make_float4(w4[0], w4[1], w4[2], w4[3]);
                        } else {
#pragma unroll
                            for (int j = 0; j < 4; ++j)
                                if (key0 + j < p.Lk) ap[key0 + j] = w4[j];
                        }
                    }
                }
            }
        }
    } else if (p.attn) {
        __syncthreads();  // merge buffer is dead; the staging images are reused
        float* ap = p.attn + (((long)b * p.H + h) * p.Lq + qrow) * p.Lk;
        const bool vec = (p.Lk & 3) == 0;
        if (wave < nkt) load_tile(kbase, p.ldk, wave, kx);
        for (int kt = wave; kt < nkt; kt += NW) {
            stage_tile(kx);
            if (kt + NW < nkt) load_tile(kbase, p.ldk, kt + NW, kx);
            float s[2][4];
            scores(kt, s);
            if (qi < p.Lq) {
#pragma unroll
                for (int sub = 0; sub < 2; ++sub) {
                    const int key0 = kt * 32 + sub * 16 + 4 * g;
                    float w4[4];
#pragma unroll
                    for (int j = 0; j < 4; ++j) w4[j] = exp_neg(s[sub][j] - m_all) * inv_l;
                    if (vec && key0 + 3 < p.Lk) {
                        *reinterpret_cast<float4*>(ap + key0) = make_float4(w4[0], w4[1], w4[2], w4[3]);
                    } else {
#pragma unroll
                        for (int j = 0; j < 4; ++j)
                            if (key0 + j < p.Lk) ap[key0 + j] = w4[j];
                    }
                }
            }
        }
    }
}


// ======================================================================================================
// Batched variant: a workgroup owns 64 query rows of one (batch, head), 16 per wave, and all four waves share every
// staged K / V tile (32 keys, loaded and split once per workgroup instead of once per 16 queries: a quarter of the
// L2 traffic and of the conversion work of the kernel above, which is what bounds it at batch 32).  Each wave keeps
// the whole softmax state of its own queries, so nothing is merged at the end.  Used when the launch has enough
// (batch, head, 64-query) workgroups to fill the chip.
//   * no weights wanted: one sweep with the online softmax; a causal launch stops at the last key tile any of the
//     workgroup's queries may see -- unless one of them has only masked keys there (its row is then uniform over ALL
//     keys, future ones included, exactly as softmax(-1e9 everywhere) is), in which case the sweep goes on to the end;
//   * weights wanted: a first sweep over K finds m and l, a second one over K and V writes exp(s - m) / l and
//     accumulates the normalised output.
// ======================================================================================================
// MAXT > 0: the weights are wanted and the keys fit MAXT tiles (Lk <= 32 MAXT): one sweep, the scaled scores stay in registers
// and the weights are written after it (as in the kernel above) instead of a second sweep that recomputes them.
// (four workgroups per CU up to depth 64: left alone the one-sweep form compiled to 156 registers -- three per CU, 1.67 rounds of
// the 1280 workgroups of a 600-query batch of 32 -- where 122 do without a spill: 600 x 100 with weights 28.6 -> 25.8 us, causal
// 600 x 600 46.5 -> 43.9 us)
template <int DH, int NS, int MAXT = 0>
__global__ __launch_bounds__(256, DH <= 64 ? 4 : 1) void attention_wide_kernel(const estts_attention_args p) {
    constexpr int KS = DH / 32;
    constexpr int NT = DH / 16;
    constexpr int LD = DH + 8;
    constexpr int IMG = 32 * LD;
    constexpr int LPR = DH / 4;        // threads per tile row (one float4 each)
    constexpr int RPP = 256 / LPR;     // tile rows per pass of the workgroup
    constexpr int NLD = 32 / RPP;      // passes per 32-key tile
    __shared__ __attribute__((aligned(16))) __bf16 smem[2][2][NS][IMG];   // [buffer][K, V][hi, lo]

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int i = lane & 15;
    const int g = lane >> 4;
    // Linear workgroup id -> (query tile, batch x head).  Workgroups go to the eight XCDs round robin, so with B H a
    // multiple of 8 every XCD gets B H / 8 whole (batch, head) pairs (their K / V stay in that XCD's L2) and the same mix
    // of query tiles; the last query tiles come first: under a causal mask they sweep the most keys, and the short ones
    // then fill the gaps the long ones leave.  (Without a mask, going through an XCD's pairs one after the other instead
    // measured 2 % slower: 600 x 600 with weights, B = 32.)
    const int BH = p.B * p.H;
    const int nqt = (p.Lq + 63) / 64;
    int bh, qt;
    if ((BH & 7) == 0) {
        const int per = BH >> 3, slot = blockIdx.x >> 3;
        bh = (blockIdx.x & 7) * per + slot % per;
        qt = nqt - 1 - slot / per;
    } else {
        bh = blockIdx.x % BH;
        qt = nqt - 1 - blockIdx.x / BH;
    }
    const int h = bh % p.H;
    const int b = bh / p.H;
    const int qi = qt * 64 + wave * 16 + i;
    const int qrow = min(qi, p.Lq - 1);
    const float* qp = p.q + ((long)b * p.Lq + qrow) * p.ldq + h * DH + 8 * g;
    const long kvb = p.kv_shared ? 0 : b;
    const float* kbase = p.k + kvb * p.Lk * p.ldk + h * DH;
    const float* vbase = p.v + kvb * p.Lk * p.ldv + h * DH;
    const float* mask = p.key_mask ? p.key_mask + (long)b * p.Lk : nullptr;
    const bool mask_quads = (p.Lk & 3) == 0 && (reinterpret_cast<uintptr_t>(p.key_mask) & 15) == 0;   // mask rows as float4
    const float scale = 1.0f / sqrtf((float)DH);

    Frag<NS> qf[KS];
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
        float t[8];
        *reinterpret_cast<float4*>(t) = *reinterpret_cast<const float4*>(qp + ks * 32);
        *reinterpret_cast<float4*>(t + 4) = *reinterpret_cast<const float4*>(qp + ks * 32 + 4);
        qf[ks] = make_frag<NS>(t);
    }

    const int ld_row = tid / LPR;
    const int ld_col = (tid % LPR) * 4;
    auto load_tile = [&](const float* base, int ld, int kt, float4 (&x)[NLD]) {
#pragma unroll
        for (int j = 0; j < NLD; ++j) {
            const int key = min(kt * 32 + j * RPP + ld_row, p.Lk - 1);
            x[j] = *reinterpret_cast<const float4*>(base + (long)key * ld + ld_col);
        }
    };
    auto stage_tile = [&](const float4 (&x)[NLD], __bf16* dst) {
#pragma unroll
        for (int j = 0; j < NLD; ++j) {
            bf16x4 hi, lo;
            split_bf16x4(x[j], hi, lo);
            const int o = (j * RPP + ld_row) * LD + ld_col;
            *reinterpret_cast<bf16x4*>(dst + o) = hi;
            if (NS == 2) *reinterpret_cast<bf16x4*>(dst + IMG + o) = lo;
        }
    };
    // scores of a staged K tile for this lane's query: s[sub * 4 + j] <-> key kt*32 + sub*16 + 4g + j
    auto scores = [&](int kt, const __bf16* kimg, float (&s)[8]) {
        // A tile that lies inside the key range and, under a causal mask, before this wave's first query needs neither the
        // range check nor the comparison with the query (a quarter of the vector instructions of a tile), and its mask
        // values come as two float4.  Both conditions are uniform over the wave.
        const bool inside = kt * 32 + 32 <= p.Lk && mask_quads;
        const bool plain = inside && !(p.causal && kt * 32 + 31 > qt * 64 + wave * 16);
        float mk[8];     // loaded up front and without conditions: see the kernel above
        if (inside) {
            float4 m0 = make_float4(0.f, 0.f, 0.f, 0.f), m1 = m0;
            if (mask) {
                m0 = *reinterpret_cast<const float4*>(mask + kt * 32 + 4 * g);
                m1 = *reinterpret_cast<const float4*>(mask + kt * 32 + 16 + 4 * g);
            }
            mk[0] = m0.x, mk[1] = m0.y, mk[2] = m0.z, mk[3] = m0.w, mk[4] = m1.x, mk[5] = m1.y, mk[6] = m1.z, mk[7] = m1.w;
        } else {
#pragma unroll
            for (int e = 0; e < 8; ++e) mk[e] = mask ? mask[min(kt * 32 + (e >> 2) * 16 + 4 * g + (e & 3), p.Lk - 1)] : 0.0f;
        }
#pragma unroll
        for (int sub = 0; sub < 2; ++sub) {
            f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) {
                Frag<NS> kf;
                const int o = (sub * 16 + i) * LD + ks * 32 + g * 8;
                kf.v[0] = *reinterpret_cast<const bf16x8*>(kimg + o);
                if (NS == 2) kf.v[NS - 1] = *reinterpret_cast<const bf16x8*>(kimg + IMG + o);
                acc = mfma_split<NS>(kf, qf[ks], acc);
            }
            if (plain) {
#pragma unroll
                for (int j = 0; j < 4; ++j) s[sub * 4 + j] = acc[j] * scale + mk[sub * 4 + j] * -1e9f;
            } else {
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int key = kt * 32 + sub * 16 + 4 * g + j;
                    const float m = (p.causal && key > qi) ? 1.0f : mk[sub * 4 + j];
                    const float v = acc[j] * scale + m * -1e9f;
                    s[sub * 4 + j] = key >= p.Lk ? -INFINITY : v;
                }
            }
        }
    };
    auto times_v = [&](const float (&pv)[8], const __bf16* vimg, f32x4 (&o)[NT]) {
        const Frag<NS> pf = make_frag<NS>(pv);
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
            Frag<NS> vf;
            const int o0 = (4 * g + (i >> 2)) * LD + nt * 16 + 4 * (i & 3);
            vf.v[0] = tr_read_pair(vimg + o0, vimg + o0 + 16 * LD);
            if (NS == 2) vf.v[NS - 1] = tr_read_pair(vimg + IMG + o0, vimg + IMG + o0 + 16 * LD);
            o[nt] = mfma_split<NS>(vf, pf, o[nt]);
        }
    };

    f32x4 o[NT];
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) o[nt] = (f32x4){0.f, 0.f, 0.f, 0.f};
    float m_run = -INFINITY, l_run = 0.0f;
    const int nkt = (p.Lk + 31) / 32;
    float4 kx[NLD], vx[NLD];
    int par = 0;   // staging buffer of the next tile (uniform over the workgroup)

    // one sweep over key tiles [t0, t1): WITH_V also stages V; the body gets the tile number and the two images
    auto sweep = [&](int t0, int t1, bool with_v, auto&& body) {
        if (t0 >= t1) return;
        load_tile(kbase, p.ldk, t0, kx);
        if (with_v) load_tile(vbase, p.ldv, t0, vx);
        for (int kt = t0; kt < t1; ++kt) {
            __bf16* kimg = &smem[par][0][0][0];
            __bf16* vimg = &smem[par][1][0][0];
            stage_tile(kx, kimg);
            if (with_v) stage_tile(vx, vimg);
            __syncthreads();   // also: every wave is done with the other buffer, which the next tile will overwrite
            if (kt + 1 < t1) {
                load_tile(kbase, p.ldk, kt + 1, kx);
                if (with_v) load_tile(vbase, p.ldv, kt + 1, vx);
            }
            body(kt, kimg, vimg);
            par ^= 1;
        }
    };
    auto update = [&](const float (&s)[8], const __bf16* vimg, bool with_v) {
        float tmax = -INFINITY;
#pragma unroll
        for (int e = 0; e < 8; ++e) tmax = fmaxf(tmax, s[e]);
        tmax = fmaxf(tmax, lane_xor16(tmax));
        tmax = fmaxf(tmax, lane_xor32(tmax));
        const float m_new = fmaxf(m_run, tmax);
        const float alpha = exp_neg(m_run - m_new);  // exp(-inf) = 0 on the first tile
        float pv[8], psum = 0.0f;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            pv[e] = exp_neg(s[e] - m_new);
            psum += pv[e];
        }
        psum += lane_xor16(psum);
        psum += lane_xor32(psum);
        l_run = l_run * alpha + psum;
        m_run = m_new;
        if (with_v) {
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) o[nt] = o[nt] * alpha;
            times_v(pv, vimg, o);
        }
    };
    auto online = [&](int kt, const __bf16* kimg, const __bf16* vimg, bool with_v) {
        float s[8];
        scores(kt, kimg, s);
        update(s, vimg, with_v);
    };

    float inv_l;
    if constexpr (MAXT > 0) {
        float sk[MAXT][8];
        load_tile(kbase, p.ldk, 0, kx);
        load_tile(vbase, p.ldv, 0, vx);
#pragma unroll
        for (int kt = 0; kt < MAXT; ++kt) {
            if (kt < nkt) {                                   // uniform over the workgroup
                __bf16* kimg = &smem[par][0][0][0];
                __bf16* vimg = &smem[par][1][0][0];
                stage_tile(kx, kimg);
                stage_tile(vx, vimg);
                __syncthreads();
                if (kt + 1 < nkt) {
                    load_tile(kbase, p.ldk, kt + 1, kx);
                    load_tile(vbase, p.ldv, kt + 1, vx);
                }
                scores(kt, kimg, sk[kt]);
                update(sk[kt], vimg, true);
                par ^= 1;
            }
        }
        inv_l = 1.0f / l_run;
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) o[nt] = o[nt] * inv_l;
        if (qi < p.Lq) {
            float* ap = p.attn + (((long)b * p.H + h) * p.Lq + qrow) * p.Lk;
            const bool vec = (p.Lk & 3) == 0;
#pragma unroll
            for (int kt = 0; kt < MAXT; ++kt) {
                if (kt < nkt) {
#pragma unroll
                    for (int sub = 0; sub < 2; ++sub) {
                        const int key0 = kt * 32 + sub * 16 + 4 * g;
                        float w4[4];
#pragma unroll
                        for (int j = 0; j < 4; ++j) w4[j] = exp_neg(sk[kt][sub * 4 + j] - m_run) * inv_l;
                        if (vec && key0 + 3 < p.Lk) {
                            *reinterpret_cast<float4*>(ap + key0) = make_float4(w4[0], w4[1], w4[2], w4[3]);
                        } else {
#pragma unroll
                            for (int j = 0; j < 4; ++j)
                                if (key0 + j < p.Lk) ap[key0 + j] = w4[j];
                        }
                    }
                }
            }
        }
    } else if (!p.attn) {
        const int last_q = min(qt * 64 + 63, p.Lq - 1);
        const int nlim = p.causal ? min(nkt, last_q / 32 + 1) : nkt;
        sweep(0, nlim, true, [&](int kt, const __bf16* kimg, const __bf16* vimg) { online(kt, kimg, vimg, true); });
        if (nlim < nkt) {
            // a query whose visible keys are all masked has every score at -1e9, the future keys' included
            const int all_masked = __syncthreads_or(qi < p.Lq && m_run < -5e8f);
            if (all_masked)
                sweep(nlim, nkt, true, [&](int kt, const __bf16* kimg, const __bf16* vimg) { online(kt, kimg, vimg, true); });
        }
        inv_l = 1.0f / l_run;
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) o[nt] = o[nt] * inv_l;
    } else {
        sweep(0, nkt, false, [&](int kt, const __bf16* kimg, const __bf16* vimg) { online(kt, kimg, vimg, false); });
        inv_l = 1.0f / l_run;
        float* ap = p.attn + (((long)b * p.H + h) * p.Lq + qrow) * p.Lk;
        const bool vec = (p.Lk & 3) == 0;
        sweep(0, nkt, true, [&](int kt, const __bf16* kimg, const __bf16* vimg) {
            float s[8], pv[8];
            scores(kt, kimg, s);
#pragma unroll
            for (int e = 0; e < 8; ++e) pv[e] = exp_neg(s[e] - m_run) * inv_l;
            if (qi < p.Lq) {
#pragma unroll
                for (int sub = 0; sub < 2; ++sub) {
                    const int key0 = kt * 32 + sub * 16 + 4 * g;
                    if (vec && key0 + 3 < p.Lk) {
                        *reinterpret_cast<float4*>(ap + key0) = make_float4(pv[sub * 4], pv[sub * 4 + 1], pv[sub * 4 + 2], pv[sub * 4 + 3]);
                    } else {
#pragma unroll
                        for (int j = 0; j < 4; ++j)
                            if (key0 + j < p.Lk) ap[key0 + j] = pv[sub * 4 + j];
                    }
                }
            }
            times_v(pv, vimg, o);
        });
    }
    if (qi < p.Lq) {
        float* op = p.ctx + ((long)b * p.Lq + qi) * p.ldo + h * DH + 4 * g;
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) *reinterpret_cast<float4*>(op + nt * 16) = make_float4(o[nt][0], o[nt][1], o[nt][2], o[nt][3]);
    }
}

template <int DH>
int launch_wide(const estts_attention_args& a, hipStream_t st) {
    dim3 grid((unsigned)((long)((a.Lq + 63) / 64) * a.H * a.B));
    if (a.attn && a.Lk <= 128) {
        if (a.precision == 3)
            ESTTS_LAUNCH((attention_wide_kernel<DH, 2, 4>), grid, dim3(256), 0, st, a);
        else
            ESTTS_LAUNCH((attention_wide_kernel<DH, 1, 4>), grid, dim3(256), 0, st, a);
    } else if (a.precision == 3)
        ESTTS_LAUNCH((attention_wide_kernel<DH, 2>), grid, dim3(256), 0, st, a);
    else
        ESTTS_LAUNCH((attention_wide_kernel<DH, 1>), grid, dim3(256), 0, st, a);
    ESTTS_CHECK_LAUNCH("estts_attention");
    return 0;
}

template <int DH, int MAXT, int NW = 4>
int launch_t(const estts_attention_args& a, hipStream_t st) {
    dim3 grid((a.Lq + 15) / 16, a.H, a.B);
    if (a.precision == 3)
        ESTTS_LAUNCH((attention_kernel<DH, 2, MAXT, NW>), grid, dim3(NW * 64), 0, st, a);
    else
        ESTTS_LAUNCH((attention_kernel<DH, 1, MAXT, NW>), grid, dim3(NW * 64), 0, st, a);
    ESTTS_CHECK_LAUNCH("estts_attention");
    return 0;
}

template <int DH>
int launch(const estts_attention_args& a, hipStream_t st) {
    // variant: 0 = by launch size, 1 = 16-query workgroups, 2 = 64-query workgroups
    const long wide_wgs = (long)((a.Lq + 63) / 64) * a.H * a.B;
    // (at depth 128 the 16-query kernel needs more than 256 registers, one workgroup per CU: the shared-tile kernel wins earlier)
    // (and so does its one-sweep form for weights over at most 128 keys: 100 x 100, B = 32: 12.4 us against 15.0)
    const long wide_min = (DH >= 128 || (a.attn && a.Lk <= 128)) ? 192 : 512;
    if (a.variant == 2 || (a.variant == 0 && wide_wgs >= wide_min && a.Lk >= 64)) return launch_wide<DH>(a, st);
    // few workgroups and many key tiles (one utterance, depth 64): eight waves split the keys, three tiles each at 600 keys
    const long narrow_wgs = (long)((a.Lq + 15) / 16) * a.H * a.B;
    if constexpr (DH == 64) {
        if (a.variant == 0 && narrow_wgs <= 256 && a.Lk > 256) {
            const int per8 = ((a.Lk + 31) / 32 + 7) / 8;
            if (a.attn && per8 <= 3) return launch_t<DH, 3, 8>(a, st);
            if (!a.attn) return launch_t<DH, 0, 8>(a, st);
        }
    }
    const int per_wave = ((a.Lk + 31) / 32 + 3) / 4;   // key tiles each wave sweeps
    if (a.attn && per_wave <= 2) return launch_t<DH, 2>(a, st);
    if (a.attn && per_wave <= 5) return launch_t<DH, 5>(a, st);
    if (a.attn && per_wave <= 8) return launch_t<DH, 8>(a, st);
    return launch_t<DH, 0>(a, st);
}

}  // namespace

extern "C" int estts_attention(const estts_attention_args* args, void* stream) {
    ESTTS_CHECK_ARG(args != nullptr, "estts_attention: null args");
    const estts_attention_args& a = *args;
    ESTTS_CHECK_ARG(a.q && a.k && a.v && a.ctx, "estts_attention: null q/k/v/ctx");
    ESTTS_CHECK_ARG(a.B > 0 && a.H > 0 && a.Lq > 0 && a.Lk > 0, "estts_attention: bad shape");
    ESTTS_CHECK_ARG(a.B <= 65535 && a.H <= 65535, "estts_attention: B / H exceed grid limits");
    ESTTS_CHECK_ARG((long)((a.Lq + 15) / 16) * a.H * a.B < (1L << 31), "estts_attention: too many workgroups");
    ESTTS_CHECK_ARG(a.precision == 1 || a.precision == 3, "estts_attention: precision must be 1 or 3");
    ESTTS_CHECK_ARG((a.ldq & 3) == 0 && (a.ldk & 3) == 0 && (a.ldv & 3) == 0 && (a.ldo & 3) == 0,
                    "estts_attention: ld not multiple of 4");
    ESTTS_CHECK_ARG(a.ldq >= a.H * a.depth && a.ldk >= a.H * a.depth && a.ldv >= a.H * a.depth && a.ldo >= a.H * a.depth,
                    "estts_attention: ld smaller than H*depth");
    ESTTS_CHECK_ARG((((uintptr_t)a.q | (uintptr_t)a.k | (uintptr_t)a.v | (uintptr_t)a.ctx | (uintptr_t)a.attn) & 15) == 0,
                    "estts_attention: q/k/v/ctx/attn not 16-byte aligned");
    hipStream_t st = (hipStream_t)stream;
    switch (a.depth) {
        case 32: return launch<32>(a, st);
        case 64: return launch<64>(a, st);
        case 128: return launch<128>(a, st);
        default: estts_set_error("estts_attention: depth %d unsupported (32, 64, 128)", a.depth); return -1;
    }
}
