// Gradients of multi-head attention from the saved weights P (estts_attention_bwd in include/estts.h).
//
// Two launches of one kernel template after a small prepass (delta[b, h, i] = dctx_i . ctx_i):
//   OWN_Q = true : a workgroup owns 16 queries of one (batch, head) and sweeps the keys          -> dQ
//   OWN_Q = false: a workgroup owns 16 keys and sweeps the queries                                -> dK, dV
// As in the forward kernel, the four waves take interleaved 32-row tiles of the swept side, every product is issued
// "swapped" so that an owned row stays on one lane column, and the swept tiles are staged (bf16 hi + lo) in a
// wave-private LDS image [32][depth + 8]:
//   dP tile  = (swept V or dctx rows, ds_read_b128 fragments)  x  (owned dctx or V rows, in registers)
//   dS       = P o (dP - delta) / sqrt(depth)                              P read from global memory
//   out^T   += (swept K or Q tile)^T (ds_read_b64_tr_b16)  x  dS           -> dQ^T or dK^T
//   dV^T    += (swept dctx tile)^T  x  P                                   (OWN_Q = false only)
// Partial results of the four waves are summed through LDS.
// Launches with at least 192 workgroups of 64 owned rows use the batched variant further down instead.
#include "common.h"

namespace {

typedef __attribute__((ext_vector_type(4))) short s16x4;
typedef __attribute__((ext_vector_type(8))) short s16x8;

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

// delta[b, h, i] = sum_f dctx[b, i, h, f] * ctx[b, i, h, f]; one wave per (b, i) row
__global__ __launch_bounds__(256) void attention_delta_kernel(const estts_attention_bwd_args p) {
    const int lane = threadIdx.x & 63;
    const long row = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= (long)p.B * p.Lq) return;
    const int b = (int)(row / p.Lq);
    const int qi = (int)(row - (long)b * p.Lq);
    const int per_head = p.depth >> 2;  // float4 chunks per head: 8, 16 or 32 lanes
    const int chunks = p.H * per_head;
    for (int c0 = 0; c0 < chunks; c0 += 64) {
        const int ch = c0 + lane;
        float s = 0.f;
        if (ch < chunks) {
            const float4 a = *reinterpret_cast<const float4*>(p.dctx + row * p.lddo + ch * 4);
            const float4 o = *reinterpret_cast<const float4*>(p.ctx + row * p.ldo + ch * 4);
            s = a.x * o.x + a.y * o.y + a.z * o.z + a.w * o.w;
        }
        for (int off = per_head >> 1; off > 0; off >>= 1) s += __shfl_xor(s, off, 64);
        if (ch < chunks && (lane & (per_head - 1)) == 0) {
            const int h = ch / per_head;
            p.delta[((long)b * p.H + h) * p.Lq + qi] = s;
        }
    }
}

template <int DH, int NS, bool OWN_Q>
__global__ __launch_bounds__(256) void attention_bwd_kernel(const estts_attention_bwd_args p) {
    constexpr int KS = DH / 32;
    constexpr int NT = DH / 16;
    constexpr int LD = DH + 8;
    constexpr int LPR = DH / 4;
    constexpr int RPI = 64 / LPR;
    constexpr int NLD = 32 / RPI;
    constexpr int IMG = 32 * LD;
    constexpr int STAGE_BYTES = 4 * NS * IMG * 2;
    constexpr int MERGE_FLOATS = NT * 256;
    constexpr int MERGE_BYTES = 4 * MERGE_FLOATS * 4;
    __shared__ __attribute__((aligned(16))) unsigned char smem[STAGE_BYTES > MERGE_BYTES ? STAGE_BYTES : MERGE_BYTES];

    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    const int i = lane & 15;
    const int g = lane >> 4;
    const int L_own = OWN_Q ? p.Lq : p.Lk;
    const int L_oth = OWN_Q ? p.Lk : p.Lq;
    // Linear workgroup id -> (batch x head, tile of 16 owned rows).  Workgroups go to the eight XCDs round robin: with
    // B H a multiple of 8 every XCD works through B H / 8 whole (batch, head) pairs one after the other, so the other
    // side's rows and the neighbouring 64-byte column blocks of the weights are found in that XCD's L2.
    const int BH = p.B * p.H;
    const int ntile = (L_own + 15) / 16;
    int bh, tile;
    if ((BH & 7) == 0) {
        const int slot = blockIdx.x >> 3;
        bh = (blockIdx.x & 7) * (BH >> 3) + slot / ntile;
        tile = slot % ntile;
    } else {
        bh = blockIdx.x / ntile;
        tile = blockIdx.x % ntile;
    }
    const int h = bh % p.H;
    const int b = bh / p.H;
    const int oi = tile * 16 + i;
    const int orow = min(oi, L_own - 1);
    const bool own_ok = oi < L_own;

    const float* rp = OWN_Q ? p.dctx + ((long)b * p.Lq + orow) * p.lddo + h * DH + 8 * g
                            : p.v + ((long)b * p.Lk + orow) * p.ldv + h * DH + 8 * g;
    const float* sbase = OWN_Q ? p.v + (long)b * p.Lk * p.ldv + h * DH : p.dctx + (long)b * p.Lq * p.lddo + h * DH;
    const int s_ld = OWN_Q ? p.ldv : p.lddo;
    const float* obase = OWN_Q ? p.k + (long)b * p.Lk * p.ldk + h * DH : p.q + (long)b * p.Lq * p.ldq + h * DH;
    const int o_ld = OWN_Q ? p.ldk : p.ldq;
    const float* P = p.attn + ((long)b * p.H + h) * p.Lq * p.Lk;
    const float* delta = p.delta + ((long)b * p.H + h) * p.Lq;
    const float scale = 1.0f / sqrtf((float)DH);
    const float own_delta = OWN_Q ? delta[orow] : 0.0f;
    const bool vec = OWN_Q && (p.Lk & 3) == 0;
    __bf16* stage = reinterpret_cast<__bf16*>(smem) + wave * (NS * IMG);

    Frag<NS> rf[KS];
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
        float t[8];
        *reinterpret_cast<float4*>(t) = *reinterpret_cast<const float4*>(rp + ks * 32);
        *reinterpret_cast<float4*>(t + 4) = *reinterpret_cast<const float4*>(rp + ks * 32 + 4);
        rf[ks] = make_frag<NS>(t);
    }

    const int ld_row = lane / LPR;
    const int ld_col = (lane % LPR) * 4;
    auto load_tile = [&](const float* base, int ld, int kt, float4 (&x)[NLD]) {
#pragma unroll
        for (int j = 0; j < NLD; ++j) {
            const int r = min(kt * 32 + j * RPI + ld_row, L_oth - 1);
            x[j] = *reinterpret_cast<const float4*>(base + (long)r * ld + ld_col);
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
    // out^T (16-feature tiles x owned rows) += (staged tile)^T x operand
    auto accumulate = [&](f32x4* o, const Frag<NS>& bf) {
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
            Frag<NS> af;
            const int o0 = (4 * g + (i >> 2)) * LD + nt * 16 + 4 * (i & 3);
            af.v[0] = tr_read_pair(stage + o0, stage + o0 + 16 * LD);
            if (NS == 2) af.v[NS - 1] = tr_read_pair(stage + IMG + o0, stage + IMG + o0 + 16 * LD);
            o[nt] = mfma_split<NS>(af, bf, o[nt]);
        }
    };

    f32x4 o1[NT], o2[OWN_Q ? 1 : NT];
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) o1[nt] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int nt = 0; nt < (OWN_Q ? 1 : NT); ++nt) o2[nt] = (f32x4){0.f, 0.f, 0.f, 0.f};

    const int nkt = (L_oth + 31) / 32;
    float4 sx[NLD], ox[NLD];
    if (wave < nkt) {
        load_tile(sbase, s_ld, wave, sx);
        load_tile(obase, o_ld, wave, ox);
    }
    for (int kt = wave; kt < nkt; kt += 4) {
        stage_tile(sx);
        if (kt + 4 < nkt) load_tile(sbase, s_ld, kt + 4, sx);
        float pv[8], dsv[8];
#pragma unroll
        for (int sub = 0; sub < 2; ++sub) {
            f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) {
                Frag<NS> sf;
                const int o = (sub * 16 + i) * LD + ks * 32 + g * 8;
                sf.v[0] = *reinterpret_cast<const bf16x8*>(stage + o);
                if (NS == 2) sf.v[NS - 1] = *reinterpret_cast<const bf16x8*>(stage + IMG + o);
                acc = mfma_split<NS>(sf, rf[ks], acc);
            }
            const int oth0 = kt * 32 + sub * 16 + 4 * g;
            float pp[4] = {0.f, 0.f, 0.f, 0.f};
            if (own_ok) {
                if (vec && oth0 + 3 < L_oth) {
                    *reinterpret_cast<float4*>(pp) = *reinterpret_cast<const float4*>(P + (long)orow * p.Lk + oth0);
                } else {
#pragma unroll
                    for (int j = 0; j < 4; ++j)
                        if (oth0 + j < L_oth) pp[j] = OWN_Q ? P[(long)orow * p.Lk + oth0 + j] : P[(long)(oth0 + j) * p.Lk + orow];
                }
            }
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                float dl = own_delta;
                if (!OWN_Q) dl = oth0 + j < L_oth ? delta[oth0 + j] : 0.0f;
                pv[sub * 4 + j] = pp[j];
                dsv[sub * 4 + j] = pp[j] * (acc[j] - dl) * scale;
            }
        }
        if constexpr (!OWN_Q) {
            const Frag<NS> pf = make_frag<NS>(pv);
            accumulate(o2, pf);
        }
        stage_tile(ox);  // same image: this wave's reads of it were issued before these writes (in-order LDS)
        if (kt + 4 < nkt) load_tile(obase, o_ld, kt + 4, ox);
        const Frag<NS> dsf = make_frag<NS>(dsv);
        accumulate(o1, dsf);
    }

    float* mg = reinterpret_cast<float*>(smem);
    auto merge_write = [&](const f32x4* o, float* dst, int ld) {
        __syncthreads();  // staging images / the previous merge are dead
        float* mine = mg + wave * MERGE_FLOATS;
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
#pragma unroll
            for (int r = 0; r < 4; ++r) mine[(nt * 4 + r) * 64 + lane] = o[nt][r];
        __syncthreads();
        if (own_ok) {
            float* op = dst + ((long)b * L_own + oi) * ld + h * DH + 4 * g;
            for (int nt = wave; nt < NT; nt += 4) {
                float r4[4];
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    float acc = 0.0f;
#pragma unroll
                    for (int w = 0; w < 4; ++w) acc += mg[w * MERGE_FLOATS + (nt * 4 + r) * 64 + lane];
                    r4[r] = acc;
                }
                *reinterpret_cast<float4*>(op + nt * 16) = make_float4(r4[0], r4[1], r4[2], r4[3]);
            }
        }
    };
    if constexpr (OWN_Q) {
        merge_write(o1, p.dq, p.lddq);
    } else {
        merge_write(o1, p.dk, p.lddk);
        merge_write(o2, p.dv, p.lddv);
    }
}

// Batched variant: a workgroup owns 64 rows of one (batch, head), 16 per wave, and the four waves share every staged tile
// of the swept side (both operands of a tile in one double-buffered LDS image pair, one barrier per tile), so a swept
// tile is fetched from L2 and split once per 64 owned rows instead of once per 16: at batch 32 the kernel above is bound
// by exactly that traffic.  Every wave keeps the whole sum of its own rows, so nothing is merged at the end.  The weights
// and delta of the next tile are fetched together with its rows, one tile ahead of their use.
// workgroups per CU the register budget must allow (depth 64, split operands: the dQ pass compiles to 132 registers and the
// dK / dV pass to 176 left alone, each one step above an occupancy threshold)
template <int DH, int NS, bool OWN_Q>
constexpr int bwd_wide_occupancy() { return (DH == 64 && NS == 2) ? (OWN_Q ? 4 : 3) : 1; }

template <int DH, int NS, bool OWN_Q>
__global__ __launch_bounds__(256, (bwd_wide_occupancy<DH, NS, OWN_Q>())) void attention_bwd_wide_kernel(const estts_attention_bwd_args p) {
    constexpr int KS = DH / 32;
    constexpr int NT = DH / 16;
    constexpr int LD = DH + 8;
    constexpr int IMG = 32 * LD;
    constexpr int LPR = DH / 4;        // threads per tile row (one float4 each)
    constexpr int RPP = 256 / LPR;     // tile rows per pass of the workgroup
    constexpr int NLD = 32 / RPP;      // passes per 32-row tile
    __shared__ __attribute__((aligned(16))) __bf16 smem[2][2][NS][IMG];   // [buffer][dP operand, out operand][hi, lo]

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int i = lane & 15;
    const int g = lane >> 4;
    const int L_own = OWN_Q ? p.Lq : p.Lk;
    const int L_oth = OWN_Q ? p.Lk : p.Lq;
    const int BH = p.B * p.H;          // workgroup id -> (batch x head, tile): as in the kernel above
    const int ntile = (L_own + 63) / 64;
    int bh, tile;
    if ((BH & 7) == 0) {
        const int slot = blockIdx.x >> 3;
        bh = (blockIdx.x & 7) * (BH >> 3) + slot / ntile;
        tile = slot % ntile;
    } else {
        bh = blockIdx.x / ntile;
        tile = blockIdx.x % ntile;
    }
    const int h = bh % p.H;
    const int b = bh / p.H;
    const int oi = tile * 64 + wave * 16 + i;
    const int orow = min(oi, L_own - 1);
    const bool own_ok = oi < L_own;

    const float* rp = OWN_Q ? p.dctx + ((long)b * p.Lq + orow) * p.lddo + h * DH + 8 * g
                            : p.v + ((long)b * p.Lk + orow) * p.ldv + h * DH + 8 * g;
    const float* sbase = OWN_Q ? p.v + (long)b * p.Lk * p.ldv + h * DH : p.dctx + (long)b * p.Lq * p.lddo + h * DH;
    const int s_ld = OWN_Q ? p.ldv : p.lddo;
    const float* obase = OWN_Q ? p.k + (long)b * p.Lk * p.ldk + h * DH : p.q + (long)b * p.Lq * p.ldq + h * DH;
    const int o_ld = OWN_Q ? p.ldk : p.ldq;
    const float* P = p.attn + ((long)b * p.H + h) * p.Lq * p.Lk;
    const float* delta = p.delta + ((long)b * p.H + h) * p.Lq;
    const float scale = 1.0f / sqrtf((float)DH);
    const float own_delta = OWN_Q ? delta[orow] : 0.0f;

    Frag<NS> rf[KS];
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
        float t[8];
        *reinterpret_cast<float4*>(t) = *reinterpret_cast<const float4*>(rp + ks * 32);
        *reinterpret_cast<float4*>(t + 4) = *reinterpret_cast<const float4*>(rp + ks * 32 + 4);
        rf[ks] = make_frag<NS>(t);
    }

    const int ld_row = tid / LPR;
    const int ld_col = (tid % LPR) * 4;
    auto load_tile = [&](const float* base, int ld, int kt, float4 (&x)[NLD]) {
#pragma unroll
        for (int j = 0; j < NLD; ++j) {
            const int r = min(kt * 32 + j * RPP + ld_row, L_oth - 1);
            x[j] = *reinterpret_cast<const float4*>(base + (long)r * ld + ld_col);
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
    // weights of this lane's row against the 8 swept rows it meets in tile kt (0 outside the map), and their delta.
    // Tiles that lie inside the map take loads without bounds checks from one base address (the checks and the 64-bit
    // address arithmetic of eight separate elements were a third of the vector instructions of a tile).  A lane whose
    // own row is past the end reads its clamped row there: its column of the products is never written.
    const bool quad_ok = (L_oth & 3) == 0;   // (OWN_Q: rows of P; otherwise: delta) can be read four at a time
    const float* pbase = OWN_Q ? P + (long)orow * p.Lk + 4 * g : P + (long)(4 * g) * p.Lk + orow;
    auto load_p = [&](int kt, float (&pp)[8], float (&dl)[8]) {
        if (kt * 32 + 32 <= L_oth && quad_ok) {   // uniform over the workgroup
            if constexpr (OWN_Q) {
#pragma unroll
                for (int sub = 0; sub < 2; ++sub) {
                    const float4 t = *reinterpret_cast<const float4*>(pbase + kt * 32 + sub * 16);
                    pp[sub * 4] = t.x, pp[sub * 4 + 1] = t.y, pp[sub * 4 + 2] = t.z, pp[sub * 4 + 3] = t.w;
#pragma unroll
                    for (int j = 0; j < 4; ++j) dl[sub * 4 + j] = own_delta;
                }
            } else {
                const float* pc = pbase + (long)kt * 32 * p.Lk;
#pragma unroll
                for (int sub = 0; sub < 2; ++sub) {
#pragma unroll
                    for (int j = 0; j < 4; ++j) pp[sub * 4 + j] = pc[(sub * 16 + j) * p.Lk];
                    const float4 t = *reinterpret_cast<const float4*>(delta + kt * 32 + sub * 16 + 4 * g);
                    dl[sub * 4] = t.x, dl[sub * 4 + 1] = t.y, dl[sub * 4 + 2] = t.z, dl[sub * 4 + 3] = t.w;
                }
            }
            return;
        }
#pragma unroll
        for (int sub = 0; sub < 2; ++sub) {
            const int oth0 = kt * 32 + sub * 16 + 4 * g;
#pragma unroll
            for (int j = 0; j < 4; ++j) pp[sub * 4 + j] = 0.0f;
            if (own_ok) {
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    if (oth0 + j < L_oth) pp[sub * 4 + j] = OWN_Q ? P[(long)orow * p.Lk + oth0 + j] : P[(long)(oth0 + j) * p.Lk + orow];
            }
#pragma unroll
            for (int j = 0; j < 4; ++j) dl[sub * 4 + j] = OWN_Q ? own_delta : delta[min(oth0 + j, L_oth - 1)];
        }
    };
    // out^T (16-feature tiles x owned rows) += (staged tile)^T x operand
    auto accumulate = [&](f32x4* o, const Frag<NS>& bf, const __bf16* img) {
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
            Frag<NS> af;
            const int o0 = (4 * g + (i >> 2)) * LD + nt * 16 + 4 * (i & 3);
            af.v[0] = tr_read_pair(img + o0, img + o0 + 16 * LD);
            if (NS == 2) af.v[NS - 1] = tr_read_pair(img + IMG + o0, img + IMG + o0 + 16 * LD);
            o[nt] = mfma_split<NS>(af, bf, o[nt]);
        }
    };

    f32x4 o1[NT], o2[OWN_Q ? 1 : NT];
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) o1[nt] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int nt = 0; nt < (OWN_Q ? 1 : NT); ++nt) o2[nt] = (f32x4){0.f, 0.f, 0.f, 0.f};

    // (dQ pass) the fast form of load_p alone, for tiles known to lie inside the map: no condition around its loads
    auto load_p_inside = [&](int kt, float (&pp)[8], float (&dl)[8]) {
#pragma unroll
        for (int sub = 0; sub < 2; ++sub) {
            const float4 t = *reinterpret_cast<const float4*>(pbase + kt * 32 + sub * 16);
            pp[sub * 4] = t.x, pp[sub * 4 + 1] = t.y, pp[sub * 4 + 2] = t.z, pp[sub * 4 + 3] = t.w;
#pragma unroll
            for (int j = 0; j < 4; ++j) dl[sub * 4 + j] = own_delta;
        }
    };

    const int nkt = (L_oth + 31) / 32;
    const int ninside = quad_ok ? L_oth / 32 : 0;   // tiles [0, ninside) take load_p_inside
    float4 sx[NLD], ox[NLD];
    // One swept tile: stage the rows held in sx / ox, meet the other waves, multiply.  `after_barrier` requests the rows of
    // the next tile; `after_p` runs once this tile's weights and delta (pv, dl) are dead: the dQ pass requests those of the
    // tile after the next into the same registers there, so that the weights -- the operand that comes from HBM, 184 MB per
    // pass at batch 32 -- have a tile and a half to arrive instead of one.
    auto step = [&](int par, float (&pv)[8], float (&dl)[8], auto&& after_barrier, auto&& after_p) {
        __bf16* simg = &smem[par][0][0][0];
        __bf16* oimg = &smem[par][1][0][0];
        stage_tile(sx, simg);
        stage_tile(ox, oimg);
        __syncthreads();   // also: every wave is done with the other buffer, which the next tile will overwrite
        after_barrier();
        if constexpr (OWN_Q) __builtin_amdgcn_sched_barrier(0);   // the requests go out here, not wherever the matrix instructions leave a gap
        float dsv[8];
#pragma unroll
        for (int sub = 0; sub < 2; ++sub) {
            f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) {
                Frag<NS> sf;
                const int o = (sub * 16 + i) * LD + ks * 32 + g * 8;
                sf.v[0] = *reinterpret_cast<const bf16x8*>(simg + o);
                if (NS == 2) sf.v[NS - 1] = *reinterpret_cast<const bf16x8*>(simg + IMG + o);
                acc = mfma_split<NS>(sf, rf[ks], acc);
            }
#pragma unroll
            for (int j = 0; j < 4; ++j) dsv[sub * 4 + j] = pv[sub * 4 + j] * (acc[j] - dl[sub * 4 + j]) * scale;
        }
        Frag<NS> pf;
        if constexpr (!OWN_Q) pf = make_frag<NS>(pv);
        after_p();
        if constexpr (OWN_Q) __builtin_amdgcn_sched_barrier(0);
        if constexpr (!OWN_Q) accumulate(o2, pf, simg);
        const Frag<NS> dsf = make_frag<NS>(dsv);
        accumulate(o1, dsf, oimg);
    };
    auto load_rows = [&](int kt) {
        load_tile(sbase, s_ld, kt, sx);
        load_tile(obase, o_ld, kt, ox);
    };

    load_rows(0);
    if constexpr (OWN_Q) {
        float pa[8], da[8], pb[8], db[8];   // weights and delta of the even / odd tiles
        int kt = 0;
        if (ninside >= 4) {
            // Steady state: no load under a condition, entered behind loads in the loop's own order.  The compiler's s_waitcnt
            // counts are static; one path on which a younger set of loads was not issued makes every iteration wait for all
            // loads in flight (vmcnt(0)), and the longer prefetch distance buys nothing.
            load_p_inside(0, pa, da);
            __builtin_amdgcn_sched_barrier(0);
            load_p_inside(1, pb, db);
            __builtin_amdgcn_sched_barrier(0);
            for (; kt + 3 < ninside; kt += 2) {
                step(0, pa, da, [&] { load_rows(kt + 1); }, [&] { load_p_inside(kt + 2, pa, da); });
                step(1, pb, db, [&] { load_rows(kt + 2); }, [&] { load_p_inside(kt + 3, pb, db); });
            }
        } else {
            load_p(0, pa, da);
            if (nkt > 1) load_p(1, pb, db);
        }
        // the last tiles (and maps too small for the loop above): the same steps with every request under its condition
        for (; kt < nkt; kt += 2) {
            step(0, pa, da, [&] { if (kt + 1 < nkt) load_rows(kt + 1); }, [&] { if (kt + 2 < nkt) load_p(kt + 2, pa, da); });
            if (kt + 1 < nkt)
                step(1, pb, db, [&] { if (kt + 2 < nkt) load_rows(kt + 2); }, [&] { if (kt + 3 < nkt) load_p(kt + 3, pb, db); });
        }
    } else {
        // the dK / dV pass has no registers to spare for a second set at three workgroups per CU (eight scalar loads with
        // 64-bit lane addresses per tile): weights and delta one tile ahead, requested together with the rows
        float pn[8], dn[8];
        load_p(0, pn, dn);
        int par = 0;
        for (int kt = 0; kt < nkt; ++kt) {
            float pv[8], dl[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) pv[e] = pn[e], dl[e] = dn[e];
            step(par, pv, dl, [&] {
                if (kt + 1 < nkt) {
                    load_rows(kt + 1);
                    load_p(kt + 1, pn, dn);
                }
            }, [] {});
            par ^= 1;
        }
    }

    if (own_ok) {
        auto write = [&](const f32x4* o, float* dst, int ld) {
            float* op = dst + ((long)b * L_own + oi) * ld + h * DH + 4 * g;
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) *reinterpret_cast<float4*>(op + nt * 16) = make_float4(o[nt][0], o[nt][1], o[nt][2], o[nt][3]);
        };
        if constexpr (OWN_Q) {
            write(o1, p.dq, p.lddq);
        } else {
            write(o1, p.dk, p.lddk);
            write(o2, p.dv, p.lddv);
        }
    }
}

template <int DH, bool OWN_Q>
void launch_side(const estts_attention_bwd_args& a, hipStream_t st) {
    const int L_own = OWN_Q ? a.Lq : a.Lk;
    const long BH = (long)a.H * a.B;
    const long wide = (long)((L_own + 63) / 64) * BH;
    if (wide >= 192) {   // enough 64-row workgroups to fill the chip
        if (a.precision == 3)
            ESTTS_LAUNCH((attention_bwd_wide_kernel<DH, 2, OWN_Q>), dim3((unsigned)wide), dim3(256), 0, st, a);
        else
            ESTTS_LAUNCH((attention_bwd_wide_kernel<DH, 1, OWN_Q>), dim3((unsigned)wide), dim3(256), 0, st, a);
    } else {
        const dim3 grid((unsigned)((long)((L_own + 15) / 16) * BH));
        if (a.precision == 3)
            ESTTS_LAUNCH((attention_bwd_kernel<DH, 2, OWN_Q>), grid, dim3(256), 0, st, a);
        else
            ESTTS_LAUNCH((attention_bwd_kernel<DH, 1, OWN_Q>), grid, dim3(256), 0, st, a);
    }
}

template <int DH>
int launch(const estts_attention_bwd_args& a, hipStream_t st) {
    launch_side<DH, true>(a, st);
    launch_side<DH, false>(a, st);
    ESTTS_CHECK_LAUNCH("estts_attention_bwd");
    return 0;
}

}  // namespace

extern "C" int estts_attention_bwd(const estts_attention_bwd_args* args, void* stream) {
    ESTTS_CHECK_ARG(args != nullptr, "estts_attention_bwd: null args");
    const estts_attention_bwd_args& a = *args;
    ESTTS_CHECK_ARG(a.q && a.k && a.v && a.ctx && a.dctx && a.attn && a.delta && a.dq && a.dk && a.dv, "estts_attention_bwd: null pointer");
    ESTTS_CHECK_ARG(a.B > 0 && a.H > 0 && a.Lq > 0 && a.Lk > 0 && a.B <= 65535 && a.H <= 65535, "estts_attention_bwd: bad shape");
    ESTTS_CHECK_ARG(a.precision == 1 || a.precision == 3, "estts_attention_bwd: precision must be 1 or 3");
    ESTTS_CHECK_ARG((long)((std::max(a.Lq, a.Lk) + 15) / 16) * a.H * a.B < (1L << 31), "estts_attention_bwd: too many workgroups");
    const int hd = a.H * a.depth;
    ESTTS_CHECK_ARG(((a.ldq | a.ldk | a.ldv | a.ldo | a.lddo | a.lddq | a.lddk | a.lddv) & 3) == 0, "estts_attention_bwd: ld not multiple of 4");
    ESTTS_CHECK_ARG(a.ldq >= hd && a.ldk >= hd && a.ldv >= hd && a.ldo >= hd && a.lddo >= hd && a.lddq >= hd && a.lddk >= hd && a.lddv >= hd,
                    "estts_attention_bwd: ld smaller than H*depth");
    ESTTS_CHECK_ARG((((uintptr_t)a.q | (uintptr_t)a.k | (uintptr_t)a.v | (uintptr_t)a.ctx | (uintptr_t)a.dctx | (uintptr_t)a.attn |
                      (uintptr_t)a.dq | (uintptr_t)a.dk | (uintptr_t)a.dv | (uintptr_t)a.delta) & 15) == 0,
                    "estts_attention_bwd: pointers not 16-byte aligned");
    hipStream_t st = (hipStream_t)stream;
    const long rows = (long)a.B * a.Lq;
    ESTTS_LAUNCH(attention_delta_kernel, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, st, a);
    switch (a.depth) {
        case 32: return launch<32>(a, st);
        case 64: return launch<64>(a, st);
        case 128: return launch<128>(a, st);
        default: estts_set_error("estts_attention_bwd: depth %d unsupported (32, 64, 128)", a.depth); return -1;
    }
}
