// Multi-head scaled-dot-product attention (see estts_attention in include/estts.h).
//
// One wave owns 16 query rows of one (batch, head); a 256-thread workgroup is four such waves.
// Both products are issued "swapped" so that a query stays on one lane column throughout:
//   S^T (16 keys x 16 queries) = K_tile . Q^T     A = K rows (key on lane&15), B = Q^T
//   O^T (16 feats x 16 queries) = V^T . P^T       A = V^T (feature on lane&15), B = P^T
// With v_mfma_f32_16x16x32_bf16 the D layout is D[row 4(l>>4)+j][col l&15], so after S^T every lane
// holds, for its query (l&15), the scores of keys 4g+j (g = l>>4) of each 16-key sub-tile; two
// sub-tiles give the 8 k-slots (8g+j') of the P^T operand without any data movement:
// slot j' <-> key 16*(j'>>2) + 4g + (j'&3).  V^T fragments are gathered with that same key map.
// Softmax is online (running max / sum per query, replicated over the 4 lane groups by two
// xor-shuffles); the optional attention-weight output is a second sweep that recomputes S.
#include "common.h"

namespace {

template <int NS>
struct Frag {
    bf16x8 v[NS];
};

template <int NS>
__device__ __forceinline__ Frag<NS> make_frag(const float* x) {
    Frag<NS> f;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        __bf16 h, l;
        split_bf16(x[j], h, l);
        f.v[0][j] = h;
        if (NS == 2) f.v[NS - 1][j] = l;
    }
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

template <int DH, int NS>
__global__ __launch_bounds__(256) void attention_kernel(const estts_attention_args p) {
    constexpr int KS = DH / 32;  // k-steps of the score product
    constexpr int NT = DH / 16;  // 16-feature tiles of the output
    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    const int i = lane & 15;
    const int g = lane >> 4;
    const int h = blockIdx.y;
    const int b = blockIdx.z;
    const int q0 = (blockIdx.x * 4 + wave) * 16;
    if (q0 >= p.Lq) return;  // whole wave out of range (no barriers in this kernel)

    const int qi = q0 + i;
    const int qrow = min(qi, p.Lq - 1);
    const float* qp = p.q + ((long)b * p.Lq + qrow) * p.ldq + h * DH + 8 * g;
    const float* kbase = p.k + (long)b * p.Lk * p.ldk + h * DH;
    const float* vbase = p.v + (long)b * p.Lk * p.ldv + h * DH;
    const float* mask = p.key_mask ? p.key_mask + (long)b * p.Lk : nullptr;
    const float scale = 1.0f / sqrtf((float)DH);

    Frag<NS> qf[KS];
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
        float t[8];
        *reinterpret_cast<float4*>(t) = *reinterpret_cast<const float4*>(qp + ks * 32);
        *reinterpret_cast<float4*>(t + 4) = *reinterpret_cast<const float4*>(qp + ks * 32 + 4);
        qf[ks] = make_frag<NS>(t);
    }

    // scores of one 32-key tile for this lane's query: s[sub][j] <-> key kt*32 + sub*16 + 4g + j
    auto scores = [&](int kt, float (&s)[2][4]) {
#pragma unroll
        for (int sub = 0; sub < 2; ++sub) {
            const int krow = min(kt * 32 + sub * 16 + i, p.Lk - 1);
            const float* kp = kbase + (long)krow * p.ldk + 8 * g;
            f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) {
                float t[8];
                *reinterpret_cast<float4*>(t) = *reinterpret_cast<const float4*>(kp + ks * 32);
                *reinterpret_cast<float4*>(t + 4) = *reinterpret_cast<const float4*>(kp + ks * 32 + 4);
                const Frag<NS> kf = make_frag<NS>(t);
                acc = mfma_split<NS>(kf, qf[ks], acc);
            }
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int key = kt * 32 + sub * 16 + 4 * g + j;
                float v = acc[j] * scale;
                if (key >= p.Lk) {
                    v = -INFINITY;
                } else {
                    float mk = mask ? mask[key] : 0.0f;
                    if (p.causal && key > qi) mk = 1.0f;
                    v += mk * -1e9f;
                }
                s[sub][j] = v;
            }
        }
    };

    f32x4 o[NT];
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) o[nt] = (f32x4){0.f, 0.f, 0.f, 0.f};
    float m_run = -INFINITY, l_run = 0.0f;
    const int nkt = (p.Lk + 31) / 32;

    for (int kt = 0; kt < nkt; ++kt) {
        float s[2][4];
        scores(kt, s);
        float tmax = -INFINITY;
#pragma unroll
        for (int sub = 0; sub < 2; ++sub)
#pragma unroll
            for (int j = 0; j < 4; ++j) tmax = fmaxf(tmax, s[sub][j]);
        tmax = fmaxf(tmax, __shfl_xor(tmax, 16, 64));
        tmax = fmaxf(tmax, __shfl_xor(tmax, 32, 64));
        const float m_new = fmaxf(m_run, tmax);
        const float alpha = expf(m_run - m_new);  // exp(-inf) = 0 on the first tile
        float pv[8], psum = 0.0f;
#pragma unroll
        for (int sub = 0; sub < 2; ++sub)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const float e = expf(s[sub][j] - m_new);
                pv[sub * 4 + j] = e;
                psum += e;
            }
        psum += __shfl_xor(psum, 16, 64);
        psum += __shfl_xor(psum, 32, 64);
        l_run = l_run * alpha + psum;
        m_run = m_new;
        const Frag<NS> pf = make_frag<NS>(pv);
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
            float t[8];
#pragma unroll
            for (int jp = 0; jp < 8; ++jp) {
                const int key = min(kt * 32 + (jp >> 2) * 16 + 4 * g + (jp & 3), p.Lk - 1);
                t[jp] = vbase[(long)key * p.ldv + nt * 16 + i];
            }
            const Frag<NS> vf = make_frag<NS>(t);
            o[nt] = o[nt] * alpha;
            o[nt] = mfma_split<NS>(vf, pf, o[nt]);
        }
    }

    const float inv_l = 1.0f / l_run;
    if (qi < p.Lq) {
        float* op = p.ctx + ((long)b * p.Lq + qi) * p.ldo + h * DH + 4 * g;
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
            const float4 r = make_float4(o[nt][0] * inv_l, o[nt][1] * inv_l, o[nt][2] * inv_l, o[nt][3] * inv_l);
            *reinterpret_cast<float4*>(op + nt * 16) = r;
        }
    }

    if (p.attn) {
        float* ap = p.attn + (((long)b * p.H + h) * p.Lq + qrow) * p.Lk;
        for (int kt = 0; kt < nkt; ++kt) {
            float s[2][4];
            scores(kt, s);
            if (qi < p.Lq) {
#pragma unroll
                for (int sub = 0; sub < 2; ++sub)
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const int key = kt * 32 + sub * 16 + 4 * g + j;
                        if (key < p.Lk) ap[key] = expf(s[sub][j] - m_run) * inv_l;
                    }
            }
        }
    }
}

template <int DH>
int launch(const estts_attention_args& a, hipStream_t st) {
    dim3 grid((a.Lq + 63) / 64, a.H, a.B);
    if (a.precision == 3)
        hipLaunchKernelGGL((attention_kernel<DH, 2>), grid, dim3(256), 0, st, a);
    else
        hipLaunchKernelGGL((attention_kernel<DH, 1>), grid, dim3(256), 0, st, a);
    ESTTS_CHECK_LAUNCH("estts_attention");
    return 0;
}

}  // namespace

extern "C" int estts_attention(const estts_attention_args* args, void* stream) {
    ESTTS_CHECK_ARG(args != nullptr, "estts_attention: null args");
    const estts_attention_args& a = *args;
    ESTTS_CHECK_ARG(a.q && a.k && a.v && a.ctx, "estts_attention: null q/k/v/ctx");
    ESTTS_CHECK_ARG(a.B > 0 && a.H > 0 && a.Lq > 0 && a.Lk > 0, "estts_attention: bad shape");
    ESTTS_CHECK_ARG(a.B <= 65535 && a.H <= 65535, "estts_attention: B / H exceed grid limits");
    ESTTS_CHECK_ARG(a.precision == 1 || a.precision == 3, "estts_attention: precision must be 1 or 3");
    ESTTS_CHECK_ARG((a.ldq & 3) == 0 && (a.ldk & 3) == 0 && (a.ldo & 3) == 0, "estts_attention: ld not multiple of 4");
    ESTTS_CHECK_ARG(a.ldq >= a.H * a.depth && a.ldk >= a.H * a.depth && a.ldv >= a.H * a.depth && a.ldo >= a.H * a.depth,
                    "estts_attention: ld smaller than H*depth");
    ESTTS_CHECK_ARG((((uintptr_t)a.q | (uintptr_t)a.k | (uintptr_t)a.ctx) & 15) == 0, "estts_attention: q/k/ctx not 16-byte aligned");
    hipStream_t st = (hipStream_t)stream;
    switch (a.depth) {
        case 32: return launch<32>(a, st);
        case 64: return launch<64>(a, st);
        case 128: return launch<128>(a, st);
        default: estts_set_error("estts_attention: depth %d unsupported (32, 64, 128)", a.depth); return -1;
    }
}
