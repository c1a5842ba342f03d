// HBM-bound row kernels: embedding + positional encoding, LayerNorm combos, length regulator.
// All of them move float4 per lane, one wave per row where a row reduction is needed.
#include "common.h"

#include <string.h>

static thread_local char g_err[512] = "";

void estts_set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

extern "C" const char* estts_last_error(void) { return g_err; }
extern "C" int estts_abi_version(void) { return ESTTS_ABI_VERSION; }
long long g_estts_launches = 0;
extern "C" long long estts_launch_count(void) { return __atomic_load_n(&g_estts_launches, __ATOMIC_RELAXED); }

namespace {

// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void embed_posenc_kernel(const int32_t* tokens, const float* emb, const float* pe,
                                                           const float* pos_scalar, float* out, int ldo,
                                                           float* key_mask, int rows, int N, int D, float scale,
                                                           int pe_stride) {
    const int dq = D >> 2;
    const long total = (long)rows * dq;
    const float ps = pos_scalar ? pos_scalar[0] : 1.0f;
    for (long idx = (long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long)gridDim.x * blockDim.x) {
        const int row = (int)(idx / dq);
        const int c = (int)(idx - (long)row * dq) * 4;
        const int n = row % N;
        const int tok = tokens[row];
        const float4 e = *reinterpret_cast<const float4*>(emb + (long)tok * D + c);
        const float4 q = *reinterpret_cast<const float4*>(pe + (long)n * pe_stride * D + c);
        float4 r;
        r.x = e.x * scale + ps * q.x;
        r.y = e.y * scale + ps * q.y;
        r.z = e.z * scale + ps * q.z;
        r.w = e.w * scale + ps * q.w;
        *reinterpret_cast<float4*>(out + (long)row * ldo + c) = r;
        if (key_mask && c == 0) key_mask[row] = tok == 0 ? 1.0f : 0.0f;
    }
}

// ------------------------------------------------------------------------------------------
// out = LNb(dropout(act(LNa(y))) + x), one wave per row, up to 4 float4 per lane (D <= 1024)
__device__ __forceinline__ void ln_inplace(float4 (&v)[4], int nch, int lane, int dq, int D, const float* gamma,
                                           const float* beta, float eps) {
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 4; ++i)
        if (i < nch && lane + 64 * i < dq) s += v[i].x + v[i].y + v[i].z + v[i].w;
    const float mean = wave_sum(s) / (float)D;
    float q = 0.f;
#pragma unroll
    for (int i = 0; i < 4; ++i)
        if (i < nch && lane + 64 * i < dq) {
            const float a = v[i].x - mean, b = v[i].y - mean, c = v[i].z - mean, d = v[i].w - mean;
            q += a * a + b * b + c * c + d * d;
        }
    const float inv = rsqrtf(wave_sum(q) / (float)D + eps);
#pragma unroll
    for (int i = 0; i < 4; ++i)
        if (i < nch && lane + 64 * i < dq) {
            const int c = (lane + 64 * i) * 4;
            const float4 ga = *reinterpret_cast<const float4*>(gamma + c);
            const float4 be = *reinterpret_cast<const float4*>(beta + c);
            v[i].x = (v[i].x - mean) * inv * ga.x + be.x;
            v[i].y = (v[i].y - mean) * inv * ga.y + be.y;
            v[i].z = (v[i].z - mean) * inv * ga.z + be.z;
            v[i].w = (v[i].w - mean) * inv * ga.w + be.w;
        }
}

__global__ __launch_bounds__(256) void rownorm_kernel(const estts_rownorm_args p) {
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= p.M) return;
    const int dq = p.D >> 2;
    const int nch = (dq + 63) >> 6;
    float4 v[4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
        if (i < nch && lane + 64 * i < dq)
            v[i] = *reinterpret_cast<const float4*>(p.y + (long)row * p.ldy + (lane + 64 * i) * 4);
    if (p.ga) ln_inplace(v, nch, lane, dq, p.D, p.ga, p.ba, p.eps);
#pragma unroll
    for (int i = 0; i < 4; ++i)
        if (i < nch && lane + 64 * i < dq) {
            v[i].x = apply_act(v[i].x, p.act);
            v[i].y = apply_act(v[i].y, p.act);
            v[i].z = apply_act(v[i].z, p.act);
            v[i].w = apply_act(v[i].w, p.act);
            if (p.drop_rate > 0.0f) {
                const int c = (lane + 64 * i) * 4;
                const float keep = 1.0f / (1.0f - p.drop_rate);
                v[i].x = dropout_keep(p.drop_seed, p.drop_site, row, c, p.drop_rate) ? v[i].x * keep : 0.0f;
                v[i].y = dropout_keep(p.drop_seed, p.drop_site, row, c + 1, p.drop_rate) ? v[i].y * keep : 0.0f;
                v[i].z = dropout_keep(p.drop_seed, p.drop_site, row, c + 2, p.drop_rate) ? v[i].z * keep : 0.0f;
                v[i].w = dropout_keep(p.drop_seed, p.drop_site, row, c + 3, p.drop_rate) ? v[i].w * keep : 0.0f;
            }
            if (p.x) {
                const float4 r = *reinterpret_cast<const float4*>(p.x + (long)row * p.ldx + (lane + 64 * i) * 4);
                v[i].x += r.x;
                v[i].y += r.y;
                v[i].z += r.z;
                v[i].w += r.w;
            }
        }
    if (p.gb) ln_inplace(v, nch, lane, dq, p.D, p.gb, p.bb, p.eps);
    float dot = 0.f;
#pragma unroll
    for (int i = 0; i < 4; ++i)
        if (i < nch && lane + 64 * i < dq) {
            const int c = (lane + 64 * i) * 4;
            if (p.out) *reinterpret_cast<float4*>(p.out + (long)row * p.ldo + c) = v[i];
            if (p.dot_w) {
                const float4 w = *reinterpret_cast<const float4*>(p.dot_w + c);
                dot += v[i].x * w.x + v[i].y * w.y + v[i].z * w.z + v[i].w * w.w;
            }
        }
    if (p.dot_w) {
        dot = wave_sum(dot);
        if (lane == 0) {
            float d = fmaxf(dot + (p.dot_b_ptr ? p.dot_b_ptr[0] : p.dot_b), 0.0f) * p.dot_scale;
            if (p.dot_mask) d *= (1.0f - p.dot_mask[row]);
            p.dot_out[row] = d;
        }
    }
}

// The same for D = 256 NCH with every operand of the row requested before the first reduction: in the general kernel above the
// residual row and the two pairs of scale / shift vectors are loaded where they are used, behind the reductions, and each of
// those round trips (~0.5-1 us) lies on the critical path of a launch that is a few microseconds long at one utterance.
// RPW rows per wave (thousands of rows: two rows in flight per wave hide the reductions of one behind the loads of the other).
template <int NCH, bool LNA, bool RES, bool LNB, int RPW>
__global__ __launch_bounds__(256) void rownorm_fast_kernel(const estts_rownorm_args p) {
    const int lane = threadIdx.x & 63;
    const int row0 = (blockIdx.x * 4 + (threadIdx.x >> 6)) * RPW;
    if (row0 >= p.M) return;
    float4 v[RPW][NCH], r[RPW][NCH], ga[NCH], ba[NCH], gb[NCH], bb[NCH];
    int rows[RPW];
#pragma unroll
    for (int q = 0; q < RPW; ++q) rows[q] = min(row0 + q, p.M - 1);     // a surplus slot repeats the last row (same values, same place)
#pragma unroll
    for (int i = 0; i < NCH; ++i) {
        const int c = (lane + 64 * i) * 4;
#pragma unroll
        for (int q = 0; q < RPW; ++q) {
            v[q][i] = *reinterpret_cast<const float4*>(p.y + (long)rows[q] * p.ldy + c);
            if (RES) r[q][i] = *reinterpret_cast<const float4*>(p.x + (long)rows[q] * p.ldx + c);
        }
        if (LNA) {
            ga[i] = *reinterpret_cast<const float4*>(p.ga + c);
            ba[i] = *reinterpret_cast<const float4*>(p.ba + c);
        }
        if (LNB) {
            gb[i] = *reinterpret_cast<const float4*>(p.gb + c);
            bb[i] = *reinterpret_cast<const float4*>(p.bb + c);
        }
    }
    auto layer_norm = [&](const float4 (&g)[NCH], const float4 (&b)[NCH]) {
        float mean[RPW], inv[RPW];
#pragma unroll
        for (int q = 0; q < RPW; ++q) {
            float s = 0.f;
#pragma unroll
            for (int i = 0; i < NCH; ++i) s += v[q][i].x + v[q][i].y + v[q][i].z + v[q][i].w;
            mean[q] = s;
        }
#pragma unroll
        for (int q = 0; q < RPW; ++q) mean[q] = wave_sum(mean[q]) / (float)p.D;
#pragma unroll
        for (int q = 0; q < RPW; ++q) {
            float t = 0.f;
#pragma unroll
            for (int i = 0; i < NCH; ++i) {
                const float a = v[q][i].x - mean[q], b2 = v[q][i].y - mean[q], c = v[q][i].z - mean[q], d = v[q][i].w - mean[q];
                t += a * a + b2 * b2 + c * c + d * d;
            }
            inv[q] = t;
        }
#pragma unroll
        for (int q = 0; q < RPW; ++q) inv[q] = rsqrtf(wave_sum(inv[q]) / (float)p.D + p.eps);
#pragma unroll
        for (int q = 0; q < RPW; ++q)
#pragma unroll
            for (int i = 0; i < NCH; ++i) {
                v[q][i].x = (v[q][i].x - mean[q]) * inv[q] * g[i].x + b[i].x;
                v[q][i].y = (v[q][i].y - mean[q]) * inv[q] * g[i].y + b[i].y;
                v[q][i].z = (v[q][i].z - mean[q]) * inv[q] * g[i].z + b[i].z;
                v[q][i].w = (v[q][i].w - mean[q]) * inv[q] * g[i].w + b[i].w;
            }
    };
    if (LNA) layer_norm(ga, ba);
#pragma unroll
    for (int q = 0; q < RPW; ++q)
#pragma unroll
        for (int i = 0; i < NCH; ++i) {
            v[q][i].x = apply_act(v[q][i].x, p.act);
            v[q][i].y = apply_act(v[q][i].y, p.act);
            v[q][i].z = apply_act(v[q][i].z, p.act);
            v[q][i].w = apply_act(v[q][i].w, p.act);
            if (RES) {
                v[q][i].x += r[q][i].x;
                v[q][i].y += r[q][i].y;
                v[q][i].z += r[q][i].z;
                v[q][i].w += r[q][i].w;
            }
        }
    if (LNB) layer_norm(gb, bb);
#pragma unroll
    for (int q = 0; q < RPW; ++q)
#pragma unroll
        for (int i = 0; i < NCH; ++i) *reinterpret_cast<float4*>(p.out + (long)rows[q] * p.ldo + (lane + 64 * i) * 4) = v[q][i];
}

template <int NCH, int RPW>
bool launch_rownorm_fast(const estts_rownorm_args& a, hipStream_t st) {
    const dim3 grid((a.M + 4 * RPW - 1) / (4 * RPW));
    const int key = (a.ga ? 4 : 0) | (a.x ? 2 : 0) | (a.gb ? 1 : 0);
    switch (key) {
        case 7: ESTTS_LAUNCH((rownorm_fast_kernel<NCH, true, true, true, RPW>), grid, dim3(256), 0, st, a); return true;
        case 3: ESTTS_LAUNCH((rownorm_fast_kernel<NCH, false, true, true, RPW>), grid, dim3(256), 0, st, a); return true;
        case 6: ESTTS_LAUNCH((rownorm_fast_kernel<NCH, true, true, false, RPW>), grid, dim3(256), 0, st, a); return true;
        case 4: ESTTS_LAUNCH((rownorm_fast_kernel<NCH, true, false, false, RPW>), grid, dim3(256), 0, st, a); return true;
        default: return false;
    }
}

// ------------------------------------------------------------------------------------------
// inclusive prefix sum of max(rint(dur), 0) along N, one 256-thread workgroup per batch row
__global__ __launch_bounds__(256) void durations_cumsum_kernel(const float* dur, int32_t* cum, int32_t* totals, int N) {
    __shared__ int part[256];
    const int b = blockIdx.x;
    const int tid = threadIdx.x;
    const int per = (N + 255) / 256;
    const int lo = tid * per;
    const int hi = min(lo + per, N);
    int s = 0;
    for (int n = lo; n < hi; ++n) s += max((int)rintf(dur[(long)b * N + n]), 0);
    part[tid] = s;
    __syncthreads();
    for (int off = 1; off < 256; off <<= 1) {
        const int v = tid >= off ? part[tid - off] : 0;
        __syncthreads();
        part[tid] += v;
        __syncthreads();
    }
    int run = part[tid] - s;  // exclusive prefix of this thread's span
    for (int n = lo; n < hi; ++n) {
        run += max((int)rintf(dur[(long)b * N + n]), 0);
        cum[(long)b * N + n] = run;
    }
    if (tid == 255) totals[b] = part[255];
}

// one wave per output frame
__global__ __launch_bounds__(256) void expand_kernel(const float* x, int ldx, const int32_t* cum, float* out, int ldo,
                                                     float* frame_mask, int B, int N, int T, int D) {
    const int lane = threadIdx.x & 63;
    const long frame = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (frame >= (long)B * T) return;
    const int b = (int)(frame / T);
    const int t = (int)(frame - (long)b * T);
    const int32_t* c = cum + (long)b * N;
    const int total = c[N - 1];
    float* o = out + frame * ldo;
    const int dq = D >> 2;
    if (t >= total) {
        for (int ch = lane; ch < dq; ch += 64) *reinterpret_cast<float4*>(o + ch * 4) = make_float4(0.f, 0.f, 0.f, 0.f);
        if (frame_mask && lane == 0) frame_mask[frame] = 1.0f;
        return;
    }
    int lo = 0, hi = N - 1;  // smallest n with cum[n] > t
    while (lo < hi) {
        const int mid = (lo + hi) >> 1;
        if (c[mid] > t) hi = mid; else lo = mid + 1;
    }
    const float* src = x + ((long)b * N + lo) * ldx;
    for (int ch = lane; ch < dq; ch += 64)
        *reinterpret_cast<float4*>(o + ch * 4) = *reinterpret_cast<const float4*>(src + ch * 4);
    if (frame_mask && lane == 0) frame_mask[frame] = 0.0f;
}

// ------------------------------------------------------------------------------------------
// 3 x 3, one input channel: one thread per (output pixel, 4 output channels)
__global__ __launch_bounds__(256) void conv2d_c1_kernel(const float* x, const float* w, const float* scale, const float* bias,
                                                        float* out, int B, int H, int W, int Ho, int Wo, int Cout, int stride,
                                                        int pad_t, int pad_l) {
    const int cq = Cout >> 2;
    const int total = B * Ho * Wo * cq;     // checked by the launcher to fit 31 bits
    for (int idx = blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += gridDim.x * blockDim.x) {
        const int c = (idx % cq) * 4;
        int pix = idx / cq;
        const int wo = pix % Wo;
        pix /= Wo;
        const int ho = pix % Ho;
        const int b = pix / Ho;
        // all nine taps are requested before the first is used, from clamped positions, and zeroed where the tap lies
        // outside the image: a load under a condition becomes a branch with a wait of its own
        float v[9];
        float4 wv[9];
#pragma unroll
        for (int kh = 0; kh < 3; ++kh) {
            const int hi = ho * stride + kh - pad_t;
#pragma unroll
            for (int kw = 0; kw < 3; ++kw) {
                const int wi = wo * stride + kw - pad_l;
                const bool ok = hi >= 0 && hi < H && wi >= 0 && wi < W;
                const float t = x[((long)b * H + min(max(hi, 0), H - 1)) * W + min(max(wi, 0), W - 1)];
                v[kh * 3 + kw] = ok ? t : 0.0f;
                wv[kh * 3 + kw] = *reinterpret_cast<const float4*>(w + (kh * 3 + kw) * Cout + c);
            }
        }
        float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
        for (int k = 0; k < 9; ++k) {
            acc.x += v[k] * wv[k].x; acc.y += v[k] * wv[k].y; acc.z += v[k] * wv[k].z; acc.w += v[k] * wv[k].w;
        }
        const float4 s = *reinterpret_cast<const float4*>(scale + c);
        const float4 bb = *reinterpret_cast<const float4*>(bias + c);
        float4 r;
        r.x = fmaxf(acc.x * s.x + bb.x, 0.f); r.y = fmaxf(acc.y * s.y + bb.y, 0.f);
        r.z = fmaxf(acc.z * s.z + bb.z, 0.f); r.w = fmaxf(acc.w * s.w + bb.w, 0.f);
        *reinterpret_cast<float4*>(out + (((long)b * Ho + ho) * Wo + wo) * Cout + c) = r;
    }
}

// one wave per row
__global__ __launch_bounds__(256) void row_zero_mask_kernel(const float* x, int ldx, float* mask, int M, int D) {
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= M) return;
    float s = 0.f;
    for (int c = lane; c < D; c += 64) s += fabsf(x[(long)row * ldx + c]);
    s = wave_sum(s);
    if (lane == 0) mask[row] = s == 0.0f ? 1.0f : 0.0f;
}

// one wave per row; one atomic per workgroup
__global__ __launch_bounds__(256) void masked_mae_kernel(const float* target, int ldt, const float* pred, int ldp, float* out,
                                                         int M, int C) {
    __shared__ float part[4];
    const int lane = threadIdx.x & 63;
    const int wv = threadIdx.x >> 6;
    // rows are strided over a bounded grid: every workgroup ends in one atomic on the same address, and thousands of those
    // serialise (4800 of them made this a 38 us kernel at 19200 x 80)
    float v = 0.f;
    for (int row = blockIdx.x * 4 + wv; row < M; row += gridDim.x * 4) {
        float s = 0.f, nz = 0.f;
        for (int c = lane; c < C; c += 64) {
            const float t = target[(long)row * ldt + c];
            s += fabsf(t - pred[(long)row * ldp + c]);
            nz += t != 0.0f ? 1.0f : 0.0f;
        }
        s = wave_sum(s);
        nz = wave_sum(nz);
        v += nz > 0.f ? s / (float)C : 0.f;
    }
    if (lane == 0) part[wv] = v;
    __syncthreads();
    if (threadIdx.x == 0) atomicAdd(out, (part[0] + part[1] + part[2] + part[3]) / (float)M);
}

}  // namespace

extern "C" int estts_embed_posenc(const int32_t* tokens, const float* emb, const float* pe, const float* pos_scalar,
                                  float* out, int32_t ldo, float* key_mask, int32_t B, int32_t N, int32_t D,
                                  float scale, int32_t pe_stride, void* stream) {
    ESTTS_CHECK_ARG(tokens && emb && pe && out, "estts_embed_posenc: null pointer");
    ESTTS_CHECK_ARG(B > 0 && N > 0 && D > 0 && (D & 3) == 0 && (ldo & 3) == 0 && ldo >= D, "estts_embed_posenc: bad shape");
    const long total = (long)B * N * (D >> 2);
    const int grid = (int)((total + 255) / 256 < 2048 ? (total + 255) / 256 : 2048);
    ESTTS_LAUNCH(embed_posenc_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, tokens, emb, pe, pos_scalar,
                       out, ldo, key_mask, B * N, N, D, scale, pe_stride);
    ESTTS_CHECK_LAUNCH("estts_embed_posenc");
    return 0;
}

extern "C" int estts_rownorm(const estts_rownorm_args* args, void* stream) {
    ESTTS_CHECK_ARG(args != nullptr, "estts_rownorm: null args");
    const estts_rownorm_args& a = *args;
    ESTTS_CHECK_ARG(a.y && (a.out || a.dot_w), "estts_rownorm: null y/out");
    ESTTS_CHECK_ARG(a.M > 0 && a.D > 0 && a.D <= 1024 && (a.D & 3) == 0, "estts_rownorm: D=%d must be <=1024, multiple of 4", a.D);
    ESTTS_CHECK_ARG((a.ldy & 3) == 0 && (!a.x || (a.ldx & 3) == 0) && (!a.out || (a.ldo & 3) == 0), "estts_rownorm: ld not multiple of 4");
    ESTTS_CHECK_ARG(!a.ga || a.ba, "estts_rownorm: ga without ba");
    ESTTS_CHECK_ARG(!a.gb || a.bb, "estts_rownorm: gb without bb");
    ESTTS_CHECK_ARG(!a.dot_w || a.dot_out, "estts_rownorm: dot_w without dot_out");
    ESTTS_CHECK_ARG(a.drop_rate >= 0.0f && a.drop_rate < 1.0f, "estts_rownorm: drop_rate must be in [0, 1)");
    const dim3 grid((a.M + 3) / 4);
    // inference shapes: the variant with every operand requested up front (3.4 vs 4.7 us at 100 rows; 11.6 vs 14.5 us,
    // 5.1 TB/s, at 19200 rows of 256)
    bool done = false;
    if (a.drop_rate == 0.0f && !a.dot_w && a.out) {
        const bool many = a.M >= 4096;      // two rows per wave: 11.1 -> 10.1 us at 19200 rows of 256 (5.8 TB/s)
        if (a.D == 256) done = many ? launch_rownorm_fast<1, 2>(a, (hipStream_t)stream) : launch_rownorm_fast<1, 1>(a, (hipStream_t)stream);
        else if (a.D == 512) done = many ? launch_rownorm_fast<2, 2>(a, (hipStream_t)stream) : launch_rownorm_fast<2, 1>(a, (hipStream_t)stream);
    }
    if (!done) ESTTS_LAUNCH(rownorm_kernel, grid, dim3(256), 0, (hipStream_t)stream, a);
    ESTTS_CHECK_LAUNCH("estts_rownorm");
    return 0;
}

extern "C" int estts_durations_cumsum(const float* dur, int32_t* cum, int32_t* totals, int32_t B, int32_t N, void* stream) {
    ESTTS_CHECK_ARG(dur && cum && totals && B > 0 && N > 0, "estts_durations_cumsum: bad arguments");
    ESTTS_LAUNCH(durations_cumsum_kernel, dim3(B), dim3(256), 0, (hipStream_t)stream, dur, cum, totals, N);
    ESTTS_CHECK_LAUNCH("estts_durations_cumsum");
    return 0;
}

extern "C" int estts_expand(const float* x, int32_t ldx, const int32_t* cum, float* out, int32_t ldo, float* frame_mask,
                            int32_t B, int32_t N, int32_t T, int32_t D, void* stream) {
    ESTTS_CHECK_ARG(x && cum && out && B > 0 && N > 0 && T > 0 && D > 0, "estts_expand: bad arguments");
    ESTTS_CHECK_ARG((D & 3) == 0 && (ldx & 3) == 0 && (ldo & 3) == 0 && ldx >= D && ldo >= D, "estts_expand: D / ld must be multiples of 4");
    const long frames = (long)B * T;
    ESTTS_LAUNCH(expand_kernel, dim3((unsigned)((frames + 3) / 4)), dim3(256), 0, (hipStream_t)stream, x, ldx, cum,
                       out, ldo, frame_mask, B, N, T, D);
    ESTTS_CHECK_LAUNCH("estts_expand");
    return 0;
}

extern "C" int estts_conv2d_c1(const float* x, const float* w, const float* scale, const float* bias, float* out, int32_t B,
                               int32_t H, int32_t W, int32_t Ho, int32_t Wo, int32_t Cout, int32_t stride, int32_t pad_t,
                               int32_t pad_l, void* stream) {
    ESTTS_CHECK_ARG(x && w && scale && bias && out, "estts_conv2d_c1: null pointer");
    ESTTS_CHECK_ARG(B > 0 && H > 0 && W > 0 && Ho > 0 && Wo > 0 && Cout > 0 && (Cout & 3) == 0 && stride > 0,
                    "estts_conv2d_c1: bad shape (Cout must be a multiple of 4)");
    ESTTS_CHECK_ARG((Ho - 1) * stride - pad_t < H && (Wo - 1) * stride - pad_l < W, "estts_conv2d_c1: output grid leaves the image");
    const long total = (long)B * Ho * Wo * (Cout >> 2);
    ESTTS_CHECK_ARG(total < (1L << 31) - (1L << 21), "estts_conv2d_c1: too many outputs");
    const int grid = (int)((total + 255) / 256 < 4096 ? (total + 255) / 256 : 4096);
    ESTTS_LAUNCH(conv2d_c1_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, x, w, scale, bias, out, B, H, W, Ho,
                       Wo, Cout, stride, pad_t, pad_l);
    ESTTS_CHECK_LAUNCH("estts_conv2d_c1");
    return 0;
}

extern "C" int estts_row_zero_mask(const float* x, int32_t ldx, float* mask, int32_t M, int32_t D, void* stream) {
    ESTTS_CHECK_ARG(x && mask && M > 0 && D > 0 && ldx >= D, "estts_row_zero_mask: bad arguments");
    ESTTS_LAUNCH(row_zero_mask_kernel, dim3((M + 3) / 4), dim3(256), 0, (hipStream_t)stream, x, ldx, mask, M, D);
    ESTTS_CHECK_LAUNCH("estts_row_zero_mask");
    return 0;
}

extern "C" int estts_masked_mae(const float* target, int32_t ldt, const float* pred, int32_t ldp, float* out, int32_t M,
                                int32_t C, void* stream) {
    ESTTS_CHECK_ARG(target && pred && out && M > 0 && C > 0 && ldt >= C && ldp >= C, "estts_masked_mae: bad arguments");
    const int wgs = (M + 3) / 4;
    ESTTS_LAUNCH(masked_mae_kernel, dim3(wgs < 256 ? wgs : 256), dim3(256), 0, (hipStream_t)stream, target, ldt, pred, ldp, out, M, C);
    ESTTS_CHECK_LAUNCH("estts_masked_mae");
    return 0;
}


// ------------------------------------------------------------------------------------------
// step programs (see estts_run_program in include/estts.h): host-side loop, no kernel of its own
extern "C" int estts_run_program(const estts_prog_op* ops, int32_t n_ops, int32_t step, void* stream) {
    ESTTS_CHECK_ARG(ops != nullptr && n_ops > 0 && step >= 0, "estts_run_program: bad arguments");
    union {
        estts_gemm_args g;
        estts_attention_args a;
        estts_rownorm_args r;
        estts_rowmask_args m;
        estts_cross_row_args c;
        estts_decoder_block_args d;
        estts_prenet_row_args p;
    } blk;
    static const size_t sizes[7] = {sizeof(estts_gemm_args), sizeof(estts_attention_args), sizeof(estts_rownorm_args),
                                    sizeof(estts_rowmask_args), sizeof(estts_cross_row_args), sizeof(estts_decoder_block_args),
                                    sizeof(estts_prenet_row_args)};
    for (int i = 0; i < n_ops; ++i) {
        const estts_prog_op& op = ops[i];
        ESTTS_CHECK_ARG(op.kind >= 0 && op.kind < 7 && op.args && op.n_patch >= 0 && op.n_patch <= ESTTS_PROG_MAX_PATCH,
                        "estts_run_program: op %d is malformed", i);
        memcpy(&blk, op.args, sizes[op.kind]);
        char* base = reinterpret_cast<char*>(&blk);
        for (int k = 0; k < op.n_patch; ++k) {
            const int off = op.patch_offset[k];
            ESTTS_CHECK_ARG(off >= 0 && (size_t)off + (op.patch_is_ptr[k] ? 8 : 4) <= sizes[op.kind], "estts_run_program: op %d patch %d out of range", i, k);
            if (op.patch_is_ptr[k]) {
                char* ptr;
                memcpy(&ptr, base + off, sizeof(ptr));
                ptr += (int64_t)step * op.patch_stride[k];
                memcpy(base + off, &ptr, sizeof(ptr));
            } else {
                int32_t v;
                memcpy(&v, base + off, sizeof(v));
                v += (int32_t)((int64_t)step * op.patch_stride[k]);
                memcpy(base + off, &v, sizeof(v));
            }
        }
        int rc = 0;
        switch (op.kind) {
            case 0: rc = estts_gemm(&blk.g, stream); break;
            case 1: rc = estts_attention(&blk.a, stream); break;
            case 2: rc = estts_rownorm(&blk.r, stream); break;
            case 4: rc = estts_cross_attention_row(&blk.c, stream); break;
            case 5: rc = estts_decoder_block_row(&blk.d, stream); break;
            case 6: rc = estts_prenet_row(&blk.p, stream); break;
            default: rc = estts_row_zero_mask(blk.m.x, blk.m.ldx, blk.m.mask, blk.m.M, blk.m.D, stream); break;
        }
        if (rc != 0) return rc;
    }
    return 0;
}
