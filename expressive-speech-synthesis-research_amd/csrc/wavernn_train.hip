// WaveRNN training kernels: GRU time step forward / backward with every state kept, the gradient of the conditioning
// assembly (upsampling filter and aux slices), softmax cross-entropy and the discretized mixture-of-logistics loss.
// (see the "WaveRNN training" block of include/estts.h).  All HBM / latency bound; the dense work runs in estts_gemm.
#include "common.h"

namespace {

__device__ __forceinline__ float sigmoidf_(float x) { return 1.0f / (1.0f + expf(-x)); }
__device__ __forceinline__ float softplusf_(float x) { return x > 20.0f ? x : log1pf(expf(x)); }
__device__ __forceinline__ float4 ld4(const float* p) { return *reinterpret_cast<const float4*>(p); }
__device__ __forceinline__ void st4(float* p, float4 v) { *reinterpret_cast<float4*>(p) = v; }
__device__ __forceinline__ float f4e(const float4& v, int e) { return e == 0 ? v.x : (e == 1 ? v.y : (e == 2 ? v.z : v.w)); }

// thread per (row, 4 columns)
__global__ __launch_bounds__(256) void gru_step_fwd_kernel(const float* gi, int ldgi, const float* gh, int ldgh, const float* hp,
                                                           int ldhp, float* ho, int ldho, const float* xr, int ldx, float* out,
                                                           int ldo, int M, int H) {
    const int hq = H >> 2;
    const long total = (long)M * hq;
    for (long idx = (long)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (long)gridDim.x * 256) {
        const int m = (int)(idx / hq);
        const int c = (int)(idx - (long)m * hq) * 4;
        const float* a = gi + (long)m * ldgi + c;
        const float* b = gh + (long)m * ldgh + c;
        const float4 ir = ld4(a), iz = ld4(a + H), in = ld4(a + 2 * H);
        const float4 hr = ld4(b), hz = ld4(b + H), hn = ld4(b + 2 * H);
        const float4 h = ld4(hp + (long)m * ldhp + c);
        float r[4], o[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const float rr = sigmoidf_(f4e(ir, e) + f4e(hr, e));
            const float zz = sigmoidf_(f4e(iz, e) + f4e(hz, e));
            const float nn = tanhf(f4e(in, e) + rr * f4e(hn, e));
            r[e] = (1.0f - zz) * nn + zz * f4e(h, e);
            o[e] = r[e];
        }
        st4(ho + (long)m * ldho + c, make_float4(r[0], r[1], r[2], r[3]));
        if (out) {
            if (xr) {
                const float4 x = ld4(xr + (long)m * ldx + c);
                o[0] += x.x; o[1] += x.y; o[2] += x.z; o[3] += x.w;
            }
            st4(out + (long)m * ldo + c, make_float4(o[0], o[1], o[2], o[3]));
        }
    }
}

__global__ __launch_bounds__(256) void gru_step_bwd_kernel(const float* gi, int ldgi, const float* gh, int ldgh, const float* hp,
                                                           int ldhp, const float* dh, int lddh, const float* dh2, int lddh2,
                                                           float* dgi, int lddgi, float* dgh,
                                                           int lddgh, float* dhp, int lddhp, int M, int H) {
    const int hq = H >> 2;
    const long total = (long)M * hq;
    for (long idx = (long)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (long)gridDim.x * 256) {
        const int m = (int)(idx / hq);
        const int c = (int)(idx - (long)m * hq) * 4;
        const float* a = gi + (long)m * ldgi + c;
        const float* b = gh + (long)m * ldgh + c;
        const float4 ir = ld4(a), iz = ld4(a + H), in = ld4(a + 2 * H);
        const float4 hr = ld4(b), hz = ld4(b + H), hn = ld4(b + 2 * H);
        const float4 h = ld4(hp + (long)m * ldhp + c);
        float4 g = ld4(dh + (long)m * lddh + c);
        if (dh2) {
            const float4 g2 = ld4(dh2 + (long)m * lddh2 + c);
            g.x += g2.x; g.y += g2.y; g.z += g2.z; g.w += g2.w;
        }
        float ar[4], az[4], an[4], anr[4], dp[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const float rr = sigmoidf_(f4e(ir, e) + f4e(hr, e));
            const float zz = sigmoidf_(f4e(iz, e) + f4e(hz, e));
            const float nn = tanhf(f4e(in, e) + rr * f4e(hn, e));
            const float d = f4e(g, e);
            const float dn = d * (1.0f - zz);
            const float dz = d * (f4e(h, e) - nn);
            const float da_n = dn * (1.0f - nn * nn);
            an[e] = da_n;
            anr[e] = da_n * rr;
            ar[e] = da_n * f4e(hn, e) * rr * (1.0f - rr);
            az[e] = dz * zz * (1.0f - zz);
            dp[e] = d * zz;
        }
        float* x = dgi + (long)m * lddgi + c;
        float* y = dgh + (long)m * lddgh + c;
        st4(x, make_float4(ar[0], ar[1], ar[2], ar[3]));
        st4(x + H, make_float4(az[0], az[1], az[2], az[3]));
        st4(x + 2 * H, make_float4(an[0], an[1], an[2], an[3]));
        st4(y, make_float4(ar[0], ar[1], ar[2], ar[3]));
        st4(y + H, make_float4(az[0], az[1], az[2], az[3]));
        st4(y + 2 * H, make_float4(anr[0], anr[1], anr[2], anr[3]));
        st4(dhp + (long)m * lddhp + c, make_float4(dp[0], dp[1], dp[2], dp[3]));
    }
}

// one workgroup per (utterance, frame)
__global__ __launch_bounds__(256) void wavernn_prep_bwd_kernel(const estts_wavernn_prep_bwd_args p) {
    const int uf = blockIdx.x;
    const int u = uf / p.frames;
    const int f = uf - u * p.frames;
    const long row0 = (long)uf * p.hop;
    const int tid = threadIdx.x;
    // aux slices: column c of slice q, summed over the hop rows of this frame
    for (int col = tid; col < 4 * p.d; col += 256) {
        const int q = col / p.d;
        const int c = col - q * p.d;
        const float* src = q == 0 ? p.dxin + 1 + p.feat : (q == 1 ? p.da2 : (q == 2 ? p.da3 : p.da4));
        const int ld = q == 0 ? p.ld_dxin : (q == 1 ? p.ld_da2 : (q == 2 ? p.ld_da3 : p.ld_da4));
        float acc = 0.0f;
        for (int ph = 0; ph < p.hop; ++ph) acc += src[(row0 + ph) * ld + c];
        p.d_aux[(long)uf * 4 * p.d + col] = acc;
    }
    // upsampling filter: one wave per phase, lanes over mel channels
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const long mrow0 = (long)p.mel_off[u] + f + p.pad - p.fir_kmin;
    for (int ph = wave; ph < p.hop; ph += 4) {
        const float* g = p.dxin + (row0 + ph) * p.ld_dxin + 1;
        for (int j = 0; j < p.ntaps; ++j) {
            const float* mr = p.mel + (mrow0 - j) * p.feat;
            float acc = 0.0f;
            for (int c = lane; c < p.feat; c += 64) acc += g[c] * mr[c];
            acc = wave_sum(acc);
            if (lane == 0) atomic_add_f32(p.dfir + ph * p.ntaps + j, acc);
        }
    }
}

// one wave per row
// weight of row m: 1, or with weighted != 0: (label != 0) + (label == w_index) * (w_scaling - 1)
__global__ __launch_bounds__(256) void softmax_xent_kernel(const float* logits, int ld, const int32_t* labels, float* loss,
                                                           float* d, int ldd, int M, int C, float scale, int weighted,
                                                           int w_index, float w_scaling) {
    __shared__ float part[4];
    const int lane = threadIdx.x & 63;
    const int wv = threadIdx.x >> 6;
    const int row = blockIdx.x * 4 + wv;
    float mine = 0.0f;
    if (row < M) {
        const float* l = logits + (long)row * ld;
        float mx = -INFINITY;
        for (int c = lane; c < C; c += 64) mx = fmaxf(mx, l[c]);
        mx = wave_max(mx);
        float s = 0.0f;
        for (int c = lane; c < C; c += 64) s += expf(l[c] - mx);
        s = wave_sum(s);
        const float lse = mx + logf(s);
        const int y = min(max(labels[row], 0), C - 1);   // a label outside [0, C) must not become an address
        const float w = weighted ? (y != 0 ? 1.0f : 0.0f) + (y == w_index ? w_scaling - 1.0f : 0.0f) : 1.0f;
        if (d)
            for (int c = lane; c < C; c += 64) d[(long)row * ldd + c] = scale * w * (expf(l[c] - lse) - (c == y ? 1.0f : 0.0f));
        mine = w * (lse - l[y]);
    }
    if (lane == 0) part[wv] = mine;
    __syncthreads();
    if (threadIdx.x == 0) atomic_add_f32(loss, scale * (part[0] + part[1] + part[2] + part[3]));
}

// one thread per row, nr_mix <= 16
__global__ __launch_bounds__(256) void mol_loss_kernel(const float* yhat, int ld, const float* y, float* loss, float* d, int ldd,
                                                       int M, int nr, float inv_bins, float log_half_bins, float lsm, float scale) {
    __shared__ float part[256];
    const int row = blockIdx.x * 256 + threadIdx.x;
    float nll = 0.0f;
    if (row < M) {
        const float* p = yhat + (long)row * ld;
        const float yy = y[row];
        float u[16], dmu[16], dls[16], sp[16];
        float pmx = -INFINITY;
        for (int k = 0; k < nr; ++k) pmx = fmaxf(pmx, p[k]);
        float ps = 0.0f;
        for (int k = 0; k < nr; ++k) ps += expf(p[k] - pmx);
        const float plse = pmx + logf(ps);
        float umx = -INFINITY;
        for (int k = 0; k < nr; ++k) {
            const float mu = p[nr + k];
            const float ls_raw = p[2 * nr + k];
            const float ls = fmaxf(ls_raw, lsm);
            const float inv = expf(-ls);
            const float c = yy - mu;
            const float plus = inv * (c + inv_bins), mn = inv * (c - inv_bins), mid = inv * c;
            float lp, gmu, gls;
            if (yy < -0.999f) {
                const float s = sigmoidf_(plus);
                lp = plus - softplusf_(plus);
                gmu = -inv * (1.0f - s);
                gls = -plus * (1.0f - s);
            } else if (yy > 0.999f) {
                const float s = sigmoidf_(mn);
                lp = -softplusf_(mn);
                gmu = inv * s;
                gls = mn * s;
            } else {
                const float sp_ = sigmoidf_(plus), sm_ = sigmoidf_(mn);
                const float cdf = sp_ - sm_;
                if (cdf > 1e-5f) {
                    const float dpl = sp_ * (1.0f - sp_), dmn = sm_ * (1.0f - sm_);
                    lp = logf(fmaxf(cdf, 1e-12f));
                    gmu = -inv * (dpl - dmn) / cdf;
                    gls = -(plus * dpl - mn * dmn) / cdf;
                } else {
                    const float s = sigmoidf_(mid);
                    lp = mid - ls - 2.0f * softplusf_(mid) - log_half_bins;
                    gmu = -inv * (1.0f - 2.0f * s);
                    gls = -mid * (1.0f - 2.0f * s) - 1.0f;
                }
            }
            if (ls_raw < lsm) gls = 0.0f;   // clamp(min=log_scale_min) passes no gradient below the bound
            sp[k] = expf(p[k] - plse);
            u[k] = lp + p[k] - plse;
            dmu[k] = gmu;
            dls[k] = gls;
            umx = fmaxf(umx, u[k]);
        }
        float us = 0.0f;
        for (int k = 0; k < nr; ++k) us += expf(u[k] - umx);
        const float L = umx + logf(us);
        nll = -L;
        float* o = d + (long)row * ldd;
        for (int k = 0; k < nr; ++k) {
            const float q = expf(u[k] - L);
            o[k] = -scale * (q - sp[k]);
            o[nr + k] = -scale * q * dmu[k];
            o[2 * nr + k] = -scale * q * dls[k];
        }
    }
    part[threadIdx.x] = nll;
    __syncthreads();
    for (int off = 128; off > 0; off >>= 1) {
        if (threadIdx.x < off) part[threadIdx.x] += part[threadIdx.x + off];
        __syncthreads();
    }
    if (threadIdx.x == 0) atomic_add_f32(loss, scale * part[0]);
}

int grid_for(long total) {
    const long b = (total + 255) / 256;
    return (int)(b < 1 ? 1 : (b > 4096 ? 4096 : b));
}

}  // namespace

extern "C" int estts_gru_step_fwd(const float* gi, int32_t ldgi, const float* gh, int32_t ldgh, const float* h_prev, int32_t ldhp,
                                  float* h_out, int32_t ldho, const float* x_res, int32_t ldx, float* out, int32_t ldo, int32_t M,
                                  int32_t H, void* stream) {
    ESTTS_CHECK_ARG(gi && gh && h_prev && h_out, "estts_gru_step_fwd: null pointer");
    ESTTS_CHECK_ARG(M > 0 && H > 0 && (H & 3) == 0, "estts_gru_step_fwd: H must be a positive multiple of 4");
    ESTTS_CHECK_ARG(ldgi >= 3 * H && ldgh >= 3 * H && ldhp >= H && ldho >= H && (!out || ldo >= H) && (!x_res || ldx >= H),
                    "estts_gru_step_fwd: ld too small");
    ESTTS_CHECK_ARG(((ldgi | ldgh | ldhp | ldho | (out ? ldo : 0) | (x_res ? ldx : 0)) & 3) == 0, "estts_gru_step_fwd: ld not multiple of 4");
    ESTTS_LAUNCH(gru_step_fwd_kernel, dim3(grid_for((long)M * (H >> 2))), dim3(256), 0, (hipStream_t)stream, gi, ldgi, gh, ldgh,
                       h_prev, ldhp, h_out, ldho, x_res, ldx, out, ldo, M, H);
    ESTTS_CHECK_LAUNCH("estts_gru_step_fwd");
    return 0;
}

extern "C" int estts_gru_step_bwd(const float* gi, int32_t ldgi, const float* gh, int32_t ldgh, const float* h_prev, int32_t ldhp,
                                  const float* dh, int32_t lddh, const float* dh_rec, int32_t lddhr, float* dgi, int32_t lddgi,
                                  float* dgh, int32_t lddgh, float* dh_prev, int32_t lddhp, int32_t M, int32_t H, void* stream) {
    ESTTS_CHECK_ARG(gi && gh && h_prev && dh && dgi && dgh && dh_prev, "estts_gru_step_bwd: null pointer");
    ESTTS_CHECK_ARG(M > 0 && H > 0 && (H & 3) == 0, "estts_gru_step_bwd: H must be a positive multiple of 4");
    ESTTS_CHECK_ARG(ldgi >= 3 * H && ldgh >= 3 * H && lddgi >= 3 * H && lddgh >= 3 * H && ldhp >= H && lddh >= H && lddhp >= H,
                    "estts_gru_step_bwd: ld too small");
    ESTTS_CHECK_ARG(((ldgi | ldgh | ldhp | lddh | lddgi | lddgh | lddhp | (dh_rec ? lddhr : 0)) & 3) == 0, "estts_gru_step_bwd: ld not multiple of 4");
    ESTTS_CHECK_ARG(!dh_rec || lddhr >= H, "estts_gru_step_bwd: ld too small");
    ESTTS_LAUNCH(gru_step_bwd_kernel, dim3(grid_for((long)M * (H >> 2))), dim3(256), 0, (hipStream_t)stream, gi, ldgi, gh, ldgh,
                       h_prev, ldhp, dh, lddh, dh_rec, lddhr, dgi, lddgi, dgh, lddgh, dh_prev, lddhp, M, H);
    ESTTS_CHECK_LAUNCH("estts_gru_step_bwd");
    return 0;
}

extern "C" int estts_wavernn_prep_bwd(const estts_wavernn_prep_bwd_args* args, void* stream) {
    ESTTS_CHECK_ARG(args != nullptr, "estts_wavernn_prep_bwd: null args");
    const estts_wavernn_prep_bwd_args& a = *args;
    ESTTS_CHECK_ARG(a.dxin && a.da2 && a.da3 && a.da4 && a.mel && a.mel_off && a.d_aux && a.dfir, "estts_wavernn_prep_bwd: null pointer");
    ESTTS_CHECK_ARG(a.n_utt > 0 && a.frames > 0 && a.feat > 0 && a.d > 0 && a.hop > 0 && a.ntaps > 0 && a.pad >= 0,
                    "estts_wavernn_prep_bwd: bad shape");
    ESTTS_CHECK_ARG(a.pad - a.fir_kmin - (a.ntaps - 1) >= 0 && a.pad + a.fir_kmin >= 0, "estts_wavernn_prep_bwd: pad too small for the filter");
    ESTTS_CHECK_ARG(a.ld_dxin >= 1 + a.feat + a.d && a.ld_da2 >= a.d && a.ld_da3 >= a.d && a.ld_da4 >= a.d, "estts_wavernn_prep_bwd: ld too small");
    ESTTS_LAUNCH(wavernn_prep_bwd_kernel, dim3(a.n_utt * a.frames), dim3(256), 0, (hipStream_t)stream, a);
    ESTTS_CHECK_LAUNCH("estts_wavernn_prep_bwd");
    return 0;
}

extern "C" int estts_softmax_xent(const float* logits, int32_t ld, const int32_t* labels, float* loss, float* dlogits, int32_t lddl,
                                  int32_t M, int32_t C, float scale, void* stream) {
    ESTTS_CHECK_ARG(logits && labels && loss && dlogits && M > 0 && C > 0 && ld >= C && lddl >= C, "estts_softmax_xent: bad arguments");
    ESTTS_LAUNCH(softmax_xent_kernel, dim3((M + 3) / 4), dim3(256), 0, (hipStream_t)stream, logits, ld, labels, loss, dlogits,
                       lddl, M, C, scale, 0, 0, 1.0f);
    ESTTS_CHECK_LAUNCH("estts_softmax_xent");
    return 0;
}

extern "C" int estts_scaled_xent(const float* logits, int32_t ld, const int32_t* labels, float* loss, float* dlogits, int32_t lddl,
                                 int32_t M, int32_t C, int32_t index, float scaling, float scale, void* stream) {
    ESTTS_CHECK_ARG(logits && labels && loss && M > 0 && C > 0 && ld >= C && (!dlogits || lddl >= C), "estts_scaled_xent: bad arguments");
    ESTTS_LAUNCH(softmax_xent_kernel, dim3((M + 3) / 4), dim3(256), 0, (hipStream_t)stream, logits, ld, labels, loss, dlogits,
                       lddl, M, C, scale, 1, index, scaling);
    ESTTS_CHECK_LAUNCH("estts_scaled_xent");
    return 0;
}

extern "C" int estts_mol_loss(const float* y_hat, int32_t ld, const float* y, float* loss, float* d, int32_t ldd, int32_t M,
                              int32_t nr_mix, int32_t num_classes, float log_scale_min, float scale, void* stream) {
    ESTTS_CHECK_ARG(y_hat && y && loss && d && M > 0, "estts_mol_loss: bad arguments");
    ESTTS_CHECK_ARG(nr_mix > 0 && nr_mix <= 16 && ld >= 3 * nr_mix && ldd >= 3 * nr_mix && num_classes > 1, "estts_mol_loss: nr_mix must be 1..16");
    ESTTS_LAUNCH(mol_loss_kernel, dim3((M + 255) / 256), dim3(256), 0, (hipStream_t)stream, y_hat, ld, y, loss, d, ldd, M, nr_mix,
                       1.0f / (float)(num_classes - 1), logf((float)(num_classes - 1) * 0.5f), log_scale_min, scale);
    ESTTS_CHECK_LAUNCH("estts_mol_loss");
    return 0;
}
