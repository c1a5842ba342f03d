// WaveRNN per-sample step kernels (see the estts_wavernn_* / estts_gru_gates entries of include/estts.h).
// The dense layers of a step run through estts_gemm; these kernels are the HBM/latency-bound glue around
// them: conditioning assembly (upsampling filter applied on the fly), GRU gate math, sampling.
#include "common.h"
#include "wavernn_cond.h"

namespace {

// ------------------------------------------------------------------------------------------------------
// one 256-thread workgroup per sequence: thread c < 1 + feat + d fills xin, the next 3 d threads the aux slices
__global__ __launch_bounds__(256) void wavernn_prep_kernel(const estts_wavernn_prep_args p) {
    const int b = blockIdx.x;
    const int c = threadIdx.x;
    const int step = p.step[0];
    const int u = p.seq_utt[b];
    const int s = p.seq_start[b] + step;
    const bool live = s < p.utt_samples[u];
    const int frame = s / p.hop;
    const int phase = s - frame * p.hop;
    const int n_in = 1 + p.feat + p.d;
    for (int col = c; col < n_in + 3 * p.d; col += 256) {
        float v = 0.0f;
        if (col == 0) {
            v = p.x_prev[b];
        } else if (live) {
            if (col <= p.feat) {
                const float* m = p.mel + (long)(p.mel_off[u] + frame + p.pad - p.fir_kmin) * p.feat + (col - 1);
                const float* f = p.fir + phase * p.ntaps;
                for (int j = 0; j < p.ntaps; ++j) v += f[j] * m[-(long)j * p.feat];
            } else {
                v = p.aux[(long)(p.aux_off[u] + frame) * (4 * p.d) + (col - 1 - p.feat)];
            }
        }
        if (col < n_in) {
            p.xin[(long)b * p.ld_xin + col] = v;
        } else {
            const int k = col - n_in;  // 0 .. 3d-1 -> slices a2, a3, a4
            const int which = k / p.d;
            const int e = k - which * p.d;
            float* dst = which == 0 ? p.a2 + (long)b * p.ld_a2 : (which == 1 ? p.a3 + (long)b * p.ld_a3 : p.a4 + (long)b * p.ld_a4);
            dst[e] = v;
        }
    }
}

// ------------------------------------------------------------------------------------------------------
__device__ __forceinline__ float sigmoidf_(float x) { return 1.0f / (1.0f + expf(-x)); }

__global__ __launch_bounds__(256) void gru_gates_kernel(const float* gi, int ldgi, const float* gh, int ldgh, float* h,
                                                        int ldh, const float* x_res, int ldx, float* out, int ldo, int M,
                                                        int H) {
    const int hq = H >> 2;
    const long total = (long)M * hq;
    for (long idx = (long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long)gridDim.x * blockDim.x) {
        const int m = (int)(idx / hq);
        const int c = (int)(idx - (long)m * hq) * 4;
        const float* gim = gi + (long)m * ldgi + c;
        const float* ghm = gh + (long)m * ldgh + c;
        const float4 ir = *reinterpret_cast<const float4*>(gim);
        const float4 iz = *reinterpret_cast<const float4*>(gim + H);
        const float4 in = *reinterpret_cast<const float4*>(gim + 2 * H);
        const float4 hr = *reinterpret_cast<const float4*>(ghm);
        const float4 hz = *reinterpret_cast<const float4*>(ghm + H);
        const float4 hn = *reinterpret_cast<const float4*>(ghm + 2 * H);
        float4 hv = *reinterpret_cast<const float4*>(h + (long)m * ldh + c);
        float4 xr = make_float4(0.f, 0.f, 0.f, 0.f);
        if (x_res) xr = *reinterpret_cast<const float4*>(x_res + (long)m * ldx + c);
        auto cell = [](float ir_, float iz_, float in_, float hr_, float hz_, float hn_, float h_) {
            const float r = sigmoidf_(ir_ + hr_);
            const float z = sigmoidf_(iz_ + hz_);
            const float n = tanhf(in_ + r * hn_);
            return (1.0f - z) * n + z * h_;
        };
        hv.x = cell(ir.x, iz.x, in.x, hr.x, hz.x, hn.x, hv.x);
        hv.y = cell(ir.y, iz.y, in.y, hr.y, hz.y, hn.y, hv.y);
        hv.z = cell(ir.z, iz.z, in.z, hr.z, hz.z, hn.z, hv.z);
        hv.w = cell(ir.w, iz.w, in.w, hr.w, hz.w, hn.w, hv.w);
        *reinterpret_cast<float4*>(h + (long)m * ldh + c) = hv;
        *reinterpret_cast<float4*>(out + (long)m * ldo + c) = make_float4(xr.x + hv.x, xr.y + hv.y, xr.z + hv.z, xr.w + hv.w);
    }
}

// ------------------------------------------------------------------------------------------------------
// counter-based uniform in (0, 1): 32-bit finaliser of (seed, step, sequence, stream)
__device__ __forceinline__ float uniform01(uint32_t seed, uint32_t step, uint32_t seq, uint32_t stream) {
    uint32_t x = seed ^ (step * 0x9E3779B9u) ^ (seq * 0x85EBCA6Bu) ^ (stream * 0xC2B2AE35u);
    x ^= x >> 16; x *= 0x7FEB352Du; x ^= x >> 15; x *= 0x846CA68Bu; x ^= x >> 16;
    x += step * 0x27D4EB2Fu + seq;
    x ^= x >> 16; x *= 0x7FEB352Du; x ^= x >> 15; x *= 0x846CA68Bu; x ^= x >> 16;
    return ((x >> 8) + 0.5f) * (1.0f / 16777216.0f);
}

// one wave per sequence, four sequences per workgroup; with_cond: four more waves, one per sequence, prepare the next step's
// conditioning (two chains of dependent loads side by side instead of one after the other: 12 -> 7 us)
__global__ __launch_bounds__(512) void wavernn_sample_kernel(const estts_wavernn_sample_args p) {
    __shared__ int s_last;
    const int lane = threadIdx.x & 63;
    const int role = threadIdx.x >> 8;           // 0: draw, 1: conditioning
    const int b = blockIdx.x * 4 + ((threadIdx.x >> 6) & 3);
    const int step = p.step[0];
    // the next step's conditioning of this sequence (everything but x_prev), one group of eight panel columns per lane of the
    // sequence's conditioning wave: it does not depend on the draw.  The wave of the last sequence also zero-fills the rest of
    // the last row block.
    if (role == 1) {
        const int groups = 15 + 3 * (p.cond.d >> 3);
        const int n_rows = ((p.n_seq + 15) >> 4) << 4;
        if (b < p.n_seq && lane >= 1 && lane <= groups) wavernn_cond_group(p.cond, b, lane, step + 1);
        if (b == p.n_seq - 1)
            for (int bb = p.n_seq; bb < n_rows; ++bb)
                if (lane >= 1 && lane <= groups) wavernn_cond_group(p.cond, bb, lane, step + 1);
    }
    if (role == 0 && b < p.n_seq) {
        const float* lg = p.logits + (long)b * p.ld_logits;
        float value;
        if (p.mode == 0) {
            // lane owns classes [lane * per, lane * per + per)
            const int per = (p.n_classes + 63) / 64;
            const int c0 = lane * per;
            const float u01 = p.u ? p.u[(long)step * p.n_seq + b] : uniform01(p.seed, step, b, 0);
            int cnt = 0;
            if (p.n_classes == 512 && (p.ld_logits & 3) == 0 && (reinterpret_cast<uintptr_t>(p.logits) & 15) == 0) {
                // the 9-bit case: the lane's eight logits in two 16-byte loads, exponentials computed once
                float v[8];
                *reinterpret_cast<float4*>(v) = *reinterpret_cast<const float4*>(lg + c0);
                *reinterpret_cast<float4*>(v + 4) = *reinterpret_cast<const float4*>(lg + c0 + 4);
                float mx = v[0];
#pragma unroll
                for (int k = 1; k < 8; ++k) mx = fmaxf(mx, v[k]);
                mx = wave_max(mx);
                float mine = 0.0f;
#pragma unroll
                for (int k = 0; k < 8; ++k) {
                    v[k] = expf(v[k] - mx);
                    mine += v[k];
                }
                const float total = wave_sum(mine);
                float incl = mine;  // inclusive prefix over lanes
#pragma unroll
                for (int o = 1; o < 64; o <<= 1) {
                    const float t = __shfl_up(incl, o, 64);
                    if (lane >= o) incl += t;
                }
                const float u = u01 * total;
                float run = incl - mine;
#pragma unroll
                for (int k = 0; k < 8; ++k) {
                    run += v[k];
                    cnt += (run <= u) ? 1 : 0;
                }
            } else {
                float mx = -INFINITY;
                for (int k = 0; k < per; ++k)
                    if (c0 + k < p.n_classes) mx = fmaxf(mx, lg[c0 + k]);
                mx = wave_max(mx);
                float mine = 0.0f;
                for (int k = 0; k < per; ++k)
                    if (c0 + k < p.n_classes) mine += expf(lg[c0 + k] - mx);
                const float total = wave_sum(mine);
                float incl = mine;  // inclusive prefix over lanes
#pragma unroll
                for (int o = 1; o < 64; o <<= 1) {
                    const float t = __shfl_up(incl, o, 64);
                    if (lane >= o) incl += t;
                }
                const float u = u01 * total;
                float run = incl - mine;
                for (int k = 0; k < per; ++k)
                    if (c0 + k < p.n_classes) {
                        run += expf(lg[c0 + k] - mx);
                        cnt += (run <= u) ? 1 : 0;
                    }
            }
            cnt = (int)wave_sum((float)cnt);
            cnt = min(cnt, p.n_classes - 1);
            value = 2.0f * (float)cnt / ((float)p.n_classes - 1.0f) - 1.0f;
        } else {
            const int nr = p.n_classes / 3;
            float best = -INFINITY;
            int arg = 0;
            if (lane < nr) {
                float um = p.u_mix ? p.u_mix[((long)step * p.n_seq + b) * nr + lane] : uniform01(p.seed, step, b, 1 + lane);
                um = fminf(fmaxf(um, 1e-5f), 1.0f - 1e-5f);
                best = lg[lane] - logf(-logf(um));
                arg = lane;
            }
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) {  // arg-max, lowest index wins ties (as torch.max)
                const float ob = __shfl_xor(best, o, 64);
                const int oa = __shfl_xor(arg, o, 64);
                if (ob > best || (ob == best && oa < arg)) { best = ob; arg = oa; }
            }
            const float mean = lg[nr + arg];
            const float log_scale = fmaxf(lg[2 * nr + arg], -32.23619130191664f);  // log(1e-14)
            float ul = p.u ? p.u[(long)step * p.n_seq + b] : uniform01(p.seed, step, b, 0);
            ul = fminf(fmaxf(ul, 1e-5f), 1.0f - 1e-5f);
            value = mean + expf(log_scale) * (logf(ul) - logf(1.0f - ul));
            value = fminf(fmaxf(value, -1.0f), 1.0f);
        }
        if (lane == 0) {
            p.wave[(long)b * p.ld_wave + step] = value;
            const float fed = p.forced ? p.forced[(long)step * p.n_seq + b] : value;
            p.x_prev[b] = fed;
            if (p.with_cond) {   // first column group of the next step's input panel: [x_prev 0 0 0 0 0 0 0] as bf16 hi | lo
                __bf16 hi, lo;
                split_bf16(fed, hi, lo);
                const bf16x8 z = {hi, 0, 0, 0, 0, 0, 0, 0}, zl = {lo, 0, 0, 0, 0, 0, 0, 0};
                unsigned char* dst = reinterpret_cast<unsigned char*>(p.cond.xin_panel) + (long)(b >> 4) * 4 * 2048 + (b & 15) * 16;
                *reinterpret_cast<bf16x8*>(dst) = z;
                *reinterpret_cast<bf16x8*>(dst + 1024) = zl;
            }
        }
    }
    // two alternating counters: nobody reads the one this call writes
    if (p.step_next) {
        if (blockIdx.x == 0 && threadIdx.x == 0) p.step_next[0] = step + 1;
        return;
    }
    // the last workgroup to finish has seen every workgroup read step[0]; it advances the counter
    __syncthreads();
    if (threadIdx.x == 0) {
        const int t = __hip_atomic_fetch_add(p.done, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        s_last = (t == (int)gridDim.x - 1);
        if (s_last) {
            __hip_atomic_store(p.done, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(p.step, step + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
}

}  // namespace

extern "C" int estts_wavernn_prep(const estts_wavernn_prep_args* args, void* stream) {
    ESTTS_CHECK_ARG(args != nullptr, "estts_wavernn_prep: null args");
    const estts_wavernn_prep_args& a = *args;
    ESTTS_CHECK_ARG(a.mel && a.mel_off && a.aux && a.aux_off && a.fir && a.seq_utt && a.seq_start && a.utt_samples &&
                    a.x_prev && a.step && a.xin && a.a2 && a.a3 && a.a4, "estts_wavernn_prep: null pointer");
    ESTTS_CHECK_ARG(a.n_seq > 0 && a.feat > 0 && a.d > 0 && a.hop > 0 && a.pad >= 0 && a.ntaps > 0,
                    "estts_wavernn_prep: bad sizes");
    ESTTS_CHECK_ARG(a.ld_xin >= 1 + a.feat + a.d && a.ld_a2 >= a.d && a.ld_a3 >= a.d && a.ld_a4 >= a.d,
                    "estts_wavernn_prep: leading dimension too small");
    ESTTS_CHECK_ARG(a.pad - a.fir_kmin - (a.ntaps - 1) >= 0 && a.pad + a.fir_kmin >= 0,
                    "estts_wavernn_prep: filter reaches outside the padded mel");
    ESTTS_LAUNCH(wavernn_prep_kernel, dim3(a.n_seq), dim3(256), 0, (hipStream_t)stream, a);
    ESTTS_CHECK_LAUNCH("estts_wavernn_prep");
    return 0;
}

extern "C" int estts_gru_gates(const float* gi, int32_t ldgi, const float* gh, int32_t ldgh, float* h, int32_t ldh,
                               const float* x_res, int32_t ldx, float* out, int32_t ldo, int32_t M, int32_t H,
                               void* stream) {
    ESTTS_CHECK_ARG(gi && gh && h && out, "estts_gru_gates: null pointer");
    ESTTS_CHECK_ARG(M > 0 && H > 0 && (H & 3) == 0, "estts_gru_gates: H must be a positive multiple of 4");
    ESTTS_CHECK_ARG(ldgi >= 3 * H && ldgh >= 3 * H && ldh >= H && ldo >= H && (!x_res || ldx >= H), "estts_gru_gates: ld too small");
    ESTTS_CHECK_ARG(((ldgi | ldgh | ldh | ldo | (x_res ? ldx : 0)) & 3) == 0, "estts_gru_gates: ld not multiple of 4");
    const long total = (long)M * (H >> 2);
    const int grid = (int)((total + 255) / 256 < 2048 ? (total + 255) / 256 : 2048);
    ESTTS_LAUNCH(gru_gates_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, gi, ldgi, gh, ldgh, h, ldh, x_res,
                       ldx, out, ldo, M, H);
    ESTTS_CHECK_LAUNCH("estts_gru_gates");
    return 0;
}

extern "C" int estts_wavernn_sample(const estts_wavernn_sample_args* args, void* stream) {
    ESTTS_CHECK_ARG(args != nullptr, "estts_wavernn_sample: null args");
    const estts_wavernn_sample_args& a = *args;
    ESTTS_CHECK_ARG(a.logits && a.wave && a.x_prev && a.step && (a.done || a.step_next), "estts_wavernn_sample: null pointer");
    ESTTS_CHECK_ARG(a.step_next != a.step, "estts_wavernn_sample: step_next must be another counter than step");
    ESTTS_CHECK_ARG(a.n_seq > 0 && a.n_classes > 0 && a.ld_logits >= a.n_classes, "estts_wavernn_sample: bad sizes");
    ESTTS_CHECK_ARG(a.mode == 0 || (a.mode == 1 && a.n_classes % 3 == 0 && a.n_classes / 3 <= 64),
                    "estts_wavernn_sample: mode must be 0 (RAW) or 1 (MOL with <= 64 components)");
    ESTTS_CHECK_ARG(a.mode == 1 || a.n_classes <= 65536, "estts_wavernn_sample: too many classes");
    ESTTS_CHECK_ARG(a.mode == 0 || !a.u_mix == !a.u, "estts_wavernn_sample: MOL needs both u and u_mix or neither");
    if (a.with_cond) {
        const estts_wavernn_cond_args& c = a.cond;
        ESTTS_CHECK_ARG(c.mel && c.aux && c.fir && c.seq_info && ((uintptr_t)c.seq_info & 15) == 0 && c.xin_panel && c.a2_panel && c.a3_panel &&
                        c.a4_panel, "estts_wavernn_sample: null or misaligned pointer in the conditioning block");
        ESTTS_CHECK_ARG(c.n_seq == a.n_seq && c.feat > 0 && c.d > 0 && (c.d & 7) == 0 && c.d <= 32 && c.hop > 0 && c.ntaps > 0 &&
                        8 + c.feat + c.d <= 128 && c.aux_kpitch > c.aux_ks && c.aux_ks >= 0,
                        "estts_wavernn_sample: bad conditioning block (see estts_wavernn_cond)");
    }
    ESTTS_LAUNCH(wavernn_sample_kernel, dim3((a.n_seq + 3) / 4), dim3(a.with_cond ? 512 : 256), 0, (hipStream_t)stream, a);
    ESTTS_CHECK_LAUNCH("estts_wavernn_sample");
    return 0;
}

// ------------------------------------------------------------------------------------------------------
// Cross-fade of the folds, mu-law expansion and closing fade in float64 (see estts_wavernn_unfold in include/estts.h): one
// thread per output sample.  The window values follow NumPy's linspace (k * step + start, the last element the end point
// itself), multiply and add kept separate so that they round as on the host.
// ------------------------------------------------------------------------------------------------------
namespace {
#pragma clang fp contract(off)
__device__ __forceinline__ double lin_t(int k, int n, double step) {      // linspace(-1, 1, n)[k]
    const double prod = (double)k * step;
    return k == n - 1 ? 1.0 : prod + -1.0;
}

__global__ __launch_bounds__(256) void wavernn_unfold_kernel(const estts_wavernn_unfold_args p) {
#pragma clang fp contract(off)
    const int u = blockIdx.y;
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= p.out_len[u]) return;
    const int stride = p.target + p.overlap;
    const int F = p.n_folds[u];
    const int silence = p.overlap / 2;
    const int fade_len = p.overlap - silence;
    const float* base = p.wave + (long)p.seq_off[u] * p.ld_wave;
    const int f1 = i / stride;
    const int j1 = i - f1 * stride;
    double v = 0.0;
    if (f1 >= 1 && j1 < p.overlap) {          // tail of the previous fold: fade_out[j1] = sqrt(0.5 (1 - t[j1])), then silence
        const double y = (double)base[(long)(f1 - 1) * p.ld_wave + j1 + stride];
        const double w = j1 < fade_len ? sqrt(0.5 * (1.0 - lin_t(j1, fade_len, p.step_in))) : 0.0;
        v = 0.0 + y * w;
    }
    if (f1 < F) {                              // this fold: silence, fade_in, then plain
        const double y = (double)base[(long)f1 * p.ld_wave + j1];
        double w = 1.0;
        if (j1 < silence) w = 0.0;
        else if (j1 < p.overlap) w = sqrt(0.5 * (1.0 + lin_t(j1 - silence, fade_len, p.step_in)));
        v = v + y * w;
    }
    if (p.mu > 0) {
        const double m = (double)(p.mu - 1);
        const double sgn = v > 0.0 ? 1.0 : (v < 0.0 ? -1.0 : 0.0);
        v = sgn / m * (pow(1.0 + m, fabs(v)) - 1.0);
    }
    const int k = i - (p.out_len[u] - p.fade_tail);
    if (p.fade_tail > 0 && k >= 0) {
        const double prod = (double)k * p.step_tail;
        v *= (k == p.fade_tail - 1) ? 0.0 : prod + 1.0;
    }
    p.out[p.out_off[u] + i] = v;
}
}  // namespace

extern "C" int estts_wavernn_unfold(const estts_wavernn_unfold_args* args, void* stream) {
    ESTTS_CHECK_ARG(args != nullptr, "estts_wavernn_unfold: null args");
    const estts_wavernn_unfold_args a = *args;
    ESTTS_CHECK_ARG(a.wave && a.seq_off && a.n_folds && a.out_len && a.out_off && a.out, "estts_wavernn_unfold: null pointer");
    ESTTS_CHECK_ARG(a.n_utt > 0 && a.n_utt <= 65535 && a.target > 0 && a.overlap >= 4 && a.max_len > 0 && a.mu >= 0 && a.fade_tail >= 0 &&
                    a.ld_wave >= a.target + 2 * a.overlap, "estts_wavernn_unfold: bad sizes (utterances %d, target %d, overlap %d, row %d)",
                    a.n_utt, a.target, a.overlap, a.ld_wave);
    ESTTS_LAUNCH(wavernn_unfold_kernel, dim3((a.max_len + 255) / 256, a.n_utt), dim3(256), 0, (hipStream_t)stream, a);
    ESTTS_CHECK_LAUNCH("estts_wavernn_unfold");
    return 0;
}
