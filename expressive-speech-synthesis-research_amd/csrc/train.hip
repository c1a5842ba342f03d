// Training-side HBM-bound kernels: weight re-packing, LayerNorm-combo backward, BatchNorm batch statistics and
// backward, dropout, gradient plumbing (embedding, length regulator, loss) and Adam.
// (see the "Training" block of include/estts.h)
#include <cstdlib>
#include "common.h"

namespace {

// ------------------------------------------------------------------------------------------
// master float32 [K][ld] -> bf16 hi/lo fragment-major.  One thread per (16-row block, 32-column step, lane): 8 values.
__device__ __forceinline__ void pack_unit(const float* src, int ld, int K, int N, int taps, int mode, uint16_t* hi, uint16_t* lo,
                                          int rows, int cols, int KS, long idx) {
    const int cin = K / taps;
    const int l = (int)(idx & 63);
    const long blk = idx >> 6;
    const int ks = (int)(blk % KS);
    const int nb = (int)(blk / KS);
    const int r = nb * 16 + (l & 15);          // destination row
    const int c0 = ks * 32 + (l >> 4) * 8;     // first destination column
    bf16x8 h, lw;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const int c = c0 + j;
        float v = 0.0f;
        if (r < rows && c < cols) {
            if (mode == 0) {
                v = src[(long)c * ld + r];
            } else {
                const int tp = c / N;
                const int n = c - tp * N;
                v = src[(long)((taps - 1 - tp) * cin + r) * ld + n];
            }
        }
        __bf16 a, b;
        split_bf16(v, a, b);
        h[j] = a;
        lw[j] = b;
    }
    *reinterpret_cast<bf16x8*>(hi + idx * 8) = h;
    *reinterpret_cast<bf16x8*>(lo + idx * 8) = lw;
}

__global__ __launch_bounds__(256) void pack_weight_kernel(const float* src, int ld, int K, int N, int taps, int mode,
                                                          uint16_t* hi, uint16_t* lo, int rows, int cols, int KS, long total) {
    for (long idx = (long)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (long)gridDim.x * 256)
        pack_unit(src, ld, K, N, taps, mode, hi, lo, rows, cols, KS, idx);
}

// every weight of a model in one launch: unit u belongs to the job with first_unit <= u < next job's first_unit
__global__ __launch_bounds__(256) void pack_weights_kernel(const estts_pack_job* jobs, int n_jobs, long total) {
    for (long u = (long)blockIdx.x * 256 + threadIdx.x; u < total; u += (long)gridDim.x * 256) {
        int lo = 0, hi = n_jobs - 1;
        while (lo < hi) {
            const int mid = (lo + hi + 1) >> 1;
            if (jobs[mid].first_unit <= u) lo = mid; else hi = mid - 1;
        }
        const estts_pack_job j = jobs[lo];
        const int rows = j.mode == 0 ? j.N : j.K / j.taps;
        const int cols = j.mode == 0 ? j.K : j.taps * j.N;
        const int cp = (cols + 63) / 64 * 64;
        pack_unit(j.src, j.ld_src, j.K, j.N, j.taps, j.mode, j.hi, j.lo, rows, cols, cp / 32, u - j.first_unit);
    }
}

// ------------------------------------------------------------------------------------------
// elementwise helpers: thread per float4, grid-stride
template <typename F>
__device__ __forceinline__ void for_each_f4(int M, int D, F f) {
    const int dq = D >> 2;
    const long total = (long)M * dq;
    for (long idx = (long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long)gridDim.x * blockDim.x) {
        const int row = (int)(idx / dq);
        const int c = (int)(idx - (long)row * dq) * 4;
        f(row, c);
    }
}

__device__ __forceinline__ float4 ld4(const float* p) { return *reinterpret_cast<const float4*>(p); }
__device__ __forceinline__ void st4(float* p, float4 v) { *reinterpret_cast<float4*>(p) = v; }

__global__ __launch_bounds__(256) void affine_act_kernel(const float* z, int ldz, const float* scale, const float* shift,
                                                         int act, const float* R, int ldr, int r_mod, float* out, int ldo, int M,
                                                         int C) {
    for_each_f4(M, C, [&](int row, int c) {
        float4 v = ld4(z + (long)row * ldz + c);
        if (scale) {
            const float4 s = ld4(scale + c), b = ld4(shift + c);
            v.x = v.x * s.x + b.x; v.y = v.y * s.y + b.y; v.z = v.z * s.z + b.z; v.w = v.w * s.w + b.w;
        }
        v.x = apply_act(v.x, act); v.y = apply_act(v.y, act); v.z = apply_act(v.z, act); v.w = apply_act(v.w, act);
        if (R) {
            const float4 r = ld4(R + (long)(r_mod > 0 ? row % r_mod : row) * ldr + c);
            v.x += r.x; v.y += r.y; v.z += r.z; v.w += r.w;
        }
        st4(out + (long)row * ldo + c, v);
    });
}

__global__ __launch_bounds__(256) void bn_bwd_kernel(const float* dy, int lddy, const float* z, int ldz, const float* mean,
                                                     const float* rstd, const float* gamma, const float* dgamma,
                                                     const float* dbeta, float* dz, int lddz, int M, int C) {
    const float inv_m = 1.0f / (float)M;
    for_each_f4(M, C, [&](int row, int c) {
        const float4 g = ld4(dy + (long)row * lddy + c), x = ld4(z + (long)row * ldz + c);
        const float4 mu = ld4(mean + c), rs = ld4(rstd + c), ga = ld4(gamma + c), dg = ld4(dgamma + c), db = ld4(dbeta + c);
        float4 r;
        r.x = ga.x * rs.x * (g.x - db.x * inv_m - (x.x - mu.x) * rs.x * dg.x * inv_m);
        r.y = ga.y * rs.y * (g.y - db.y * inv_m - (x.y - mu.y) * rs.y * dg.y * inv_m);
        r.z = ga.z * rs.z * (g.z - db.z * inv_m - (x.z - mu.z) * rs.z * dg.z * inv_m);
        r.w = ga.w * rs.w * (g.w - db.w * inv_m - (x.w - mu.w) * rs.w * dg.w * inv_m);
        st4(dz + (long)row * lddz + c, r);
    });
}

// the same with dbeta / dgamma still spread over COLREDUCE_SLOTS copies (estts_colreduce with defer): a thread keeps its
// four columns, adds the copies up once and walks rows blockIdx.y * 4 + (its row lane), + 4 gridDim.y, ...; the first row
// split also adds the totals to the parameter gradients
__global__ __launch_bounds__(256) void bn_bwd_slots_kernel(const float* dy, int lddy, const float* z, int ldz, const float* mean,
                                                           const float* rstd, const float* gamma, const float* slots,
                                                           long cp, float* dgamma, float* dbeta, float* dz, int lddz, int M,
                                                           int C, int act, const float* scale, const float* shift) {
    const int chunk = threadIdx.x & 63;
    const int rl = threadIdx.x >> 6;
    const int c = (blockIdx.x * 64 + chunk) * 4;
    if (c >= C) return;
    float4 db = make_float4(0.f, 0.f, 0.f, 0.f), dg = db;
    // scale != NULL: dy is the gradient w.r.t. act(z scale + shift) and goes through the activation here, with the arithmetic of
    // the reduction that built the sums (which then need not store its dy)
    const bool through_act = scale != nullptr && act != ESTTS_ACT_NONE;
    float4 sc = db, sh = db;
    if (through_act) { sc = ld4(scale + c); sh = ld4(shift + c); }
    {
        float4 v0[16], v1[16];
#pragma unroll
        for (int k = 0; k < 16; ++k) {
            v0[k] = ld4(slots + (long)k * 2 * cp + c);
            v1[k] = ld4(slots + (long)k * 2 * cp + cp + c);
        }
#pragma unroll
        for (int k = 0; k < 16; ++k) {
            db.x += v0[k].x; db.y += v0[k].y; db.z += v0[k].z; db.w += v0[k].w;
            dg.x += v1[k].x; dg.y += v1[k].y; dg.z += v1[k].z; dg.w += v1[k].w;
        }
    }
    if (blockIdx.y == 0 && rl == 0) {
        float4 a = ld4(dbeta + c), b = ld4(dgamma + c);
        a.x += db.x; a.y += db.y; a.z += db.z; a.w += db.w;
        b.x += dg.x; b.y += dg.y; b.z += dg.z; b.w += dg.w;
        st4(dbeta + c, a);
        st4(dgamma + c, b);
    }
    const float inv_m = 1.0f / (float)M;
    const float4 mu = ld4(mean + c), rs = ld4(rstd + c), ga = ld4(gamma + c);
    auto one = [&](float4 g, const float4& x) {
        if (through_act) {
            g.x *= act_grad_from_output(apply_act(x.x * sc.x + sh.x, act), act);
            g.y *= act_grad_from_output(apply_act(x.y * sc.y + sh.y, act), act);
            g.z *= act_grad_from_output(apply_act(x.z * sc.z + sh.z, act), act);
            g.w *= act_grad_from_output(apply_act(x.w * sc.w + sh.w, act), act);
        }
        float4 r;
        r.x = ga.x * rs.x * (g.x - db.x * inv_m - (x.x - mu.x) * rs.x * dg.x * inv_m);
        r.y = ga.y * rs.y * (g.y - db.y * inv_m - (x.y - mu.y) * rs.y * dg.y * inv_m);
        r.z = ga.z * rs.z * (g.z - db.z * inv_m - (x.z - mu.z) * rs.z * dg.z * inv_m);
        r.w = ga.w * rs.w * (g.w - db.w * inv_m - (x.w - mu.w) * rs.w * dg.w * inv_m);
        return r;
    };
    const int stride = gridDim.y * 4;
    int row = blockIdx.y * 4 + rl;
    for (; row + stride < M; row += 2 * stride) {   // two rows in flight (dz may alias dy: loads first)
        const float4 g0 = ld4(dy + (long)row * lddy + c), x0 = ld4(z + (long)row * ldz + c);
        const float4 g1 = ld4(dy + (long)(row + stride) * lddy + c), x1 = ld4(z + (long)(row + stride) * ldz + c);
        st4(dz + (long)row * lddz + c, one(g0, x0));
        st4(dz + (long)(row + stride) * lddz + c, one(g1, x1));
    }
    if (row < M) st4(dz + (long)row * lddz + c, one(ld4(dy + (long)row * lddy + c), ld4(z + (long)row * ldz + c)));
}

__global__ __launch_bounds__(256) void dropout_kernel(const float* x, int ldx, float* out, int ldo, int M, int D, float rate,
                                                      uint32_t seed, uint32_t site) {
    const float keep = 1.0f / (1.0f - rate);
    for_each_f4(M, D, [&](int row, int c) {
        float4 v = ld4(x + (long)row * ldx + c);
        v.x = dropout_keep(seed, site, row, c, rate) ? v.x * keep : 0.0f;
        v.y = dropout_keep(seed, site, row, c + 1, rate) ? v.y * keep : 0.0f;
        v.z = dropout_keep(seed, site, row, c + 2, rate) ? v.z * keep : 0.0f;
        v.w = dropout_keep(seed, site, row, c + 3, rate) ? v.w * keep : 0.0f;
        st4(out + (long)row * ldo + c, v);
    });
}

__global__ __launch_bounds__(256) void axpby_kernel(const float* x, int ldx, float alpha, float* out, int ldo, float beta,
                                                    int M, int D) {
    for_each_f4(M, D, [&](int row, int c) {
        const float4 a = ld4(x + (long)row * ldx + c);
        float4 o = make_float4(0.f, 0.f, 0.f, 0.f);
        if (beta != 0.0f) o = ld4(out + (long)row * ldo + c);
        o.x = beta * o.x + alpha * a.x; o.y = beta * o.y + alpha * a.y;
        o.z = beta * o.z + alpha * a.z; o.w = beta * o.w + alpha * a.w;
        st4(out + (long)row * ldo + c, o);
    });
}

__global__ __launch_bounds__(256) void head_scale_kernel(float* x, int ldx, const float* scale, int rows, int L, int H,
                                                         int depth) {
    for_each_f4(rows, H * depth, [&](int row, int c) {
        const float s = scale[(row / L) * H + c / depth];
        float4 v = ld4(x + (long)row * ldx + c);
        v.x *= s; v.y *= s; v.z *= s; v.w *= s;
        st4(x + (long)row * ldx + c, v);
    });
}

__global__ __launch_bounds__(256) void embed_bwd_kernel(const int32_t* tokens, const float* dx, int lddx, float* demb, int rows,
                                                        int D, float scale) {
    for_each_f4(rows, D, [&](int row, int c) {
        const float4 g = ld4(dx + (long)row * lddx + c);
        float* d = demb + (long)tokens[row] * D + c;
        atomic_add_f32(d, g.x * scale);
        atomic_add_f32(d + 1, g.y * scale);
        atomic_add_f32(d + 2, g.z * scale);
        atomic_add_f32(d + 3, g.w * scale);
    });
}

// one wave per token
__global__ __launch_bounds__(256) void expand_bwd_kernel(const float* dout, int lddo, const int32_t* cum, float* dx, int lddx,
                                                         int B, int N, int T, int D) {
    const int lane = threadIdx.x & 63;
    const long tok = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (tok >= (long)B * N) return;
    const int b = (int)(tok / N);
    const int n = (int)(tok - (long)b * N);
    const int t0 = n > 0 ? cum[tok - 1] : 0;
    const int t1 = min(cum[tok], T);
    const int dq = D >> 2;
    for (int ch = lane; ch < dq; ch += 64) {
        float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
        for (int t = t0; t < t1; t += 4) {   // four frames in flight (the last group re-reads the last frame with weight 0)
            float4 g[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) g[u] = ld4(dout + ((long)b * T + min(t + u, t1 - 1)) * lddo + ch * 4);
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const float w = t + u < t1 ? 1.0f : 0.0f;
                acc.x += w * g[u].x; acc.y += w * g[u].y; acc.z += w * g[u].z; acc.w += w * g[u].w;
            }
        }
        st4(dx + tok * lddx + ch * 4, acc);
    }
}

// one wave per row
__global__ __launch_bounds__(256) void masked_mae_bwd_kernel(const float* target, int ldt, const float* pred, int ldp,
                                                             float* dpred, int lddp, int M, int C, float scale) {
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= M) return;
    float nz = 0.f;
    for (int c = lane; c < C; c += 64) nz += target[(long)row * ldt + c] != 0.0f ? 1.0f : 0.0f;
    nz = wave_sum(nz);
    const float w = nz > 0.f ? scale / ((float)M * (float)C) : 0.0f;
    for (int c = lane; c < C; c += 64) {
        const float d = pred[(long)row * ldp + c] - target[(long)row * ldt + c];
        dpred[(long)row * lddp + c] = d > 0.f ? w : (d < 0.f ? -w : 0.0f);
    }
}

__global__ __launch_bounds__(256) void adam_kernel(float* p, const float* grad, float* m, float* v, long n4, float lr_t,
                                                   float b1, float b2, float eps, float gs) {
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n4; i += (long)gridDim.x * 256) {
        float4 g = ld4(grad + i * 4), mm = ld4(m + i * 4), vv = ld4(v + i * 4), pp = ld4(p + i * 4);
        g.x *= gs; g.y *= gs; g.z *= gs; g.w *= gs;
        mm.x = b1 * mm.x + (1.f - b1) * g.x; mm.y = b1 * mm.y + (1.f - b1) * g.y;
        mm.z = b1 * mm.z + (1.f - b1) * g.z; mm.w = b1 * mm.w + (1.f - b1) * g.w;
        vv.x = b2 * vv.x + (1.f - b2) * g.x * g.x; vv.y = b2 * vv.y + (1.f - b2) * g.y * g.y;
        vv.z = b2 * vv.z + (1.f - b2) * g.z * g.z; vv.w = b2 * vv.w + (1.f - b2) * g.w * g.w;
        pp.x -= lr_t * mm.x / (sqrtf(vv.x) + eps); pp.y -= lr_t * mm.y / (sqrtf(vv.y) + eps);
        pp.z -= lr_t * mm.z / (sqrtf(vv.z) + eps); pp.w -= lr_t * mm.w / (sqrtf(vv.w) + eps);
        st4(m + i * 4, mm); st4(v + i * 4, vv); st4(p + i * 4, pp);
    }
}

// ------------------------------------------------------------------------------------------
constexpr int COLREDUCE_SLOTS = 16;

// column reductions: a workgroup covers 256 columns (64 float4 chunks) x 4 row lanes; gridDim.y splits the rows
__global__ __launch_bounds__(256) void colreduce_kernel(const estts_colreduce_args p) {
    __shared__ float4 red[2][4][64];
    const int chunk = threadIdx.x & 63;
    const int rl = threadIdx.x >> 6;
    const int c = (blockIdx.x * 64 + chunk) * 4;
    const bool live = c < p.C;
    float4 s0 = make_float4(0.f, 0.f, 0.f, 0.f), s1 = s0;
    if (p.clear) {   // zero the other region
        const long step = (long)gridDim.x * gridDim.y * 256;
        for (long i = ((long)blockIdx.y * gridDim.x + blockIdx.x) * 256 + threadIdx.x; i < p.clear_floats; i += step) p.clear[i] = 0.0f;
    }
    if (live) {
        float4 k0 = s0, mu = s0, rs = s0, sc = s0, sh = s0;
        if (p.mode == 0 && !p.defer) k0 = ld4(p.z + c);
        if (p.mode == 0 && p.defer && p.shift) k0 = ld4(p.shift + c);
        if (p.mode == 2) { mu = ld4(p.mean + c); rs = ld4(p.rstd + c); }
        const bool recompute = p.mode != 0 && !p.a && p.scale && p.act != ESTTS_ACT_NONE;
        if (recompute) { sc = ld4(p.scale + c); sh = ld4(p.shift + c); }
        // two rows in flight per thread: the loads of the second are requested before the first is consumed (the optional
        // in-place store of a row keeps the compiler from doing that itself)
        struct RowIn {
            float4 g, z, a;
        };
        const float4 zero4 = make_float4(0.f, 0.f, 0.f, 0.f);
        auto fetch = [&](int row) {
            RowIn v = {zero4, zero4, zero4};
            if (p.mode == 0) {
                v.z = ld4(p.z + (long)row * p.ldz + c);
            } else {
                v.g = ld4(p.g + (long)row * p.ldg + c);
                if (p.mode == 2 || recompute) v.z = ld4(p.z + (long)row * p.ldz + c);
                if (p.a) v.a = ld4(p.a + (long)row * p.lda + c);
            }
            return v;
        };
        auto consume = [&](int row, const RowIn& v) {
            if (p.mode == 0) {
                const float a = v.z.x - k0.x, b = v.z.y - k0.y, cc = v.z.z - k0.z, d = v.z.w - k0.w;
                s0.x += a; s0.y += b; s0.z += cc; s0.w += d;
                s1.x += a * a; s1.y += b * b; s1.z += cc * cc; s1.w += d * d;
            } else {
                float4 g = v.g;
                const float4 z = v.z;
                if (p.a) {
                    g.x *= act_grad_from_output(v.a.x, p.act); g.y *= act_grad_from_output(v.a.y, p.act);
                    g.z *= act_grad_from_output(v.a.z, p.act); g.w *= act_grad_from_output(v.a.w, p.act);
                } else if (recompute) {
                    g.x *= act_grad_from_output(apply_act(z.x * sc.x + sh.x, p.act), p.act);
                    g.y *= act_grad_from_output(apply_act(z.y * sc.y + sh.y, p.act), p.act);
                    g.z *= act_grad_from_output(apply_act(z.z * sc.z + sh.z, p.act), p.act);
                    g.w *= act_grad_from_output(apply_act(z.w * sc.w + sh.w, p.act), p.act);
                }
                if (p.dy_out) st4(p.dy_out + (long)row * p.lddy + c, g);
                s0.x += g.x; s0.y += g.y; s0.z += g.z; s0.w += g.w;
                if (p.mode == 2) {
                    s1.x += g.x * (z.x - mu.x) * rs.x; s1.y += g.y * (z.y - mu.y) * rs.y;
                    s1.z += g.z * (z.z - mu.z) * rs.z; s1.w += g.w * (z.w - mu.w) * rs.w;
                }
            }
        };
        const int stride = gridDim.y * 4;
        int row = blockIdx.y * 4 + rl;
        for (; row + stride < p.M; row += 2 * stride) {
            const RowIn v0 = fetch(row), v1 = fetch(row + stride);
            consume(row, v0);
            consume(row + stride, v1);
        }
        if (row < p.M) consume(row, fetch(row));
    }
    red[0][rl][chunk] = s0;
    red[1][rl][chunk] = s1;
    __syncthreads();
    if (rl == 0 && live) {
        float4 a = red[0][0][chunk], b = red[1][0][chunk];
#pragma unroll
        for (int r = 1; r < 4; ++r) {
            const float4 x = red[0][r][chunk], y = red[1][r][chunk];
            a.x += x.x; a.y += x.y; a.z += x.z; a.w += x.w;
            b.x += y.x; b.y += y.y; b.z += y.z; b.w += y.w;
        }
        // Float atomics on one address from hundreds of workgroups serialise at ~0.1 us each: with a workspace the row
        // splits are spread over COLREDUCE_SLOTS copies of the sums, which colreduce_finish_kernel adds up.
        float* o0 = p.out0;
        float* o1 = p.out1;
        if (p.ws) {
            const long cp = (long)gridDim.x * 256;
            o0 = p.ws + (long)(blockIdx.y % COLREDUCE_SLOTS) * 2 * cp;
            o1 = o0 + cp;
        }
        atomic_add_f32(o0 + c, a.x); atomic_add_f32(o0 + c + 1, a.y);
        atomic_add_f32(o0 + c + 2, a.z); atomic_add_f32(o0 + c + 3, a.w);
        if (p.mode != 1) {
            atomic_add_f32(o1 + c, b.x); atomic_add_f32(o1 + c + 1, b.y);
            atomic_add_f32(o1 + c + 2, b.z); atomic_add_f32(o1 + c + 3, b.w);
        }
    }
}

// adds the slots to out0 / out1 and leaves the workspace zero for the next call
__global__ __launch_bounds__(256) void colreduce_finish_kernel(float* ws, long cp, int C, float* out0, float* out1) {
    const int c = blockIdx.x * 256 + threadIdx.x;
    if (c >= C) return;
    float v0[COLREDUCE_SLOTS], v1[COLREDUCE_SLOTS];
#pragma unroll
    for (int k = 0; k < COLREDUCE_SLOTS; ++k) {
        v0[k] = ws[(long)k * 2 * cp + c];
        v1[k] = out1 ? ws[(long)k * 2 * cp + cp + c] : 0.0f;
    }
    float s0 = 0.0f, s1 = 0.0f;
#pragma unroll
    for (int k = 0; k < COLREDUCE_SLOTS; ++k) {
        s0 += v0[k];
        s1 += v1[k];
        ws[(long)k * 2 * cp + c] = 0.0f;
        if (out1) ws[(long)k * 2 * cp + cp + c] = 0.0f;
    }
    out0[c] += s0;
    if (out1) out1[c] += s1;
}

__global__ __launch_bounds__(256) void bn_finalize_kernel(const float* s0, const float* s1, const float* z0, int M, int C,
                                                          float eps, float momentum, float mv_scale, const float* gamma,
                                                          const float* beta,
                                                          float* moving_mean, float* moving_var, float* mean, float* rstd,
                                                          float* scale, float* shift, float* slots, long slot_stride,
                                                          float* about_next) {
    const int c = blockIdx.x * 256 + threadIdx.x;
    if (c >= C) return;
    float t0 = s0[c], t1 = s1[c];
    if (slots) {   // 16 copies of both sums, one thread per column: add up, then clear
        float v0[COLREDUCE_SLOTS], v1[COLREDUCE_SLOTS];
#pragma unroll
        for (int k = 0; k < COLREDUCE_SLOTS; ++k) {
            v0[k] = slots[(long)k * 2 * slot_stride + c];
            v1[k] = slots[(long)k * 2 * slot_stride + slot_stride + c];
        }
        t0 = t1 = 0.0f;
#pragma unroll
        for (int k = 0; k < COLREDUCE_SLOTS; ++k) {
            t0 += v0[k];
            t1 += v1[k];
            slots[(long)k * 2 * slot_stride + c] = 0.0f;
            slots[(long)k * 2 * slot_stride + slot_stride + c] = 0.0f;
        }
    }
    const float a = t0 / (float)M;
    const float mu = z0[c] + a;
    const float var = fmaxf(t1 / (float)M - a * a, 0.0f);
    const float rs = rsqrtf(var + eps);
    mean[c] = mu;
    if (about_next) about_next[c] = mu;   // (may be z0 itself: read above, by this thread alone)
    rstd[c] = rs;
    const float sc = gamma[c] * rs;
    scale[c] = sc;
    shift[c] = beta[c] - mu * sc;
    if (moving_mean) {
        moving_mean[c] = moving_mean[c] * momentum + mu * (1.0f - momentum);
        moving_var[c] = moving_var[c] * momentum + var * mv_scale * (1.0f - momentum);
    }
}

// ------------------------------------------------------------------------------------------
// backward of out = LNb(dropout(act(LNa(y))) + x) (+ the dot output); one wave per row, grid-stride over rows,
// parameter gradients kept per lane and reduced through LDS at the end
struct RowStats {
    float mean, rstd;
};

template <int NCH>
__device__ __forceinline__ RowStats row_stats(const float4 (&v)[NCH], int lane, int dq, int D, float eps) {
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < NCH; ++i)
        if (lane + 64 * i < dq) s += v[i].x + v[i].y + v[i].z + v[i].w;
    RowStats r;
    r.mean = wave_sum(s) / (float)D;
    float q = 0.f;
#pragma unroll
    for (int i = 0; i < NCH; ++i)
        if (lane + 64 * i < dq) {
            const float a = v[i].x - r.mean, b = v[i].y - r.mean, c = v[i].z - r.mean, d = v[i].w - r.mean;
            q += a * a + b * b + c * c + d * d;
        }
    r.rstd = rsqrtf(wave_sum(q) / (float)D + eps);
    return r;
}

#define F4_OP(dst, expr)                 \
    do {                                 \
        { const int e = 0; dst.x = expr; } \
        { const int e = 1; dst.y = expr; } \
        { const int e = 2; dst.z = expr; } \
        { const int e = 3; dst.w = expr; } \
    } while (0)

__device__ __forceinline__ float f4e(const float4& v, int e) { return e == 0 ? v.x : (e == 1 ? v.y : (e == 2 ? v.z : v.w)); }

// LayerNorm input gradient for one row: g = dout * gamma (in), xh = normalised input; returns rstd * (g - mean(g) - xh * mean(g xh))
template <int NCH>
__device__ __forceinline__ void ln_bwd_row(float4 (&g)[NCH], const float4 (&xh)[NCH], int lane, int dq, int D, float rstd) {
    float a = 0.f, b = 0.f;
#pragma unroll
    for (int i = 0; i < NCH; ++i)
        if (lane + 64 * i < dq) {
            a += g[i].x + g[i].y + g[i].z + g[i].w;
            b += g[i].x * xh[i].x + g[i].y * xh[i].y + g[i].z * xh[i].z + g[i].w * xh[i].w;
        }
    a = wave_sum(a) / (float)D;
    b = wave_sum(b) / (float)D;
#pragma unroll
    for (int i = 0; i < NCH; ++i)
        if (lane + 64 * i < dq) F4_OP(g[i], rstd * (f4e(g[i], e) - a - f4e(xh[i], e) * b));
}

constexpr int ROWNORM_SLOTS = 16;

// NCH: float4 chunks per lane (D <= 256 NCH); the per-row arrays are sized by it, so D = 256 runs with a quarter of the
// registers of D = 1024 and several times the waves in flight
template <int NCH>
__global__ __launch_bounds__(256) void rownorm_bwd_kernel(const estts_rownorm_bwd_args p) {
    __shared__ float red[4][1024];
    const estts_rownorm_args& f = p.f;
    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    const int dq = f.D >> 2;
    const float4 zero = make_float4(0.f, 0.f, 0.f, 0.f);
    float4 acc_ga[NCH], acc_ba[NCH], acc_gb[NCH], acc_bb[NCH], acc_dw[NCH];
#pragma unroll
    for (int i = 0; i < NCH; ++i) acc_ga[i] = acc_ba[i] = acc_gb[i] = acc_bb[i] = acc_dw[i] = zero;
    float acc_db = 0.f;
    const float keep = f.drop_rate > 0.f ? 1.0f / (1.0f - f.drop_rate) : 1.0f;

    // the LayerNorm parameters of this lane's columns stay in registers for all rows
    float4 p_ga[NCH], p_ba[NCH], p_gb[NCH], p_bb[NCH];
#pragma unroll
    for (int i = 0; i < NCH; ++i) {
        p_ga[i] = p_ba[i] = p_gb[i] = p_bb[i] = zero;
        if (lane + 64 * i < dq) {
            const int c = (lane + 64 * i) * 4;
            if (f.ga) { p_ga[i] = ld4(f.ga + c); p_ba[i] = ld4(f.ba + c); }
            if (f.gb) { p_gb[i] = ld4(f.gb + c); p_bb[i] = ld4(f.bb + c); }
        }
    }

    for (int row = blockIdx.x * 4 + wave; row < f.M; row += gridDim.x * 4) {
        float4 xh1[NCH], r[NCH], xh2[NCH], d[NCH];
        float rstd1 = 1.f, rstd2 = 1.f;
        // every operand of the row is requested here, before the first reduction waits for one of them: left where they are
        // used, the loads of x, dout and the old dx each start only after the wave reductions before them (three to four
        // dependent round trips per row)
        float4 x_in[NCH], dx_old[NCH];
#pragma unroll
        for (int i = 0; i < NCH; ++i) {
            xh1[i] = r[i] = xh2[i] = d[i] = x_in[i] = dx_old[i] = zero;
            if (lane + 64 * i < dq) {
                const int c = (lane + 64 * i) * 4;
                r[i] = ld4(f.y + (long)row * f.ldy + c);
                if (f.x) x_in[i] = ld4(f.x + (long)row * f.ldx + c);
                if (p.dout) d[i] = ld4(p.dout + (long)row * p.lddout + c);
                if (p.dx && p.dx_accumulate) dx_old[i] = ld4(p.dx + (long)row * p.lddx + c);
            }
        }
        // ---- recompute the forward pass
        if (f.ga) {
            const RowStats s = row_stats<NCH>(r, lane, dq, f.D, f.eps);
            rstd1 = s.rstd;
#pragma unroll
            for (int i = 0; i < NCH; ++i)
                if (lane + 64 * i < dq) {
                    const float4 ga = p_ga[i], ba = p_ba[i];
                    F4_OP(xh1[i], (f4e(r[i], e) - s.mean) * s.rstd);
                    F4_OP(r[i], f4e(xh1[i], e) * f4e(ga, e) + f4e(ba, e));
                }
        }
        // r = act(a); km = dropout factor; s = r * km + x
        float4 km[NCH];
#pragma unroll
        for (int i = 0; i < NCH; ++i) {
            km[i] = make_float4(1.f, 1.f, 1.f, 1.f);
            if (lane + 64 * i < dq) {
                const int c = (lane + 64 * i) * 4;
                F4_OP(r[i], apply_act(f4e(r[i], e), f.act));
                if (f.drop_rate > 0.f) F4_OP(km[i], dropout_keep(f.drop_seed, f.drop_site, row, c + e, f.drop_rate) ? keep : 0.0f);
                F4_OP(xh2[i], f4e(r[i], e) * f4e(km[i], e));
                if (f.x) {
                    const float4 x = x_in[i];
                    F4_OP(xh2[i], f4e(xh2[i], e) + f4e(x, e));
                }
            }
        }
        float4 out[NCH];
#pragma unroll
        for (int i = 0; i < NCH; ++i) out[i] = xh2[i];
        if (f.gb) {
            const RowStats s = row_stats<NCH>(xh2, lane, dq, f.D, f.eps);
            rstd2 = s.rstd;
#pragma unroll
            for (int i = 0; i < NCH; ++i)
                if (lane + 64 * i < dq) {
                    const float4 gb = p_gb[i], bb = p_bb[i];
                    F4_OP(xh2[i], (f4e(xh2[i], e) - s.mean) * s.rstd);
                    F4_OP(out[i], f4e(xh2[i], e) * f4e(gb, e) + f4e(bb, e));
                }
        }
        // ---- gradient w.r.t. out (d holds dout already)
        if (p.ddot) {
            float dot = 0.f;
#pragma unroll
            for (int i = 0; i < NCH; ++i)
                if (lane + 64 * i < dq) {
                    const float4 w = ld4(f.dot_w + (lane + 64 * i) * 4);
                    dot += out[i].x * w.x + out[i].y * w.y + out[i].z * w.z + out[i].w * w.w;
                }
            dot = wave_sum(dot) + (f.dot_b_ptr ? f.dot_b_ptr[0] : f.dot_b);
            float gm = dot > 0.f ? p.ddot[row] * f.dot_scale : 0.0f;
            if (f.dot_mask) gm *= 1.0f - f.dot_mask[row];
#pragma unroll
            for (int i = 0; i < NCH; ++i)
                if (lane + 64 * i < dq) {
                    const float4 w = ld4(f.dot_w + (lane + 64 * i) * 4);
                    F4_OP(acc_dw[i], f4e(acc_dw[i], e) + gm * f4e(out[i], e));
                    F4_OP(d[i], f4e(d[i], e) + gm * f4e(w, e));
                }
            if (lane == 0) acc_db += gm;
        }
        // ---- LNb
        if (f.gb) {
#pragma unroll
            for (int i = 0; i < NCH; ++i)
                if (lane + 64 * i < dq) {
                    const float4 gb = p_gb[i];
                    F4_OP(acc_gb[i], f4e(acc_gb[i], e) + f4e(d[i], e) * f4e(xh2[i], e));
                    F4_OP(acc_bb[i], f4e(acc_bb[i], e) + f4e(d[i], e));
                    F4_OP(d[i], f4e(d[i], e) * f4e(gb, e));
                }
            ln_bwd_row<NCH>(d, xh2, lane, dq, f.D, rstd2);
        }
        // ---- residual branch, dropout and activation
#pragma unroll
        for (int i = 0; i < NCH; ++i)
            if (lane + 64 * i < dq) {
                const int c = (lane + 64 * i) * 4;
                if (p.dx) {
                    float4 o = d[i];
                    if (p.dx_accumulate) {
                        const float4 old = dx_old[i];
                        o.x += old.x; o.y += old.y; o.z += old.z; o.w += old.w;
                    }
                    st4(p.dx + (long)row * p.lddx + c, o);
                }
                F4_OP(d[i], f4e(d[i], e) * f4e(km[i], e) * act_grad_from_output(f4e(r[i], e), f.act));
            }
        // ---- LNa
        if (f.ga) {
#pragma unroll
            for (int i = 0; i < NCH; ++i)
                if (lane + 64 * i < dq) {
                    const float4 ga = p_ga[i];
                    F4_OP(acc_ga[i], f4e(acc_ga[i], e) + f4e(d[i], e) * f4e(xh1[i], e));
                    F4_OP(acc_ba[i], f4e(acc_ba[i], e) + f4e(d[i], e));
                    F4_OP(d[i], f4e(d[i], e) * f4e(ga, e));
                }
            ln_bwd_row<NCH>(d, xh1, lane, dq, f.D, rstd1);
        }
        if (p.dy) {
#pragma unroll
            for (int i = 0; i < NCH; ++i)
                if (lane + 64 * i < dq) st4(p.dy + (long)row * p.lddy + (lane + 64 * i) * 4, d[i]);
        }
    }

    // ---- parameter gradients: sum the four waves through LDS, then one atomic per column and workgroup -- into one of
    // ROWNORM_SLOTS copies in the workspace when there is one (hundreds of workgroups adding to one address serialise at
    // ~0.1 us each); rownorm_bwd_finish_kernel adds the copies up
    const int slot = blockIdx.x % ROWNORM_SLOTS;
    auto flush = [&](const float4 (&acc)[NCH], float* dst, int which) {
        if (dst == nullptr) return;  // uniform across the workgroup
        if (p.ws) dst = p.ws + ((long)slot * 6 + which) * 1024;
        __syncthreads();
#pragma unroll
        for (int i = 0; i < NCH; ++i)
            if (lane + 64 * i < dq) *reinterpret_cast<float4*>(&red[wave][(lane + 64 * i) * 4]) = acc[i];
        __syncthreads();
        for (int c = threadIdx.x; c < f.D; c += 256) atomic_add_f32(dst + c, red[0][c] + red[1][c] + red[2][c] + red[3][c]);
    };
    flush(acc_ga, f.ga ? p.dga : nullptr, 0);
    flush(acc_ba, f.ga ? p.dba : nullptr, 1);
    flush(acc_gb, f.gb ? p.dgb : nullptr, 2);
    flush(acc_bb, f.gb ? p.dbb : nullptr, 3);
    flush(acc_dw, p.ddot ? p.ddot_w : nullptr, 4);
    if (p.ddot && p.ddot_b) {
        __syncthreads();
        if (lane == 0) red[0][wave] = acc_db;
        __syncthreads();
        if (threadIdx.x == 0)
            atomic_add_f32(p.ws ? p.ws + ((long)slot * 6 + 5) * 1024 : p.ddot_b, red[0][0] + red[0][1] + red[0][2] + red[0][3]);
    }
}

// adds the workspace copies to the gradient vectors and leaves the workspace zero; blockIdx.y = which vector
__global__ __launch_bounds__(256) void rownorm_bwd_finish_kernel(const estts_rownorm_bwd_args p) {
    float* dsts[6] = {p.f.ga ? p.dga : nullptr, p.f.ga ? p.dba : nullptr, p.f.gb ? p.dgb : nullptr, p.f.gb ? p.dbb : nullptr,
                      p.ddot ? p.ddot_w : nullptr, p.ddot ? p.ddot_b : nullptr};
    const int which = blockIdx.y;
    float* dst = dsts[which];
    if (dst == nullptr) return;
    const int n = which == 5 ? 1 : p.f.D;
    const int c = blockIdx.x * 256 + threadIdx.x;
    if (c >= n) return;
    float s = 0.0f;
#pragma unroll
    for (int k = 0; k < ROWNORM_SLOTS; ++k) {
        float* w = p.ws + ((long)k * 6 + which) * 1024 + c;
        s += *w;
        *w = 0.0f;
    }
    dst[c] += s;
}

int ew_grid(long total4) {
    const long b = (total4 + 255) / 256;
    return (int)(b < 1 ? 1 : (b > 8192 ? 8192 : b));
}

}  // namespace

extern "C" int estts_pack_weight(const float* src, int32_t ld_src, int32_t K, int32_t N, int32_t taps, int32_t mode,
                                 uint16_t* hi, uint16_t* lo, void* stream) {
    ESTTS_CHECK_ARG(src && hi && lo, "estts_pack_weight: null pointer");
    ESTTS_CHECK_ARG(K > 0 && N > 0 && taps >= 1 && K % taps == 0 && ld_src >= N && (mode == 0 || mode == 1),
                    "estts_pack_weight: bad shape K=%d N=%d taps=%d mode=%d", K, N, taps, mode);
    const int rows = mode == 0 ? N : K / taps;
    const int cols = mode == 0 ? K : taps * N;
    const int rp = (rows + 63) / 64 * 64, cp = (cols + 63) / 64 * 64;
    const long total = (long)(rp / 16) * (cp / 32) * 64;
    ESTTS_LAUNCH(pack_weight_kernel, dim3(ew_grid(total)), dim3(256), 0, (hipStream_t)stream, src, ld_src, K, N, taps,
                       mode, hi, lo, rows, cols, cp / 32, total);
    ESTTS_CHECK_LAUNCH("estts_pack_weight");
    return 0;
}

extern "C" int estts_pack_weights(const estts_pack_job* jobs, int32_t n_jobs, int64_t total_units, void* stream) {
    ESTTS_CHECK_ARG(jobs && n_jobs > 0 && total_units > 0, "estts_pack_weights: bad arguments");
    ESTTS_LAUNCH(pack_weights_kernel, dim3(ew_grid(total_units)), dim3(256), 0, (hipStream_t)stream, jobs, n_jobs, (long)total_units);
    ESTTS_CHECK_LAUNCH("estts_pack_weights");
    return 0;
}

extern "C" int estts_affine_act(const float* z, int32_t ldz, const float* scale, const float* shift, int32_t act,
                                const float* R, int32_t ldr, int32_t r_mod, float* out, int32_t ldo, int32_t M, int32_t C,
                                void* stream) {
    ESTTS_CHECK_ARG(z && out && M > 0 && C > 0 && (C & 3) == 0 && (ldz & 3) == 0 && (ldo & 3) == 0 && (!R || (ldr & 3) == 0),
                    "estts_affine_act: bad arguments (C and ld must be multiples of 4)");
    ESTTS_CHECK_ARG((scale == nullptr) == (shift == nullptr), "estts_affine_act: scale and shift go together");
    ESTTS_LAUNCH(affine_act_kernel, dim3(ew_grid((long)M * (C >> 2))), dim3(256), 0, (hipStream_t)stream, z, ldz, scale,
                       shift, act, R, ldr, r_mod, out, ldo, M, C);
    ESTTS_CHECK_LAUNCH("estts_affine_act");
    return 0;
}

extern "C" int estts_bn_bwd(const float* dy, int32_t lddy, const float* z, int32_t ldz, const float* mean, const float* rstd,
                            const float* gamma, const float* dgamma, const float* dbeta, float* dz, int32_t lddz, int32_t M,
                            int32_t C, void* stream) {
    ESTTS_CHECK_ARG(dy && z && mean && rstd && gamma && dgamma && dbeta && dz, "estts_bn_bwd: null pointer");
    ESTTS_CHECK_ARG(M > 0 && C > 0 && (C & 3) == 0 && (lddy & 3) == 0 && (ldz & 3) == 0 && (lddz & 3) == 0, "estts_bn_bwd: bad shape");
    ESTTS_LAUNCH(bn_bwd_kernel, dim3(ew_grid((long)M * (C >> 2))), dim3(256), 0, (hipStream_t)stream, dy, lddy, z, ldz,
                       mean, rstd, gamma, dgamma, dbeta, dz, lddz, M, C);
    ESTTS_CHECK_LAUNCH("estts_bn_bwd");
    return 0;
}

extern "C" int estts_bn_bwd_slots(const float* dy, int32_t lddy, const float* z, int32_t ldz, const float* mean, const float* rstd,
                                  const float* gamma, const float* slots, int64_t slot_stride, float* dgamma, float* dbeta,
                                  float* dz, int32_t lddz, int32_t M, int32_t C, int32_t act, const float* scale,
                                  const float* shift, void* stream) {
    ESTTS_CHECK_ARG((scale == nullptr) == (shift == nullptr), "estts_bn_bwd_slots: scale and shift go together");
    ESTTS_CHECK_ARG(dy && z && mean && rstd && gamma && slots && dgamma && dbeta && dz && M > 0 && C > 0 && (C & 3) == 0 &&
                        (lddy & 3) == 0 && (ldz & 3) == 0 && (lddz & 3) == 0,
                    "estts_bn_bwd_slots: bad arguments");
    ESTTS_CHECK_ARG(slot_stride == ((int64_t)(C + 255) / 256) * 256, "estts_bn_bwd_slots: slot_stride must be C rounded up to 256");
    const int gx = (C + 255) / 256;
    int ysplit = (M + 15) / 16;          // at least 16 rows per workgroup, about 1024 workgroups
    const int cap = (1024 + gx - 1) / gx;
    if (ysplit > cap) ysplit = cap;
    ESTTS_LAUNCH(bn_bwd_slots_kernel, dim3(gx, ysplit), dim3(256), 0, (hipStream_t)stream, dy, lddy, z, ldz, mean, rstd, gamma,
                       slots, (long)slot_stride, dgamma, dbeta, dz, lddz, M, C, act, scale, shift);
    ESTTS_CHECK_LAUNCH("estts_bn_bwd_slots");
    return 0;
}

extern "C" int estts_dropout(const float* x, int32_t ldx, float* out, int32_t ldo, int32_t M, int32_t D, float rate,
                             uint32_t seed, uint32_t site, void* stream) {
    ESTTS_CHECK_ARG(x && out && M > 0 && D > 0 && (D & 3) == 0 && (ldx & 3) == 0 && (ldo & 3) == 0, "estts_dropout: bad shape");
    ESTTS_CHECK_ARG(rate >= 0.0f && rate < 1.0f, "estts_dropout: rate must be in [0, 1)");
    ESTTS_LAUNCH(dropout_kernel, dim3(ew_grid((long)M * (D >> 2))), dim3(256), 0, (hipStream_t)stream, x, ldx, out, ldo, M,
                       D, rate, seed, site);
    ESTTS_CHECK_LAUNCH("estts_dropout");
    return 0;
}

extern "C" int estts_axpby(const float* x, int32_t ldx, float alpha, float* out, int32_t ldo, float beta, int32_t M, int32_t D,
                           void* stream) {
    ESTTS_CHECK_ARG(x && out && M > 0 && D > 0 && (D & 3) == 0 && (ldx & 3) == 0 && (ldo & 3) == 0, "estts_axpby: bad shape");
    ESTTS_LAUNCH(axpby_kernel, dim3(ew_grid((long)M * (D >> 2))), dim3(256), 0, (hipStream_t)stream, x, ldx, alpha, out,
                       ldo, beta, M, D);
    ESTTS_CHECK_LAUNCH("estts_axpby");
    return 0;
}

// ------------------------------------------------------------------------------------------------------
// Patches of a 2-D convolution over NHWC images (training of the GST reference encoder: the weight gradient and the data
// gradient of Conv2D become a dense wgrad / GEMM over these rows).  One thread per patch value / per input value.
// ------------------------------------------------------------------------------------------------------
__global__ void im2col_kernel(const float* __restrict__ x, float* __restrict__ out, int ldo, int B, int H, int W, int C, int Ho,
                              int Wo, int kh_n, int kw_n, int stride, int pad_t, int pad_l) {
    const long K = (long)kh_n * kw_n * C;
    const long total = (long)B * Ho * Wo * ldo;
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const long m = i / ldo;
        const int col = (int)(i - m * ldo);
        float v = 0.0f;
        if (col < K) {
            const int c = col % C, tap = col / C;
            const int kw = tap % kw_n, kh = tap / kw_n;
            const int wo = (int)(m % Wo);
            const long t = m / Wo;
            const int ho = (int)(t % Ho), b = (int)(t / Ho);
            const int h = ho * stride + kh - pad_t, w = wo * stride + kw - pad_l;
            if ((unsigned)h < (unsigned)H && (unsigned)w < (unsigned)W) v = x[(((long)b * H + h) * W + w) * C + c];
        }
        out[i] = v;
    }
}

// gather form of the transposed scatter: every input value sums the patch entries it was copied to (no atomics)
__global__ void col2im_kernel(const float* __restrict__ dp, int ldp, float* __restrict__ dx, int B, int H, int W, int C, int Ho,
                              int Wo, int kh_n, int kw_n, int stride, int pad_t, int pad_l) {
    const long total = (long)B * H * W * C;
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int c = (int)(i % C);
        long t = i / C;
        const int w = (int)(t % W);
        t /= W;
        const int h = (int)(t % H), b = (int)(t / H);
        float acc = 0.0f;
        for (int kh = 0; kh < kh_n; ++kh) {
            const int hn = h + pad_t - kh;
            if (hn < 0 || hn % stride) continue;
            const int ho = hn / stride;
            if (ho >= Ho) continue;
            for (int kw = 0; kw < kw_n; ++kw) {
                const int wn = w + pad_l - kw;
                if (wn < 0 || wn % stride) continue;
                const int wo = wn / stride;
                if (wo >= Wo) continue;
                acc += dp[(((long)b * Ho + ho) * Wo + wo) * ldp + (kh * kw_n + kw) * C + c];
            }
        }
        dx[i] = acc;
    }
}

extern "C" int estts_im2col(const float* x, float* out, int32_t ldo, int32_t B, int32_t H, int32_t W, int32_t C, int32_t Ho,
                            int32_t Wo, int32_t kh, int32_t kw, int32_t stride, int32_t pad_t, int32_t pad_l, void* stream) {
    ESTTS_CHECK_ARG(x && out && B > 0 && H > 0 && W > 0 && C > 0 && Ho > 0 && Wo > 0 && kh > 0 && kw > 0 && stride > 0 &&
                        pad_t >= 0 && pad_l >= 0 && ldo >= kh * kw * C,
                    "estts_im2col: bad arguments");
    ESTTS_CHECK_ARG((Ho - 1) * stride + kh - pad_t <= H + kh - 1 && (Wo - 1) * stride + kw - pad_l <= W + kw - 1,
                    "estts_im2col: output size does not fit the padded input");
    ESTTS_LAUNCH(im2col_kernel, dim3(ew_grid((long)B * Ho * Wo * ldo)), dim3(256), 0, (hipStream_t)stream, x, out, ldo, B, H,
                       W, C, Ho, Wo, kh, kw, stride, pad_t, pad_l);
    ESTTS_CHECK_LAUNCH("estts_im2col");
    return 0;
}

extern "C" int estts_col2im(const float* dp, int32_t ldp, float* dx, int32_t B, int32_t H, int32_t W, int32_t C, int32_t Ho,
                            int32_t Wo, int32_t kh, int32_t kw, int32_t stride, int32_t pad_t, int32_t pad_l, void* stream) {
    ESTTS_CHECK_ARG(dp && dx && B > 0 && H > 0 && W > 0 && C > 0 && Ho > 0 && Wo > 0 && kh > 0 && kw > 0 && stride > 0 &&
                        pad_t >= 0 && pad_l >= 0 && ldp >= kh * kw * C,
                    "estts_col2im: bad arguments");
    ESTTS_LAUNCH(col2im_kernel, dim3(ew_grid((long)B * H * W * C)), dim3(256), 0, (hipStream_t)stream, dp, ldp, dx, B, H, W,
                       C, Ho, Wo, kh, kw, stride, pad_t, pad_l);
    ESTTS_CHECK_LAUNCH("estts_col2im");
    return 0;
}

// ------------------------------------------------------------------------------------------------------
// Mutual-information estimators over the scores of one batch (a few hundred values at most): one workgroup.
// ------------------------------------------------------------------------------------------------------
constexpr int MI_THREADS = 256;

__device__ float mi_block_sum(float v, float* red) {
    v = wave_sum(v);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
    __syncthreads();
    float t = 0.0f;
    for (int w = 0; w < MI_THREADS / 64; ++w) t += red[w];
    return t;
}

__device__ float mi_block_max(float v, float* red) {
    v = wave_max(v);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
    __syncthreads();
    float t = red[0];
    for (int w = 1; w < MI_THREADS / 64; ++w) t = fmaxf(t, red[w]);
    return t;
}

// log(s * mean(exp(a x - max(a x))) + (1 - s) * prev) + max(a x) with its gradient coef * d/dx added to g:
// d/dx_i = a * [ (s / E) * exp(a x_i - mx) / n + [i == argmax] * (1 - s * e / E) ]   (the maximum carries gradient too, as
// tf.reduce_max does; with s = 1 that part vanishes).  Returns the value; *e_out receives E.
__device__ float mi_smoothed_lme(const float* x, int n, float a, float s, float prev, float coef, float* g, float* red, float* e_out) {
    float mx = -INFINITY;
    for (int i = threadIdx.x; i < n; i += MI_THREADS) mx = fmaxf(mx, a * x[i]);
    mx = mi_block_max(mx, red);
    float e = 0.0f;
    for (int i = threadIdx.x; i < n; i += MI_THREADS) e += __expf(a * x[i] - mx);
    e = mi_block_sum(e, red) / n;
    const float E = s * e + (1.0f - s) * prev;
    // the first index that attains the maximum takes its gradient
    int arg = n;
    for (int i = threadIdx.x; i < n; i += MI_THREADS)
        if (a * x[i] == mx) { arg = i; break; }
    arg = -(int)mi_block_max(-(float)arg, red);
    for (int i = threadIdx.x; i < n; i += MI_THREADS) {
        float d = (s / E) * __expf(a * x[i] - mx) / n;
        if (i == arg) d += 1.0f - s * e / E;
        g[i] += coef * a * d;
    }
    *e_out = E;
    return logf(E) + mx;
}

// MINE.measure_mi (TransformerTTS/model/MINE.py:40-95).  out[0] = mi, d_joint / d_marginal = gscale * d mi / d score.
__global__ __launch_bounds__(MI_THREADS) void mine_divergence_kernel(const float* __restrict__ joint, const float* __restrict__ marginal,
                                                                     int n, int div_type, const float* __restrict__ betas, int n_beta,
                                                                     float s, const float* __restrict__ prev, float* __restrict__ next,
                                                                     float* __restrict__ out, float* __restrict__ dj,
                                                                     float* __restrict__ dm, float gscale) {
    __shared__ float red[MI_THREADS / 64];
    for (int i = threadIdx.x; i < n; i += MI_THREADS) { dj[i] = 0.0f; dm[i] = 0.0f; }
    __syncthreads();
    float mean_j = 0.0f, mean_m = 0.0f;
    for (int i = threadIdx.x; i < n; i += MI_THREADS) { mean_j += joint[i]; mean_m += marginal[i]; }
    mean_j = mi_block_sum(mean_j, red) / n;
    mean_m = mi_block_sum(mean_m, red) / n;
    float loss = 0.0f;
    if (div_type == 0) {   // KL: mean(joint) - log(smoothed mean(exp(marginal))), no maximum taken out (MINE.py:46-55)
        float e = 0.0f;
        for (int i = threadIdx.x; i < n; i += MI_THREADS) e += __expf(marginal[i]);
        e = mi_block_sum(e, red) / n;
        const float E = s * e + (1.0f - s) * prev[1];
        for (int i = threadIdx.x; i < n; i += MI_THREADS) {
            dj[i] += gscale / n;
            dm[i] -= gscale * (s / E) * __expf(marginal[i]) / n;
        }
        loss = mean_j - logf(E);
        if (threadIdx.x == 0) { next[0] = 0.0f; next[1] = E; }
    } else {               // Renyi divergences, one term per beta (MINE.py:56-93)
        for (int b = 0; b < n_beta; ++b) {
            const float beta = betas[b];
            float term1, term2, e1 = 0.0f, e2 = 0.0f;
            if (beta == 0.0f) {
                term1 = mean_j;
                for (int i = threadIdx.x; i < n; i += MI_THREADS) dj[i] += gscale / n;
            } else {
                term1 = -(1.0f / beta) * mi_smoothed_lme(joint, n, -beta, s, prev[2 * b], -gscale / beta, dj, red, &e1);
            }
            if (beta == 1.0f) {
                term2 = mean_m;
                for (int i = threadIdx.x; i < n; i += MI_THREADS) dm[i] -= gscale / n;
            } else {
                term2 = (1.0f / (1.0f - beta)) * mi_smoothed_lme(marginal, n, 1.0f - beta, s, prev[2 * b + 1], -gscale / (1.0f - beta), dm,
                                                                 red, &e2);
            }
            loss += term1 - term2;
            if (threadIdx.x == 0) { next[2 * b] = e1; next[2 * b + 1] = e2; }
        }
    }
    if (threadIdx.x == 0) out[0] = loss;
}

// CLUB.call (TransformerTTS/model/MINE.py:160-200): positive = -(mu - y)^2 / 2 / exp(log_var), negative the same against the
// shuffled y; lld = mean_b sum_c positive; bound = mean_b (sum_c positive - sum_c negative).  d_mu / d_lv = gscale * d lld / d(.)
__global__ __launch_bounds__(MI_THREADS) void club_terms_kernel(const float* __restrict__ mu, const float* __restrict__ lv,
                                                                const float* __restrict__ y, const float* __restrict__ ys, int B, int D,
                                                                float* __restrict__ out, float* __restrict__ dmu, float* __restrict__ dlv,
                                                                float gscale) {
    __shared__ float red[MI_THREADS / 64];
    float pos = 0.0f, neg = 0.0f;
    for (int i = threadIdx.x; i < B * D; i += MI_THREADS) {
        const float iv = __expf(-lv[i]);
        const float dp = mu[i] - y[i], dn = mu[i] - ys[i];
        pos += -0.5f * dp * dp * iv;
        neg += -0.5f * dn * dn * iv;
        dmu[i] = gscale * (-dp * iv) / B;
        dlv[i] = gscale * (0.5f * dp * dp * iv) / B;
    }
    pos = mi_block_sum(pos, red);
    neg = mi_block_sum(neg, red);
    if (threadIdx.x == 0) { out[0] = pos / B; out[1] = (pos - neg) / B; }
}

// l2_loss of the style loss (TransformerTTS/utils/losses.py:65-68): Keras MeanSquaredError over every element, one workgroup.
__global__ __launch_bounds__(MI_THREADS) void mse_kernel(const float* __restrict__ a, const float* __restrict__ b, int n,
                                                         float* __restrict__ loss, float* __restrict__ da, float scale) {
    __shared__ float red[MI_THREADS / 64];
    float s = 0.0f;
    for (int i = threadIdx.x; i < n; i += MI_THREADS) {
        const float d = a[i] - b[i];
        s += d * d;
        da[i] = scale * 2.0f * d / n;
    }
    s = mi_block_sum(s, red);
    if (threadIdx.x == 0) loss[0] += s / n;
}

extern "C" int estts_mse(const float* a, const float* b, int32_t n, float* loss, float* da, float scale, void* stream) {
    ESTTS_CHECK_ARG(a && b && loss && da && n > 0, "estts_mse: bad arguments");
    ESTTS_LAUNCH(mse_kernel, dim3(1), dim3(MI_THREADS), 0, (hipStream_t)stream, a, b, n, loss, da, scale);
    ESTTS_CHECK_LAUNCH("estts_mse");
    return 0;
}

extern "C" int estts_mine_divergence(const float* joint, const float* marginal, int32_t n, int32_t div_type, const float* betas,
                                     int32_t n_beta, float smoothing, const float* prev_exp, float* next_exp, float* mi,
                                     float* d_joint, float* d_marginal, float gscale, void* stream) {
    ESTTS_CHECK_ARG(joint && marginal && prev_exp && next_exp && mi && d_joint && d_marginal && n > 0, "estts_mine_divergence: bad arguments");
    ESTTS_CHECK_ARG(div_type == 0 || (div_type == 1 && betas && n_beta > 0), "estts_mine_divergence: divergence 0 (KL) or 1 (Renyi with betas)");
    ESTTS_LAUNCH(mine_divergence_kernel, dim3(1), dim3(MI_THREADS), 0, (hipStream_t)stream, joint, marginal, n, div_type, betas,
                       n_beta, smoothing, prev_exp, next_exp, mi, d_joint, d_marginal, gscale);
    ESTTS_CHECK_LAUNCH("estts_mine_divergence");
    return 0;
}

extern "C" int estts_club_terms(const float* mu, const float* log_var, const float* y, const float* y_shuffled, int32_t B, int32_t D,
                                float* out, float* d_mu, float* d_log_var, float gscale, void* stream) {
    ESTTS_CHECK_ARG(mu && log_var && y && y_shuffled && out && d_mu && d_log_var && B > 0 && D > 0, "estts_club_terms: bad arguments");
    ESTTS_LAUNCH(club_terms_kernel, dim3(1), dim3(MI_THREADS), 0, (hipStream_t)stream, mu, log_var, y, y_shuffled, B, D, out,
                       d_mu, d_log_var, gscale);
    ESTTS_CHECK_LAUNCH("estts_club_terms");
    return 0;
}

extern "C" int estts_head_scale(float* x, int32_t ldx, const float* scale, int32_t B, int32_t L, int32_t H, int32_t depth,
                                void* stream) {
    ESTTS_CHECK_ARG(x && scale && B > 0 && L > 0 && H > 0 && depth > 0 && (depth & 3) == 0 && (ldx & 3) == 0 && ldx >= H * depth,
                    "estts_head_scale: bad arguments");
    ESTTS_LAUNCH(head_scale_kernel, dim3(ew_grid((long)B * L * (H * depth >> 2))), dim3(256), 0, (hipStream_t)stream, x, ldx,
                       scale, B * L, L, H, depth);
    ESTTS_CHECK_LAUNCH("estts_head_scale");
    return 0;
}

extern "C" int estts_embed_bwd(const int32_t* tokens, const float* dx, int32_t lddx, float* demb, int32_t B, int32_t N, int32_t D,
                               float scale, void* stream) {
    ESTTS_CHECK_ARG(tokens && dx && demb && B > 0 && N > 0 && D > 0 && (D & 3) == 0 && (lddx & 3) == 0, "estts_embed_bwd: bad arguments");
    ESTTS_LAUNCH(embed_bwd_kernel, dim3(ew_grid((long)B * N * (D >> 2))), dim3(256), 0, (hipStream_t)stream, tokens, dx,
                       lddx, demb, B * N, D, scale);
    ESTTS_CHECK_LAUNCH("estts_embed_bwd");
    return 0;
}

extern "C" int estts_expand_bwd(const float* dout, int32_t lddo, const int32_t* cum, float* dx, int32_t lddx, int32_t B, int32_t N,
                                int32_t T, int32_t D, void* stream) {
    ESTTS_CHECK_ARG(dout && cum && dx && B > 0 && N > 0 && T > 0 && D > 0 && (D & 3) == 0 && (lddo & 3) == 0 && (lddx & 3) == 0,
                    "estts_expand_bwd: bad arguments");
    const long toks = (long)B * N;
    ESTTS_LAUNCH(expand_bwd_kernel, dim3((unsigned)((toks + 3) / 4)), dim3(256), 0, (hipStream_t)stream, dout, lddo, cum, dx,
                       lddx, B, N, T, D);
    ESTTS_CHECK_LAUNCH("estts_expand_bwd");
    return 0;
}

extern "C" int estts_masked_mae_bwd(const float* target, int32_t ldt, const float* pred, int32_t ldp, float* dpred, int32_t lddp,
                                    int32_t M, int32_t C, float scale, void* stream) {
    ESTTS_CHECK_ARG(target && pred && dpred && M > 0 && C > 0 && ldt >= C && ldp >= C && lddp >= C, "estts_masked_mae_bwd: bad arguments");
    ESTTS_LAUNCH(masked_mae_bwd_kernel, dim3((M + 3) / 4), dim3(256), 0, (hipStream_t)stream, target, ldt, pred, ldp, dpred,
                       lddp, M, C, scale);
    ESTTS_CHECK_LAUNCH("estts_masked_mae_bwd");
    return 0;
}

extern "C" int estts_adam(float* p, const float* grad, float* m, float* v, int64_t n, float lr_t, float beta1, float beta2,
                          float eps, float grad_scale, void* stream) {
    ESTTS_CHECK_ARG(p && grad && m && v && n > 0 && (n & 3) == 0, "estts_adam: n must be a positive multiple of 4");
    ESTTS_CHECK_ARG((((uintptr_t)p | (uintptr_t)grad | (uintptr_t)m | (uintptr_t)v) & 15) == 0, "estts_adam: buffers must be 16-byte aligned");
    ESTTS_LAUNCH(adam_kernel, dim3(ew_grid(n / 4)), dim3(256), 0, (hipStream_t)stream, p, grad, m, v, (long)(n / 4), lr_t,
                       beta1, beta2, eps, grad_scale);
    ESTTS_CHECK_LAUNCH("estts_adam");
    return 0;
}

extern "C" int estts_colreduce(const estts_colreduce_args* args, void* stream) {
    ESTTS_CHECK_ARG(args != nullptr, "estts_colreduce: null args");
    const estts_colreduce_args& a = *args;
    ESTTS_CHECK_ARG(a.M > 0 && a.C > 0 && (a.C & 3) == 0 && a.mode >= 0 && a.mode <= 2, "estts_colreduce: bad shape / mode");
    ESTTS_CHECK_ARG(a.defer || (a.out0 && (a.mode == 1 || a.out1)), "estts_colreduce: null output");
    if (a.mode == 0) {
        ESTTS_CHECK_ARG(a.z && (a.ldz & 3) == 0, "estts_colreduce: mode 0 needs z");
    } else {
        ESTTS_CHECK_ARG(a.g && (a.ldg & 3) == 0 && (!a.a || (a.lda & 3) == 0) && (!a.dy_out || (a.lddy & 3) == 0),
                        "estts_colreduce: modes 1/2 need g (ld multiples of 4)");
        ESTTS_CHECK_ARG(a.mode == 1 || (a.z && a.mean && a.rstd && (a.ldz & 3) == 0), "estts_colreduce: mode 2 needs z, mean, rstd");
        ESTTS_CHECK_ARG((a.scale == nullptr) == (a.shift == nullptr), "estts_colreduce: scale and shift go together");
        ESTTS_CHECK_ARG(!a.scale || a.a || (a.z && (a.ldz & 3) == 0), "estts_colreduce: recomputing the activation needs z");
    }
    // enough workgroups to keep every CU's memory pipeline busy (the row loop is a chain of dependent loads), at least 16
    // rows per workgroup (about 512 workgroups measured best; ESTTS_COLREDUCE_WGS overrides it for measurements); without a
    // workspace all row splits meet on the same addresses, which caps their useful number
    static const int target = [] { const char* e = getenv("ESTTS_COLREDUCE_WGS"); return e ? atoi(e) : 512; }();
    const int gx = (a.C + 255) / 256;
    const long cp = (long)gx * 256;
    estts_colreduce_args k = a;
    if (a.defer) {
        ESTTS_CHECK_ARG(a.mode != 1, "estts_colreduce: defer is for modes 0 and 2");
        ESTTS_CHECK_ARG(a.ws && a.ws_bytes >= (int64_t)COLREDUCE_SLOTS * 2 * cp * 4,
                        "estts_colreduce: defer needs ws of %ld bytes", (long)COLREDUCE_SLOTS * 2 * cp * 4);
        ESTTS_CHECK_ARG(!a.clear || (a.clear_floats >= 0 && (a.clear + a.clear_floats <= a.ws || a.clear >= a.ws + (long)COLREDUCE_SLOTS * 2 * cp)),
                        "estts_colreduce: clear must not overlap the sums at ws");
        int ys = (a.M + 15) / 16;
        const int cap_d = (target + gx - 1) / gx;
        if (ys > cap_d) ys = cap_d;
        ESTTS_LAUNCH(colreduce_kernel, dim3(gx, ys), dim3(256), 0, (hipStream_t)stream, k);
        ESTTS_CHECK_LAUNCH("estts_colreduce");
        return 0;
    }
    k.clear = nullptr;
    if (k.ws && k.ws_bytes < (int64_t)COLREDUCE_SLOTS * 2 * cp * 4) k.ws = nullptr;
    int ysplit = (a.M + 15) / 16;
    const int cap = k.ws ? (target + gx - 1) / gx : 128;
    if (ysplit > cap) ysplit = cap;
    if (ysplit <= 32) k.ws = nullptr;
    dim3 grid(gx, ysplit);
    ESTTS_LAUNCH(colreduce_kernel, grid, dim3(256), 0, (hipStream_t)stream, k);
    if (k.ws) {
        ESTTS_CHECK_LAUNCH("estts_colreduce");
        ESTTS_LAUNCH(colreduce_finish_kernel, dim3(gx), dim3(256), 0, (hipStream_t)stream, k.ws, cp, a.C, a.out0,
                           a.mode != 1 ? a.out1 : nullptr);
    }
    ESTTS_CHECK_LAUNCH("estts_colreduce");
    return 0;
}

extern "C" int estts_bn_finalize(const float* s0, const float* s1, const float* z0, int32_t M, int32_t C, float eps,
                                 float momentum, float moving_var_scale, const float* gamma, const float* beta,
                                 float* moving_mean, float* moving_var, float* mean, float* rstd, float* scale, float* shift,
                                 void* stream) {
    ESTTS_CHECK_ARG(s0 && s1 && z0 && gamma && beta && mean && rstd && scale && shift && M > 0 && C > 0, "estts_bn_finalize: bad arguments");
    ESTTS_CHECK_ARG((moving_mean == nullptr) == (moving_var == nullptr), "estts_bn_finalize: moving statistics go together");
    ESTTS_LAUNCH(bn_finalize_kernel, dim3((C + 255) / 256), dim3(256), 0, (hipStream_t)stream, s0, s1, z0, M, C, eps, momentum,
                       moving_var_scale, gamma, beta, moving_mean, moving_var, mean, rstd, scale, shift, (float*)nullptr, 0L,
                       (float*)nullptr);
    ESTTS_CHECK_LAUNCH("estts_bn_finalize");
    return 0;
}

extern "C" int estts_bn_finalize_slots(float* slots, int64_t slot_stride, const float* z0, int32_t M, int32_t C, float eps,
                                       float momentum, float moving_var_scale, const float* gamma, const float* beta,
                                       float* moving_mean, float* moving_var, float* mean, float* rstd, float* scale,
                                       float* shift, float* about_next, void* stream) {
    ESTTS_CHECK_ARG(slots && z0 && gamma && beta && mean && rstd && scale && shift && M > 0 && C > 0, "estts_bn_finalize_slots: bad arguments");
    ESTTS_CHECK_ARG(slot_stride == ((int64_t)(C + 255) / 256) * 256, "estts_bn_finalize_slots: slot_stride must be C rounded up to 256");
    ESTTS_CHECK_ARG((moving_mean == nullptr) == (moving_var == nullptr), "estts_bn_finalize_slots: moving statistics go together");
    ESTTS_LAUNCH(bn_finalize_kernel, dim3((C + 255) / 256), dim3(256), 0, (hipStream_t)stream, slots, slots, z0, M, C, eps,
                       momentum, moving_var_scale, gamma, beta, moving_mean, moving_var, mean, rstd, scale, shift, slots,
                       (long)slot_stride, about_next);
    ESTTS_CHECK_LAUNCH("estts_bn_finalize_slots");
    return 0;
}

extern "C" int estts_rownorm_bwd(const estts_rownorm_bwd_args* args, void* stream) {
    ESTTS_CHECK_ARG(args != nullptr, "estts_rownorm_bwd: null args");
    const estts_rownorm_bwd_args& a = *args;
    const estts_rownorm_args& f = a.f;
    ESTTS_CHECK_ARG(f.y && f.M > 0 && f.D > 0 && f.D <= 1024 && (f.D & 3) == 0, "estts_rownorm_bwd: D=%d must be <=1024, multiple of 4", f.D);
    ESTTS_CHECK_ARG(a.dout || a.ddot, "estts_rownorm_bwd: no incoming gradient");
    ESTTS_CHECK_ARG(!a.ddot || f.dot_w, "estts_rownorm_bwd: ddot without dot_w");
    ESTTS_CHECK_ARG((f.ldy & 3) == 0 && (!f.x || (f.ldx & 3) == 0) && (!a.dout || (a.lddout & 3) == 0) && (!a.dy || (a.lddy & 3) == 0) &&
                        (!a.dx || (a.lddx & 3) == 0),
                    "estts_rownorm_bwd: ld not multiple of 4");
    ESTTS_CHECK_ARG(!a.dx || f.x, "estts_rownorm_bwd: dx without a residual input");
    const int nch = (f.D / 4 + 63) / 64;
    // workgroups the instantiation keeps resident (four per CU at D <= 256, two above): rows beyond that are looped over
    const int cap = nch == 1 ? 1024 : 512;
    int grid = (f.M + 3) / 4;
    if (grid > cap) grid = cap;
    estts_rownorm_bwd_args k = a;
    if (k.ws && (k.ws_bytes < (int64_t)ROWNORM_SLOTS * 6 * 1024 * 4 || grid <= 32)) k.ws = nullptr;
    if (nch == 1) ESTTS_LAUNCH(rownorm_bwd_kernel<1>, dim3(grid), dim3(256), 0, (hipStream_t)stream, k);
    else if (nch == 2) ESTTS_LAUNCH(rownorm_bwd_kernel<2>, dim3(grid), dim3(256), 0, (hipStream_t)stream, k);
    else ESTTS_LAUNCH(rownorm_bwd_kernel<4>, dim3(grid), dim3(256), 0, (hipStream_t)stream, k);
    if (k.ws) {
        ESTTS_CHECK_LAUNCH("estts_rownorm_bwd");
        ESTTS_LAUNCH(rownorm_bwd_finish_kernel, dim3((f.D + 255) / 256, 6), dim3(256), 0, (hipStream_t)stream, k);
    }
    ESTTS_CHECK_LAUNCH("estts_rownorm_bwd");
    return 0;
}
