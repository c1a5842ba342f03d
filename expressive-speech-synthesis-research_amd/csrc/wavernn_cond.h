// Conditioning values of one WaveRNN step for one (sequence, group of eight panel columns), shared by the conditioning
// kernel and the sampler (which prepares the next step's conditioning right after its draw).
#pragma once
#include "common.h"

// columns in the order [x_prev group (8) | mel | aux slice 1 | zeros up to 128 | a2 | a3 | a4]; group q covers columns
// 8 q .. 8 q + 7 of the step input (q < 16) or of the aux slices (q >= 16).  Rows b >= n_seq (the tail of the last row
// block) are written as zeros so that whole blocks stay defined.
__device__ __forceinline__ void wavernn_cond_group(const estts_wavernn_cond_args& p, int b, int q, int step) {
    float v[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) v[e] = 0.0f;
    const int dq = p.d >> 3;
    if (b < p.n_seq) {
        // one 16-byte record per sequence instead of the chain sequence -> utterance -> offsets (each link a memory round trip)
        const int4 info = reinterpret_cast<const int4*>(p.seq_info)[b];   // (first mel row, first aux row, start sample, samples)
        const int s = info.z + step;
        const bool live = s < info.w;
        const int frame = s / p.hop;
        const int phase = s - frame * p.hop;
        const int n_in = 8 + p.feat + p.d;
        const int col0 = q < 16 ? 8 * q : 128 + 8 * (q - 16);
        if (live && (p.feat & 7) == 0) {
            // whole groups of eight columns never straddle the mel / aux boundary: two 16-byte loads per filter tap or aux row
            if (col0 >= 8 && col0 < 8 + p.feat) {
                const float* m = p.mel + (long)(info.x + frame) * p.feat + (col0 - 8);
                const float* f = p.fir + phase * p.ntaps;
                for (int j = 0; j < p.ntaps; ++j) {
                    const float fj = f[j];
                    const float4 a = *reinterpret_cast<const float4*>(m - (long)j * p.feat);
                    const float4 c = *reinterpret_cast<const float4*>(m - (long)j * p.feat + 4);
                    v[0] += fj * a.x; v[1] += fj * a.y; v[2] += fj * a.z; v[3] += fj * a.w;
                    v[4] += fj * c.x; v[5] += fj * c.y; v[6] += fj * c.z; v[7] += fj * c.w;
                }
            } else if ((col0 >= 8 + p.feat && col0 < n_in) || col0 >= 128) {
                const float* src = p.aux + (long)(info.y + frame) * (4 * p.d) + (col0 >= 128 ? p.d + (col0 - 128) : col0 - 8 - p.feat);
                const float4 a = *reinterpret_cast<const float4*>(src);
                const float4 c = *reinterpret_cast<const float4*>(src + 4);
                v[0] = a.x; v[1] = a.y; v[2] = a.z; v[3] = a.w; v[4] = c.x; v[5] = c.y; v[6] = c.z; v[7] = c.w;
            }
        } else {
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const int col = col0 + e;
                float x = 0.0f;
                if (live) {
                    if (col >= 8 && col < 8 + p.feat) {
                        const float* m = p.mel + (long)(info.x + frame) * p.feat + (col - 8);
                        const float* f = p.fir + phase * p.ntaps;
                        for (int j = 0; j < p.ntaps; ++j) x += f[j] * m[-(long)j * p.feat];
                    } else if (col >= 8 + p.feat && col < n_in) {
                        x = p.aux[(long)(info.y + frame) * (4 * p.d) + (col - 8 - p.feat)];
                    } else if (col >= 128) {
                        x = p.aux[(long)(info.y + frame) * (4 * p.d) + p.d + (col - 128)];
                    }
                }
                v[e] = x;
            }
        }
    }
    bf16x4 h0, l0, h1, l1;
    split_bf16x4(make_float4(v[0], v[1], v[2], v[3]), h0, l0);
    split_bf16x4(make_float4(v[4], v[5], v[6], v[7]), h1, l1);
    const bf16x8 hi = {h0[0], h0[1], h0[2], h0[3], h1[0], h1[1], h1[2], h1[3]};
    const bf16x8 lo = {l0[0], l0[1], l0[2], l0[3], l1[0], l1[1], l1[2], l1[3]};
    uint16_t* panel;
    int kpitch, col;
    if (q < 16) {
        panel = p.xin_panel; kpitch = 4; col = 8 * q;
    } else {
        const int which = (q - 16) / dq;
        panel = which == 0 ? p.a2_panel : (which == 1 ? p.a3_panel : p.a4_panel);
        kpitch = p.aux_kpitch;
        col = 32 * p.aux_ks + 8 * ((q - 16) - which * dq);
    }
    unsigned char* dst = reinterpret_cast<unsigned char*>(panel) + ((long)(b >> 4) * kpitch + (col >> 5)) * 2048 +
                         ((b & 15) + 16 * ((col & 31) >> 3)) * 16;
    *reinterpret_cast<bf16x8*>(dst) = hi;
    *reinterpret_cast<bf16x8*>(dst + 1024) = lo;
}
