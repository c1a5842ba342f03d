// A-operand addressing shared by the forward / data-gradient product (gemm.hip) and the weight gradient (gemm_wgrad.hip).
#pragma once
#include "common.h"

// ---- A operand addressing.  Output row m is pixel (img, ho, wo) of a [*, Ho, Wo] output grid; K index kk is
// (tap, channel) with tap = (kh, kw); the source pixel is (ho * stride + kh - pad_top, wo * stride + kw - pad_left) of a
// [*, H, W] input grid with lda floats per pixel; pixels outside the grid read as zero.  A dense layer is the 1 x 1 case
// (H = Ho = 1, W = Wo = M), a 1-D convolution over rows has H = 1, W = seg_len, kw = taps.
struct ARow {
    int pix0;   // img * H * W
    int h0, w0; // ho * stride - pad_top, wo * stride - pad_left
    bool ok;
};
struct ATap {
    int dh, dw, c;
    bool ok;
};

__device__ __forceinline__ ARow make_arow(const estts_gemm_args& p, int m) {
    ARow r;
    r.ok = m < p.M;
    const int img = m / p.conv_howo;
    const int rem = m - img * p.conv_howo;
    const int ho = rem / p.conv_wo;
    const int wo = rem - ho * p.conv_wo;
    r.pix0 = img * p.conv_h * p.conv_w;
    r.h0 = ho * p.conv_stride - p.conv_pt;
    r.w0 = wo * p.conv_stride - p.conv_pl;
    return r;
}

__device__ __forceinline__ ATap atap_from(const estts_gemm_args& p, int kk, int tap, int c) {
    ATap t;
    t.ok = kk < p.K;
    t.c = c;
    if (p.taps > p.conv_kw) {   // 2-D kernel: tap = (kh, kw)
        t.dh = tap / p.conv_kw;
        t.dw = tap - t.dh * p.conv_kw;
    } else {                    // dense layer or 1-D convolution: one kernel row
        t.dh = 0;
        t.dw = tap;
    }
    return t;
}

__device__ __forceinline__ ATap make_atap(const estts_gemm_args& p, int kk) {
    int tap = 0, c = kk;
    if (p.taps > 1) {
        tap = kk / p.c_per_tap;
        c = kk - tap * p.c_per_tap;
    }
    return atap_from(p, kk, tap, c);
}

// (tap, channel) of a K index that only ever advances by a fixed step: one division at the start, none in the K loop
struct KCursor {
    int kk, tap, c;
};
__device__ __forceinline__ KCursor make_kcursor(const estts_gemm_args& p, int kk) {
    KCursor k;
    k.kk = kk;
    k.tap = p.taps > 1 ? kk / p.c_per_tap : 0;
    k.c = kk - k.tap * p.c_per_tap;
    return k;
}
__device__ __forceinline__ void advance(const estts_gemm_args& p, KCursor& k, int step) {
    k.kk += step;
    k.c += step;
    if (p.taps > 1)
        while (k.c >= p.c_per_tap) {
            k.c -= p.c_per_tap;
            ++k.tap;
        }
}

// The load is unconditional (a padded position reads element 0, which exists) and the caller zeroes the value where it
// USES it (zero_unless): a load under a condition becomes a branch around it, and a select right behind the load makes
// the wave wait for the data at the point of issue; either way the K loop would run at memory latency.
__device__ __forceinline__ float4 load_a_raw(const estts_gemm_args& p, const ARow& r, const ATap& t, bool& ok) {
    const int hi = r.h0 + t.dh;
    const int wi = r.w0 + t.dw;
    ok = r.ok && t.ok && hi >= 0 && hi < p.conv_h && wi >= 0 && wi < p.conv_w;
    const long idx = ok ? (long)(r.pix0 + hi * p.conv_w + wi) * p.lda + t.c : 0L;
    return *reinterpret_cast<const float4*>(p.A + idx);
}

__device__ __forceinline__ float4 zero_unless(bool ok, float4 v) {
    v.x = ok ? v.x : 0.0f;
    v.y = ok ? v.y : 0.0f;
    v.z = ok ? v.z : 0.0f;
    v.w = ok ? v.w : 0.0f;
    return v;
}

__device__ __forceinline__ float4 load_a(const estts_gemm_args& p, const ARow& r, const ATap& t) {
    bool ok;
    const float4 v = load_a_raw(p, r, t, ok);
    return zero_unless(ok, v);
}
