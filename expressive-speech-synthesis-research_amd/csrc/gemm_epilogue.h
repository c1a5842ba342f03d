// Epilogue and tile-order helpers shared by the estts_gemm kernels (gemm.hip, gemm_oneshot.hip).
#pragma once
#include "common.h"

// `res`: the value of R for this element, loaded by the caller (ignored when there is no R)
__device__ __forceinline__ float epilogue_value_r(const estts_gemm_args& p, float acc, int row, int col, float s1,
                                                  float b1, float s2, float b2, float res) {
    float v = acc * s1 + b1;
    v = apply_act(v, p.act);
    if (p.drop_rate > 0.0f)
        v = dropout_keep(p.drop_seed, p.drop_site, (uint32_t)(p.drop_row0 + row), (uint32_t)col, p.drop_rate) ? v / (1.0f - p.drop_rate) : 0.0f;
    v *= p.alpha;
    if (p.R) v += res;
    if (p.s2) v = v * s2 + b2;
    return v;
}

__device__ __forceinline__ float epilogue_residual(const estts_gemm_args& p, int row, int col) {
    const int rr = p.r_mod > 0 ? row % p.r_mod : row;
    return p.R[(long)rr * p.ldr + col];
}

__device__ __forceinline__ float epilogue_value(const estts_gemm_args& p, float acc, int row, int col, float s1,
                                                float b1, float s2, float b2) {
    return epilogue_value_r(p, acc, row, col, s1, b1, s2, b2, p.R ? epilogue_residual(p, row, col) : 0.0f);
}



// true when the activation panel (M rows) is the smaller operand: an L2 then holds all of it plus a strip of W columns
__device__ __forceinline__ bool gemm_y_first(const estts_gemm_args& p) {
    return (long)p.M * p.c_per_tap * 4 <= (long)p.N * p.K * (p.precision == 3 ? 4 : 2);
}

// One-shot kernel for launches of a few hundred small tiles (gemm_oneshot.hip).  Returns 1 when the shape is outside its
// envelope (the caller then takes another kernel), 0 after a launch, negative on a launch error.
// tile16: 16 x 16 output tiles instead of 32 x 32; waves: 4 or 8 per workgroup, 0 = by K.
int estts_launch_gemm_oneshot(const estts_gemm_args& a, int kind, int tile16, int waves, hipStream_t st);
// The same for 1-D convolutions with the tile's source rows staged once in LDS (eight waves); same return values.
int estts_launch_gemm_oneshot_conv(const estts_gemm_args& a, int kind, int tile16, hipStream_t st);
