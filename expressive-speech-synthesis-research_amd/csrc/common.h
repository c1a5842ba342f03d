// Shared device/host helpers for libestts (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdarg.h>

#include "../../include/estts.h"

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;
typedef __attribute__((ext_vector_type(4))) float f32x4;

#define ESTTS_WAVE 64

// thread-local error text returned by estts_last_error()
void estts_set_error(const char* fmt, ...);

#define ESTTS_CHECK_ARG(cond, ...)              \
    do {                                        \
        if (!(cond)) {                          \
            estts_set_error(__VA_ARGS__);       \
            return -1;                          \
        }                                       \
    } while (0)

// every kernel launch of the library goes through this macro, which counts it (estts_launch_count(): the benchmark's
// launches-per-step figure)
extern long long g_estts_launches;
#define ESTTS_LAUNCH(...)                                                       \
    do {                                                                         \
        __atomic_fetch_add(&g_estts_launches, 1LL, __ATOMIC_RELAXED);            \
        hipLaunchKernelGGL(__VA_ARGS__);                                         \
    } while (0)

#define ESTTS_CHECK_LAUNCH(name)                                                 \
    do {                                                                         \
        hipError_t e__ = hipGetLastError();                                      \
        if (e__ != hipSuccess) {                                                 \
            estts_set_error("%s: launch failed: %s", name, hipGetErrorString(e__)); \
            return -2;                                                           \
        }                                                                        \
    } while (0)

// Wave-wide reductions on the data-parallel-primitive path of the vector ALU (DPP operand modifiers: no LDS crossbar, no
// waits): quad swaps, half-row and row mirrors give every lane the result of its row of 16; row_bcast15 / row_bcast31 carry
// the rows' results up so that lane 63 holds the wave's, which v_readlane hands to every lane as a scalar.  The butterfly
// of ds_bpermute steps (__shfl_xor) costs a round trip through the LDS unit per step, ~0.3 us per reduction.
#define ESTTS_DPP(v, ctrl, row_mask) \
    __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(__builtin_bit_cast(int, ident), __builtin_bit_cast(int, v), ctrl, row_mask, 0xf, false))
__device__ __forceinline__ float wave_sum(float v) {
    const float ident = 0.0f;
    v += ESTTS_DPP(v, 0xB1, 0xf);    // quad_perm [1, 0, 3, 2]
    v += ESTTS_DPP(v, 0x4E, 0xf);    // quad_perm [2, 3, 0, 1]
    v += ESTTS_DPP(v, 0x141, 0xf);   // row_half_mirror
    v += ESTTS_DPP(v, 0x140, 0xf);   // row_mirror: every lane holds the sum of its row
    v += ESTTS_DPP(v, 0x142, 0xa);   // row_bcast15 into rows 1 and 3
    v += ESTTS_DPP(v, 0x143, 0xc);   // row_bcast31 into rows 2 and 3
    return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 63));
}

__device__ __forceinline__ float wave_max(float v) {
    const float ident = -INFINITY;
    v = fmaxf(v, ESTTS_DPP(v, 0xB1, 0xf));
    v = fmaxf(v, ESTTS_DPP(v, 0x4E, 0xf));
    v = fmaxf(v, ESTTS_DPP(v, 0x141, 0xf));
    v = fmaxf(v, ESTTS_DPP(v, 0x140, 0xf));
    v = fmaxf(v, ESTTS_DPP(v, 0x142, 0xa));
    v = fmaxf(v, ESTTS_DPP(v, 0x143, 0xc));
    return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 63));
}
#undef ESTTS_DPP

// The value held by lane ^ 16 / lane ^ 32, through the row-swap instructions of gfx950 (v_permlane16_swap exchanges the odd
// rows of its first operand with the even rows of its second, v_permlane32_swap the upper half with the lower half; given the
// same register twice, one of the two results holds the partner's value, depending on which side the lane is on).  Vector
// ALU work, where __shfl_xor takes a round trip through the LDS unit (ds_bpermute).
__device__ __forceinline__ float lane_xor16(float v) {
    const int b = __builtin_bit_cast(int, v);
    const auto r = __builtin_amdgcn_permlane16_swap(b, b, false, false);
    return __builtin_bit_cast(float, (threadIdx.x & 16) ? r[0] : r[1]);
}
__device__ __forceinline__ float lane_xor32(float v) {
    const int b = __builtin_bit_cast(int, v);
    const auto r = __builtin_amdgcn_permlane32_swap(b, b, false, false);
    return __builtin_bit_cast(float, (threadIdx.x & 32) ? r[0] : r[1]);
}

__device__ __forceinline__ float apply_act(float v, int act) {
    if (act == ESTTS_ACT_RELU) return fmaxf(v, 0.0f);
    if (act == ESTTS_ACT_TANH) return tanhf(v);
    return v;
}

// split x into bf16 hi and bf16 lo with hi + lo ~= x to ~16 mantissa bits
__device__ __forceinline__ void split_bf16(float x, __bf16& hi, __bf16& lo) {
    hi = (__bf16)x;
    lo = (__bf16)(x - (float)hi);
}

// the same for four values at once, with the conversions issued in pairs (one v_cvt_pk_bf16_f32 per two values)
typedef __attribute__((ext_vector_type(2))) float f32x2_t;
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2_t;
__device__ __forceinline__ void split_bf16x4(float4 x, bf16x4& hi, bf16x4& lo) {
    const f32x2_t a = {x.x, x.y}, b = {x.z, x.w};
    const bf16x2_t ha = __builtin_convertvector(a, bf16x2_t), hb = __builtin_convertvector(b, bf16x2_t);
    const f32x2_t ra = a - __builtin_convertvector(ha, f32x2_t), rb = b - __builtin_convertvector(hb, f32x2_t);
    const bf16x2_t la = __builtin_convertvector(ra, bf16x2_t), lb = __builtin_convertvector(rb, bf16x2_t);
    hi = (bf16x4){ha[0], ha[1], hb[0], hb[1]};
    lo = (bf16x4){la[0], la[1], lb[0], lb[1]};
}

__device__ __forceinline__ uint32_t mix32(uint32_t x) {
    x ^= x >> 16; x *= 0x7FEB352Du; x ^= x >> 15; x *= 0x846CA68Bu; x ^= x >> 16;
    return x;
}

// keep decision of the counter-based dropout generator for element (row, col) of call site `site`
__device__ __forceinline__ bool dropout_keep(uint32_t seed, uint32_t site, uint32_t row, uint32_t col, float rate) {
    uint32_t x = mix32(seed ^ (site * 0x9E3779B9u));
    x = mix32(x ^ (row * 0x85EBCA6Bu + 0x27D4EB2Fu));
    x = mix32(x ^ (col * 0xC2B2AE35u + 0x165667B1u));
    return (x >> 8) * (1.0f / 16777216.0f) >= rate;
}

// Tile numbering that follows the L2s.  Workgroups whose linear dispatch ids agree mod 8 are observed to share an XCD
// and therefore a 4 MiB L2 (placement is a speed matter only: nothing here relies on it for correctness).  Every such
// group is given one contiguous run of tile numbers (bijective for any grid size), the split-K slices of a tile are
// adjacent in it, and a run walks the tile grid along y first (a strip of columns: all of the row operand, 1/8 of the
// column operand per L2) or along x first (a strip of rows), whichever operand the caller says is the smaller one.
// panel > 0 (grids with gz = 1): the run of an L2 walks panels of `panel` columns, row by row inside a panel, so that it
// covers a block of rows x columns instead of a whole strip: with both operands of similar size every L2 then pulls a
// fraction of each instead of all of one (the LayerNorm-prologue convolution of a 600-frame decoder block: 13.5 -> about
// 10 MB of L2 fills per launch; with eight L2s no order gets below ~9.5 MB against 4.6 MB read and written once).
struct TileCoord { int tx, ty, tz; };
__device__ __forceinline__ TileCoord xcd_tile_coord(bool y_first, bool remap = true, int panel = 0) {
    const int gx = gridDim.x, gy = gridDim.y, gz = gridDim.z;
    const int nwg = gx * gy * gz;
    int id = blockIdx.x + gx * (blockIdx.y + gy * blockIdx.z);
    if (remap && nwg >= 16) {
        const int q = nwg >> 3, r = nwg & 7, xcd = id & 7;
        id = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (id >> 3);
    }
    TileCoord t;
    t.tz = id % gz;
    const int tile = id / gz;
    if (panel > 0 && gz == 1) {
        const int per = panel * gy;
        const int pn = tile / per, rem = tile - pn * per;
        const int w = min(panel, gx - pn * panel);       // (the last panel may be narrower)
        t.ty = rem / w;
        t.tx = pn * panel + (rem - t.ty * w);
        return t;
    }
    if (y_first) { t.ty = tile % gy; t.tx = tile / gy; }
    else         { t.tx = tile % gx; t.ty = tile / gx; }
    return t;
}

// derivative of the activation expressed through its OUTPUT r
__device__ __forceinline__ float act_grad_from_output(float r, int act) {
    if (act == ESTTS_ACT_RELU) return r > 0.0f ? 1.0f : 0.0f;
    if (act == ESTTS_ACT_TANH) return 1.0f - r * r;
    return 1.0f;
}

__device__ __forceinline__ void atomic_add_f32(float* p, float v) {
    __hip_atomic_fetch_add(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
