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

#define ESTTS_CHECK_LAUNCH(name)                                                 \
    do {                                                                         \
        hipError_t e__ = hipGetLastError();                                      \
        if (e__ != hipSuccess) {                                                 \
            estts_set_error("%s: launch failed: %s", name, hipGetErrorString(e__)); \
            return -2;                                                           \
        }                                                                        \
    } while (0)

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    return v;
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
