// First-touch latency of a cold kernel: one wave stamps the 100 MHz clock around dependent loads (measurement tool, not part
// of the library).
#include <hip/hip_runtime.h>
#include <stdint.h>

__global__ void probe_kernel(const float* a, long far_elems, unsigned long long* out, float* sink) {
    const int lane = threadIdx.x;
    unsigned long long t0 = wall_clock64();
    float v = a[lane];
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    unsigned long long t1 = wall_clock64();
    float w = a[far_elems + lane + (v == 12345.f ? 1 : 0)];      // another page of the same buffer
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    unsigned long long t2 = wall_clock64();
    float u = a[lane + 64 + (w == 12345.f ? 1 : 0)];             // the line next to the first one
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    unsigned long long t3 = wall_clock64();
    float x = a[lane + (u == 12345.f ? 1 : 0)];                  // the first line again
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    unsigned long long t4 = wall_clock64();
    if (lane == 0) { out[0] = t1 - t0; out[1] = t2 - t1; out[2] = t3 - t2; out[3] = t4 - t3; }
    if (v + w + u + x == 12345.f) sink[0] = v;
}

__global__ void sweep_kernel(const float4* a, long n, float* sink) {
    float s = 0.f;
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) { float4 v = a[i]; s += v.x + v.y + v.z + v.w; }
    if (s == 12345.f) sink[0] = s;
}

extern "C" int probe(const float* a, long far_elems, unsigned long long* out, float* sink, void* stream) {
    hipLaunchKernelGGL(probe_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, a, far_elems, out, sink);
    return (int)hipGetLastError();
}
extern "C" int sweep(const float* a, long n4, float* sink, void* stream) {
    hipLaunchKernelGGL(sweep_kernel, dim3(2048), dim3(256), 0, (hipStream_t)stream, (const float4*)a, n4, sink);
    return (int)hipGetLastError();
}
