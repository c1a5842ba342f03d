// Diagnostic (GPU box): which half of the v_permlane16_swap / v_permlane32_swap result pair holds the value of lane ^ 16 / lane ^ 32.
#include <hip/hip_runtime.h>
#include <stdio.h>
__global__ void k(int* o) {
    const int v = threadIdx.x;
    const auto a = __builtin_amdgcn_permlane16_swap(v, v, false, false);
    const auto b = __builtin_amdgcn_permlane32_swap(v, v, false, false);
    o[threadIdx.x] = a[0]; o[64 + threadIdx.x] = a[1]; o[128 + threadIdx.x] = b[0]; o[192 + threadIdx.x] = b[1];
}
int main() {
    int* d; int h[256];
    hipMalloc(&d, sizeof(h));
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d);
    hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    const char* names[4] = {"p16[0]", "p16[1]", "p32[0]", "p32[1]"};
    for (int j = 0; j < 4; ++j) { printf("%s:", names[j]); for (int i = 0; i < 64; i += 8) printf(" %d", h[j * 64 + i]); printf("\n"); }
    return 0;
}
