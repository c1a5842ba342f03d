// Diagnostic harness (GPU box): times estts_gemm tile / split choices from C++ and, when built with
// -DESTTS_GEMM_STAMPS, prints where one workgroup's K loop spends its cycles.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 [-DESTTS_GEMM_STAMPS] tools/gemm_lab.hip -o /tmp/gemm_lab
#include "../expressive-speech-synthesis-research_amd/csrc/gemm.hip"
#include "../expressive-speech-synthesis-research_amd/csrc/rowops.hip"
#include <vector>
#include <cstdlib>
#include <cmath>

#define HC(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

static uint16_t f2bf(float f) { uint32_t u; memcpy(&u, &f, 4); u += 0x7fff + ((u >> 16) & 1); return (uint16_t)(u >> 16); }
static float bf2f(uint16_t h) { uint32_t u = (uint32_t)h << 16; float f; memcpy(&f, &u, 4); return f; }

int main(int argc, char** argv) {
    int M = argc > 1 ? atoi(argv[1]) : 600, N = argc > 2 ? atoi(argv[2]) : 256, K = argc > 3 ? atoi(argv[3]) : 1536;
    int taps = argc > 4 ? atoi(argv[4]) : 3, prec = argc > 5 ? atoi(argv[5]) : 3;
    int padw = argc > 6 ? atoi(argv[6]) : 0, pada = argc > 7 ? atoi(argv[7]) : 0;
    int Np = (N + 63) / 64 * 64, Kp = (K + 63) / 64 * 64, cin = K / taps, lda = cin + pada;
    std::vector<float> hA((size_t)M * lda), hW((size_t)N * K);
    srand(1);
    for (auto& v : hA) v = (rand() / (float)RAND_MAX - 0.5f) * 2;
    for (auto& v : hW) v = (rand() / (float)RAND_MAX - 0.5f) * 2 / sqrtf((float)K);
    std::vector<uint16_t> hi((size_t)Np * Kp, 0), lo((size_t)Np * Kp, 0);
    for (int n = 0; n < N; ++n) for (int k = 0; k < K; ++k) {   // fragment-major packing
        float w = hW[(size_t)n * K + k]; uint16_t h = f2bf(w);
        size_t o = ((size_t)(n / 16) * (Kp / 32) + k / 32) * 512 + ((n % 16) + 16 * ((k % 32) / 8)) * 8 + k % 8;
        hi[o] = h; lo[o] = f2bf(w - bf2f(h)); }
    float *dA, *dC, *ws; uint16_t *dhi, *dlo; int* cnt;
    HC(hipMalloc(&dA, hA.size() * 4)); HC(hipMalloc(&dC, (size_t)M * N * 4)); HC(hipMalloc(&ws, 64 << 20));
    HC(hipMalloc(&dhi, hi.size() * 2)); HC(hipMalloc(&dlo, lo.size() * 2)); HC(hipMalloc(&cnt, 4096 * 4));
    HC(hipMemset(cnt, 0, 4096 * 4));
    HC(hipMemcpy(dA, hA.data(), hA.size() * 4, hipMemcpyHostToDevice));
    HC(hipMemcpy(dhi, hi.data(), hi.size() * 2, hipMemcpyHostToDevice));
    HC(hipMemcpy(dlo, lo.data(), lo.size() * 2, hipMemcpyHostToDevice));
    estts_gemm_args a = {};
    a.A = dA; a.lda = lda; a.seg_len = M; a.taps = taps; a.c_per_tap = cin; a.tap_lpad = (taps - 1) / 2;
    a.W_hi = dhi; a.W_lo = dlo; a.ldw = Kp; a.M = M; a.N = N; a.K = K; a.alpha = 1.f; a.C = dC; a.ldc = N; a.precision = prec;
    a.ws = ws; a.ws_bytes = 64 << 20; a.counters = cnt; a.n_counters = 4096;
    hipStream_t st; HC(hipStreamCreate(&st));
    hipEvent_t e0, e1; HC(hipEventCreate(&e0)); HC(hipEventCreate(&e1));
    for (int tile = 1; tile <= 6; ++tile) for (int sk = 1; sk <= 8; sk *= 2) {
        a.tune = tile | (sk << 8);
        if (tile >= 4) a.tune = 4 | (sk << 8) | ((4 << (tile - 4)) << 16);   // tiles 4,5,6 = small kernel with 4/8/16 waves
        if (estts_gemm(&a, st) != 0) { printf("tile %d sk %d: %s\n", tile, sk, estts_last_error()); continue; }
        HC(hipStreamSynchronize(st));
        const int reps = 200;
        hipGraph_t g; hipGraphExec_t ge;
        HC(hipStreamBeginCapture(st, hipStreamCaptureModeGlobal));
        for (int r = 0; r < reps; ++r) estts_gemm(&a, st);
        HC(hipStreamEndCapture(st, &g)); HC(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
        HC(hipGraphLaunch(ge, st)); HC(hipStreamSynchronize(st));
        float best = 1e9;
        for (int round = 0; round < 5; ++round) {
            HC(hipEventRecord(e0, st)); HC(hipGraphLaunch(ge, st)); HC(hipEventRecord(e1, st)); HC(hipEventSynchronize(e1));
            float ms; HC(hipEventElapsedTime(&ms, e0, e1)); if (ms < best) best = ms;
        }
        printf("M=%d N=%d K=%d taps=%d prec=%d padw=%d pada=%d tile=%d splitk=%d : %7.2f us/launch  %6.1f TF/s", M, N, K, taps, prec, padw, pada, tile, sk,
               best * 1e3 / reps, 2.0 * M * N * K / (best * 1e-3 / reps) / 1e12);
#ifdef ESTTS_GEMM_STAMPS
        if (sk == 1) { float h[6]; HC(hipMemcpy(h, ws, 24, hipMemcpyDeviceToHost));
            printf("  | stages %g: per stage wait %.0f store %.0f barrier %.0f compute %.0f ; total %.0f cyc", h[5], h[0] / h[5], h[1] / h[5], h[2] / h[5], h[3] / h[5], h[4]); }
#endif
        printf("\n");
        HC(hipGraphExecDestroy(ge)); HC(hipGraphDestroy(g));
    }
    return 0;
}
