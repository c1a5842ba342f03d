// Mel front-end on the device (see the audio block of include/estts.h): the short-time Fourier transform as a float32
// matrix product of overlapping frames with a windowed DFT matrix, magnitudes, the mel filterbank as a second product, and
// the reference's dB / log normalisers.
//
// The products run on v_mfma_f32_16x16x4_f32 (exact float32 multiply-add; lane l holds A[row l & 15][k l >> 4],
// B[k l >> 4][col l & 15], D[row 4 (l >> 4) + j][col l & 15]): the split-bf16 products of gemm.hip carry ~1e-5 of the
// frame's norm as error, which is audible in the quiet bins of a log spectrum (60-80 dB below the peak), float32 is not.
// A workgroup owns a 64 x 64 tile of C, its four waves 32 x 32 each; K advances in steps of 16 through LDS.
#include "common.h"

namespace {

constexpr int FT = 64;     // tile edge
constexpr int FK = 16;     // K per stage
constexpr int FLD = FK + 1;

__global__ __launch_bounds__(256) void gemm_f32_kernel(const float* A, long lda, const float* W, long ldw, float* C, long ldc, int M,
                                                       int N, int K) {
    __shared__ float As[2][FT * FLD];
    __shared__ float Ws[2][FT * FLD];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int i16 = lane & 15, g = lane >> 4;
    const int m0 = blockIdx.y * FT, n0 = blockIdx.x * FT;
    const int lr = tid >> 2, lc = (tid & 3) * 4;                 // this thread's float4 of a 64 x 16 stage
    const float* ap = A + (long)min(m0 + lr, M - 1) * lda + lc;  // rows / columns past the edge re-read the last one
    const float* wp = W + (long)min(n0 + lr, N - 1) * ldw + lc;  // (their results are never stored)

    f32x4 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

    float4 ar = *reinterpret_cast<const float4*>(ap);
    float4 wr = *reinterpret_cast<const float4*>(wp);
    const int nk = K / FK;
    for (int kt = 0; kt < nk; ++kt) {
        const int buf = kt & 1;
        float* as = &As[buf][lr * FLD + lc];
        float* ws = &Ws[buf][lr * FLD + lc];
        as[0] = ar.x; as[1] = ar.y; as[2] = ar.z; as[3] = ar.w;
        ws[0] = wr.x; ws[1] = wr.y; ws[2] = wr.z; ws[3] = wr.w;
        __syncthreads();
        if (kt + 1 < nk) {
            ar = *reinterpret_cast<const float4*>(ap + (long)(kt + 1) * FK);
            wr = *reinterpret_cast<const float4*>(wp + (long)(kt + 1) * FK);
        }
#pragma unroll
        for (int ks = 0; ks < FK / 4; ++ks) {
            float af[2], wf[2];
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                af[i] = As[buf][(wm * 32 + i * 16 + i16) * FLD + ks * 4 + g];
                wf[i] = Ws[buf][(wn * 32 + i * 16 + i16) * FLD + ks * 4 + g];
            }
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[i], wf[j], acc[i][j], 0, 0, 0);
        }
    }
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int col = n0 + wn * 32 + j * 16 + i16;
            if (col >= N) continue;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int row = m0 + wm * 32 + i * 16 + 4 * g + r;
                if (row < M) C[(long)row * ldc + col] = acc[i][j][r];
            }
        }
}

// mag[m, c] = sqrt(re^2 + im^2) with re = x[m, c], im = x[m, nb + c]
__global__ __launch_bounds__(256) void spec_magnitude_kernel(const float* x, long ldx, int nb, float* mag, long ldm, int M) {
    const long total = (long)M * nb;
    for (long e = (long)blockIdx.x * 256 + threadIdx.x; e < total; e += (long)gridDim.x * 256) {
        const long m = e / nb;
        const int c = (int)(e - m * nb);
        const float re = x[m * ldx + c], im = x[m * ldx + nb + c];
        mag[m * ldm + c] = sqrtf(re * re + im * im);
    }
}

__global__ __launch_bounds__(256) void mel_normalize_kernel(const float* x, long ldx, float* out, long ldo, int M, int C, int mode,
                                                            float ref_db, float min_db, float scale, float shift) {
    const long total = (long)M * C;
    for (long e = (long)blockIdx.x * 256 + threadIdx.x; e < total; e += (long)gridDim.x * 256) {
        const long m = e / C;
        const int c = (int)(e - m * C);
        const float v = fmaxf(x[m * ldx + c], 1e-5f);
        float r;
        if (mode == 1) {
            r = logf(v);
        } else {
            const float db = 20.0f * log10f(v) - ref_db;
            r = fminf(fmaxf((db - min_db) / -min_db, 0.0f), 1.0f) * scale + shift;
        }
        out[m * ldo + c] = r;
    }
}

int ew_blocks(long total) {
    const long b = (total + 255) / 256;
    return (int)(b < 1 ? 1 : (b > 4096 ? 4096 : b));
}

}  // namespace

extern "C" int estts_gemm_f32(const float* A, int64_t lda, const float* W, int64_t ldw, float* C, int64_t ldc, int32_t M, int32_t N,
                              int32_t K, void* stream) {
    ESTTS_CHECK_ARG(A && W && C && M > 0 && N > 0 && K > 0, "estts_gemm_f32: bad arguments");
    ESTTS_CHECK_ARG(K % 16 == 0 && (lda & 3) == 0 && (ldw & 3) == 0, "estts_gemm_f32: K must be a multiple of 16, lda / ldw of 4");
    ESTTS_CHECK_ARG((((uintptr_t)A | (uintptr_t)W) & 15) == 0, "estts_gemm_f32: A / W must be 16-byte aligned");
    ESTTS_CHECK_ARG(ldw >= K && ldc >= N, "estts_gemm_f32: ldw < K or ldc < N");
    dim3 grid((N + FT - 1) / FT, (M + FT - 1) / FT);
    ESTTS_CHECK_ARG(grid.y <= 65535, "estts_gemm_f32: too many rows");
    ESTTS_LAUNCH(gemm_f32_kernel, grid, dim3(256), 0, (hipStream_t)stream, A, (long)lda, W, (long)ldw, C, (long)ldc, M, N, K);
    ESTTS_CHECK_LAUNCH("estts_gemm_f32");
    return 0;
}

extern "C" int estts_spec_magnitude(const float* reim, int64_t ld, int32_t nb, float* mag, int64_t ldm, int32_t M, void* stream) {
    ESTTS_CHECK_ARG(reim && mag && M > 0 && nb > 0 && ld >= 2 * (int64_t)nb && ldm >= nb, "estts_spec_magnitude: bad arguments");
    ESTTS_LAUNCH(spec_magnitude_kernel, dim3(ew_blocks((long)M * nb)), dim3(256), 0, (hipStream_t)stream, reim, (long)ld, nb, mag,
                       (long)ldm, M);
    ESTTS_CHECK_LAUNCH("estts_spec_magnitude");
    return 0;
}

extern "C" int estts_mel_normalize(const float* x, int64_t ldx, float* out, int64_t ldo, int32_t M, int32_t C, int32_t mode,
                                   float ref_level_db, float min_level_db, float scale, float shift, void* stream) {
    ESTTS_CHECK_ARG(x && out && M > 0 && C > 0 && ldx >= C && ldo >= C, "estts_mel_normalize: bad arguments");
    ESTTS_CHECK_ARG(mode == 0 || mode == 1, "estts_mel_normalize: mode must be 0 (dB) or 1 (log)");
    ESTTS_CHECK_ARG(mode == 1 || min_level_db < 0.0f, "estts_mel_normalize: min_level_db must be negative");
    ESTTS_LAUNCH(mel_normalize_kernel, dim3(ew_blocks((long)M * C)), dim3(256), 0, (hipStream_t)stream, x, (long)ldx, out, (long)ldo,
                       M, C, mode, ref_level_db, min_level_db, scale, shift);
    ESTTS_CHECK_LAUNCH("estts_mel_normalize");
    return 0;
}
