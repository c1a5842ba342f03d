// Duration extraction: the reductions over the decoder's cross-attention maps (see the alignment block of include/estts.h).
//
// The reference copies every (heads, frames, tokens) map to the host and scans it in NumPy.  What the sequential part of that
// procedure needs is small: one diagonal-distance score per head, one column-sum vector per head, and for every frame the
// position / value / multiplicity of the attention peak.  These kernels produce exactly that, so a batch of B = 32 maps
// (31 MB) leaves the device as ~0.3 MB.  All sums run in float64 like the reference's (its distance mask is float64).
#include "common.h"

namespace {

__device__ __forceinline__ double wave_sum_f64(double v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    return v;
}

// One workgroup per utterance: frames whose channels are all zero and tokens equal to zero are padding.
__global__ __launch_bounds__(256) void alignment_lengths_kernel(const float* mels, long ldm, int T, int C, const int* tokens, int N,
                                                                int* mel_len, int* phon_len) {
    __shared__ int zeros[2];
    const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    if (tid < 2) zeros[tid] = 0;
    __syncthreads();
    int zr = 0;
    for (int t = wave; t < T; t += 4) {
        const float* row = mels + ((long)b * T + t) * ldm;
        float s = 0.0f;
        for (int c = lane; c < C; c += 64) s += fabsf(row[c]);
        s = wave_sum(s);
        zr += (s == 0.0f);
    }
    if (lane == 0 && zr) atomicAdd(&zeros[0], zr);
    int zt = 0;
    for (int n = tid; n < N; n += 256) zt += (tokens[(long)b * N + n] == 0);
    if (zt) atomicAdd(&zeros[1], zt);
    __syncthreads();
    if (tid == 0) {
        mel_len[b] = T - zeros[0];
        phon_len[b] = N - zeros[1];
    }
}

// One workgroup per (head, utterance): score = sum_{t, n} |n / N' - t / T'| A[t + 1, n + 1] and colsum[n] = sum_t A[t + 1, n + 1]
// over the window of T' = mel_len - 2 frames and N' = phon_len - 2 tokens.  Lane = token, the four waves take frames t = w, w + 4, ...
__global__ __launch_bounds__(256) void alignment_stats_kernel(const float* attn, int H, int Lq, int Lk, const int* mel_len,
                                                              const int* phon_len, double* score, double* colsum) {
    __shared__ double part[4][64];
    __shared__ double spart[4];
    const int h = blockIdx.x, b = blockIdx.y, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int Tw = min(mel_len[b] - 2, Lq - 1), Nw = min(phon_len[b] - 2, Lk - 1);   // the host rejects lengths past the map
    const float* A = attn + ((long)b * H + h) * Lq * Lk;
    double* cs = colsum + ((long)b * H + h) * Lk;
    for (int n = tid; n < Lk; n += 256) cs[n] = 0.0;
    double sc = 0.0;
    if (Tw > 0 && Nw > 0) {
        for (int n0 = 0; n0 < Nw; n0 += 64) {
            const int n = n0 + lane;
            const bool in = n < Nw;
            const double fn = (double)n / (double)Nw;
            double col = 0.0;
            for (int t = wave; t < Tw; t += 4) {
                const double a = in ? (double)A[(long)(t + 1) * Lk + n + 1] : 0.0;
                col += a;
                sc += fabs(fn - (double)t / (double)Tw) * a;
            }
            part[wave][lane] = col;
            __syncthreads();
            if (wave == 0 && in) cs[n] = (part[0][lane] + part[1][lane]) + (part[2][lane] + part[3][lane]);
            __syncthreads();
        }
    }
    sc = wave_sum_f64(sc);
    if (lane == 0) spart[wave] = sc;
    __syncthreads();
    if (tid == 0) score[(long)b * H + h] = (spart[0] + spart[1]) + (spart[2] + spart[3]);
}

// One wave per frame of the window.  ref[t, n] = A[best head] (weighted == 0; best = first head with the smallest score) or
// sum_h A[h] / score[h] (weighted != 0).  Writes the first position of the row maximum, its value and how many entries equal it.
__global__ __launch_bounds__(256) void alignment_peaks_kernel(const float* attn, int H, int Lq, int Lk, const int* mel_len,
                                                              const int* phon_len, const double* score, int weighted, int* peak_idx,
                                                              double* peak_val, int* peak_ties, int* best_head) {
    const int b = blockIdx.y, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int t = blockIdx.x * 4 + wave;
    const int Tw = min(mel_len[b] - 2, Lq - 1), Nw = min(phon_len[b] - 2, Lk - 1);
    const double* sb = score + (long)b * H;
    int best = 0;
    for (int h = 1; h < H; ++h)
        if (sb[h] < sb[best]) best = h;
    if (blockIdx.x == 0 && tid == 0) best_head[b] = best;
    const long o = (long)b * Lq + t;
    if (t >= Lq) return;
    if (t >= Tw || Nw <= 0) {
        if (lane == 0) {
            peak_idx[o] = -1;
            peak_val[o] = 0.0;
            peak_ties[o] = 0;
        }
        return;
    }
    const float* A = attn + (long)b * H * Lq * Lk + (long)(t + 1) * Lk + 1;
    double vmax = -1.0;
    int imax = 0x7fffffff, cnt = 0;
    for (int n = lane; n < Nw; n += 64) {
        double v;
        if (weighted) {
            v = 0.0;
            for (int h = 0; h < H; ++h) v += (double)A[(long)h * Lq * Lk + n] / sb[h];
        } else {
            v = (double)A[(long)best * Lq * Lk + n];
        }
        if (v > vmax) {
            vmax = v;
            imax = n;
            cnt = 1;
        } else if (v == vmax) {
            ++cnt;
        }
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        const double ov = __shfl_xor(vmax, off);
        const int oi = __shfl_xor(imax, off);
        const int oc = __shfl_xor(cnt, off);
        if (ov > vmax) {
            vmax = ov;
            imax = oi;
            cnt = oc;
        } else if (ov == vmax) {
            imax = min(imax, oi);
            cnt += oc;
        }
    }
    if (lane == 0) {
        peak_idx[o] = imax;
        peak_val[o] = vmax;
        peak_ties[o] = cnt;
    }
}

}  // namespace

extern "C" int estts_alignment_lengths(const float* mels, int64_t ldm, int32_t T, int32_t C, const int32_t* tokens, int32_t N, int32_t B,
                                       int32_t* mel_len, int32_t* phon_len, void* stream) {
    ESTTS_CHECK_ARG(mels && tokens && mel_len && phon_len && B > 0 && T > 0 && C > 0 && N > 0 && ldm >= C,
                    "estts_alignment_lengths: bad arguments");
    ESTTS_LAUNCH(alignment_lengths_kernel, dim3(B), dim3(256), 0, (hipStream_t)stream, mels, (long)ldm, T, C, tokens, N, mel_len,
                       phon_len);
    ESTTS_CHECK_LAUNCH("estts_alignment_lengths");
    return 0;
}

extern "C" int estts_alignment_stats(const float* attn, int32_t B, int32_t H, int32_t Lq, int32_t Lk, const int32_t* mel_len,
                                     const int32_t* phon_len, double* score, double* colsum, void* stream) {
    ESTTS_CHECK_ARG(attn && mel_len && phon_len && score && colsum && B > 0 && H > 0 && Lq > 0 && Lk > 0,
                    "estts_alignment_stats: bad arguments");
    ESTTS_CHECK_ARG(B <= 65535, "estts_alignment_stats: B = %d exceeds 65535", B);
    ESTTS_LAUNCH(alignment_stats_kernel, dim3(H, B), dim3(256), 0, (hipStream_t)stream, attn, H, Lq, Lk, mel_len, phon_len, score,
                       colsum);
    ESTTS_CHECK_LAUNCH("estts_alignment_stats");
    return 0;
}

extern "C" int estts_alignment_peaks(const float* attn, int32_t B, int32_t H, int32_t Lq, int32_t Lk, const int32_t* mel_len,
                                     const int32_t* phon_len, const double* score, int32_t weighted, int32_t* peak_idx, double* peak_val,
                                     int32_t* peak_ties, int32_t* best_head, void* stream) {
    ESTTS_CHECK_ARG(attn && mel_len && phon_len && score && peak_idx && peak_val && peak_ties && best_head && B > 0 && H > 0 && Lq > 0 &&
                        Lk > 0,
                    "estts_alignment_peaks: bad arguments");
    ESTTS_CHECK_ARG(B <= 65535, "estts_alignment_peaks: B = %d exceeds 65535", B);
    ESTTS_LAUNCH(alignment_peaks_kernel, dim3((Lq + 3) / 4, B), dim3(256), 0, (hipStream_t)stream, attn, H, Lq, Lk, mel_len,
                       phon_len, score, weighted, peak_idx, peak_val, peak_ties, best_head);
    ESTTS_CHECK_LAUNCH("estts_alignment_peaks");
    return 0;
}
