/*
 * estts.h -- C ABI of libestts.so, the MI355X (gfx950) kernel library behind the
 * expressive-TTS acoustic + vocoder hot path.
 *
 * The reference (sankar-mukherjee/Expressive-Speech-Synthesis-Research) has no
 * FFI / plugin interface: its hot path is reached through Python objects that
 * call TensorFlow-Keras / PyTorch layers.  Each entry point below therefore
 * replaces one (or one fused group) of those framework-layer calls; the comment
 * on every declaration names the reference call sites it stands in for
 * (paths relative to the reference root).
 *
 * Conventions
 *   - every pointer is a DEVICE pointer owned by the caller (allocated by
 *     PyTorch-ROCm or hipMalloc); nothing here allocates, frees or synchronises,
 *     so every call may be captured into a hipGraph
 *   - `stream` is a hipStream_t passed as void*; work is asynchronous on it
 *   - return value 0 = launched; negative = argument / launch error, text via
 *     estts_last_error() (thread-local)
 *   - activations are float32 row-major [rows, ld]; rows = batch * seq_len with
 *     row m = b * seq_len + t; weights are bf16 (hi [+ lo] split), laid out
 *     [N_padded][K_padded] with K contiguous
 */
#ifndef ESTTS_H
#define ESTTS_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define ESTTS_ABI_VERSION 5

/* activation codes for estts_gemm / estts_rownorm */
#define ESTTS_ACT_NONE 0
#define ESTTS_ACT_RELU 1
#define ESTTS_ACT_TANH 2

int estts_abi_version(void);
const char* estts_last_error(void);

/* ------------------------------------------------------------------------------------------
 * Fused dense / 1-D convolution on the matrix cores.
 *
 *   acc[m, n] = sum_{tap, c} A[row(m) + tap - tap_lpad, c] * W[n, tap * c_per_tap + c]
 *               (rows outside [0, seg_len) of m's own segment contribute zero)
 *   v = acc * s1[n] + b1[n];  v = act(v);  v = v * alpha;
 *   if (R)  v += R[(r_mod ? m % r_mod : m), n];
 *   if (s2) v = v * s2[n] + b2[n];
 *   C[m, n] = v
 *
 * Replaces: tf.keras.layers.Dense / Conv1D (+ folded inference BatchNormalization, activation,
 * residual add, trailing BatchNormalization) at TransformerTTS/model/layers.py:19-47 (CNNResNorm),
 * :58-67 (FFNResNorm d1/d2), :112-116,129-131,145 (MultiHeadAttention wq/wk/wv/dense),
 * :387-401 (DecoderPrenet), :636 (DurationPredictor.linear),
 * pretrained_models/for_text_encoder/models.py:350 (out);
 * nn.Linear / nn.Conv1d at WaveRNN/models/fatchord_version.py:13-25,32-45,109-114;
 * with the 2-D geometry fields: Conv2D(3x3, stride 2, 'same') + BatchNormalization + ReLU of the GST
 * reference encoder, TransformerTTS/model/layers.py:538-559 (layers with at least 4 input channels).
 *
 * precision: 1 = bf16 operands (one MFMA per product); 3 = split-bf16 (hi*hi + lo*hi + hi*lo,
 * fp32 accumulate; needs W_lo).  K, c_per_tap and lda must be multiples of 4; W rows are padded
 * with zeros to ldw = K rounded up to 64, W has ceil(N/64)*64 rows, and its storage order is
 * fragment-major: 1 KiB blocks indexed (n / 16, k / 32), each holding for lane l = 0..63 the eight
 * values W[16 (n/16) + (l & 15)][32 (k/32) + 8 (l >> 4) .. + 7].
 *
 * Split-K: when ws / counters are given and the tile count is small, several workgroups share one
 * output tile through float32 slabs in ws (ws_bytes >= splitk * tiles * tile_elems * 4) and one
 * int32 ticket per tile in counters (n_counters of them, zero before the first launch; the kernel
 * leaves them zero).  Calls that may overlap on the GPU must not share ws / counters.
 * tune: 0 = choose tile and split automatically; else (tile id 1..4) | (splitk << 8) | (waves << 16,
 * tile 4 only: 4, 8 or 16 waves per workgroup, 0 = by K).
 * ------------------------------------------------------------------------------------------ */
typedef struct estts_gemm_args {
    const float* A;
    int32_t lda;
    int32_t seg_len;     /* rows per sequence (conv boundary); 0 = one segment of M rows */
    int32_t taps;        /* 1 = dense */
    int32_t c_per_tap;   /* input channels per tap; K = taps * c_per_tap */
    int32_t tap_lpad;    /* left padding in rows: (k-1)/2 for 'same', k-1 for 'causal' */
    const uint16_t* W_hi;
    const uint16_t* W_lo; /* may be NULL when precision == 1 */
    int32_t ldw;
    int32_t M, N, K;
    const float* s1;     /* NULL = 1 */
    const float* b1;     /* NULL = 0 */
    int32_t act;
    float alpha;
    const float* R;      /* NULL = none */
    int32_t ldr;
    int32_t r_mod;
    const float* s2;     /* NULL = none (then b2 ignored) */
    const float* b2;
    float* C;
    int32_t ldc;
    int32_t precision;
    float* ws;           /* split-K slabs, may be NULL (then no split) */
    int64_t ws_bytes;
    int32_t* counters;   /* split-K tickets, may be NULL */
    int32_t n_counters;
    int32_t tune;
    /* 2-D convolution geometry (NHWC, lda floats per pixel); conv_kw = 0 selects the dense / 1-D form above.
     * Output row m = (image, ho, wo) on a conv_howo / conv_wo grid, tap = (kh, kw) with kw fastest,
     * source pixel (ho * stride + kh - conv_pt, wo * stride + kw - conv_pl) of a conv_h x conv_w image. */
    int32_t conv_h, conv_w, conv_wo, conv_howo, conv_kw, conv_stride, conv_pt, conv_pl;
} estts_gemm_args;

int estts_gemm(const estts_gemm_args* args, void* stream);

/* ------------------------------------------------------------------------------------------
 * Multi-head scaled-dot-product attention over already-projected q / k / v.
 *
 *   S = q k^T / sqrt(depth) + mask * (-1e9);  P = softmax(S);  ctx = P v
 *   mask[b, i, j] = max(key_mask[b, j], causal ? (j > i) : 0)
 *
 * q rows: [B*Lq, ldq] (head h at columns h*depth .. ), k / v rows: [B*Lk, ldk / ldv];
 * ctx is written to [B*Lq, ldo] at column h*depth; attn (optional) is [B, H, Lq, Lk] float32.
 * depth must be 32, 64 or 128.
 *
 * Replaces: scaled_dot_product_attention + split_heads / merge at
 * TransformerTTS/model/transformer_utils.py:24-58 and model/layers.py:118-143; the masks of
 * transformer_utils.py:61-74 enter as key_mask (float 0/1 per key) and the causal flag.
 * ------------------------------------------------------------------------------------------ */
typedef struct estts_attention_args {
    const float* q; int32_t ldq;
    const float* k; int32_t ldk;
    const float* v; int32_t ldv;
    float* ctx; int32_t ldo;
    float* attn;             /* NULL = do not materialise the attention weights */
    const float* key_mask;   /* [B, Lk] 1.0 = masked; NULL = none */
    int32_t B, H, Lq, Lk, depth;
    int32_t causal;
    int32_t precision;       /* 1 or 3, as in estts_gemm */
    int32_t kv_shared;       /* 1 = k / v hold Lk rows used by every batch entry (style tokens) */
} estts_attention_args;

int estts_attention(const estts_attention_args* args, void* stream);

/* ------------------------------------------------------------------------------------------
 * Row-wise normalisation combos:   out = LNb( act( LNa(y) ) + x )
 * each stage optional; LayerNorm over the last axis (biased variance, eps inside the sqrt).
 * Optionally also emits dot[m] = relu(out[m,:] . dot_w + dot_b) * dot_scale * (dot_mask ? 1 - dot_mask[m] : 1).
 *
 * Replaces: LayerNormalization chains at TransformerTTS/model/layers.py:65-71 (FFNResNorm),
 * :163-168 (SelfAttentionResNorm), :270-274 (CrossAttentionResnorm), :35-47 with
 * normalization='layer' (DurationPredictor), and the Dense(1, relu) + padding zeroing at
 * layers.py:636-642 / pretrained_models/for_text_encoder/models.py:439-440.
 * ------------------------------------------------------------------------------------------ */
typedef struct estts_rownorm_args {
    const float* y; int32_t ldy;
    const float* x; int32_t ldx;        /* residual, NULL = none */
    const float* ga; const float* ba;   /* LNa gamma/beta, NULL = skip LNa */
    const float* gb; const float* bb;   /* LNb gamma/beta, NULL = skip LNb */
    int32_t act;
    float eps;
    float* out; int32_t ldo;
    int32_t M, D;                       /* D <= 1024, multiple of 4 */
    const float* dot_w;                 /* NULL = no dot output */
    float dot_b; float dot_scale;
    const float* dot_mask;              /* [M] 1.0 = padded row */
    float* dot_out;                     /* [M] */
} estts_rownorm_args;

int estts_rownorm(const estts_rownorm_args* args, void* stream);

/* out[b, n, :] = emb[tokens[b, n], :] * scale + pos_scalar[0] * pe[n * pe_stride, :];
 * also writes key_mask[b, n] = (tokens[b, n] == 0) when key_mask != NULL.
 * Replaces: Embedding + the positional add at TransformerTTS/model/layers.py:242-246 and
 * create_encoder_padding_mask (model/transformer_utils.py:61-63). */
int estts_embed_posenc(const int32_t* tokens, const float* emb, const float* pe, const float* pos_scalar,
                       float* out, int32_t ldo, float* key_mask, int32_t B, int32_t N, int32_t D,
                       float scale, int32_t pe_stride, void* stream);

/* Length regulator, step 1: reps = max(round_half_even(dur), 0) per token, inclusive prefix sums
 * per batch row into cum[b, 0..N-1] and the row total into totals[b].  One workgroup per batch row.
 * Replaces: tf.math.round + the row-length bookkeeping of Expand.call, model/layers.py:667-677. */
int estts_durations_cumsum(const float* dur, int32_t* cum, int32_t* totals, int32_t B, int32_t N, void* stream);

/* Length regulator, step 2: out[b, t, :] = x[b, n(t), :] where n(t) is the token whose span covers
 * frame t (binary search in cum); frames t >= totals[b] are zero and get frame_mask[b, t] = 1.
 * Replaces: the tile / mask / ragged-to-dense of Expand.call (model/layers.py:679-683) and
 * create_mel_padding_mask on its output (pretrained_models/for_text_encoder/models.py:445). */
int estts_expand(const float* x, int32_t ldx, const int32_t* cum, float* out, int32_t ldo, float* frame_mask,
                 int32_t B, int32_t N, int32_t T, int32_t D, void* stream);

/* First layer of the GST reference encoder: 3 x 3 convolution with ONE input channel over the mel "image"
 * x [B, H = frames, W = mel bins], stride `stride`, TensorFlow 'SAME' padding (pad_t / pad_l on the low side),
 * followed by the folded inference BatchNormalization and ReLU:
 *   out[b, ho, wo, co] = relu(scale[co] * sum_{kh,kw} x[b, ho s + kh - pad_t, wo s + kw - pad_l] w[kh*3+kw, co] + bias[co])
 * out is NHWC [B, Ho, Wo, Cout].  HBM-bound (reads the mel once, writes Cout floats per output pixel).
 * Replaces: Conv2D + BatchNormalization + ReLU, first iteration of ReferenceEncoderGST.call_convs,
 * TransformerTTS/model/layers.py:554-559 after the expand_dims of :563. */
int estts_conv2d_c1(const float* x, const float* w, const float* scale, const float* bias, float* out, int32_t B,
                    int32_t H, int32_t W, int32_t Ho, int32_t Wo, int32_t Cout, int32_t stride, int32_t pad_t,
                    int32_t pad_l, void* stream);

/* mask[m] = 1.0 if every element of row m of x [M, D] is zero, else 0.0.
 * Replaces: create_mel_padding_mask, TransformerTTS/model/transformer_utils.py:66-69. */
int estts_row_zero_mask(const float* x, int32_t ldx, float* mask, int32_t M, int32_t D, void* stream);

/* Masked mean absolute error, forward value:  out[0] = sum_rows( any(target_row != 0) * mean_c |target - pred| ) / M
 * (Keras MeanAbsoluteError with sample_weight under SUM_OVER_BATCH_SIZE: padded rows count in the denominator).
 * out[0] must be zero on entry (accumulated with one float atomic per workgroup).
 * Replaces: masked_mean_absolute_error, TransformerTTS/utils/losses.py:41-47. */
int estts_masked_mae(const float* target, int32_t ldt, const float* pred, int32_t ldp, float* out, int32_t M,
                     int32_t C, void* stream);

/* ------------------------------------------------------------------------------------------
 * WaveRNN generation, per-sample step kernels.  One "sequence" is one fold of one utterance
 * (fold_with_overlap, WaveRNN/models/fatchord_version.py:272-319); all sequences advance together.
 * The step index lives in device memory (step[0]) so that a run of steps can be captured once into
 * a hipGraph and replayed; estts_wavernn_sample advances it.
 * ------------------------------------------------------------------------------------------ */

/* Builds the inputs of one step for every sequence, without ever materialising the upsampled
 * conditioning tensors: for sequence b at global sample s = seq_start[b] + step[0] of utterance
 * u = seq_utt[b]:
 *   xin[b, 0]            = x_prev[b]
 *   xin[b, 1 .. feat]    = sum_j fir[s % hop, j] * mel[u, s / hop + pad - (fir_kmin + j), :]
 *   xin[b, 1+feat ..+d]  = aux[u, s / hop, 0:d];  a2/a3/a4[b, 0:d] = aux[u, s / hop, d:2d / 2d:3d / 3d:4d]
 * (all zero when s >= utt_samples[u]: the zero padding fold_with_overlap appends).
 * mel: [sum frames + 2 pad per utterance][feat] padded frames, utterance u starting at row mel_off[u];
 * aux: MelResNet output at frame rate [sum frames][4 d], utterance u starting at row aux_off[u];
 * fir: composite response of the three repeat + box-filter stages, [hop][ntaps].
 * Replaces: UpsampleNetwork.forward's Stretch2d / Conv2d chain and the aux repeat
 * (fatchord_version.py:79-86), fold_with_overlap (:272-319) and the per-step slicing and
 * concatenations of generate (:182-199). */
typedef struct estts_wavernn_prep_args {
    const float* mel; const int32_t* mel_off;
    const float* aux; const int32_t* aux_off;
    const float* fir; int32_t ntaps; int32_t fir_kmin;
    const int32_t* seq_utt; const int32_t* seq_start; const int32_t* utt_samples;
    const float* x_prev; const int32_t* step;
    float* xin; int32_t ld_xin;
    float* a2; int32_t ld_a2;
    float* a3; int32_t ld_a3;
    float* a4; int32_t ld_a4;
    int32_t n_seq, feat, d, hop, pad;
} estts_wavernn_prep_args;

int estts_wavernn_prep(const estts_wavernn_prep_args* args, void* stream);

/* torch-order GRU gates (r, z, n) on precomputed projections gi = W_ih x + b_ih, gh = W_hh h + b_hh:
 *   r = sigmoid(gi_r + gh_r); z = sigmoid(gi_z + gh_z); n = tanh(gi_n + r * gh_n); h' = (1 - z) n + z h
 * h is updated in place; out[m, :] = (x_res ? x_res[m, :] : 0) + h'[m, :].
 * Replaces: nn.GRUCell at fatchord_version.py:189-195 (and nn.GRU, :135-142, one time step at a time). */
int estts_gru_gates(const float* gi, int32_t ldgi, const float* gh, int32_t ldgh, float* h, int32_t ldh,
                    const float* x_res, int32_t ldx, float* out, int32_t ldo, int32_t M, int32_t H, void* stream);

/* Draws the next sample of every sequence from its logits and advances step[0].
 * mode 0 (RAW): softmax, inverse CDF at u (class = number of cdf entries <= u), value 2 c / (n - 1) - 1.
 * mode 1 (MOL): mixture of logistics with nr_mix = n_classes / 3 components, Gumbel-max component pick from
 *               u_mix, logistic inverse CDF at u, clamp to [-1, 1].
 * Uniform numbers: u[step * n_seq + b] (and u_mix[(step * n_seq + b) * nr_mix + k]) when given, else a
 * counter-based generator keyed by (seed, step, b).  wave[b * ld_wave + step] receives the value;
 * x_prev[b] receives forced[step * n_seq + b] when forced != NULL, else the value.
 * `done` is one zero-initialised int32 used to find the last workgroup, which advances step[0].
 * Replaces: F.softmax + Categorical.sample (fatchord_version.py:210-216) and
 * sample_from_discretized_mix_logistic (utility/distribution.py:87-123). */
typedef struct estts_wavernn_sample_args {
    const float* logits; int32_t ld_logits; int32_t n_classes; int32_t mode;
    const float* u; const float* u_mix; const float* forced;
    uint32_t seed;
    float* wave; int32_t ld_wave;
    float* x_prev;
    int32_t* step; int32_t* done;
    int32_t n_seq;
} estts_wavernn_sample_args;

int estts_wavernn_sample(const estts_wavernn_sample_args* args, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* ESTTS_H */
