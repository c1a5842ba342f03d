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

#define ESTTS_ABI_VERSION 36

/* activation codes for estts_gemm / estts_rownorm */
#define ESTTS_ACT_NONE 0
#define ESTTS_ACT_RELU 1
#define ESTTS_ACT_TANH 2

int estts_abi_version(void);
const char* estts_last_error(void);
/* Number of kernels this library has launched in the calling process so far (every entry point counts its launches).
 * No counterpart in the reference: it is what a benchmark divides by its step count to report launches per step. */
long long estts_launch_count(void);

/* ------------------------------------------------------------------------------------------
 * Fused dense / 1-D convolution on the matrix cores.
 *
 *   acc[m, n] = sum_{tap, c} A[row(m) + tap - tap_lpad, c] * W[n, tap * c_per_tap + c]
 *               (rows outside [0, seg_len) of m's own segment contribute zero)
 *   v = acc * s1[n] + b1[n];  v = act(v);
 *   if (drop_rate > 0)  v = keep(drop_seed, drop_site, drop_row0 + m, n) ? v / (1 - drop_rate) : 0
 *                       (estts_dropout's counter-based generator: DecoderPrenet's always-on dropout, layers.py:394-402)
 *   v = v * alpha;
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
 * tune: 0 = choose tile and split automatically; else (tile id) | (splitk << 8) | (waves << 16).  Tile ids: 1..3 LDS-staged
 * 64 x 64 / 32 x 64 / 32 x 32 with split-K; 4 register-ring 32 x 32 (waves: 4, 8 or 16 per workgroup, 0 = by K; split-K);
 * 5..8 LDS-staged 128 x 128 / 64 x 64 / 128 x 64 / 64 x 128 for thousands of rows (no split-K); 9 / 10 one-shot 16 x 16 /
 * 32 x 32 (every load issued up front: dense layers and 1-D convolutions with K <= 192 x waves, K and c_per_tap multiples
 * of 8; waves 4 or 8); 11 / 12 the same for convolutions with the tile's source rows staged once in LDS; 13 the 64 x 128
 * tile with W read from memory straight into the matrix instruction's registers and only A staged (what tune = 0 picks
 * for thousands of rows; results identical to id 8; no batch, no split-K); 14 the same kernel with eight waves and 64 x 256
 * tiles (what ln_epilogue runs on; tune = 0 picks it from 512 columns up); 15 the 64 x 128 tile with eight waves, two groups
 * of four on the two halves of K (K a multiple of 64; for launches of at most one tile per CU).  Launches of ids 13 / 14 with
 * more tiles than CUs send the rows of a last round that would be at most half full out as 32-row tiles (bit 26 of tune:
 * equal tiles only).  An explicit id outside its kernel's envelope is an error; tune = 0 never fails on shape.
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
    float drop_rate;     /* 0 = no dropout */
    uint32_t drop_seed, drop_site;
    int32_t drop_row0;   /* generator row of output row 0 (an incremental decoder computes row i alone) */
    /* batch > 1: that many independent products of one shape in a single launch (e.g. the two recurrent products of a
     * WaveRNN step); product z uses A + z * batch_stride_a, W_hi / W_lo + z * batch_stride_w (uint16 elements), C and R +
     * z * batch_stride_c, and s1 / b1 / s2 / b2 + z * batch_stride_v (floats).  LDS-staged tiles only (tune 0 or 5..8). */
    int32_t batch;
    int64_t batch_stride_a, batch_stride_w, batch_stride_c, batch_stride_v;
    /* LayerNorm prologue (ln_y != NULL): the A operand is not read from A but computed per source row, over c_per_tap
     * columns, as  a = LNb( act_ln( LNa(ln_y[row]) ) + ln_x[row] )  -- LNa = (ln_ga, ln_ba) or none (then act_ln must be
     * none: a = LNb(ln_y + ln_x), CrossAttentionResnorm, layers.py:270-274), LNb = (ln_gb, ln_bb) or none (at least one of
     * the two), ln_x or none, act_ln one of ESTTS_ACT_NONE / _RELU -- i.e. estts_rownorm fused into the layer that consumes
     * it (SelfAttentionResNorm's LN(LN(y) + x) in front of CNNResNorm's first Conv1D, TransformerTTS/model/layers.py:163-168
     * and :35-40; FFNResNorm's LN(relu(LN(y)) + x) in front of the next block's query / key / value projection, :65-71).
     * Every row is also written once to ln_out (float32, ln_ldo floats per row), which must not alias ln_y / ln_x.
     * Dense layers and 1-D convolutions (up to 8 taps) over few rows with c_per_tap <= 512 can take it (the kernel that
     * stages a tile's source rows in LDS; every column tile recomputes the norms of its rows, so it pays where the rows are
     * few -- the one-row products of incremental decoding -- or the column tiles are): estts_gemm fails with an error
     * otherwise and the caller runs estts_rownorm itself. */
    const float* ln_y; const float* ln_x;
    const float* ln_ga; const float* ln_ba; const float* ln_gb; const float* ln_bb;
    float* ln_out;
    int32_t ln_ldy, ln_ldx, ln_ldo, ln_act;
    float ln_eps;
    /* Column moments of the output (stat_slots != NULL), for a BatchNormalization in training mode that follows the layer
     * (CNNResNorm, TransformerTTS/model/layers.py:35-47): sum over the M rows of (C[m, c] - k[c]) and of its square, k =
     * stat_shift or 0, added with float atomics to 16 copies laid out as estts_colreduce's deferred sums
     * ([copy][sum 0 / 1][ceil(N / 256) * 256] floats, zero before the launch) and consumed by estts_bn_finalize_slots.
     * The large-M tiles (5..8) sum inside their epilogue; any other kernel is followed by a reduction pass over C
     * (same result layout).  Not with batch > 1 or a LayerNorm prologue. */
    float* stat_slots; const float* stat_shift;
    /* LayerNorm epilogue (ln_epilogue != 0; ln_y and ln_out NULL): C = LNb( act_ln( LNa(acc * s1 + b1) ) + ln_x ) with the
     * ln_* fields above read as in the prologue form -- estts_rownorm fused into the layer that PRODUCES its input (the
     * output projections of SelfAttentionResNorm / CrossAttentionResnorm and the folded FFNResNorm product at batch size,
     * TransformerTTS/model/layers.py:65-71, 163-168, 270-274), which saves writing and re-reading the un-normalised rows.
     * Needs N = 256 (a workgroup of the 64 x 256 tile owns whole rows), one product, and none of R / s2 / act / alpha /
     * dropout / stat_slots; it pays from a few hundred row tiles up (the caller decides: estts_gemm does what it is told or
     * fails with an error).  C may be ln_x, and for a dense layer A, row for row. */
    int32_t ln_epilogue;
} estts_gemm_args;

int estts_gemm(const estts_gemm_args* args, void* stream);

/* estts_gemm for callers that cannot allocate device memory (code under HIP graph capture, tight serving loops, harnesses
 * that forbid hipMalloc).  No counterpart in the reference.  args->ws / ws_bytes / counters / n_counters are ignored: the
 * split-K slabs and tickets live in a static device array inside the library (4 MiB of slabs + 4096 tickets per GPU, held
 * until the process ends).  A launch whose split needs more is rejected with an error naming both sizes (tune = 0 never
 * splits beyond 32 tiles, i.e. 1 MiB).  The call allocates nothing: the launch, preceded by one hipMemsetAsync of the
 * tickets only when the launch really splits K.  The first call on a device looks the array's address up
 * (hipGetSymbolAddress): make that call outside stream capture; later calls may be captured.  `stream` must belong to the
 * current device (a stream of another GPU is rejected: it would be handed this GPU's array).  Every call on one GPU
 * shares that GPU's array: calls that may overlap on one GPU (different streams) race and must use estts_gemm with
 * workspaces of their own instead. */
int estts_gemm_noalloc(const estts_gemm_args* args, void* stream);

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
    int32_t variant;         /* 0 = choose by launch size; 1 = 16-query workgroups whose waves split the keys (few
                                queries: batch 1); 2 = 64-query workgroups sharing each staged K / V tile (batches) */
} estts_attention_args;

int estts_attention(const estts_attention_args* args, void* stream);

/* ------------------------------------------------------------------------------------------
 * Row-wise normalisation combos:   out = LNb( dropout( act( LNa(y) ) ) + x )
 * each stage optional; LayerNorm over the last axis (biased variance, eps inside the sqrt).
 * dropout keeps element (m, c) when estts_dropout's generator says so for (drop_seed, drop_site, m, c) and
 * scales kept values by 1 / (1 - drop_rate) (tf.keras.layers.Dropout in training mode).
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
    float drop_rate;                    /* training: Dropout between act and the residual add; 0 = none */
    uint32_t drop_seed; uint32_t drop_site;
    const float* dot_b_ptr;             /* when not NULL, dot_b is read from device memory (a trainable bias) */
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

/* ------------------------------------------------------------------------------------------
 * WaveRNN generation, one launch per layer.  Between the layers an activation matrix X[n_seq, K] is an "activation
 * panel": bf16 hi + lo (x = hi + lo to ~16 mantissa bits) in 1 KiB blocks, one per (16 sequences, 32 columns); block
 * (rb, kb) starts at ((rb * kpitch + kb) * 2 + s) KiB (s = 0 hi, 1 lo) and holds for lane l = 0..63 the eight values
 * X[16 rb + (l & 15)][32 kb + 8 (l >> 4) .. + 7], the A operand of v_mfma_f32_16x16x32_bf16.  A panel has
 * ceil(n_seq / 16) row blocks of kpitch column blocks; columns no layer writes must be zero (allocate zeroed).
 *
 * estts_wavernn_layer, products in split-bf16 (hi * hi + lo * hi + hi * lo, fp32 accumulate):
 *   mode 0 (dense: fc1, fc2, fc3):  y = act(W a0 + bias), a0 = panel of ks0 K-steps;
 *          written to out_f32 [n_seq, ldo] and / or to columns [0, n_units) of out_panel (either may be NULL).
 *          W: for each group of unit_blocks * 16 units, the fragment images (K-step, unit block, hi | lo); rows are padded
 *          with zeros to a whole group, bias likewise.  A workgroup owns 128 sequences x one group, or with seq_tile 64
 *          (unit_blocks 2 only) 64 sequences x 32 units: the same results, a third fewer operand bytes per workgroup.
 *   mode 1 (rnn1 with the input layer) / mode 2 (rnn2), torch's gate order (r, z, n), x = a0 row, h = a1 row (the previous
 *          state, ks1 = n_units / 32 K-steps):
 *            r = sigmoid(Wx_r x + Wh_r h + b_r);  z = sigmoid(Wx_z x + Wh_z h + b_z);  n = tanh(Wx_n x + b_in + r (Wh_n h + b_hn))
 *            h' = (1 - z) n + z h_prev;   y = h' + res
 *          mode 1: x = the step input, Wx_* = W_ih W_I composed by the host, res = W_I x + b_I (the input layer's output,
 *          a fifth sum); accumulators xI r z gi_n gh_n, bias rows [5][n_units] in that order (b_r, b_z = input-side + recurrent).
 *          mode 2: res = x_res [n_seq, ldr]; accumulators r z gi_n gh_n, bias [4][n_units].
 *          h' goes to h_out (must differ from h_prev) and h_panel (must differ from a1); y to out_f32 (may be NULL) and
 *          out_panel.  The float32 arrays of a GRU call -- h_prev, h_out, x_res, out_f32 -- are private to the layer pair and
 *          "tile-native": ceil(n_seq / 128) * (n_units / 16) * 2048 floats, element ((workgroup * 8 + wave) * 64 + lane) * 4 + j
 *          being the (sequence, unit) pair that lane owns (the ld* fields are ignored); rnn2's x_res is rnn1's out_f32.
 *          Zero them before the first step.  W: for each 16-unit block (phase 0: K-step, accumulator it feeds, hi | lo), then
 *          (phase 1: K-step, r z gh_n, hi | lo).
 * Replaces: self.I, nn.GRUCell x 2 with their residual adds, fc1 / fc2 (+ ReLU) and fc3 of one generation step,
 * WaveRNN/models/fatchord_version.py:188-208.
 * ------------------------------------------------------------------------------------------ */
typedef struct estts_wavernn_layer_args {
    const uint16_t* a0; int32_t ks0; int32_t kpitch0;
    const uint16_t* a1; int32_t ks1; int32_t kpitch1;      /* GRU: previous state panel */
    const uint16_t* W;
    const float* bias;
    int32_t n_seq, n_units, mode, act, unit_blocks;        /* unit_blocks (dense only): 1, 2 or 4 blocks of 16 units per workgroup */
    int32_t seq_tile;                                      /* sequences per workgroup: 0 or 128; 64 (dense with unit_blocks 2 only) */
    const float* h_prev; int32_t ldh;
    const float* x_res; int32_t ldr;
    float* h_out; int32_t ldho;
    uint16_t* h_panel; int32_t h_kpitch;
    float* out_f32; int32_t ldo;
    uint16_t* out_panel; int32_t out_kpitch;
} estts_wavernn_layer_args;

int estts_wavernn_layer(const estts_wavernn_layer_args* args, void* stream);

/* The conditioning of one step as panels (the arithmetic of estts_wavernn_prep): for sequence b at step s = step[0], with
 * seq_info[b] = (m0, a0, start, live): sample n = start + s of its utterance, frame = n / hop, phase = n % hop,
 *   mel columns:  sum_j fir[phase, j] * mel[m0 + frame - j, :]      (m0 = first padded mel row of the utterance + pad - fir_kmin)
 *   aux columns:  aux[a0 + frame, :]                                 (a0 = first MelResNet row of the utterance)
 * all zero when n >= live (the zero padding fold_with_overlap appends).
 * xin_panel (kpitch 4: 128 columns) = [x_prev 0 0 0 0 0 0 0 | mel filtered to sample rate | aux slice 1 | zeros], of which this
 * call writes everything but the first group of eight columns (estts_wavernn_sample writes that one); the aux slices 2, 3, 4
 * (d columns each, d a multiple of 8, at most 32) go to K-step aux_ks of a2_panel, a3_panel, a4_panel (kpitch aux_kpitch):
 * the inputs of rnn2, fc1 and fc2, whose first aux_ks K-steps the preceding layers fill.  The call keeps a step counter of
 * its own (`step`, advanced by one per call).  The caller checks once that the filter stays inside the padded mels
 * (pad - fir_kmin - (ntaps - 1) >= 0 and pad + fir_kmin >= 0).
 * Replaces: what estts_wavernn_prep replaces. */
typedef struct estts_wavernn_cond_args {
    const float* mel;
    const float* aux;
    const float* fir; int32_t ntaps;
    const int32_t* seq_info;  /* [n_seq][4], 16-byte aligned */
    int32_t* step;            /* this call's own step counter (not the sampler's): read, then advanced by one */
    int32_t* done;            /* one zero-initialised int32 used to find the last workgroup, which advances step[0] */
    uint16_t* xin_panel;
    uint16_t* a2_panel; uint16_t* a3_panel; uint16_t* a4_panel;
    int32_t aux_ks, aux_kpitch;
    int32_t n_seq, feat, d, hop;
} estts_wavernn_cond_args;

int estts_wavernn_cond(const estts_wavernn_cond_args* args, void* stream);

/* Draws the next sample of every sequence from its logits and advances step[0].
 * mode 0 (RAW): softmax, inverse CDF at u (class = number of cdf entries <= u), value 2 c / (n - 1) - 1.
 * mode 1 (MOL): mixture of logistics with nr_mix = n_classes / 3 components, Gumbel-max component pick from
 *               u_mix, logistic inverse CDF at u, clamp to [-1, 1].
 * Uniform numbers: u[step * n_seq + b] (and u_mix[(step * n_seq + b) * nr_mix + k]) when given, else a
 * counter-based generator keyed by (seed, step, b).  wave[b * ld_wave + step] receives the value;
 * x_prev[b] receives forced[step * n_seq + b] when forced != NULL, else the value.
 * `done` is one zero-initialised int32 used to find the last workgroup, which advances step[0].
 * With step_next != NULL (a different int32 from step) the call writes step[0] + 1 there instead and leaves step and done
 * alone: consecutive calls that alternate between two counters (each reading what the previous one wrote) need no
 * workgroup to be the last -- the ~200 returning atomics on `done` otherwise end every call of a generation step.
 * Replaces: F.softmax + Categorical.sample (fatchord_version.py:210-216) and
 * sample_from_discretized_mix_logistic (utility/distribution.py:87-123). */
typedef struct estts_wavernn_sample_args {
    const float* logits; int32_t ld_logits; int32_t n_classes; int32_t mode;
    const float* u; const float* u_mix; const float* forced;
    uint32_t seed;
    float* wave; int32_t ld_wave;
    float* x_prev;
    int32_t* step; int32_t* done;
    int32_t* step_next;      /* NULL, or where step[0] + 1 goes (then step[0] and done are left alone) */
    int32_t n_seq;
    int32_t with_cond;       /* 1: also prepare the NEXT step's conditioning (step[0] + 1) into the panels of `cond`, exactly as
                                estts_wavernn_cond would, plus the value fed back as bf16 hi | lo followed by seven zeros in the
                                first column group of cond.xin_panel; cond.step / cond.done are not used */
    estts_wavernn_cond_args cond;
} estts_wavernn_sample_args;

int estts_wavernn_sample(const estts_wavernn_sample_args* args, void* stream);

/* What WaveRNN.generate does to the generated folds on the host (WaveRNN/models/fatchord_version.py:222-241 and
 * xfade_and_unfold, :321-383; utility/dsp.py:100-105), as one float64 pass on the device: for utterance u and output sample i
 *   v = sum over the (at most two) folds that cover i of wave[seq_off[u] + fold, i - fold (target + overlap)] * fade
 *       (fade: 0 over the first overlap / 2 samples of a fold's head, then sqrt(0.5 (1 + t)), t = linspace(-1, 1, n) -- passed
 *        as step_in = 2 / (n - 1) --; the mirror image over the fold's tail; 1 in between);
 *   v = sign(v) / (mu - 1) * (mu ^ |v| - 1)                      when mu > 0 (decode_mu_law(., n_classes, False));
 *   v *= 1 + k step_tail over the last fade_tail samples (k = 0 .. fade_tail - 1; the closing linspace(1, 0, 20 hop));
 *   out[out_off[u] + i] = v,  0 <= i < out_len[u].
 * wave holds float32 sample values, one row of ld_wave steps per fold.  Sums, windows and the expansion are computed in
 * float64 as NumPy does on the host (the windows bit for bit; pow() of the device library may differ in the last bit). */
typedef struct estts_wavernn_unfold_args {
    const float* wave; int32_t ld_wave;
    const int32_t* seq_off;      /* [n_utt] row of every utterance's first fold */
    const int32_t* n_folds;      /* [n_utt] */
    const int32_t* out_len;      /* [n_utt] samples kept */
    const int64_t* out_off;      /* [n_utt] start of every utterance in out */
    double* out;
    int32_t n_utt, target, overlap, max_len;
    int32_t mu;                  /* 0 = no expansion, else the number of classes */
    int32_t fade_tail;           /* 0 = none */
    double step_in, step_tail;
} estts_wavernn_unfold_args;

int estts_wavernn_unfold(const estts_wavernn_unfold_args* args, void* stream);



/* ==========================================================================================
 * Training (ForwardTransformer._train_step, pretrained_models/for_text_encoder/models.py:389-404:
 * forward with training=True, loss, gradients, Adam).  Master weights are float32 in the Keras layout
 * (kernel [K = taps * cin][N = cout], row-major), gradients have the same layout.
 * ========================================================================================== */

/* Re-packs a float32 master weight into the bf16 hi/lo fragment-major operand of estts_gemm.
 * mode 0 (forward):  W[n][k]          = src[k * ld_src + n]                       n < N,  k < K
 * mode 1 (data gradient of a dense / 1-D conv layer): with cin = K / taps,
 *                    W'[c][t' * N + n] = src[((taps - 1 - t') * cin + c) * ld_src + n]   c < cin, t' < taps, n < N
 *   so that dX = estts_gemm(A = dY, W', taps, c_per_tap = N, tap_lpad = taps - 1 - forward tap_lpad).
 * Destination rows / columns are padded with zeros to multiples of 64 (hi and lo each
 * ceil64(rows) * ceil64(cols) uint16). */
int estts_pack_weight(const float* src, int32_t ld_src, int32_t K, int32_t N, int32_t taps, int32_t mode,
                      uint16_t* hi, uint16_t* lo, void* stream);

/* The same for many weights in one launch (after an optimizer step every operand of the model is re-packed).
 * `jobs` is a DEVICE array sorted by first_unit; a unit is 8 consecutive destination values, job j owns units
 * [first_unit, first_unit + ceil64(rows) / 16 * ceil64(cols) / 32 * 64) with rows / cols as in estts_pack_weight, and
 * total_units is the end of the last job.  Nothing is validated on the device: the caller builds the table once. */
typedef struct estts_pack_job {
    const float* src;
    uint16_t* hi;
    uint16_t* lo;
    int64_t first_unit;
    int32_t ld_src, K, N, taps, mode, pad_;
} estts_pack_job;

int estts_pack_weights(const estts_pack_job* jobs, int32_t n_jobs, int64_t total_units, void* stream);

/* Weight gradient of a dense / 1-D conv layer on the matrix cores:
 *   dW[tap * c_per_tap + c, n] += sum_m A[row(m) + tap - tap_lpad, c] * dY[m, n]
 * (same row addressing as estts_gemm; float32 atomic accumulation: dW must be zeroed by the caller and the
 * summation order is not fixed).  splits: number of row ranges worked on concurrently, 0 = automatic.
 * Convolutions of 3 or 5 taps over a multiple of 64 channels run, when a workgroup gets enough rows (automatic) or splits
 * is given, as a kernel that stages every input tile once for all taps.
 * Replaces: the kernel gradients tf.GradientTape produces for Dense / Conv1D (models.py:402). */
typedef struct estts_wgrad_args {
    const float* A; int32_t lda;
    int32_t seg_len, taps, c_per_tap, tap_lpad;
    const float* dY; int32_t lddy;
    int32_t M, N, K;
    float* dW; int32_t lddw;
    int32_t precision;
    int32_t splits;
    float* db;           /* optional: db[n] += sum_m dY[m, n], the bias gradient, from the rows the kernel stages anyway
                            (zeroed by the caller like dW); NULL = not wanted */
} estts_wgrad_args;

int estts_gemm_wgrad(const estts_wgrad_args* args, void* stream);

/* Gradients of estts_attention given the saved weights P = attn [B, H, Lq, Lk] and dctx:
 *   dV = P^T dctx;  dP = dctx V^T;  dS = P o (dP - delta);  dQ = dS K / sqrt(depth);  dK = dS^T Q / sqrt(depth)
 * with delta[b, h, i] = sum_f dctx[b, i, h, f] * ctx[b, i, h, f]  (computed into `delta`, [B, H, Lq] floats, 16-byte
 * aligned like the other buffers).
 * q/k/v/ctx/dctx and dq/dk/dv use the row layouts of estts_attention. */
typedef struct estts_attention_bwd_args {
    const float* q; int32_t ldq;
    const float* k; int32_t ldk;
    const float* v; int32_t ldv;
    const float* ctx; int32_t ldo;
    const float* dctx; int32_t lddo;
    const float* attn;
    float* delta;
    float* dq; int32_t lddq;
    float* dk; int32_t lddk;
    float* dv; int32_t lddv;
    int32_t B, H, Lq, Lk, depth;
    int32_t precision;
} estts_attention_bwd_args;

int estts_attention_bwd(const estts_attention_bwd_args* args, void* stream);

/* Backward of estts_rownorm (`f` describes the forward call; its `out` / `dot_out` are not read: everything is
 * recomputed from y and x).  dout may be NULL when only the dot output carries gradient (ddot [M]).
 * dy receives the gradient w.r.t. y; dx the gradient w.r.t. the residual x (added to dx when dx_accumulate).
 * Parameter gradients are accumulated with float atomics (buffers zeroed by the caller; NULL = not wanted). */
typedef struct estts_rownorm_bwd_args {
    estts_rownorm_args f;
    const float* dout; int32_t lddout;
    const float* ddot;
    float* dy; int32_t lddy;
    float* dx; int32_t lddx; int32_t dx_accumulate;
    float* dga; float* dba; float* dgb; float* dbb;
    float* ddot_w; float* ddot_b;
    float* ws; int64_t ws_bytes;   /* optional scratch, ZERO before the first call (every call leaves it zero), 16 * 6 * 1024
                                      floats: the workgroups spread the float atomics of the parameter gradients over 16
                                      copies; NULL = every workgroup adds into the gradient vectors directly */
} estts_rownorm_bwd_args;

int estts_rownorm_bwd(const estts_rownorm_bwd_args* args, void* stream);

/* Column reductions over the rows of [M, C] matrices (float atomics into out0 / out1, zeroed by the caller).
 * With dy[m, c] = g[m, c] * act'(a[m, c]), where a = the activation OUTPUT: the stored tensor `a`, or, when a is
 * NULL and scale / shift are given, recomputed as act(z * scale[c] + shift[c]); identity when neither is given:
 *   mode 0 (moments):   out0[c] += sum_m (z - z[0, c]);  out1[c] += sum_m (z - z[0, c])^2
 *   mode 1 (bias grad): out0[c] += sum_m dy
 *   mode 2 (BatchNorm): out0[c] += sum_m dy;  out1[c] += sum_m dy * (z - mean[c]) * rstd[c]
 * modes 1 / 2 also store dy to dy_out when it is not NULL (may alias g).
 * Replaces: the batch statistics of tf.keras.layers.BatchNormalization(training=True) (layers.py:31,38,45,47)
 * and the bias / gamma / beta gradients of the tape. */
typedef struct estts_colreduce_args {
    const float* z; int32_t ldz;
    const float* g; int32_t ldg;
    const float* a; int32_t lda;
    int32_t act;
    const float* mean; const float* rstd;
    float* out0; float* out1;
    float* dy_out; int32_t lddy;
    int32_t M, C, mode;
    const float* scale; const float* shift;
    float* ws; int64_t ws_bytes;   /* optional scratch, ZERO before the first call (every call leaves it zero): 16 copies of
                                      the two sums, 128 * ceil(C / 256) * 256 bytes, over which the row splits spread their
                                      float atomics; NULL = all splits add into out0 / out1 directly */
    int32_t defer;                 /* 1 (modes 0 / 2): leave the sums in the 16 copies at ws, laid out [copy][sum 0 / 1]
                                      [ceil(C / 256) * 256], for the launch that consumes them (estts_bn_finalize_slots,
                                      estts_bn_bwd_slots); out0 / out1 are not touched and ws is required.  In mode 0 the
                                      moments are then taken about shift[c], or about 0 when shift is NULL. */
    float* clear; int64_t clear_floats;   /* with defer: another region, of clear_floats floats, that this launch sets to
                                      zero (copies an earlier consumer has finished reading), or NULL */
} estts_colreduce_args;

int estts_colreduce(const estts_colreduce_args* args, void* stream);

/* Batch statistics from mode-0 sums:  mean = z0 + s0 / M;  var = s1 / M - (s0 / M)^2 (biased);  rstd = (var + eps)^-1/2;
 * scale = gamma * rstd;  shift = beta - mean * scale;  moving_mean = moving_mean * momentum + mean * (1 - momentum);
 * moving_var = moving_var * momentum + var * moving_var_scale * (1 - momentum)   (moving_var_scale: 1 for Keras,
 * M / (M - 1) for torch.nn.BatchNorm1d, which stores the unbiased variance).
 * z0 is row 0 of the reduced matrix (the shift used by mode 0). */
int estts_bn_finalize(const float* s0, const float* s1, const float* z0, int32_t M, int32_t C, float eps, float momentum,
                      float moving_var_scale, const float* gamma, const float* beta, float* moving_mean, float* moving_var,
                      float* mean, float* rstd, float* scale, float* shift, void* stream);

/* The same from sums still spread over 16 copies (estts_colreduce with defer, or estts_gemm with stat_slots): copy k of
 * s0 / s1 for column c is slots[(2 k + 0 / 1) * slot_stride + c], slot_stride = ceil(C / 256) * 256.  Adds the copies up and
 * sets them to zero.  z0 [C]: what the moments were taken about (row 0 of the matrix, or the shift vector); it may be
 * moving_mean itself (read before it is updated).  about_next [C] (or NULL) also receives the batch mean: the shift for the
 * same layer's next step (the closer the shift to the mean, the less the variance loses in its subtraction); it may be z0. */
int estts_bn_finalize_slots(float* slots, int64_t slot_stride, const float* z0, int32_t M, int32_t C, float eps, float momentum,
                            float moving_var_scale, const float* gamma, const float* beta, float* moving_mean,
                            float* moving_var, float* mean, float* rstd, float* scale, float* shift, float* about_next,
                            void* stream);

/* out[m, c] = act(z[m, c] * scale[c] + shift[c]) + (R ? R[r_mod ? m % r_mod : m, c] : 0)   (scale / shift NULL = 1 / 0) */
int estts_affine_act(const float* z, int32_t ldz, const float* scale, const float* shift, int32_t act, const float* R,
                     int32_t ldr, int32_t r_mod, float* out, int32_t ldo, int32_t M, int32_t C, void* stream);

/* BatchNorm input gradient:  dz = gamma * rstd * (dy - dbeta / M - (z - mean) * rstd * dgamma / M). */
int estts_bn_bwd(const float* dy, int32_t lddy, const float* z, int32_t ldz, const float* mean, const float* rstd,
                 const float* gamma, const float* dgamma, const float* dbeta, float* dz, int32_t lddz, int32_t M,
                 int32_t C, void* stream);

/* The same with dbeta / dgamma = the mode-2 sums still spread over 16 copies (estts_colreduce with defer; layout as for
 * estts_bn_finalize_slots, sum 0 = dbeta, sum 1 = dgamma).  The totals are also ADDED to dgamma [C] / dbeta [C] (the
 * parameter gradients).  The copies are left as they are: hand the region to a later estts_colreduce as `clear`.
 * scale / shift != NULL: dy is the gradient w.r.t. act(z scale + shift) and is taken through the activation here (dy <- dy *
 * act'), with the arithmetic of the estts_colreduce call that built the sums from the same operands -- that call then needs no
 * dy_out, which saves a store and a load of the tensor. */
int estts_bn_bwd_slots(const float* dy, int32_t lddy, const float* z, int32_t ldz, const float* mean, const float* rstd,
                       const float* gamma, const float* slots, int64_t slot_stride, float* dgamma, float* dbeta, float* dz,
                       int32_t lddz, int32_t M, int32_t C, int32_t act, const float* scale, const float* shift, void* stream);

/* out[m, c] = x[m, c] * keep(seed, site, m, c) / (1 - rate): tf.keras.layers.Dropout in training mode with a
 * counter-based generator (the same call on a gradient is the backward pass).  May run in place. */
int estts_dropout(const float* x, int32_t ldx, float* out, int32_t ldo, int32_t M, int32_t D, float rate, uint32_t seed,
                  uint32_t site, void* stream);

/* out[m, c] = beta * out[m, c] + alpha * x[m, c]   (gradient accumulation on residual paths) */
int estts_axpby(const float* x, int32_t ldx, float alpha, float* out, int32_t ldo, float beta, int32_t M, int32_t D,
                void* stream);

/* Patches of a 2-D convolution over an NHWC image x [B, H, W, C] (TensorFlow 'SAME' geometry: pad_t / pad_l on the low
 * side): out[(b, ho, wo), (kh_i * kw + kw_i) * C + c] = x[b, ho * stride + kh_i - pad_t, wo * stride + kw_i - pad_l, c], zero
 * outside the image; columns from kh * kw * C up to ldo are zeroed too (K rounded up for estts_gemm).
 * estts_col2im is its transpose: dx[b, h, w, c] = the sum of the patch entries that value was copied to.
 * With them the gradients of Conv2D(3 x 3, stride 2, 'same') in ReferenceEncoderGST.call_convs (TransformerTTS/model/layers.py:554-559)
 * are estts_gemm_wgrad(patches, dZ) and estts_col2im(estts_gemm(dZ, W^T)); the reference gets them from tf.GradientTape
 * (model/traning_steps.py:54). */
int estts_im2col(const float* x, float* out, int32_t ldo, int32_t B, int32_t H, int32_t W, int32_t C, int32_t Ho, int32_t Wo,
                 int32_t kh, int32_t kw, int32_t stride, int32_t pad_t, int32_t pad_l, void* stream);
int estts_col2im(const float* dp, int32_t ldp, float* dx, int32_t B, int32_t H, int32_t W, int32_t C, int32_t Ho, int32_t Wo,
                 int32_t kh, int32_t kw, int32_t stride, int32_t pad_t, int32_t pad_l, void* stream);

/* Mean squared error over n contiguous elements (small tensors: one workgroup): loss[0] += mean((a - b)^2),
 * da[i] = scale * 2 (a[i] - b[i]) / n (the gradient with respect to a; the one with respect to b is its negative).
 * Replaces: l2_loss between the two style embeddings, TransformerTTS/utils/losses.py:65-68 as used at
 * TransformerTTS/model/traning_steps.py:41-49. */
int estts_mse(const float* a, const float* b, int32_t n, float* loss, float* da, float scale, void* stream);

/* Mutual-information estimate of MINE from the statistics network's scores on the joint and the marginal samples of one
 * batch (contiguous [n] each), with its gradient, in one single-workgroup launch.
 *   div_type 0 (KL):    mi = mean(joint) - log(E),  E = s * mean(exp(marginal)) + (1 - s) * prev_exp[1]
 *   div_type 1 (Renyi): mi = sum over betas of term1 - term2 with the max-shifted, smoothed log-mean-exp terms of
 *                       TransformerTTS/model/MINE.py:56-93 (beta == 0 / beta == 1 use the plain means there)
 * prev_exp / next_exp: [n_beta][2] device floats, the smoothed exponential terms of the previous / this step (KL uses row 0);
 * mi: device float; d_joint / d_marginal [n] = gscale * d mi / d score (gscale = -1: the ascent direction's negative loss).
 * Replaces: MINE.measure_mi, TransformerTTS/model/MINE.py:40-95, and its tape gradient (model/traning_steps.py:78-81). */
int estts_mine_divergence(const float* joint, const float* marginal, int32_t n, int32_t div_type, const float* betas,
                          int32_t n_beta, float smoothing, const float* prev_exp, float* next_exp, float* mi, float* d_joint,
                          float* d_marginal, float gscale, void* stream);

/* CLUB's log-likelihood and upper bound from mu, log_var, y and the batch-shuffled y (contiguous [B, D] each):
 *   positive = -(mu - y)^2 / 2 / exp(log_var);  negative = -(mu - y_shuffled)^2 / 2 / exp(log_var)
 *   out[0] = lld = mean_b sum_d positive;  out[1] = bound = mean_b (sum_d positive - sum_d negative)
 *   d_mu / d_log_var = gscale * d lld / d(mu / log_var)
 * Replaces: CLUB.call after its two networks, TransformerTTS/model/MINE.py:181-200, and the tape gradient of lld. */
int estts_club_terms(const float* mu, const float* log_var, const float* y, const float* y_shuffled, int32_t B, int32_t D,
                     float* out, float* d_mu, float* d_log_var, float gscale, void* stream);

/* Per-(batch, head) scaling of the attention context (HeadDrop, layers.py:74-97): x[b*L + t, h*depth + f] *= scale[b*H + h]. */
int estts_head_scale(float* x, int32_t ldx, const float* scale, int32_t B, int32_t L, int32_t H, int32_t depth, void* stream);

/* demb[tokens[b, n], :] += dx[b * N + n, :] * scale  (float atomics): gradient of the Embedding lookup. */
int estts_embed_bwd(const int32_t* tokens, const float* dx, int32_t lddx, float* demb, int32_t B, int32_t N, int32_t D,
                    float scale, void* stream);

/* Gradient of estts_expand: dx[b, n, :] = sum of dout[b, t, :] over the frames t of token n (cum[n-1] <= t < cum[n]). */
int estts_expand_bwd(const float* dout, int32_t lddo, const int32_t* cum, float* dx, int32_t lddx, int32_t B, int32_t N,
                     int32_t T, int32_t D, void* stream);

/* Gradient of estts_masked_mae times `scale`: dpred[m, c] = scale * sign(pred - target) * any(target_row != 0) / (M * C). */
int estts_masked_mae_bwd(const float* target, int32_t ldt, const float* pred, int32_t ldp, float* dpred, int32_t lddp,
                         int32_t M, int32_t C, float scale, void* stream);

/* Keras Adam (utils/config_manager.py:172-176) over flat buffers, gradient pre-scaled by grad_scale:
 *   g = grad * grad_scale;  m = b1 m + (1 - b1) g;  v = b2 v + (1 - b2) g^2;  p -= lr_t * m / (sqrt(v) + eps)
 * lr_t = lr * sqrt(1 - b2^t) / (1 - b1^t) is computed by the caller. */
int estts_adam(float* p, const float* grad, float* m, float* v, int64_t n, float lr_t, float beta1, float beta2,
               float eps, float grad_scale, void* stream);

/* ------------------------------------------------------------------------------------------
 * WaveRNN training (WaveRNN/train_wavernn.py:27-42: teacher-forced forward, loss, backward, Adam).
 * ------------------------------------------------------------------------------------------ */

/* One GRU time step on precomputed projections (gate order r, z, n as in estts_gru_gates), previous state read from
 * h_prev and the new one written to h_out (the training pass keeps every state):
 *   h_out = (1 - z) n + z h_prev;   out = h_out + (x_res ? x_res : 0)   (out may be NULL)
 * Replaces: one time step of nn.GRU, fatchord_version.py:135-142. */
int estts_gru_step_fwd(const float* gi, int32_t ldgi, const float* gh, int32_t ldgh, const float* h_prev, int32_t ldhp,
                       float* h_out, int32_t ldho, const float* x_res, int32_t ldx, float* out, int32_t ldo, int32_t M,
                       int32_t H, void* stream);

/* Backward of that step.  The gradient w.r.t. h_out is dh (from the layer output) + dh_rec (from the next time step,
 * NULL at the last one); writes dgi, dgh [M, 3H] (gradients of the two projections) and dh_prev = (dh + dh_rec) * z
 * (the direct path; the caller adds dgh W_hh with estts_gemm to obtain the next dh_rec). */
int estts_gru_step_bwd(const float* gi, int32_t ldgi, const float* gh, int32_t ldgh, const float* h_prev, int32_t ldhp,
                       const float* dh, int32_t lddh, const float* dh_rec, int32_t lddhr, float* dgi, int32_t lddgi,
                       float* dgh, int32_t lddgh, float* dh_prev, int32_t lddhp, int32_t M, int32_t H, void* stream);

/* One GRU time step of a training recurrence in a single launch: the recurrent product with the gate arithmetic as its
 * epilogue (torch.nn.GRU gate order r, z, n: WaveRNN/models/fatchord_version.py:117-125, and the GST reference encoder's GRU,
 * TransformerTTS/model/layers.py:437-451).  A recurrence is a chain of dependent launches that costs per launch, so one launch
 * per step instead of two (estts_gemm + estts_gru_step_fwd / _bwd) shortens the chain by about 40 %.
 *   estts_gru_fused_fwd:  gh = h_prev W_hh^T + bias;  r = s(gi_r + gh_r);  z = s(gi_z + gh_z);  n = tanh(gi_n + r gh_n);
 *                         h_out = (1 - z) n + z h_prev;  out = h_out (+ x_res) when out != NULL.  gh [M, 3H] is stored (the
 *                         backward pass recomputes the gates from gi and gh).  W: W_hh [3H x H] packed as an estts_gemm weight.
 *   estts_gru_fused_bwd:  d = dh + dgh_next W + dhd  (the gradient of step t's state: from above, through the gates of step
 *                         t + 1, and through its z path), then the gate gradients of step t exactly as estts_gru_step_bwd:
 *                         dgi, dgh [M, 3H] and dhd <- d z (same buffer: read, then overwritten).  W: the transposed pack
 *                         [H x 3H] (the weight estts_gemm uses for the data gradient).  gi, gh, h_prev, dh: step t's.
 * H a multiple of 16, H <= 1536 forward / 3H <= 1536 backward; h_out != h_prev and dgh != dgh_next (other workgroups of the
 * launch still read them). */
typedef struct estts_gru_fused_args {
    const float* gi; int32_t ldgi;
    float* gh; int32_t ldgh;
    const float* h_prev; int32_t ldhp;
    const uint16_t* W_hi; const uint16_t* W_lo; int32_t ldw;
    const float* bias;
    float* h_out; int32_t ldho;
    const float* x_res; int32_t ldx;
    float* out; int32_t ldo;
    const float* dgh_next; int32_t lddghn;
    const float* dh; int32_t lddh;
    float* dgi; int32_t lddgi;
    float* dgh; int32_t lddgh;
    float* dhd; int32_t lddhd;
    int32_t M, H, precision;
} estts_gru_fused_args;

int estts_gru_fused_fwd(const estts_gru_fused_args* args, void* stream);
int estts_gru_fused_bwd(const estts_gru_fused_args* args, void* stream);
/* Two independent steps of the same width H and precision in one launch (the two layers of a WaveRNN window, one a chunk of
 * steps behind the other: two recurrences then cost one chain of launches instead of two that run side by side). */
int estts_gru_fused_fwd2(const estts_gru_fused_args* a, const estts_gru_fused_args* b, void* stream);
int estts_gru_fused_bwd2(const estts_gru_fused_args* a, const estts_gru_fused_args* b, void* stream);

/* Backward of estts_wavernn_prep for teacher-forced windows: every utterance has `frames` frames of `hop` samples and
 * row (u * frames + f) * hop + ph of the gradient matrices belongs to frame f, phase ph.
 *   d_aux[u * frames + f, q * d + c] = sum_ph dq[row, c]        dq = (dxin + 1 + feat, da2, da3, da4) for q = 0..3
 *   dfir[ph, j] += sum_{u, f, c} dxin[row, 1 + c] * mel[mel_off[u] + f + pad - (fir_kmin + j), c]     (float atomics)
 * Replaces: the gradients autograd sends through UpsampleNetwork.forward's repeat / Conv2d chain and the aux slicing
 * (fatchord_version.py:79-86,126-131). */
typedef struct estts_wavernn_prep_bwd_args {
    const float* dxin; int32_t ld_dxin;
    const float* da2; int32_t ld_da2;
    const float* da3; int32_t ld_da3;
    const float* da4; int32_t ld_da4;
    const float* mel; const int32_t* mel_off;
    float* d_aux;             /* [n_utt * frames, 4 d] */
    float* dfir;              /* [hop, ntaps], zeroed by the caller */
    int32_t ntaps, fir_kmin;
    int32_t n_utt, frames, feat, d, hop, pad;
} estts_wavernn_prep_bwd_args;

int estts_wavernn_prep_bwd(const estts_wavernn_prep_bwd_args* args, void* stream);

/* Softmax cross-entropy over the last axis, forward and backward in one pass:
 *   loss[0] += scale * sum_m (logsumexp(logits[m, :]) - logits[m, labels[m]]);  dlogits[m, c] = scale * (softmax - onehot)
 * (scale = 1 / M gives F.cross_entropy's mean, train_wavernn.py:39,118).  loss[0] zeroed by the caller. */
int estts_softmax_xent(const float* logits, int32_t ld, const int32_t* labels, float* loss, float* dlogits, int32_t lddl,
                       int32_t M, int32_t C, float scale, void* stream);

/* Stop-token loss of the autoregressive model: new_scaled_crossentropy / masked_crossentropy (TransformerTTS/utils/losses.py:4-29),
 * SparseCategoricalCrossentropy(from_logits) with sample weights w[m] = (label != 0) + (label == index) * (scaling - 1):
 *   loss[0] += scale * sum_m w[m] * ce[m];  dlogits[m, c] = scale * w[m] * (softmax - onehot)   (dlogits may be NULL)
 * scale = 1 / M reproduces Keras' SUM_OVER_BATCH_SIZE (padded positions count in the denominator); scaling = 1 is
 * masked_crossentropy.  loss[0] zeroed by the caller. */
int estts_scaled_xent(const float* logits, int32_t ld, const int32_t* labels, float* loss, float* dlogits, int32_t lddl,
                      int32_t M, int32_t C, int32_t index, float scaling, float scale, void* stream);

/* Discretized mixture-of-logistics negative log-likelihood and its gradient (WaveRNN/utility/distribution.py:16-84,
 * reduce=True): y_hat [M, 3 nr_mix] = (mixture logits | means | log scales), y [M] in [-1, 1].
 *   loss[0] += scale * sum_m nll_m;  d[m, :] = scale * d nll_m / d y_hat[m, :]      nr_mix <= 16 */
int estts_mol_loss(const float* y_hat, int32_t ld, const float* y, float* loss, float* d, int32_t ldd, int32_t M,
                   int32_t nr_mix, int32_t num_classes, float log_scale_min, float scale, void* stream);

/* ------------------------------------------------------------------------------------------
 * One decoder position through a CrossAttentionResnorm block in one launch, for incremental decoding
 * (TransformerTTS/model/layers.py:258-274 with transformer_utils.py:24-44 for a single query row; the reference re-runs the
 * whole decoder per step, model/AutoregressiveTransformer.py:279-296).  Per row m < M:
 *   x   = ln_y ? LNb(act_ln(LNa(ln_y[m])) + ln_x[m]) : x[m]       (the ln_* fields as in estts_gemm's LayerNorm prologue; the
 *                                                                  normalised row is also written to ln_out[m] when given)
 *   q   = x Wq + bq;   p[h, j] = softmax_j(q_h . k[j, h] / sqrt(depth) - 1e9 key_mask[m, j]);   ctx_h = sum_j p[h, j] v[j, h]
 *   y[m] = [x | ctx] Wo + bo
 * Wq ([d, d]) and Wo ([d, 2 d]) in the packed bf16 hi + lo layout of estts_gemm (ldw = K rounded up to 64), k / v float32
 * [Lk, >= d] rows of row m at k / v + m * kv_batch_stride (floats; 0 = shared), attn (may be NULL) receives p as [H, Lk] at
 * attn + m * attn_row_stride.  This kernel takes d = 256 with heads of depth 64 and Lk <= 256; y may be the buffer ln_y points
 * to (a row is read before it is written).  With q given the kernel is attention + output projection only: a self-attention
 * block's step after the product that wrote the position's q | k | v row (k / v = the cache up to and including it, Lk = the
 * position + 1; TransformerTTS/model/layers.py:217-233).  A row is four workgroups (one per head) whose shares of y meet in ws; launches
 * that share ws / counters must be ordered on one stream, as with estts_gemm's.  The projections use the split bf16 operands of estts_gemm
 * (hi hi + hi lo + lo hi in float32) in another order of summation; attention in float32: equal to the three launches to 1e-5
 * of the largest value, not bit for bit.
 * ------------------------------------------------------------------------------------------ */
typedef struct estts_cross_row_args {
    const float* x; int32_t ldx;
    const float* ln_y; const float* ln_x;
    const float* ln_ga; const float* ln_ba; const float* ln_gb; const float* ln_bb;
    float* ln_out;
    int32_t ln_ldy, ln_ldx, ln_ldo, ln_act;
    float ln_eps;
    const uint16_t* wq_hi; const uint16_t* wq_lo; int32_t ldwq; const float* bq;
    const float* k; const float* v; int32_t ldk, ldv; int64_t kv_batch_stride;
    const float* key_mask;
    float* attn; int64_t attn_row_stride;
    const uint16_t* wo_hi; const uint16_t* wo_lo; int32_t ldwo; const float* bo;
    float* y; int32_t ldy;
    int32_t M, d, H, Lk;
    float* ws; int64_t ws_bytes;             /* slab for the heads' shares of y: M * 4 * 256 floats */
    int32_t* counters; int32_t n_counters;   /* >= M, zero before the first launch; every launch leaves them at zero */
    const float* q; int32_t ldq;             /* given: the query rows themselves (bias included); wq_hi / wq_lo / bq are not read */
} estts_cross_row_args;

int estts_cross_attention_row(const estts_cross_row_args* args, void* stream);

/* ------------------------------------------------------------------------------------------
 * One decoder position through a whole decoder block in one launch: SelfAttentionResNorm, CrossAttentionResnorm and the dense
 * product of FFNResNorm (TransformerTTS/model/layers.py:217-233, 258-274, 104-124; the block, layers.py:277-297), for
 * incremental decoding.  Per row m < M, with LN(v; g, b) the LayerNormalization over the 256 columns:
 *   x    = ln_y ? LNb(act_ln(LNa(ln_y[m])) + ln_x[m]) : x[m]                     (as in estts_gemm's LayerNorm prologue)
 *   q | k | v = x Wqkv + bqkv  -> row `pos` of the cache qkv ([rows, >= 768] per m);  ctx = attention of q over cache rows
 *          0 .. pos (keys at columns 256.., values at 512..; -1e9 self_mask[m, j] added to the scores)
 *   y_s  = [x | ctx] Wo_s + bos;        x'  = LN(LN(y_s; sa_ga, sa_ba) + x; sa_gb, sa_bb)
 *   q'   = x' Wq_c + bqc;  ctx' = attention of q' over k / v (Lk rows, key_mask) -> weights to attn ([H, Lk] at
 *          attn + m * attn_row_stride, may be NULL);     y_c = [x' | ctx'] Wo_c + boc
 *   x2[m] = LN(y_c + x'; ca_g, ca_b);   y2[m] = x2 W_ff + bff
 * W_ff is the block's two dense layers folded into one 256 x 256 product (the reference applies no activation between them,
 * layers.py:116-119); the block's closing LN(relu(LN(y2)) + x2) is left to the consumer of (y2, x2), which is the next block's
 * ln_y / ln_x or estts_gemm's prologue.  All weights in the packed bf16 hi + lo layout of estts_gemm.  Width 256, 4 heads,
 * pos < 256, Lk <= 256, M <= 64.  ws: M * 2 * 4 * 256 floats of scratch; counters: 2 M int32 that belong to this entry point
 * alone, zero at allocation (a launch leaves them ready for the next one; they are not the zero-after-use counters of
 * estts_gemm).  The four workgroups of a row wait for one another inside the kernel: launches that share ws / counters must be
 * ordered on one stream.  Same sums as the separate launches up to float32 summation order (1e-5 of the largest value).
 * ------------------------------------------------------------------------------------------ */
typedef struct estts_decoder_block_args {
    const float* x; int32_t ldx;
    const float* ln_y; const float* ln_x;
    const float* ln_ga; const float* ln_ba; const float* ln_gb; const float* ln_bb;
    int32_t ln_ldy, ln_ldx, ln_act;
    float ln_eps;                              /* epsilon of every LayerNormalization of the block */
    const uint16_t* wqkv_hi; const uint16_t* wqkv_lo; int32_t ldwqkv; const float* bqkv;
    float* qkv; int32_t ldqkv; int64_t qkv_batch_stride; int32_t pos;
    const float* self_mask; int64_t self_mask_stride;
    const uint16_t* wos_hi; const uint16_t* wos_lo; int32_t ldwos; const float* bos;
    const float* sa_ga; const float* sa_ba; const float* sa_gb; const float* sa_bb;
    const uint16_t* wqc_hi; const uint16_t* wqc_lo; int32_t ldwqc; const float* bqc;
    const float* k; const float* v; int32_t ldk, ldv; int64_t kv_batch_stride; int32_t Lk;
    const float* key_mask;
    float* attn; int64_t attn_row_stride;
    const uint16_t* woc_hi; const uint16_t* woc_lo; int32_t ldwoc; const float* boc;
    const float* ca_g; const float* ca_b;
    const uint16_t* wff_hi; const uint16_t* wff_lo; int32_t ldwff; const float* bff;
    float* y2; int32_t ldy2; float* x2; int32_t ldx2;
    float* ws; int64_t ws_bytes; int32_t* counters; int32_t n_counters;
    int32_t M, d, H;
} estts_decoder_block_args;

int estts_decoder_block_row(const estts_decoder_block_args* args, void* stream);

/* ------------------------------------------------------------------------------------------
 * The decoder's pre-net for one position per row in one launch (TransformerTTS/model/layers.py:394-402 with the padding mask
 * of model/transformer_utils.py:47-50), for incremental decoding.  Per row m < M:
 *   mask[m] = all C channels of x[m] are zero ? 1 : 0                                  (mask may be NULL)
 *   p1  = drop_1(relu(x[m] W1 + b1));      out[m] = alpha * drop_2(relu(p1 W2 + b2)) + R[m]        (R may be NULL)
 * drop_s(v)[n] = keep(drop_seed, drop_site_s, drop_row0 + m, n) ? v[n] / (1 - drop_rate) : 0, the generator of estts_gemm, so
 * the result is estts_row_zero_mask + two estts_gemm launches with the same dropout arguments, up to float32 summation order.
 * W1 [256, C] and W2 [256, 256] in the packed bf16 hi + lo layout of estts_gemm; C <= 128, a multiple of 4.
 * ------------------------------------------------------------------------------------------ */
typedef struct estts_prenet_row_args {
    const float* x; int32_t ldx; int32_t C;
    float* mask;
    const uint16_t* w1_hi; const uint16_t* w1_lo; int32_t ldw1; const float* b1; int32_t N1;
    const uint16_t* w2_hi; const uint16_t* w2_lo; int32_t ldw2; const float* b2; int32_t N2;
    float alpha; const float* R; int32_t ldr;
    float drop_rate; uint32_t drop_seed, drop_site1, drop_site2; int32_t drop_row0;
    float* out; int32_t ldo;
    int32_t M;
} estts_prenet_row_args;

int estts_prenet_row(const estts_prenet_row_args* args, void* stream);

/* ------------------------------------------------------------------------------------------
 * Step programs: a fixed sequence of launches whose arguments depend affinely on a step index (an incremental decoder:
 * row i of a cache, a key range of i + 1, ...).  Each op carries the argument block of step 0 and a patch list; at step s
 * the field at byte `offset` of a copy of the block is advanced by s * stride (a pointer, in bytes, or an int32), then the
 * op's entry point is called.  This issues a whole decoder step from one host call, without rebuilding argument blocks
 * in the host language.  No counterpart in the reference (its loop, model/AutoregressiveTransformer.py:279-296, re-runs
 * the whole Keras decoder every step).
 * kind: 0 estts_gemm, 1 estts_attention, 2 estts_rownorm, 3 estts_row_zero_mask (args = estts_rowmask_args),
 * 4 estts_cross_attention_row, 5 estts_decoder_block_row, 6 estts_prenet_row.
 * ------------------------------------------------------------------------------------------ */
typedef struct estts_rowmask_args {
    const float* x; int32_t ldx;
    float* mask;
    int32_t M, D;
} estts_rowmask_args;

#define ESTTS_PROG_MAX_PATCH 8
typedef struct estts_prog_op {
    int32_t kind;
    int32_t n_patch;
    const void* args;
    int32_t patch_offset[ESTTS_PROG_MAX_PATCH];
    int32_t patch_is_ptr[ESTTS_PROG_MAX_PATCH];
    int64_t patch_stride[ESTTS_PROG_MAX_PATCH];
} estts_prog_op;

int estts_run_program(const estts_prog_op* ops, int32_t n_ops, int32_t step, void* stream);

/* ------------------------------------------------------------------------------------------
 * Mel front-end (what turns a reference wav into the mel the style encoder / vocoder training read).
 * Replaces librosa.stft + librosa.feature.melspectrogram + the normalisers at TransformerTTS/utils/audio.py:20-40,87-119,
 * WaveRNN/utility/dsp.py:40-80 and gst_tacotron/util/audio.py:27-53,118-148.
 *
 * estts_gemm_f32:  C[m, n] = sum_k A[m, k] * W[n, k] in exact float32 on the matrix cores (row strides in floats; A rows may
 *   overlap: frame m of a signal is A + m * hop, so lda = hop_length).  K % 16 == 0, lda / ldw % 4 == 0.  With W = the
 *   window-weighted DFT matrix (rows [0, nb): w[k] cos(2 pi f k / n_fft); rows [nb, 2 nb): -w[k] sin(...)) this is the STFT;
 *   with W = the mel filterbank it is linear_to_mel.
 * estts_spec_magnitude:  mag[m, c] = hypot(x[m, c], x[m, nb + c])                                    c < nb
 * estts_mel_normalize:   v = max(x, 1e-5);
 *   mode 0: out = clip((20 log10 v - ref_level_db - min_level_db) / -min_level_db, 0, 1) * scale + shift
 *           (vocoder mels: ref 0, min -100, scale 1, shift 0; the TTS "WaveRNN" normaliser: scale 8, shift -4;
 *            gst_tacotron: ref 20)
 *   mode 1: out = log v   (the "MelGAN" normaliser)
 * ------------------------------------------------------------------------------------------ */
int estts_gemm_f32(const float* A, int64_t lda, const float* W, int64_t ldw, float* C, int64_t ldc, int32_t M, int32_t N, int32_t K,
                   void* stream);
int estts_spec_magnitude(const float* reim, int64_t ld, int32_t nb, float* mag, int64_t ldm, int32_t M, void* stream);
int estts_mel_normalize(const float* x, int64_t ldx, float* out, int64_t ldo, int32_t M, int32_t C, int32_t mode, float ref_level_db,
                        float min_level_db, float scale, float shift, void* stream);

/* ------------------------------------------------------------------------------------------
 * Duration extraction from the decoder's cross-attention maps.
 * Replaces the map reductions of get_durations_from_alignment (TransformerTTS/utils/alignments.py:87-165: padding lengths
 * :103-112, weight_mask and the per-head scores :116-122 with :31-36, best head / weighted reference map :124-128,
 * binary_attention :78-84, the column sums of the non-binary branch :140), called by TransformerTTS/extract_durations.py:177,212
 * on maps it first copies to the host (:129-153).  The sequential part (jump repair, rounding repair, gap filling) stays on the
 * host and reads only these reductions.  All sums are float64 like the reference's.
 *
 * With T' = mel_len[b] - 2 and N' = phon_len[b] - 2 (start / end vectors and tokens removed) and W[b, h] the window
 * attn[b, h, 1 : 1 + T', 1 : 1 + N'] of the (B, H, Lq, Lk) float32 maps:
 * estts_alignment_lengths:  mel_len[b] = T - #{t : sum_c |mels[b, t, c]| == 0},  phon_len[b] = N - #{n : tokens[b, n] == 0}
 * estts_alignment_stats:    score[b, h] = sum_{t, n} |n / N' - t / T'| W[b, h][t, n];  colsum[b, h, n] = sum_t W[b, h][t, n]
 *                           (colsum has Lk entries per head, zero past N')
 * estts_alignment_peaks:    ref = W[b, best_head[b]] with best_head[b] = first argmin_h score[b, h] (weighted == 0), or
 *                           ref = sum_h W[b, h] / score[b, h] (weighted != 0); for t < T':  peak_idx[b, t] = first argmax_n ref[t, n],
 *                           peak_val[b, t] = that maximum, peak_ties[b, t] = number of entries equal to it (the reference asserts 1);
 *                           rows t >= T' get -1 / 0 / 0.  peak_* have Lq entries per utterance.
 * Lengths with mel_len - 1 > Lq or phon_len > Lk are clamped to the map on the device; the host layer rejects them.
 * ------------------------------------------------------------------------------------------ */
int estts_alignment_lengths(const float* mels, int64_t ldm, int32_t T, int32_t C, const int32_t* tokens, int32_t N, int32_t B,
                            int32_t* mel_len, int32_t* phon_len, void* stream);
int estts_alignment_stats(const float* attn, int32_t B, int32_t H, int32_t Lq, int32_t Lk, const int32_t* mel_len,
                          const int32_t* phon_len, double* score, double* colsum, void* stream);
int estts_alignment_peaks(const float* attn, int32_t B, int32_t H, int32_t Lq, int32_t Lk, const int32_t* mel_len,
                          const int32_t* phon_len, const double* score, int32_t weighted, int32_t* peak_idx, double* peak_val,
                          int32_t* peak_ties, int32_t* best_head, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* ESTTS_H */
