// The dense work of a WaveRNN generation step (see estts_wavernn_layer / estts_wavernn_cond in include/estts.h): each
// layer -- rnn1 with the input layer, rnn2, fc1, fc2, fc3 -- is ONE launch that multiplies, applies its epilogue (GRU gates and
// state update, or bias + ReLU) and hands its output to the next layer in the matrix cores' own operand layout.
//
// Activation panels.  Between the layers an activation matrix X[n_seq, K] lives as bf16 hi + lo (x = hi + lo to ~16 mantissa
// bits) in 1 KiB blocks, one per (16 sequences, 32 columns): block (rb, kb) starts at ((rb * kpitch + kb) * 2 + s) KiB
// (s = 0 hi, 1 lo) and holds, for lane l = 0..63, the eight values X[16 rb + (l & 15)][32 kb + 8 (l >> 4) .. + 7]: exactly
// the A operand of v_mfma_f32_16x16x32_bf16.  The producer splits each value once; every consumer fetches a fragment with
// one fully coalesced 16-byte-per-lane load and feeds it to the matrix cores as it is.  (Float32 row-major activations
// loaded fragment-wise are 64 scattered 16-byte pieces per instruction, and splitting them in the K loop costs as many
// vector instructions as the matrix instructions take cycles: that form ran the two GRU layers at 25 + 18 us per step.)
//
// Tiling.  A workgroup owns 128 sequences x one group of 16-unit blocks (dense layers also: 64 sequences x 32 units) and keeps
// NT accumulator tiles per sequence block:
//   GRU layers: the NT sums of its 16 units, [xI] r z gi_n gh_n; phase 0 (the layer input) feeds [xI] r z gi_n, phase 1
//               (the previous state) feeds r z gh_n, so r and z are sums over both phases and nothing zero is multiplied;
//   dense layers: NT consecutive 16-unit blocks, one phase.
// v_mfma_f32_16x16x32_bf16 leaves D[row 4 (l >> 4) + j][col l & 15] in lane l: a lane ends up with all sums of its (sequence, unit)
// pairs, so the gate math needs no exchange.  Eight waves: wave w works on sequence blocks of group w & 3 and on every second
// K-step (parity w >> 2), so that each SIMD holds two waves whose loads, LDS reads and matrix instructions overlap; the two
// partial sums meet in LDS at the end.
//
// Weights.  The host packs, per workgroup, the 1 KiB fragment images of its weight rows in exactly the order the K loop
// consumes them (phase, K-step, accumulator, hi | lo), so a workgroup's weights are one contiguous stream, shared by the
// eight waves through a two-slot LDS ring.
//
// Pipeline.  The work is cut into chunks of four K-steps.  In iteration c a wave requests chunk c + 2 (its own A
// fragments and its 16-byte pieces of the weight stream, all into registers: three register sets rotate), multiplies
// chunk c, and only then writes the weight pieces of chunk c + 1 -- requested one iteration earlier -- into the free LDS
// slot; one barrier per chunk.  Every request is therefore two chunks of matrix work old when its data is needed.  All
// loads are unconditional (indices past the end are clamped to valid memory and their data is ignored) and sit outside every
// branch, so that the compiler can count the outstanding requests and wait for the oldest only: a request inside one arm of
// an if / else is laid out as two skippable blocks with a path past both, and every wait behind it becomes a wait for all.
// (With the chunk's loads issued right after the barrier that the next chunk's wait sits in front of -- the direct-to-LDS
// form drains every outstanding load at each barrier -- every chunk cost one full memory round trip.)
//
// What bounds the K loop (in-kernel stamps, build with -DESTTS_WL_STAMPS and run tools/wl_stamps.py).  A chunk of a GRU layer
// is 96 KB of operands for the CU -- 1536 cycles of its 64 B / clock fetch path -- and 72 to 96 matrix instructions per
// SIMD, 1150 to 1540 cycles: the two are balanced, so they must overlap.  The waves of a SIMD therefore take request and
// multiply in opposite order (see `iteration`), and a K-step's LDS reads are issued under the previous K-step's matrix
// instructions (see `multiply`): 1.0 to 1.2 us per chunk where requests first, then multiply in every wave took 1.3 to 1.55.
// A dense layer has a quarter of the matrix work per byte and runs at the fetch rate: 0.6 us per 72 KB chunk at 128 x 16,
// which is why its tile is 64 x 32 (48 KB per chunk).  What is left of a layer's time at 832 sequences is mostly not this loop:
// a dense layer takes ~6.5 us of which the loop is 2 to 3, the rest being dispatch and the end-of-kernel release (~2), the
// first round trip (1.2), the exchange and the drain of the stores.
#include <type_traits>

#include "common.h"
#include "wavernn_cond.h"

#ifdef ESTTS_WL_STAMPS
__device__ unsigned long long wl_stamps[3][2][8];
#define WL_STAMP(i) do { if (threadIdx.x == 0 && (blockIdx.x == 0 || blockIdx.x == gridDim.x - 1)) wl_stamps[MODE][blockIdx.x ? 1 : 0][i] = wall_clock64(); } while (0)
extern "C" int estts_debug_wl_stamps(unsigned long long* out) { return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(wl_stamps), sizeof(wl_stamps)); }
#else
#define WL_STAMP(i)
#endif

namespace {

constexpr int WL_CK = 4;                       // K-steps per chunk
constexpr int WL_SLOT = WL_CK * 4 * 2 * 1024;  // bytes per LDS slot: 4 K-steps x up to 4 accumulators x (hi | lo) x 1 KiB

typedef const __attribute__((address_space(1))) void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;

template <int NAX, int MB>
struct WlSet {
    uint4 a[WL_CK / 2][MB][2];  // A: [the wave's K-steps of a chunk][16-row block][hi | lo]
    uint4 w[NAX];               // the thread's pieces of the chunk's weight stream (piece i * 512 + tid)
};

// sigmoid and tanh through v_exp_f32 / v_rcp_f32 (about 1e-7 absolute; the library forms cost ~10x the instructions, and the
// gate math of 832 x 512 cells otherwise took 1.7 us of a layer's 15)
__device__ __forceinline__ float sigmoid_(float x) {
    return __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(-1.4426950408889634f * x));
}
__device__ __forceinline__ float tanh_(float x) {
    return 1.0f - 2.0f * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(2.8853900817779268f * x));
}

// MODE 1: rnn1 (NT 5: xI r z gi_n gh_n; phase 0 feeds 0..3, phase 1 feeds 1, 2, 4)
// MODE 2: rnn2 (NT 4: r z gi_n gh_n;    phase 0 feeds 0..2, phase 1 feeds 0, 1, 3)
// MODE 0: dense, NT unit blocks, one phase
// MB: 16-sequence blocks per wave.  2: the workgroup owns 128 sequences.  1 (dense with NT 2): 64 sequences x 32 units, which
//     fetches a third fewer bytes per matrix instruction than 128 x 16 -- a dense layer's K loop runs at the rate its CU
//     fetches operands (64 KB of A and 8 KB of W per chunk against 24 matrix instructions per wave).
template <int MODE, int NT, int MB>
__global__ __launch_bounds__(512) void wavernn_layer_kernel(const estts_wavernn_layer_args p) {
    constexpr bool GRU = MODE != 0;
    constexpr int NA0 = MODE == 1 ? 4 : (MODE == 2 ? 3 : NT);   // accumulators fed by phase 0
    constexpr int NA1 = GRU ? 3 : 0;
    constexpr int NAX = NA0 > NA1 ? NA0 : NA1;
    __shared__ __attribute__((aligned(1024))) unsigned char lds[2][WL_SLOT];

    WL_STAMP(0);
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);   // (uniform, and known to the compiler as such: branches on it are scalar)
    const int mg = wave & 3;
    const int kpar = wave >> 2;
    const int i16 = lane & 15;
    const int g = lane >> 4;
    const int n_groups = (p.n_units + 16 * (GRU ? 1 : NT) - 1) / (16 * (GRU ? 1 : NT));
    const int ug = blockIdx.x % n_groups;
    const int m0 = (blockIdx.x / n_groups) * (64 * MB) + mg * (16 * MB);
    const int n_rb = (p.n_seq + 15) >> 4;

    const int nks0 = p.ks0;
    const int nks1 = GRU ? p.ks1 : 0;
    const int nc0 = (nks0 + WL_CK - 1) / WL_CK;
    const int nc1 = (nks1 + WL_CK - 1) / WL_CK;
    const int nc = nc0 + nc1;
    const unsigned char* wbase = reinterpret_cast<const unsigned char*>(p.W) + (long)ug * (nks0 * NA0 + nks1 * NA1) * 2048;

    // the wave's sequence blocks in each panel (blocks past the end re-read the last one; their rows are never stored)
    const unsigned char* a0[MB];
    const unsigned char* a1[MB];
#pragma unroll
    for (int mb = 0; mb < MB; ++mb) {
        const int rb = min((m0 >> 4) + mb, n_rb - 1);
        a0[mb] = reinterpret_cast<const unsigned char*>(p.a0) + (long)rb * p.kpitch0 * 2048 + lane * 16;
        a1[mb] = GRU ? reinterpret_cast<const unsigned char*>(p.a1) + (long)rb * p.kpitch1 * 2048 + lane * 16 : a0[mb];
    }

    // chunk c: K-steps [ks0, ks0 + nsteps) of its phase (nsteps <= 0 past the last chunk)
    auto chunk_steps = [&](int c, int& ks0, bool& ph1) {
        ph1 = c >= nc0;
        ks0 = (ph1 ? c - nc0 : c) * WL_CK;
        return min(WL_CK, (ph1 ? nks1 : nks0) - ks0);
    };
    // request everything chunk c needs.  Unconditional: a K-step past the chunk's end or a chunk past the last one reads
    // valid memory that nobody uses (the panel's last K-step; the stream's following bytes, which the host pads).
    auto request = [&](int c, WlSet<NAX, MB>& set) {
        c = min(c, nc - 1);
        int ks0;
        bool ph1;
        chunk_steps(c, ks0, ph1);
        const int klast = (ph1 ? nks1 : nks0) - 1;
#pragma unroll
        for (int q = 0; q < WL_CK / 2; ++q) {
            const int ks = min(ks0 + 2 * q + kpar, klast);
#pragma unroll
            for (int mb = 0; mb < MB; ++mb) {
                const unsigned char* src = (ph1 ? a1[mb] : a0[mb]) + (long)ks * 2048;
                set.a[q][mb][0] = *reinterpret_cast<const uint4*>(src);
                set.a[q][mb][1] = *reinterpret_cast<const uint4*>(src + 1024);
            }
        }
        const unsigned char* src = wbase + (long)(ph1 ? nks0 * NA0 + ks0 * NA1 : ks0 * NA0) * 2048 + tid * 16;
#pragma unroll
        for (int i = 0; i < NAX; ++i) set.w[i] = *reinterpret_cast<const uint4*>(src + i * 8192);
    };
    auto stage_weights = [&](int c, const WlSet<NAX, MB>& set) {
        unsigned char* dst = lds[c & 1] + tid * 16;
#pragma unroll
        for (int i = 0; i < NAX; ++i) *reinterpret_cast<uint4*>(dst + i * 8192) = set.w[i];
    };

    f32x4 acc[NT][MB];
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int mb = 0; mb < MB; ++mb) acc[t][mb] = (f32x4){0.f, 0.f, 0.f, 0.f};

    // The matrix work of chunk c for this wave: K-steps kpar and 2 + kpar of the chunk.  Per K-step three groups of matrix
    // instructions, small terms first, consecutive instructions on different accumulators: A hi x B lo, A lo x B hi, A hi x B hi.
    // B lo is dead after the first group, so the next K-step's B lo is read from LDS into its registers while the other two
    // groups run, and its B hi while that K-step's first group runs: only the chunk's first reads are waited for with nothing else to do.  Reads are unconditional (a K-step past the
    // chunk's end reads stale bytes that no instruction uses); the groups of such a K-step are skipped.
    struct Bfrag { bf16x8 hi[NAX], lo[NAX]; };
    auto multiply = [&](int c, const WlSet<NAX, MB>& set, auto phase_tag) {
        constexpr bool P1 = decltype(phase_tag)::value;
        constexpr int NACT = P1 ? NA1 : NA0;
        constexpr int first = MODE == 1 ? 1 : 0;
        int ks0;
        bool ph1;
        const int nsteps = chunk_steps(c, ks0, ph1);
        const unsigned char* blk0 = lds[c & 1] + kpar * NACT * 2048 + lane * 16;
        const unsigned char* blk1 = blk0 + 2 * NACT * 2048;
        Bfrag f;
#pragma unroll
        for (int a = 0; a < NACT; ++a) f.lo[a] = *reinterpret_cast<const bf16x8*>(blk0 + a * 2048 + 1024);
#pragma unroll
        for (int a = 0; a < NACT; ++a) f.hi[a] = *reinterpret_cast<const bf16x8*>(blk0 + a * 2048);
        __builtin_amdgcn_sched_barrier(0);   // (left alone, the scheduler sinks these reads into the first group one by one)
        auto group = [&](const bf16x8 (&a)[MB], const bf16x8 (&b)[NAX]) {
#pragma unroll
            for (int i = 0; i < NACT; ++i) {
                const int t = P1 ? (i < 2 ? first + i : first + 3) : i;
#pragma unroll
                for (int mb = 0; mb < MB; ++mb) acc[t][mb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[mb], b[i], acc[t][mb], 0, 0, 0);
            }
        };
        bf16x8 ahi[MB], alo[MB];
#pragma unroll
        for (int mb = 0; mb < MB; ++mb) {
            ahi[mb] = __builtin_bit_cast(bf16x8, set.a[0][mb][0]);
            alo[mb] = __builtin_bit_cast(bf16x8, set.a[0][mb][1]);
        }
        const bool v0 = kpar < nsteps, v1 = 2 + kpar < nsteps;
        if (v0) group(ahi, f.lo);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int a = 0; a < NACT; ++a) f.lo[a] = *reinterpret_cast<const bf16x8*>(blk1 + a * 2048 + 1024);
        __builtin_amdgcn_sched_barrier(0);
        if (v0) {
            group(alo, f.hi);
            group(ahi, f.hi);
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int a = 0; a < NACT; ++a) f.hi[a] = *reinterpret_cast<const bf16x8*>(blk1 + a * 2048);
        __builtin_amdgcn_sched_barrier(0);
        if (v1) {
#pragma unroll
            for (int mb = 0; mb < MB; ++mb) {
                ahi[mb] = __builtin_bit_cast(bf16x8, set.a[1][mb][0]);
                alo[mb] = __builtin_bit_cast(bf16x8, set.a[1][mb][1]);
            }
            group(ahi, f.lo);
            group(alo, f.hi);
            group(ahi, f.hi);
        }
    };
    auto multiply_phase = [&](int c, const WlSet<NAX, MB>& set) {
        if constexpr (GRU) {
            if (c >= nc0) multiply(c, set, std::true_type{});
            else multiply(c, set, std::false_type{});
        } else {
            multiply(c, set, std::false_type{});
        }
    };

    // iteration c with the register sets of chunks c, c + 1, c + 2
    auto iteration = [&](int c, WlSet<NAX, MB>& cur, WlSet<NAX, MB>& nxt, WlSet<NAX, MB>& far) {
        __syncthreads();                 // chunk c's weights are in slot c & 1; every wave has left chunk c - 1, whose slot is free
        // The two waves of a SIMD (K parities 0 and 1) take the two halves of the iteration in opposite order: while one
        // half of the workgroup sends its twelve 1 KiB requests down the CU's one address path -- which takes as long as the
        // chunk's matrix work -- the other half multiplies.  (With every wave requesting first, all eight queue there and no
        // one multiplies: the two times add up.)  The requests stay outside the branches, so that every path issues the same
        // loads and the compiler can count them.
        if (kpar) multiply_phase(c, cur);
        request(c + 2, far);
        if (!kpar) multiply_phase(c, cur);
        stage_weights(c + 1, nxt);       // (requested one iteration ago)
    };
    WlSet<NAX, MB> s0, s1, s2;
    request(0, s0);
    request(1, s1);
    stage_weights(0, s0);
    WL_STAMP(1);
    for (int c = 0; c < nc; c += 3) {
        iteration(c, s0, s1, s2);
        if (c + 1 >= nc) break;
        iteration(c + 1, s1, s2, s0);
        if (c + 2 >= nc) break;
        iteration(c + 2, s2, s0, s1);
    }

    WL_STAMP(2);
    // Epilogue.  The two waves of a pair swap halves: the wave with K parity q keeps sequence block q (MB 2; with MB 1: unit
    // block q) of their NT x MB tiles and adds its partner's partial sums for it, so all eight waves share the gate math, the
    // loads and the stores.  Own tile x of NX: MB 2: accumulator x, sequence block kpar; MB 1: accumulator kpar.
    static_assert(MB == 2 || (MB == 1 && MODE == 0 && NT == 2), "one sequence block per wave: dense with two unit blocks only");
    constexpr int NX = NT * MB / 2;
    const int unit0 = ug * 16 * (GRU ? 1 : NT);
    const int row_own = m0 + (MB == 2 ? 16 * kpar : 0);
    const int tile_own = MB == 2 ? 0 : kpar;     // unit block of own tile x: tile_own + x
    auto own = [&](int x, int j) {
        return MB == 2 ? (kpar == 0 ? acc[x][0][j] : acc[x][MB - 1][j]) : (kpar == 0 ? acc[0][0][j] : acc[NT - 1][0][j]);
    };
    auto given = [&](int x, int j) {
        return MB == 2 ? (kpar == 0 ? acc[x][MB - 1][j] : acc[x][0][j]) : (kpar == 0 ? acc[NT - 1][0][j] : acc[0][0][j]);
    };
    // what the epilogue reads besides the sums, requested before the exchange.  The float32 side arrays of the GRU layers
    // (previous state, residual, and their outputs) are "tile-native": [workgroup][wave][lane] float4 = the four values the
    // lane owns, so that they move as whole 16-byte pieces; only the layer itself (and its twin: rnn2 reads what rnn1 wrote
    // for the same workgroup / wave / lane) ever looks at them.
    float b[NX];
#pragma unroll
    for (int t = 0; t < NX; ++t) b[t] = GRU ? p.bias[t * p.n_units + unit0 + i16] : p.bias[unit0 + 16 * (tile_own + t) + i16];
    const long native = ((long)blockIdx.x * 8 + wave) * 64 + lane;
    float4 hp = make_float4(0.f, 0.f, 0.f, 0.f), res = hp;
    if (GRU) {
        hp = reinterpret_cast<const float4*>(p.h_prev)[native];
        if (MODE == 2) res = reinterpret_cast<const float4*>(p.x_res)[native];
    }
    __syncthreads();              // every wave is done with the weight ring: reuse it for the exchange
    float* red = reinterpret_cast<float*>(&lds[0][0]);
    {
        float* mine = red + ((mg * 2 + kpar) * NX * 4) * 64;      // what this wave hands to its partner
#pragma unroll
        for (int t = 0; t < NX; ++t)
#pragma unroll
            for (int j = 0; j < 4; ++j) mine[(t * 4 + j) * 64 + lane] = given(t, j);
    }
    __syncthreads();
    f32x4 sum[NX];
    {
        const float* theirs = red + ((mg * 2 + (kpar ^ 1)) * NX * 4) * 64;
#pragma unroll
        for (int t = 0; t < NX; ++t)
#pragma unroll
            for (int j = 0; j < 4; ++j) sum[t][j] = own(t, j) + theirs[(t * 4 + j) * 64 + lane];
    }

    WL_STAMP(3);
    // Results per lane: NOUT x 4 values, rows row_own + 4 g + j, unit unit0 + 16 (tile_own + o) + i16.
    // GRU: output 0 = the new state, output 1 = state + residual (the layer output); dense: output o = own unit block o.
    constexpr int NOUT = GRU ? 2 : NX;
    float val[NOUT][4];
    if constexpr (GRU) {
        constexpr int R = MODE == 1 ? 1 : 0;
        const float hpv[4] = {hp.x, hp.y, hp.z, hp.w}, resv[4] = {res.x, res.y, res.z, res.w};
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const float r = sigmoid_(sum[R][j] + b[R]);
            const float z = sigmoid_(sum[R + 1][j] + b[R + 1]);
            const float n = tanh_(sum[R + 2][j] + b[R + 2] + r * (sum[R + 3][j] + b[R + 3]));
            const float hn = (1.0f - z) * n + z * hpv[j];
            val[0][j] = hn;
            val[1][j] = (MODE == 1 ? sum[0][j] + b[0] : resv[j]) + hn;
        }
        reinterpret_cast<float4*>(p.h_out)[native] = make_float4(val[0][0], val[0][1], val[0][2], val[0][3]);
        if (p.out_f32) reinterpret_cast<float4*>(p.out_f32)[native] = make_float4(val[1][0], val[1][1], val[1][2], val[1][3]);
    } else {
#pragma unroll
        for (int t = 0; t < NX; ++t)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const float v = sum[t][j] + b[t];
                val[t][j] = p.act == ESTTS_ACT_RELU ? fmaxf(v, 0.0f) : v;
            }
        // float32 row-major copy (16 lanes = 64 contiguous bytes of a row)
        if (p.out_f32) {
#pragma unroll
            for (int t = 0; t < NX; ++t) {
                const int unit = unit0 + 16 * (tile_own + t) + i16;
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int row = row_own + 4 * g + j;
                    if (row < p.n_seq && unit < p.n_units) p.out_f32[(long)row * p.ldo + unit] = val[t][j];
                }
            }
        }
    }

    WL_STAMP(4);
    // panel copies: a wave's 16 sequences x 16 units go through its own 1 KiB of LDS (row-major) so that a lane can pick up
    // eight consecutive units of one sequence, split them and store 16 bytes of hi and 16 bytes of lo (lanes 0..31)
    __syncthreads();              // the exchange area is read: reuse it
    float* tr = reinterpret_cast<float*>(&lds[0][0]) + wave * 256;
    const int trow = (lane >> 1) & 15, thalf = lane & 1;
#pragma unroll
    for (int o = 0; o < NOUT; ++o) {
        uint16_t* panel = GRU ? (o == 0 ? p.h_panel : p.out_panel) : p.out_panel;
        const int kpitch = GRU ? (o == 0 ? p.h_kpitch : p.out_kpitch) : p.out_kpitch;
        if (!panel) continue;
#pragma unroll
        for (int j = 0; j < 4; ++j) tr[(4 * g + j) * 16 + i16] = val[o][j];
        const float4 v0 = *reinterpret_cast<const float4*>(tr + trow * 16 + thalf * 8);
        const float4 v1 = *reinterpret_cast<const float4*>(tr + trow * 16 + thalf * 8 + 4);
        bf16x4 h0, l0, h1, l1;
        split_bf16x4(v0, h0, l0);
        split_bf16x4(v1, h1, l1);
        const bf16x8 hi = {h0[0], h0[1], h0[2], h0[3], h1[0], h1[1], h1[2], h1[3]};
        const bf16x8 lo = {l0[0], l0[1], l0[2], l0[3], l1[0], l1[1], l1[2], l1[3]};
        const int row = row_own + trow;
        const int col = unit0 + (GRU ? 0 : 16 * (tile_own + o)) + 8 * thalf;
        if (lane < 32 && row < n_rb * 16 && col < p.n_units) {   // whole blocks are kept defined: rows up to the end of the last block
            unsigned char* dst = reinterpret_cast<unsigned char*>(panel) + ((long)(row >> 4) * kpitch + (col >> 5)) * 2048 +
                                 ((row & 15) + 16 * ((col & 31) >> 3)) * 16;
            *reinterpret_cast<bf16x8*>(dst) = hi;
            *reinterpret_cast<bf16x8*>(dst + 1024) = lo;
        }
    }
    WL_STAMP(5);
#ifdef ESTTS_WL_STAMPS
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    WL_STAMP(6);
#endif
}

// ------------------------------------------------------------------------------------------------------
// Conditioning of one step, written as panels: one thread per (sequence, group of eight columns).  Groups 1..15 are the step
// input behind its first group: [x_prev 0 0 0 0 0 0 0 | mel (FIR over frames) | aux slice 1 | zeros] (128 columns; group 0
// belongs to the sampler, which writes x_prev there); the next 3 d / 8 groups are the aux slices that follow the layer
// outputs in the inputs of rnn2, fc1 and fc2 (K-step `aux_ks` of their panels).
__global__ __launch_bounds__(256) void wavernn_cond_kernel(const estts_wavernn_cond_args p) {
    __shared__ int s_step;
    if (threadIdx.x == 0) s_step = __hip_atomic_load(p.step, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __syncthreads();
    const int step_now = s_step;
    // the last workgroup to arrive has seen every workgroup read step[0]; it advances the counter (this kernel's own)
    if (threadIdx.x == 0) {
        const int t = __hip_atomic_fetch_add(p.done, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (t == (int)gridDim.x - 1) {
            __hip_atomic_store(p.done, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(p.step, step_now + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
    const int groups = 15 + 3 * (p.d >> 3);
    const long idx = (long)blockIdx.x * 256 + threadIdx.x;
    const int n_rows = ((p.n_seq + 15) >> 4) << 4;
    if (idx >= (long)n_rows * groups) return;
    const int b = (int)(idx / groups);
    wavernn_cond_group(p, b, (int)(idx - (long)b * groups) + 1, step_now);
}

template <int MODE, int NT, int MB = 2>
int launch_layer(const estts_wavernn_layer_args& a, hipStream_t st) {
    const int per = 16 * (MODE ? 1 : NT);
    const int grid = ((a.n_seq + 64 * MB - 1) / (64 * MB)) * ((a.n_units + per - 1) / per);
    ESTTS_LAUNCH((wavernn_layer_kernel<MODE, NT, MB>), dim3(grid), dim3(512), 0, st, a);
    ESTTS_CHECK_LAUNCH("estts_wavernn_layer");
    return 0;
}

}  // namespace

extern "C" int estts_wavernn_layer(const estts_wavernn_layer_args* args, void* stream) {
    ESTTS_CHECK_ARG(args != nullptr, "estts_wavernn_layer: null args");
    const estts_wavernn_layer_args& a = *args;
    ESTTS_CHECK_ARG(a.a0 && a.W && a.bias, "estts_wavernn_layer: null pointer");
    ESTTS_CHECK_ARG(a.mode >= 0 && a.mode <= 2, "estts_wavernn_layer: mode must be 0 (dense), 1 (rnn1) or 2 (rnn2)");
    ESTTS_CHECK_ARG(a.n_seq > 0 && a.n_units > 0 && a.ks0 > 0 && a.kpitch0 >= a.ks0, "estts_wavernn_layer: bad sizes");
    ESTTS_CHECK_ARG((((uintptr_t)a.a0 | (uintptr_t)a.a1 | (uintptr_t)a.W | (uintptr_t)a.out_panel | (uintptr_t)a.h_panel) & 15) == 0,
                    "estts_wavernn_layer: panels and W must be 16-byte aligned");
    ESTTS_CHECK_ARG(!a.out_f32 || a.mode != 0 || a.ldo >= a.n_units, "estts_wavernn_layer: ldo too small");
    ESTTS_CHECK_ARG(!a.out_panel || a.out_kpitch * 32 >= a.n_units, "estts_wavernn_layer: output panel too narrow");
    hipStream_t st = (hipStream_t)stream;
    if (a.mode != 0) {
        ESTTS_CHECK_ARG((a.n_units & 31) == 0, "estts_wavernn_layer: a GRU needs a multiple of 32 units");
        ESTTS_CHECK_ARG(a.a1 && a.ks1 * 32 == a.n_units && a.kpitch1 >= a.ks1, "estts_wavernn_layer: the state panel must have n_units columns");
        ESTTS_CHECK_ARG(a.h_prev && a.h_out && a.h_prev != a.h_out && (((uintptr_t)a.h_prev | (uintptr_t)a.h_out | (uintptr_t)a.x_res | (uintptr_t)a.out_f32) & 15) == 0,
                        "estts_wavernn_layer: a GRU needs distinct, 16-byte aligned float32 state arrays");
        ESTTS_CHECK_ARG(a.h_panel && a.h_kpitch >= a.ks1 && a.h_panel != a.a1, "estts_wavernn_layer: a GRU needs a new state panel distinct from the old one");
        ESTTS_CHECK_ARG(a.mode == 1 || a.x_res, "estts_wavernn_layer: rnn2 needs the residual input");
        return a.mode == 1 ? launch_layer<1, 5>(a, st) : launch_layer<2, 4>(a, st);
    }
    ESTTS_CHECK_ARG(a.out_f32 || a.out_panel, "estts_wavernn_layer: no output");
    ESTTS_CHECK_ARG(a.act == ESTTS_ACT_NONE || a.act == ESTTS_ACT_RELU, "estts_wavernn_layer: activation must be none or relu");
    ESTTS_CHECK_ARG(a.seq_tile == 0 || a.seq_tile == 128 || (a.seq_tile == 64 && a.unit_blocks == 2),
                    "estts_wavernn_layer: seq_tile must be 0 / 128, or 64 with unit_blocks 2");
    if (a.seq_tile == 64) return launch_layer<0, 2, 1>(a, st);
    switch (a.unit_blocks) {
        case 1: return launch_layer<0, 1>(a, st);
        case 2: return launch_layer<0, 2>(a, st);
        case 4: return launch_layer<0, 4>(a, st);
        default: estts_set_error("estts_wavernn_layer: unit_blocks must be 1, 2 or 4"); return -1;
    }
}

extern "C" int estts_wavernn_cond(const estts_wavernn_cond_args* args, void* stream) {
    ESTTS_CHECK_ARG(args != nullptr, "estts_wavernn_cond: null args");
    const estts_wavernn_cond_args& a = *args;
    ESTTS_CHECK_ARG(a.mel && a.aux && a.fir && a.seq_info && a.step && a.done && a.xin_panel && a.a2_panel && a.a3_panel && a.a4_panel,
                    "estts_wavernn_cond: null pointer");
    ESTTS_CHECK_ARG(((uintptr_t)a.seq_info & 15) == 0, "estts_wavernn_cond: seq_info must be 16-byte aligned");
    ESTTS_CHECK_ARG(a.n_seq > 0 && a.feat > 0 && a.d > 0 && (a.d & 7) == 0 && a.d <= 32 && a.hop > 0 && a.ntaps > 0,
                    "estts_wavernn_cond: bad sizes (aux slices must be a multiple of 8 and at most 32 wide)");
    ESTTS_CHECK_ARG(8 + a.feat + a.d <= 128, "estts_wavernn_cond: the step input must fit 128 columns");
    ESTTS_CHECK_ARG(a.aux_kpitch > a.aux_ks && a.aux_ks >= 0, "estts_wavernn_cond: aux K-step outside the panels");
    const long threads = (long)(((a.n_seq + 15) >> 4) << 4) * (15 + 3 * (a.d >> 3));
    ESTTS_LAUNCH(wavernn_cond_kernel, dim3((unsigned)((threads + 255) / 256)), dim3(256), 0, (hipStream_t)stream, a);
    ESTTS_CHECK_LAUNCH("estts_wavernn_cond");
    return 0;
}
