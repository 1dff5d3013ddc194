// Fused single-head self-attention core for the U-Net's 16x16 attention blocks (N = 256 tokens, C = 256 channels), fp32:
//     out[i][:] = (sum_j softmax_j(q_i . k_j * C^-1/2) v_j  + bias [+ res[i][:]]) * out_scale
// (the value bias moves to the output because softmax rows sum to one; in the planner's folded form -- Plan::attn -- k is the GroupNorm
// output itself, v already carries the output projection, and the block's residual, 1/sqrt(2) and the fused GroupNorm statistics of the
// result live in this epilogue)
// (cld_jax/models/layerspp.py:74-78: the two einsums and the softmax of AttnBlockpp).  The unfused path is three launches -- QK^T,
// softmax_rows_kernel over a B N N scratch in HBM, PV -- this kernel keeps a 64-query block of the scores in registers / LDS:
//   phase A  S = Q_blk K^T        64 x 256 x 256, fp32 MFMA 32x32x2; 8 wave64 as 2 (rows) x 4 (keys), two accumulators per wave
//   softmax  row maxima and sums: in-lane over the wave's two tiles, across the 32 column lanes by shuffles, across the four key waves
//            through LDS; P = exp2((S - max) C^-1/2 log2 e) / sum written to LDS row-major (the A operand layout of phase B)
//   phase B  O = P V              64 x 256 x 256, V read as V^T[c][token] (what the engine's v projection emits): k-contiguous
// Both phases stream their B operand (K rows, V^T rows) through two LDS stages of 32-deep chunks with a two-deep register prefetch,
// staging pieces and next-step fragment reads hand-placed between the MFMA pairs as in the convolution kernels; rows are
// padded to 36 floats (260 for P), an odd number of 16-byte slots, so fragment ds_read_b128 and staging ds_write_b128 are conflict-free.
// One workgroup per (image, 64-query block): 4 B workgroups, the four of an image on one XCD; LDS 139 KB, one workgroup per CU.
#include <cstdio>
#include <cstdlib>

#include "../common.h"
#include "../options.h"

#ifndef AT_OUT_NT
#define AT_OUT_NT 0  // nontemporal output stores (A/B, round 5)
#endif

namespace gd {

typedef float af32x16 __attribute__((ext_vector_type(16)));
typedef float af32x4 __attribute__((ext_vector_type(4)));

constexpr int AT_N = 256, AT_C = 256;   // tokens, channels (fixed: the 16x16 level of the benchmarked family)
constexpr int AT_LD = 36;               // LDS row stride of a 32-deep chunk
constexpr int AT_PLD = 260;             // LDS row stride of P
constexpr int AT_P = 64 * AT_PLD;       // floats of the P region (also the start of the phase-B stages)
constexpr int AT_KST = 256 * AT_LD;     // one B-operand stage (256 rows)
constexpr int AT_QST = 64 * AT_LD;      // one Q stage
constexpr int AT_RED = AT_P + 2 * AT_KST;  // softmax scratch [64 rows][4 waves]
constexpr int AT_SMEM = AT_RED + 64 * 4;

struct AttnParams {
    const float *q;   // q[(img * N + token) * ld_q + c]
    const float *k;   // k[(img * N + token) * ld_k + c]
    int ld_q, ld_k;
    const float *vT;  // vT[(img * C + c) * N + token]
    const float *bias;  // [C] or null
    const float *res;   // res[(img * N + token) * ld_res + c] or null
    int ld_res;
    float out_scale;
    float *qstats;    // qstats[img][8 slices of 32 tokens][C / 4] = (sum, sum of squares) of out (kernels/qstats.h) or null
    float *out;       // out[(img * N + token) * ld_o + c]
    int ld_o;
    float scale_log2e;  // C^-1/2 * log2(e)
    int batch;
};

__global__ __launch_bounds__(512, 2) void attn256_f32_kernel(const AttnParams p) {
    __shared__ __attribute__((aligned(16))) float smem[AT_SMEM];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 2, wn = wave & 3;
    const int li = lane & 31, lh = lane >> 5;
    // XCD-aware workgroup -> (image, query block): consecutive workgroup ids go round-robin over the 8 XCDs, each with an L2 of its own, and
    // the four query blocks of an image stream the same K rows and V^T.  With id -> (xcd = id % 8, slot = id / 8, image = (slot / 4) * 8
    // + xcd, block = slot % 4) the four land on ONE XCD: K / V^T come from HBM once per image instead of once per query block
    // (round 2's dim3(4, batch) grid: 671 MB per launch against 268 MB algorithmic).
    const int slot = blockIdx.x >> 3;
    const int img = (slot >> 2) * 8 + (blockIdx.x & 7), q0 = (slot & 3) * 64;
    if (img >= p.batch) return;

    // staging roles: row (tid >> 3) [+ 64 j], 16-byte piece (tid & 7) of the 32-deep chunk
    const int srow = tid >> 3, spc = tid & 7;
    const float *gq = p.q + ((size_t)img * AT_N + q0 + srow) * p.ld_q + spc * 4;
    const float *gk = p.k + ((size_t)img * AT_N + srow) * p.ld_k + spc * 4;
    const float *gv = p.vT + ((size_t)img * AT_C + srow) * AT_N + spc * 4;
    const int s_lds = srow * AT_LD + spc * 4;

    // two register sets by chunk parity: a chunk is requested two iterations (of ~1 us) before it is written to LDS
    af32x4 rq[2], rb[2][4];
    // (piece j = 0..3: K / V^T rows srow + 64 j; piece 4: the Q row, phase A only)
    auto load_a1 = [&](int kc, int j) __attribute__((always_inline)) {  // phase A chunk kc
        if (j == 4) rq[kc & 1] = *reinterpret_cast<const af32x4 *>(gq + kc * 32);
        else rb[kc & 1][j] = *reinterpret_cast<const af32x4 *>(gk + (size_t)j * 64 * p.ld_k + kc * 32);
    };
    auto store_a1 = [&](int kc, int j) __attribute__((always_inline)) {  // ... into LDS stage kc & 1
        float *qs = smem + (kc & 1) * (AT_QST + AT_KST), *ks = qs + AT_QST;
        if (j == 4) *reinterpret_cast<af32x4 *>(qs + s_lds) = rq[kc & 1];
        else *reinterpret_cast<af32x4 *>(ks + s_lds + j * 64 * AT_LD) = rb[kc & 1][j];
    };
    auto load_b1 = [&](int kc, int j) __attribute__((always_inline)) {  // phase B chunk kc: V^T pieces (rows = channels)
        rb[kc & 1][j] = *reinterpret_cast<const af32x4 *>(gv + (size_t)j * 64 * AT_N + kc * 32);
    };
    auto store_b1 = [&](int kc, int j) __attribute__((always_inline)) {
        *reinterpret_cast<af32x4 *>(smem + AT_P + (kc & 1) * AT_KST + s_lds + j * 64 * AT_LD) = rb[kc & 1][j];
    };
    auto load_a = [&](int kc) __attribute__((always_inline)) {
#pragma unroll
        for (int j = 0; j < 5; ++j) load_a1(kc, j);
    };
    auto store_a = [&](int kc) __attribute__((always_inline)) {
#pragma unroll
        for (int j = 0; j < 5; ++j) store_a1(kc, j);
    };
    auto load_b = [&](int kc) __attribute__((always_inline)) {
#pragma unroll
        for (int j = 0; j < 4; ++j) load_b1(kc, j);
    };
    auto store_b = [&](int kc) __attribute__((always_inline)) {
#pragma unroll
        for (int j = 0; j < 4; ++j) store_b1(kc, j);
    };

    af32x16 acc[2];
#pragma unroll
    for (int c = 0; c < 2; ++c)
#pragma unroll
        for (int v = 0; v < 16; ++v) acc[c][v] = 0.f;

    // ---- phase A: S = Q_blk K^T over 8 chunks of 32 channels ----
    const int a_frag = (wm * 32 + li) * AT_LD + lh * 4;           // + kk * 8
    const int b_frag = (wn * 64 + li) * AT_LD + lh * 4;           // + c * 32 rows + kk * 8
    load_a(0);
    load_a(1);
    store_a(0);
    load_a(2);
    __syncthreads();
#pragma unroll
    for (int kc = 0; kc < 8; ++kc) {  // (unrolled: the register set of a chunk is a compile-time choice)
        const float *qs = smem + (kc & 1) * (AT_QST + AT_KST), *ks = qs + AT_QST;
        af32x4 a[2], b0[2], b1[2];
        a[0] = *reinterpret_cast<const af32x4 *>(qs + a_frag);
        b0[0] = *reinterpret_cast<const af32x4 *>(ks + b_frag);
        b1[0] = *reinterpret_cast<const af32x4 *>(ks + b_frag + 32 * AT_LD);
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) {
#pragma unroll
            for (int jj = 0; jj < 4; ++jj) {
                acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[kk & 1][jj], b0[kk & 1][jj], acc[0], 0, 0, 0);
                acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[kk & 1][jj], b1[kk & 1][jj], acc[1], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
                const int gap = kk * 4 + jj;
                if (jj == 0 && kk < 3) {  // fragments of the next 8-deep step, one step ahead of their use
                    a[(kk + 1) & 1] = *reinterpret_cast<const af32x4 *>(qs + a_frag + (kk + 1) * 8);
                    b0[(kk + 1) & 1] = *reinterpret_cast<const af32x4 *>(ks + b_frag + (kk + 1) * 8);
                    b1[(kk + 1) & 1] = *reinterpret_cast<const af32x4 *>(ks + b_frag + 32 * AT_LD + (kk + 1) * 8);
                }
                // staging in the shadow of the MFMA pair: chunk kc+1 registers -> the other stage, chunk kc+3 requested into them
                if (gap >= 1 && gap <= 5 && kc + 1 < 8) store_a1(kc + 1, gap - 1);
                if (gap >= 6 && gap <= 10 && kc + 3 < 8) load_a1(kc + 3, gap - 6);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        __syncthreads();
    }
    load_b(0);  // the first V^T chunks travel while the softmax runs
    load_b(1);

    // ---- softmax over the 256 keys of every row.  Accumulator element v of lane (li, lh): row 8 (v >> 2) + (v & 3) + 4 lh of the
    // wave's 32, column li of tile c ----
    float *red = smem + AT_RED;
    float mx[16];
#pragma unroll
    for (int v = 0; v < 16; ++v) {
        float m = fmaxf(acc[0][v], acc[1][v]);
#pragma unroll
        for (int o = 1; o < 32; o <<= 1) m = fmaxf(m, __shfl_xor(m, o));
        mx[v] = m;
    }
    if (li == 0) {
#pragma unroll
        for (int v = 0; v < 16; ++v) red[(wm * 32 + 8 * (v >> 2) + (v & 3) + 4 * lh) * 4 + wn] = mx[v];
    }
    __syncthreads();
#pragma unroll
    for (int v = 0; v < 16; ++v) {
        const af32x4 r = *reinterpret_cast<const af32x4 *>(red + (wm * 32 + 8 * (v >> 2) + (v & 3) + 4 * lh) * 4);
        mx[v] = fmaxf(fmaxf(r[0], r[1]), fmaxf(r[2], r[3]));
    }
    __syncthreads();  // (red is rewritten with the sums)
    float sm[16];
#pragma unroll
    for (int v = 0; v < 16; ++v) {
        acc[0][v] = __builtin_amdgcn_exp2f((acc[0][v] - mx[v]) * p.scale_log2e);
        acc[1][v] = __builtin_amdgcn_exp2f((acc[1][v] - mx[v]) * p.scale_log2e);
        float s = acc[0][v] + acc[1][v];
#pragma unroll
        for (int o = 1; o < 32; o <<= 1) s += __shfl_xor(s, o);
        sm[v] = s;
    }
    if (li == 0) {
#pragma unroll
        for (int v = 0; v < 16; ++v) red[(wm * 32 + 8 * (v >> 2) + (v & 3) + 4 * lh) * 4 + wn] = sm[v];
    }
    __syncthreads();
#pragma unroll
    for (int v = 0; v < 16; ++v) {
        const af32x4 r = *reinterpret_cast<const af32x4 *>(red + (wm * 32 + 8 * (v >> 2) + (v & 3) + 4 * lh) * 4);
        const float inv = 1.0f / ((r[0] + r[1]) + (r[2] + r[3]));
        const int row = wm * 32 + 8 * (v >> 2) + (v & 3) + 4 * lh;
        smem[row * AT_PLD + wn * 64 + li] = acc[0][v] * inv;
        smem[row * AT_PLD + wn * 64 + 32 + li] = acc[1][v] * inv;
    }
    store_b(0);
    load_b(2);
    __syncthreads();

    // the block's residual tile (32 values per lane) and the bias travel during phase B: one workgroup per CU leaves the registers for it
    float rv[2][16], bvv[2] = {0.f, 0.f};
#pragma unroll
    for (int c = 0; c < 2; ++c) {
        const int col = wn * 64 + c * 32 + li;
        if (p.bias) bvv[c] = p.bias[col];
        if (p.res) {
#pragma unroll
            for (int v = 0; v < 16; ++v) {
                const int row = wm * 32 + 8 * (v >> 2) + (v & 3) + 4 * lh;
                rv[c][v] = p.res[((size_t)img * AT_N + q0 + row) * p.ld_res + col];
            }
        }
    }

    // ---- phase B: O = P V over 8 chunks of 32 keys ----
#pragma unroll
    for (int c = 0; c < 2; ++c)
#pragma unroll
        for (int v = 0; v < 16; ++v) acc[c][v] = 0.f;
    const int p_frag = (wm * 32 + li) * AT_PLD + lh * 4;  // + kc * 32 + kk * 8
#pragma unroll
    for (int kc = 0; kc < 8; ++kc) {
        const float *vs = smem + AT_P + (kc & 1) * AT_KST;
        af32x4 a[2], b0[2], b1[2];
        a[0] = *reinterpret_cast<const af32x4 *>(smem + p_frag + kc * 32);
        b0[0] = *reinterpret_cast<const af32x4 *>(vs + b_frag);
        b1[0] = *reinterpret_cast<const af32x4 *>(vs + b_frag + 32 * AT_LD);
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) {
#pragma unroll
            for (int jj = 0; jj < 4; ++jj) {
                acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[kk & 1][jj], b0[kk & 1][jj], acc[0], 0, 0, 0);
                acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[kk & 1][jj], b1[kk & 1][jj], acc[1], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
                const int gap = kk * 4 + jj;
                if (jj == 0 && kk < 3) {
                    a[(kk + 1) & 1] = *reinterpret_cast<const af32x4 *>(smem + p_frag + kc * 32 + (kk + 1) * 8);
                    b0[(kk + 1) & 1] = *reinterpret_cast<const af32x4 *>(vs + b_frag + (kk + 1) * 8);
                    b1[(kk + 1) & 1] = *reinterpret_cast<const af32x4 *>(vs + b_frag + 32 * AT_LD + (kk + 1) * 8);
                }
                if (gap >= 1 && gap <= 4 && kc + 1 < 8) store_b1(kc + 1, gap - 1);
                if (gap >= 6 && gap <= 9 && kc + 3 < 8) load_b1(kc + 3, gap - 6);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        __syncthreads();
    }

    // ---- epilogue: row = query, column = channel.  Element v of lane (li, lh): row 8 (v >> 2) + (v & 3) + 4 lh of the wave's 32 ----
#pragma unroll
    for (int c = 0; c < 2; ++c) {
        const int col = wn * 64 + c * 32 + li;
        const float bv = bvv[c];
        float s = 0.f, q = 0.f;
#pragma unroll
        for (int v = 0; v < 16; ++v) {
            const int row = wm * 32 + 8 * (v >> 2) + (v & 3) + 4 * lh;
            float val = acc[c][v] + bv;
            if (p.res) val += rv[c][v];
            val *= p.out_scale;
            if (AT_OUT_NT) __builtin_nontemporal_store(val, &p.out[((size_t)img * AT_N + q0 + row) * p.ld_o + col]);
            else p.out[((size_t)img * AT_N + q0 + row) * p.ld_o + col] = val;
            s += val;
            q += val * val;
        }
        if (p.qstats) {  // the wave's 32 rows are one slice; four consecutive columns = four consecutive lanes, rows + 4 in the other half-wave
            s += __shfl_xor(s, 1); q += __shfl_xor(q, 1);
            s += __shfl_xor(s, 2); q += __shfl_xor(q, 2);
            s += __shfl_xor(s, 32); q += __shfl_xor(q, 32);
            if ((li & 3) == 0 && lh == 0)
                reinterpret_cast<float2 *>(p.qstats)[((size_t)img * (AT_N / 32) + ((q0 + wm * 32) >> 5)) * (AT_C / 4) + (col >> 2)] = make_float2(s, q);
        }
    }
}

// ---- the 32-query form: two workgroups per CU ---------------------------------------------------------------------------------------
// The 64-query kernel above holds 139 KB of LDS: one workgroup per CU, so its softmax (every wave of the workgroup waits on the row maxima
// and sums), its first loads and its stores -- a quarter of a workgroup's 44 us -- run with the matrix pipe idle.  Here a workgroup is
// 32 queries on FOUR waves (each wave: all 32 rows x 64 keys / channels, two accumulators) and the B operand streams through 16-deep chunks:
// P 32 x 260 floats + two 256 x 20 stages = 75 KB, so TWO workgroups share a CU and one's softmax / prologue / epilogue runs under the
// other's MFMAs.  Same arithmetic per output element as the 64-query form (same k order inside a row): bit-identical.
// Measured (profiles/r04_notes.md): 100.3 against 99.0 TFLOP/s at B = 256 -- the launcher uses it where the 64-query grid cannot fill the chip.
constexpr int A2_Q = 32;                    // queries per workgroup
constexpr int A2_D = 16;                    // chunk depth
constexpr int A2_LD = 20;                   // LDS row stride of a 16-deep chunk (5 16-byte units: odd)
constexpr int A2_P = A2_Q * AT_PLD;         // floats of the P region; phase A keeps its two Q stages inside it (P is written after phase A)
constexpr int A2_KST = 256 * A2_LD;         // one B-operand stage
constexpr int A2_QST = A2_Q * A2_LD;        // one Q stage
constexpr int A2_RED = A2_P + 2 * A2_KST;   // softmax scratch [32 rows][4 waves]
constexpr int A2_SMEM = A2_RED + A2_Q * 4;  // 18 688 floats = 74 752 bytes

__global__ __launch_bounds__(256, 2) void attn256_q32_f32_kernel(const AttnParams p) {
    __shared__ __attribute__((aligned(16))) float smem[A2_SMEM];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wn = __builtin_amdgcn_readfirstlane(tid >> 6);  // the wave's block of 64 keys (phase A) / channels (phase B)
    const int li = lane & 31, lh = lane >> 5;
    // XCD-aware workgroup -> (image, query block): the eight query blocks of an image on ONE XCD (they stream the same K rows and V^T)
    const int slot = blockIdx.x >> 3;
    const int img = (slot >> 3) * 8 + (blockIdx.x & 7), q0 = (slot & 7) * A2_Q;
    if (img >= p.batch) return;

    // staging roles: B-operand rows (tid >> 2) + 64 j, 16-byte piece (tid & 3) of the 16-deep chunk; the first 128 threads also a Q piece
    const int srow = tid >> 2, spc = tid & 3;
    const bool has_q = tid < 4 * A2_Q;
    const float *gq = p.q + ((size_t)img * AT_N + q0 + (has_q ? srow : 0)) * p.ld_q + spc * 4;
    const float *gk = p.k + ((size_t)img * AT_N + srow) * p.ld_k + spc * 4;
    const float *gv = p.vT + ((size_t)img * AT_C + srow) * AT_N + spc * 4;
    const int s_lds = srow * A2_LD + spc * 4;

    af32x4 rq[2], rb[2][4];  // two register sets by chunk parity: a chunk is requested two iterations before it is written to LDS
    auto load_a1 = [&](int kc, int j) __attribute__((always_inline)) {
        if (j == 4) { if (has_q) rq[kc & 1] = *reinterpret_cast<const af32x4 *>(gq + kc * A2_D); }
        else rb[kc & 1][j] = *reinterpret_cast<const af32x4 *>(gk + (size_t)j * 64 * p.ld_k + kc * A2_D);
    };
    auto store_a1 = [&](int kc, int j) __attribute__((always_inline)) {
        if (j == 4) { if (has_q) *reinterpret_cast<af32x4 *>(smem + (kc & 1) * A2_QST + s_lds) = rq[kc & 1]; }
        else *reinterpret_cast<af32x4 *>(smem + A2_P + (kc & 1) * A2_KST + s_lds + j * 64 * A2_LD) = rb[kc & 1][j];
    };
    auto load_b1 = [&](int kc, int j) __attribute__((always_inline)) {
        rb[kc & 1][j] = *reinterpret_cast<const af32x4 *>(gv + (size_t)j * 64 * AT_N + kc * A2_D);
    };
    auto store_b1 = [&](int kc, int j) __attribute__((always_inline)) {
        *reinterpret_cast<af32x4 *>(smem + A2_P + (kc & 1) * A2_KST + s_lds + j * 64 * A2_LD) = rb[kc & 1][j];
    };

    af32x16 acc[2];
#pragma unroll
    for (int c = 0; c < 2; ++c)
#pragma unroll
        for (int v = 0; v < 16; ++v) acc[c][v] = 0.f;

    // ---- phase A: S = Q_blk K^T over 16 chunks of 16 channels; 16 MFMAs per wave between barriers, in 8 gaps of two ----
    const int a_frag = li * A2_LD + lh * 4;                // + kk * 8
    const int b_frag = (wn * 64 + li) * A2_LD + lh * 4;    // + c * 32 rows + kk * 8
    constexpr int NKC = AT_C / A2_D;
#pragma unroll
    for (int j = 0; j < 5; ++j) load_a1(0, j);
#pragma unroll
    for (int j = 0; j < 5; ++j) load_a1(1, j);
#pragma unroll
    for (int j = 0; j < 5; ++j) store_a1(0, j);
#pragma unroll
    for (int j = 0; j < 5; ++j) load_a1(2, j);
    __syncthreads();
#pragma unroll
    for (int kc = 0; kc < NKC; ++kc) {  // (unrolled: the register set of a chunk is a compile-time choice)
        const float *qs = smem + (kc & 1) * A2_QST, *ks = smem + A2_P + (kc & 1) * A2_KST;
        af32x4 a[2], b0[2], b1[2];
        a[0] = *reinterpret_cast<const af32x4 *>(qs + a_frag);
        b0[0] = *reinterpret_cast<const af32x4 *>(ks + b_frag);
        b1[0] = *reinterpret_cast<const af32x4 *>(ks + b_frag + 32 * A2_LD);
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) {
#pragma unroll
            for (int jj = 0; jj < 4; ++jj) {
                acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[kk][jj], b0[kk][jj], acc[0], 0, 0, 0);
                acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[kk][jj], b1[kk][jj], acc[1], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
                const int gap = kk * 4 + jj;
                if (gap == 0) {  // the chunk's second 8-deep step
                    a[1] = *reinterpret_cast<const af32x4 *>(qs + a_frag + 8);
                    b0[1] = *reinterpret_cast<const af32x4 *>(ks + b_frag + 8);
                    b1[1] = *reinterpret_cast<const af32x4 *>(ks + b_frag + 32 * A2_LD + 8);
                }
                // staging in the shadow of the MFMA pairs: chunk kc+1 registers -> the other stage, chunk kc+3 requested into them
                if (gap >= 1 && gap <= 4 && kc + 1 < NKC) store_a1(kc + 1, gap - 1);
                if (gap == 4 && kc + 1 < NKC) store_a1(kc + 1, 4);
                if (gap >= 5 && gap <= 7 && kc + 3 < NKC) load_a1(kc + 3, gap - 5);
                if (gap == 7 && kc + 3 < NKC) { load_a1(kc + 3, 3); load_a1(kc + 3, 4); }
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        __syncthreads();
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) load_b1(0, j);  // the first V^T chunks travel while the softmax runs
#pragma unroll
    for (int j = 0; j < 4; ++j) load_b1(1, j);

    // ---- softmax over the 256 keys of every row.  Accumulator element v of lane (li, lh): row 8 (v >> 2) + (v & 3) + 4 lh, column li of
    // tile c ----
    float *red = smem + A2_RED;
    float mx[16];
#pragma unroll
    for (int v = 0; v < 16; ++v) {
        float m = fmaxf(acc[0][v], acc[1][v]);
#pragma unroll
        for (int o = 1; o < 32; o <<= 1) m = fmaxf(m, __shfl_xor(m, o));
        mx[v] = m;
    }
    if (li == 0) {
#pragma unroll
        for (int v = 0; v < 16; ++v) red[(8 * (v >> 2) + (v & 3) + 4 * lh) * 4 + wn] = mx[v];
    }
    __syncthreads();
#pragma unroll
    for (int v = 0; v < 16; ++v) {
        const af32x4 r = *reinterpret_cast<const af32x4 *>(red + (8 * (v >> 2) + (v & 3) + 4 * lh) * 4);
        mx[v] = fmaxf(fmaxf(r[0], r[1]), fmaxf(r[2], r[3]));
    }
    __syncthreads();  // (red is rewritten with the sums)
    float sm[16];
#pragma unroll
    for (int v = 0; v < 16; ++v) {
        acc[0][v] = __builtin_amdgcn_exp2f((acc[0][v] - mx[v]) * p.scale_log2e);
        acc[1][v] = __builtin_amdgcn_exp2f((acc[1][v] - mx[v]) * p.scale_log2e);
        float s = acc[0][v] + acc[1][v];
#pragma unroll
        for (int o = 1; o < 32; o <<= 1) s += __shfl_xor(s, o);
        sm[v] = s;
    }
    if (li == 0) {
#pragma unroll
        for (int v = 0; v < 16; ++v) red[(8 * (v >> 2) + (v & 3) + 4 * lh) * 4 + wn] = sm[v];
    }
    __syncthreads();
#pragma unroll
    for (int v = 0; v < 16; ++v) {
        const af32x4 r = *reinterpret_cast<const af32x4 *>(red + (8 * (v >> 2) + (v & 3) + 4 * lh) * 4);
        const float inv = 1.0f / ((r[0] + r[1]) + (r[2] + r[3]));
        const int row = 8 * (v >> 2) + (v & 3) + 4 * lh;
        smem[row * AT_PLD + wn * 64 + li] = acc[0][v] * inv;       // (the Q stages under it were last read before the softmax's barriers)
        smem[row * AT_PLD + wn * 64 + 32 + li] = acc[1][v] * inv;
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) store_b1(0, j);
#pragma unroll
    for (int j = 0; j < 4; ++j) load_b1(2, j);
    __syncthreads();

    // the block's residual tile (32 values per lane) and the bias travel during phase B
    float rv[2][16], bvv[2] = {0.f, 0.f};
#pragma unroll
    for (int c = 0; c < 2; ++c) {
        const int col = wn * 64 + c * 32 + li;
        if (p.bias) bvv[c] = p.bias[col];
        if (p.res) {
#pragma unroll
            for (int v = 0; v < 16; ++v) {
                const int row = 8 * (v >> 2) + (v & 3) + 4 * lh;
                rv[c][v] = p.res[((size_t)img * AT_N + q0 + row) * p.ld_res + col];
            }
        }
    }

    // ---- phase B: O = P V over 16 chunks of 16 keys ----
#pragma unroll
    for (int c = 0; c < 2; ++c)
#pragma unroll
        for (int v = 0; v < 16; ++v) acc[c][v] = 0.f;
    const int p_frag = li * AT_PLD + lh * 4;  // + kc * 16 + kk * 8
#pragma unroll
    for (int kc = 0; kc < NKC; ++kc) {
        const float *vs = smem + A2_P + (kc & 1) * A2_KST;
        af32x4 a[2], b0[2], b1[2];
        a[0] = *reinterpret_cast<const af32x4 *>(smem + p_frag + kc * A2_D);
        b0[0] = *reinterpret_cast<const af32x4 *>(vs + b_frag);
        b1[0] = *reinterpret_cast<const af32x4 *>(vs + b_frag + 32 * A2_LD);
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) {
#pragma unroll
            for (int jj = 0; jj < 4; ++jj) {
                acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[kk][jj], b0[kk][jj], acc[0], 0, 0, 0);
                acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[kk][jj], b1[kk][jj], acc[1], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
                const int gap = kk * 4 + jj;
                if (gap == 0) {
                    a[1] = *reinterpret_cast<const af32x4 *>(smem + p_frag + kc * A2_D + 8);
                    b0[1] = *reinterpret_cast<const af32x4 *>(vs + b_frag + 8);
                    b1[1] = *reinterpret_cast<const af32x4 *>(vs + b_frag + 32 * A2_LD + 8);
                }
                if (gap >= 1 && gap <= 4 && kc + 1 < NKC) store_b1(kc + 1, gap - 1);
                if (gap >= 5 && gap <= 7 && kc + 3 < NKC) load_b1(kc + 3, gap - 5);
                if (gap == 7 && kc + 3 < NKC) load_b1(kc + 3, 3);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        __syncthreads();
    }

    // ---- epilogue: row = query, column = channel ----
#pragma unroll
    for (int c = 0; c < 2; ++c) {
        const int col = wn * 64 + c * 32 + li;
        const float bv = bvv[c];
        float s = 0.f, q = 0.f;
#pragma unroll
        for (int v = 0; v < 16; ++v) {
            const int row = 8 * (v >> 2) + (v & 3) + 4 * lh;
            float val = acc[c][v] + bv;
            if (p.res) val += rv[c][v];
            val *= p.out_scale;
            if (AT_OUT_NT) __builtin_nontemporal_store(val, &p.out[((size_t)img * AT_N + q0 + row) * p.ld_o + col]);
            else p.out[((size_t)img * AT_N + q0 + row) * p.ld_o + col] = val;
            s += val;
            q += val * val;
        }
        if (p.qstats) {  // the workgroup's 32 rows are one statistics slice
            s += __shfl_xor(s, 1); q += __shfl_xor(q, 1);
            s += __shfl_xor(s, 2); q += __shfl_xor(q, 2);
            s += __shfl_xor(s, 32); q += __shfl_xor(q, 32);
            if ((li & 3) == 0 && lh == 0)
                reinterpret_cast<float2 *>(p.qstats)[((size_t)img * (AT_N / 32) + (q0 >> 5)) * (AT_C / 4) + (col >> 2)] = make_float2(s, q);
        }
    }
}

bool attn_fused_ok(int N, int C) {
    return !opts().no_fused_attn && N == AT_N && C == AT_C;
}

// The query block of a launch: 64 (one workgroup per CU) when its 4 B workgroups fill the chip, 32 (two per CU, 8 B workgroups) below that --
// at the production batch the two forms measure the same (round 4: 78.1 vs 77.0 ms per step, what the overlap of one workgroup's softmax with
// the other's MFMAs gains goes to streaming K and V^T twice as often and twice the barriers), at small batches the 32-query form has twice the
// workgroups to spread.  GDDIM_ATTN_Q=32 / 64 forces a form (A/B).  Bit-identical outputs.
static int attn_qblock(int batch) {
    if (opts().attn_q == 32 || opts().attn_q == 64) return opts().attn_q;
    return 4 * batch >= device_cus() ? 64 : 32;
}
const char *attn_fused_name(int batch) { return attn_qblock(batch) == 64 ? "attn256_f32_kernel" : "attn256_q32_f32_kernel"; }

int launch_attn_fused(const float *q, int ld_q, const float *k, int ld_k, const float *vT, const float *bias, const float *res, int ld_res,
                      float out_scale, float *qstats, float *out, int ld_o, int batch, int N, int C, hipStream_t s) {
    GD_CHECK(q && k && vT && out && N == AT_N && C == AT_C && ld_q % 4 == 0 && ld_k % 4 == 0 && batch >= 1,
             "attn_fused: unsupported problem (N=%d C=%d)", N, C);
    AttnParams p;
    p.q = q; p.k = k; p.ld_q = ld_q; p.ld_k = ld_k; p.vT = vT; p.bias = bias; p.res = res; p.ld_res = ld_res; p.out_scale = out_scale;
    p.qstats = qstats; p.out = out; p.ld_o = ld_o;
    p.scale_log2e = (float)(1.0 / 16.0 * 1.4426950408889634);  // C^-1/2 = 1/16
    p.batch = batch;
    if (attn_qblock(batch) == 64) hipLaunchKernelGGL(attn256_f32_kernel, dim3(((batch + 7) / 8) * 8 * (AT_N / 64)), dim3(512), 0, s, p);
    else hipLaunchKernelGGL(attn256_q32_f32_kernel, dim3(((batch + 7) / 8) * 8 * (AT_N / A2_Q)), dim3(256), 0, s, p);
    GD_HIP(hipGetLastError());
    return 0;
}

}  // namespace gd
