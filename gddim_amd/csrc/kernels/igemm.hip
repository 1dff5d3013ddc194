// Implicit-GEMM convolution / GEMM on the gfx950 fp32 matrix cores.
//
// One kernel serves every contraction of the NCSN++ forward:
// 3x3 / 1x1 / 6x6-stride-2 convolutions (nn.Conv call sites, cld_jax/models/layers.py:66-107,
// up_or_down_sampling.py:168-209), NIN / Dense layers (layers.py:467-493, ncsnpp.py:88-89,
// layerspp.py:216) and the two attention einsums (layerspp.py:74,78).
//
//   C[m][n] = (alpha * sum_k A[m][k] B[n][k] + bias[n] + rowbias[img(m)][n] + res[m][n]) * out_scale
//
// A is never materialised: row m is output pixel (img, oy, ox), column k = (dy*KW+dx)*Cin + c is
// gathered from up to two NHWC sources (channel concat of the U-Net skip, ncsnpp.py:185) with
// zero fill outside the image ('SAME' padding).  B is [N][K] row-major (packed weights, or the K
// matrix of attention) or k-major (the V matrix of attention).
//
// Tiling (CDNA4): 256 threads = 4 wave64 as 2x2; each wave owns a (BM/2)x(BN/2) sub-tile as RMxRN
// v_mfma_f32_32x32x2_f32 accumulators (exact fp32, 64 cycles each, 157 TFLOP/s chip peak).  K is
// consumed in chunks of 32: the A and B chunk tiles are staged through LDS with a 36-float row
// stride so that both the 16-byte staging stores and the per-lane ds_read_b128 fragment reads are
// bank-conflict free.  Lane (i = lane&31, h = lane>>5) reads 4 consecutive k values
// [kb*8 + 4h, kb*8 + 4h + 4) of row i; MFMA number jj of that group therefore contracts
// k = kb*8 + jj (lanes 0-31) and k = kb*8 + 4 + jj (lanes 32-63) -- A and B use the same
// assignment, so the (k0,k1) pairing is consistent and the result is the plain k-sum.
// Staging is register based and branch-free: masked lanes (image border, ragged M/N/K) read a device zero buffer, so no
// select follows the load.  Global loads run two chunks ahead of the MFMAs and, together with the LDS writes, are
// interleaved by hand into the gaps of the MFMA stream (see the kernel comment).  LDS is double buffered.
//
// Block -> tile mapping is XCD-aware: consecutive block ids round-robin over the 8 XCDs, so ids
// are grouped in runs of 8*n_tiles with the M tile varying fastest inside a run; the n_tiles
// blocks that share one A tile then carry ids congruent mod 8 and run on the same XCD (shared L2).
#include <cstdio>
#include <cstdlib>

#include "../common.h"
#include "../options.h"
#include "act.h"
#include "qstats.h"

namespace gd {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int KC = 32;   // K chunk
constexpr int LDT = 36;  // LDS row stride (floats): 32 + 4 pad -> 144 B, odd multiple of 16 B

__device__ __forceinline__ float swish_f(float x) { return swish_e(x); }
#ifndef IGEMM_OUT_AUX
#define IGEMM_OUT_AUX 2  // cache policy of the persistent kernel's output / GroupNorm side-output stores: nt (streaming).  Round 5, three alternating
                         // repetitions in one box: 138.11 -> 138.63 images/s (+0.4 %), the persistent launches 243.7 -> 241.8 ms per step (0 = default policy, A/B)
#endif

// BM x BN block tile, WM x WN wave grid (64*WM*WN threads), DB: double-buffered LDS (one barrier per K chunk)
// BM x BN block tile, WM x WN wave grid (64*WM*WN threads).
//
// K loop, software pipelined by hand.  At the top of iteration kc the staging registers hold chunk kc+1 (its global
// loads were issued during iteration kc-1, i.e. a full chunk of latency budget) and LDS stage kc&1 holds chunk kc.
// The iteration issues its 64/(WM*WN/4..) MFMAs in 16 groups (4 k-groups x 4 k-pairs); each gap between groups is
// given one "piece" of staging work for the SAME wave -- write one staged float4 of chunk kc+1 to the other LDS stage,
// then compute the address of and issue the global load of the same register for chunk kc+2 -- because a wave issues
// in order and only instructions placed between two MFMAs execute in the 64-cycle shadow of the first one.
// __builtin_amdgcn_sched_barrier(0) pins that order.  One barrier per chunk.
template <int BM, int BN, int WM, int WN, bool B_KMAJOR, bool IN_ACT>
__global__ __launch_bounds__(64 * WM * WN) void igemm_f32_kernel(const GemmParams p) {
    constexpr int NT = 64 * WM * WN;
    constexpr int RM = BM / (32 * WM), RN = BN / (32 * WN);  // 32x32 MFMA tiles per wave
    constexpr int RPP = NT / 8;                              // tile rows staged per pass (8 threads x 16 B per row)
    constexpr int AJ = BM / RPP, BJ = BN / RPP;              // float4 staged per thread (A / B)
    constexpr int NP = AJ + BJ;                              // staging pieces per chunk
    constexpr int BUF = (BM + BN) * LDT;                     // floats per LDS stage
    static_assert(BM % RPP == 0 && BN % RPP == 0 && RM >= 1 && RN >= 1 && NP <= 14, "bad tile geometry");
    __shared__ __attribute__((aligned(16))) float smem[2 * BUF];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);  // wave-uniform: keeps the epilogue row offsets in SGPRs
    const int wm = wave / WN, wn = wave % WN;

    // ---- XCD-aware block -> (mtile, ntile) ----
    const int nt = p.n_tiles;
    const int id = blockIdx.x;
    int ntile, mtile;
    if (p.flat_tiles) {  // small grids: no padding of the M-tile count to the XCD group size
        mtile = id / nt;
        ntile = id - mtile * nt;
    } else {
        const int grp = id / (8 * nt);
        const int rem = id - grp * 8 * nt;
        ntile = rem >> 3;
        mtile = grp * 8 + (rem & 7);
    }
    if (mtile >= p.m_tiles) return;
    const int m0 = mtile * BM, n0 = ntile * BN;
    const int z = blockIdx.z;

    const float *a0 = p.a0 + (size_t)z * p.sA;
    const float *a1 = p.a1;
    const float *bptr = p.b + (size_t)z * p.sB;
    const float *zeros = p.zeros;  // >= 16 B of zeros: masked lanes load from here, so no select after the load

    // ---- per-thread row bookkeeping (fixed over the K loop); offsets are 32-bit element counts ----
    const int arow = tid >> 3;        // 0..RPP-1, +RPP per j
    const int kcol = (tid & 7) * 4;   // float offset inside the chunk
    int a_pix[AJ], a_iy0[AJ], a_ix0[AJ];
    {
        const int OHW = p.OH * p.OW;
#pragma unroll
        for (int j = 0; j < AJ; ++j) {
            const int m = m0 + arow + RPP * j;
            if (m < p.M) {
                const int img = m / OHW;
                const int r = m - img * OHW;
                const int oy = r / p.OW;
                const int ox = r - oy * p.OW;
                a_pix[j] = img * p.H * p.W;
                a_iy0[j] = oy * p.stride - p.pad;
                a_ix0[j] = ox * p.stride - p.pad;
            } else {
                a_pix[j] = 0;
                a_iy0[j] = -(1 << 28);  // always outside the image: masked
                a_ix0[j] = 0;
            }
        }
    }
    int b_off[BJ];   // n-major: n * ldb (or -1 when n >= N); k-major: n (or -1)
#pragma unroll
    for (int j = 0; j < BJ; ++j) {
        if (!B_KMAJOR) {
            const int n = n0 + arow + RPP * j;
            b_off[j] = n < p.N ? n * p.ldb : -1;
        } else {
            const int n = n0 + ((tid >> 5) + (NT / 32) * j) * 4;
            b_off[j] = n < p.N ? n : -1;
        }
    }
    const int Cin = p.C0 + p.C1;
    // decomposition of this thread's k = kc*32 + kcol into (dy, dx, c) for the chunk whose loads are issued next
    int c_run = kcol, dy = 0, dx = 0, k_next = kcol;
    while (c_run >= Cin) {
        c_run -= Cin;
        if (++dx == p.KW) { dx = 0; ++dy; }
    }
    auto advance_k = [&]() {
        c_run += KC;
        k_next += KC;
        while (c_run >= Cin) {
            c_run -= Cin;
            if (++dx == p.KW) { dx = 0; ++dy; }
        }
    };

    f32x4 ra[AJ], rb[BJ];

    // issue the global load of staged register `pc` (0..AJ-1: A rows, AJ..NP-1: B rows) for the chunk at k_next
    auto load_piece = [&](int pc) {
        if (pc < AJ) {
            const int j = pc;
            const bool first = c_run < p.C0;
            const float *src = first ? a0 : a1;
            const int ld = first ? p.lda0 : p.lda1;
            const int cc = first ? c_run : c_run - p.C0;
            const int iy = a_iy0[j] + dy, ix = a_ix0[j] + dx;
            const bool ok = k_next < p.K && (unsigned)iy < (unsigned)p.H && (unsigned)ix < (unsigned)p.W;
            const int off = (a_pix[j] + iy * p.W + ix) * ld + cc;
            const float *ptr = ok ? src + off : zeros;
            ra[j] = *reinterpret_cast<const f32x4 *>(ptr);
        } else {
            const int j = pc - AJ;
            if (!B_KMAJOR) {
                const bool ok = b_off[j] >= 0 && k_next < p.K;
                const float *ptr = ok ? bptr + (b_off[j] + k_next) : zeros;
                rb[j] = *reinterpret_cast<const f32x4 *>(ptr);
            } else {
                const int kk = k_next - kcol + (tid & 31);
                const bool ok = b_off[j] >= 0 && kk < p.K;
                const float *ptr = ok ? bptr + (kk * p.ldb + b_off[j]) : zeros;
                rb[j] = *reinterpret_cast<const f32x4 *>(ptr);
            }
        }
    };
    // write staged register `pc` into LDS stage `st`
    auto store_piece = [&](int pc, float *st) {
        if (pc < AJ) {
            f32x4 v = ra[pc];
            if (IN_ACT) { v[0] = swish_f(v[0]); v[1] = swish_f(v[1]); v[2] = swish_f(v[2]); v[3] = swish_f(v[3]); }
            *reinterpret_cast<f32x4 *>(&st[(arow + RPP * pc) * LDT + kcol]) = v;
        } else {
            const int j = pc - AJ;
            float *Bs = st + BM * LDT;
            if (!B_KMAJOR) {
                *reinterpret_cast<f32x4 *>(&Bs[(arow + RPP * j) * LDT + kcol]) = rb[j];
            } else {
                const int kk = tid & 31;
                const int nl = ((tid >> 5) + (NT / 32) * j) * 4;
                Bs[(nl + 0) * LDT + kk] = rb[j][0];
                Bs[(nl + 1) * LDT + kk] = rb[j][1];
                Bs[(nl + 2) * LDT + kk] = rb[j][2];
                Bs[(nl + 3) * LDT + kk] = rb[j][3];
            }
        }
    };

    f32x16 acc[RM][RN];
#pragma unroll
    for (int r = 0; r < RM; ++r)
#pragma unroll
        for (int c = 0; c < RN; ++c)
#pragma unroll
            for (int v = 0; v < 16; ++v) acc[r][c][v] = 0.f;

    const int nk = (p.K + KC - 1) / KC;
    const int li = lane & 31, lh = lane >> 5;
    const int a_off = (wm * (BM / WM) + li) * LDT + lh * 4;
    const int b_offl = BM * LDT + (wn * (BN / WN) + li) * LDT + lh * 4;

    // ---- prologue: chunk 0 -> LDS stage 0, chunk 1 -> registers (in flight) ----
#pragma unroll
    for (int pc = 0; pc < NP; ++pc) load_piece(pc);
    advance_k();
#pragma unroll
    for (int pc = 0; pc < NP; ++pc) store_piece(pc, smem);
#pragma unroll
    for (int pc = 0; pc < NP; ++pc) load_piece(pc);  // k_next >= K loads zeros when nk == 1
    advance_k();
    __syncthreads();

    for (int kc = 0; kc < nk; ++kc) {
        const float *cur = smem + (kc & 1) * BUF;
        float *nxt = smem + ((kc + 1) & 1) * BUF;
        f32x4 af[RM], bf[RN];
#pragma unroll
        for (int r = 0; r < RM; ++r) af[r] = *reinterpret_cast<const f32x4 *>(cur + a_off + r * 32 * LDT);
#pragma unroll
        for (int c = 0; c < RN; ++c) bf[c] = *reinterpret_cast<const f32x4 *>(cur + b_offl + c * 32 * LDT);
#pragma unroll
        for (int kb = 0; kb < KC / 8; ++kb) {
            f32x4 afn[RM], bfn[RN];
#pragma unroll
            for (int jj = 0; jj < 4; ++jj) {
#pragma unroll
                for (int r = 0; r < RM; ++r)
#pragma unroll
                    for (int c = 0; c < RN; ++c)
                        acc[r][c] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[r][jj], bf[c][jj], acc[r][c], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
                // ---- work placed in the shadow of the MFMA group above ----
                const int gap = kb * 4 + jj;
                if (gap < NP) {
                    // staged register `gap`: chunk kc+1 -> LDS, then reload it with chunk kc+2
                    store_piece(gap, nxt);
                    load_piece(gap);
                } else if (gap == NP) {
                    advance_k();
                }
                if (jj == 2 && kb + 1 < KC / 8) {  // fragments of the next k-group, one gap ahead of their use
#pragma unroll
                    for (int r = 0; r < RM; ++r) afn[r] = *reinterpret_cast<const f32x4 *>(cur + a_off + r * 32 * LDT + (kb + 1) * 8);
#pragma unroll
                    for (int c = 0; c < RN; ++c) bfn[c] = *reinterpret_cast<const f32x4 *>(cur + b_offl + c * 32 * LDT + (kb + 1) * 8);
                }
                __builtin_amdgcn_sched_barrier(0);
            }
            if (kb + 1 < KC / 8) {
#pragma unroll
                for (int r = 0; r < RM; ++r) af[r] = afn[r];
#pragma unroll
                for (int c = 0; c < RN; ++c) bf[c] = bfn[c];
            }
        }
        __syncthreads();
    }

    // ---- epilogue: D[row][col], col = lane&31, row = (v&3) + 8*(v>>2) + 4*(lane>>5) ----
    // Residual / per-image bias loads are issued 16 at a time (one 32x32 tile) before any is consumed, so the
    // epilogue pays one memory round trip per tile instead of one per element.
    float *outp = p.out + (size_t)z * p.sC;
    const float *resp = p.res ? p.res + (size_t)z * p.sR : nullptr;
    const int rpi = p.rows_per_image;
#pragma unroll
    for (int c = 0; c < RN; ++c) {
        const int n = n0 + wn * (BN / WN) + c * 32 + li;
        const bool nok = n < p.N;
        const float bias = (p.bias && nok) ? p.bias[n] : 0.f;
#pragma unroll
        for (int r = 0; r < RM; ++r) {
            const int mb = m0 + wm * (BM / WM) + r * 32 + 4 * lh;
            float addv[16];
#pragma unroll
            for (int v = 0; v < 16; ++v) addv[v] = bias;
            if (p.res) {
                float t[16];
#pragma unroll
                for (int v = 0; v < 16; ++v) {
                    const int m = mb + (v & 3) + 8 * (v >> 2);
                    const float *ptr = (nok && m < p.M) ? resp + (size_t)m * p.ldres + n : resp;
                    t[v] = *ptr;
                }
#pragma unroll
                for (int v = 0; v < 16; ++v) addv[v] += t[v];
            }
            if (p.rowbias) {
                const int img_b = mb / rpi;
                const int rem_b = mb - img_b * rpi;
                float t[16];
#pragma unroll
                for (int v = 0; v < 16; ++v) {
                    int q = rem_b + (v & 3) + 8 * (v >> 2), img = img_b;
                    while (q >= rpi) { q -= rpi; ++img; }
                    const int m = mb + (v & 3) + 8 * (v >> 2);
                    const float *ptr = (nok && m < p.M) ? p.rowbias + (size_t)img * p.ld_rowbias + n : p.rowbias;
                    t[v] = *ptr;
                }
#pragma unroll
                for (int v = 0; v < 16; ++v) addv[v] += t[v];
            }
            float vals[16];
#pragma unroll
            for (int v = 0; v < 16; ++v) {
                const int m = mb + (v & 3) + 8 * (v >> 2);
                const float val = (p.alpha * acc[r][c][v] + addv[v]) * p.out_scale;
                const bool ok = nok && m < p.M;
                if (ok) outp[(size_t)m * p.ldc + n] = val;
                vals[v] = ok ? val : 0.f;
            }
            if (p.qstats) emit_qstats(p, vals, mb - 4 * lh, n, nok, li, lh, p.qstats + (size_t)z * p.sQ);
        }
    }
}

// =====================================================================================================================
// Fast path.  Counter evidence (profiles/): on gfx950 the fp32 MFMA never co-executes with VALU work
// (SQ_VALU_MFMA_COEXEC_CYCLES = 0, MFMA-busy + VALU-busy ~ 100 %): every VALU instruction in the K loop costs matrix
// throughput, whichever wave issues it.  This kernel therefore keeps the K loop (almost) VALU-free:
//   * chunk bookkeeping (tap, channel base, source) is wave-uniform and lives in SGPRs (needs Cin % 32 == 0);
//   * each staged row has a precomputed byte offset and a precomputed bit mask of the taps that fall outside the
//     image; the per-chunk address is  voffset = rowbase | sext(invalid bit)  (2 VALU), the tap/channel displacement
//     rides in the scalar soffset of a raw buffer load, and the buffer's bounds check returns the zero padding;
//   * B (weights / K matrix) rows need no VALU at all: voffset is constant, soffset = chunk offset.
// Same tiling, LDS layout, software pipeline and epilogue as the generic kernel above.
// Requirements (checked by launch_igemm): (C0+C1) % 32 == 0, C0 % 32 == 0 when C1 > 0, KH*KW <= 32 (<= 64: the T64 instances), every operand
// smaller than 2 GiB, no input activation.
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

// n / d with a host-computed (magic, shift): exact for n < 2^31 (launch_igemm: fill_div)
__device__ __forceinline__ unsigned fast_div(unsigned n, unsigned d, unsigned magic, unsigned shift) {
    if (d == 1) return n;
    const unsigned t = __umulhi(magic, n);
    return (t + ((n - t) >> 1)) >> (shift - 1);
}

__device__ __forceinline__ f32x4 buf_load16(__amdgpu_buffer_rsrc_t rsrc, int voffset, int soffset) {
    u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(rsrc, voffset, soffset, 0);
    return __builtin_bit_cast(f32x4, v);
}

// T64: kernels with 33..64 taps (the 6x6 stride-2 pyramid convolutions) keep a second mask word per row
// KS > 1: intra-workgroup split-K.  KS wave groups of WM x WN waves each own every KS-th K chunk (their own LDS stages, their
// own accumulators over the SAME output tile); after the loop groups 1..KS-1 hand their accumulators to group 0 through LDS
// and group 0 runs the epilogue.  For launches with too few tiles to give every SIMD two waves (the 4x4 level of the U-Net:
// 256 tiles of K = 2304..4608) this doubles the resident waves without a second pass or atomics; summation order over the
// chunks changes (even/odd partial sums), results stay deterministic.
template <int BM, int BN, int WM, int WN, bool B_KMAJOR, bool T64 = false, int KS = 1>
__global__ __launch_bounds__(64 * WM * WN * KS) void igemm_f32_fast_kernel(const GemmParams p) {
    constexpr int NT = 64 * WM * WN;  // threads of one wave group
    constexpr int RM = BM / (32 * WM), RN = BN / (32 * WN);
    constexpr int RPP = NT / 8;
    constexpr int AJ = BM / RPP, BJ = BN / RPP;
    constexpr int NP = AJ + BJ;
    constexpr int BUF = (BM + BN) * LDT;
    constexpr int OOB = (int)0x80000000u;  // voffset that fails the bounds check of every (< 2 GiB) buffer
    static_assert(BM % RPP == 0 && BN % RPP == 0 && RM >= 1 && RN >= 1 && NP <= 14, "bad tile geometry");
    static_assert(KS == 1 || (KS - 1) * RM * RN * 16 * NT <= 2 * KS * BUF, "split-K reduction buffer does not fit the stages");
    __shared__ __attribute__((aligned(16))) float smem[2 * KS * BUF];

    const int lane = threadIdx.x & 63;
    const int wave_all = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);  // wave-uniform: keeps the epilogue row offsets in SGPRs
    const int ks = KS == 1 ? 0 : wave_all / (WM * WN);                       // K-split group of this wave
    const int wave = KS == 1 ? wave_all : wave_all - ks * (WM * WN);
    const int tid = wave * 64 + lane;                                       // thread index inside the wave group
    const int wm = wave / WN, wn = wave % WN;

    const int nt = p.n_tiles;
    const int id = blockIdx.x;
    int ntile, mtile;
    if (p.flat_tiles) {  // small grids: no padding of the M-tile count to the XCD group size
        mtile = id / nt;
        ntile = id - mtile * nt;
    } else {
        const int grp = id / (8 * nt);
        const int rem = id - grp * 8 * nt;
        ntile = rem >> 3;
        mtile = grp * 8 + (rem & 7);
    }
    if (mtile >= p.m_tiles) return;
    const int m0 = mtile * BM, n0 = ntile * BN;
    const int z = blockIdx.z;

    // ---- buffer descriptors (wave-uniform).  The A descriptors start `bias` bytes BEFORE the tensor so that the
    // precomputed row offsets (which address tap (0,0), up to `pad` rows/columns outside the image) are non-negative.
    const int Cin = p.C0 + p.C1;
    const int taps = p.KH * p.KW;
    const int a_imgs = (p.M + p.OH * p.OW - 1) / (p.OH * p.OW);
    const int bias0 = (p.pad * p.W + p.pad) * p.lda0 * 4;
    const int bias1 = (p.pad * p.W + p.pad) * p.lda1 * 4;
    const char *a0c = reinterpret_cast<const char *>(p.a0 + (size_t)z * p.sA);
    const __amdgpu_buffer_rsrc_t srd_a0 =
        __builtin_amdgcn_make_buffer_rsrc(const_cast<char *>(a0c - bias0), 0, a_imgs * p.H * p.W * p.lda0 * 4 + bias0, 0x00020000);
    const char *a1c = reinterpret_cast<const char *>(p.C1 ? p.a1 : p.a0);
    const __amdgpu_buffer_rsrc_t srd_a1 =
        __builtin_amdgcn_make_buffer_rsrc(const_cast<char *>(a1c - bias1), 0, a_imgs * p.H * p.W * p.lda1 * 4 + bias1, 0x00020000);
    const char *bc = reinterpret_cast<const char *>(p.b + (size_t)z * p.sB);
    const __amdgpu_buffer_rsrc_t srd_b =
        __builtin_amdgcn_make_buffer_rsrc(const_cast<char *>(bc), 0, (B_KMAJOR ? p.K : p.N) * p.ldb * 4, 0x00020000);

    // ---- per-thread, loop-invariant row data ----
    const int arow = tid >> 3;
    const int kcol = (tid & 7) * 4;
    int rb0[AJ], rb1[AJ];      // byte offset of (row's tap (0,0) pixel, channel kcol) in source 0 / 1, + bias
    unsigned inv[AJ];          // bit t set: tap t of this row lies outside the image (or the row is >= M)
    unsigned inv_hi[T64 ? AJ : 1];  // taps 32..63
    {
        const unsigned OHW = p.OH * p.OW;
        const unsigned rowbits = (1u << p.KW) - 1u;
#pragma unroll
        for (int j = 0; j < AJ; ++j) {
            const int m = m0 + arow + RPP * j;
            rb0[j] = 0; rb1[j] = 0; inv[j] = 0xFFFFFFFFu;
            if (T64) inv_hi[j] = 0xFFFFFFFFu;
            if (m < p.M) {
                const int img = (int)fast_div((unsigned)m, OHW, p.mg_ohw, p.sh_ohw);
                const int r = m - img * (int)OHW;
                const int oy = (int)fast_div((unsigned)r, (unsigned)p.OW, p.mg_ow, p.sh_ow);
                const int ox = r - oy * p.OW;
                const int iy0 = oy * p.stride - p.pad, ix0 = ox * p.stride - p.pad;
                const int pix = (img * p.H + iy0) * p.W + ix0;  // may be "negative" by at most pad*W + pad: the bias covers it
                rb0[j] = pix * p.lda0 * 4 + bias0 + kcol * 4;
                rb1[j] = pix * p.lda1 * 4 + bias1 + kcol * 4;
                unsigned xb = 0;  // columns of the window outside the image
                for (int tx = 0; tx < p.KW; ++tx) xb |= ((unsigned)(ix0 + tx) < (unsigned)p.W ? 0u : 1u) << tx;
                unsigned long long mk = 0;
                for (int ty = 0; ty < p.KH; ++ty)
                    mk |= (unsigned long long)((unsigned)(iy0 + ty) < (unsigned)p.H ? xb : rowbits) << (ty * p.KW);
                inv[j] = (unsigned)mk;
                if (T64) inv_hi[j] = (unsigned)(mk >> 32);
            }
        }
    }
    int bvo[BJ];  // B voffset (bytes), OOB when the row / column group is outside N
#pragma unroll
    for (int j = 0; j < BJ; ++j) {
        if (!B_KMAJOR) {
            const int n = n0 + arow + RPP * j;
            bvo[j] = n < p.N ? (n * p.ldb + kcol) * 4 : OOB;
        } else {
            const int n = n0 + ((tid >> 5) + (NT / 32) * j) * 4;
            bvo[j] = n < p.N ? ((tid & 31) * p.ldb + n) * 4 : OOB;
        }
    }

    // ---- wave-uniform state of the chunk whose loads are issued next: tap (ty, tx), channel base c0 ----
    int s_tap = 0, s_ty = 0, s_tx = 0, s_c0 = 0, s_kc = 0;
    const int nk = p.diag == 2 ? 1 : p.K / KC;  // diag: diagnostic only (GDDIM_IGEMM_DIAG), results wrong
    const int nit = (nk + KS - 1) / KS;         // loop iterations: group ks owns chunks ks, ks + KS, ... (chunks >= nk read zeros)
    // K order: 32-channel slab outermost, the KH*KW taps of that slab innermost.  The taps re-read (shifted) the same
    // activation lines, so keeping them adjacent in time turns 8 of 9 tap loads into L1/L2 hits (with taps outermost
    // the lines are evicted between taps: ~4x HBM over-fetch in the PMC pass).  Any K permutation is legal as long as
    // A and B agree; B is addressed through the scalar soffset, so the packed weight layout [n][tap][c] is unchanged.
    auto advance_1 = [&]() {
        ++s_kc;
        ++s_tap;
        if (++s_tx == p.KW) { s_tx = 0; ++s_ty; }
        if (s_tap == taps) { s_tap = 0; s_ty = 0; s_tx = 0; s_c0 += KC; }
    };
    auto advance_k = [&]() {
#pragma unroll
        for (int i = 0; i < KS; ++i) advance_1();
    };
    for (int i = 0; i < ks; ++i) advance_1();  // first chunk of this group

    f32x4 ra[AJ], rb[BJ];

    auto load_piece = [&](int pc) {
        if (pc < AJ) {
            const int j = pc;
            // sign-extended "invalid" bit of the current tap: -1 pushes the offset out of bounds -> hardware returns 0
            const unsigned word = (T64 && s_tap >= 32) ? inv_hi[j] : inv[j];
            const int bad = __builtin_amdgcn_sbfe(word, T64 ? (s_tap & 31) : (s_tap < 32 ? s_tap : 31), 1) | (s_kc < nk ? 0 : -1);
            if (s_c0 < p.C0) {
                const int soff = ((s_ty * p.W + s_tx) * p.lda0 + s_c0) * 4;
                ra[j] = buf_load16(srd_a0, rb0[j] | (bad & OOB), soff);
            } else {
                const int soff = ((s_ty * p.W + s_tx) * p.lda1 + (s_c0 - p.C0)) * 4;
                ra[j] = buf_load16(srd_a1, rb1[j] | (bad & OOB), soff);
            }
        } else {
            const int j = pc - AJ;
            const int kb = s_tap * Cin + s_c0;  // first k of this chunk in the packed [tap][c] order
            const int soff = (B_KMAJOR ? kb * p.ldb : kb) * 4;
            rb[j] = buf_load16(srd_b, s_kc < nk ? bvo[j] : OOB, soff);
        }
    };
    auto store_piece = [&](int pc, float *st) {
        if (pc < AJ) {
            *reinterpret_cast<f32x4 *>(&st[(arow + RPP * pc) * LDT + kcol]) = ra[pc];
        } else {
            const int j = pc - AJ;
            float *Bs = st + BM * LDT;
            if (!B_KMAJOR) {
                *reinterpret_cast<f32x4 *>(&Bs[(arow + RPP * j) * LDT + kcol]) = rb[j];
            } else {
                const int kk = tid & 31;
                const int nl = ((tid >> 5) + (NT / 32) * j) * 4;
                Bs[(nl + 0) * LDT + kk] = rb[j][0];
                Bs[(nl + 1) * LDT + kk] = rb[j][1];
                Bs[(nl + 2) * LDT + kk] = rb[j][2];
                Bs[(nl + 3) * LDT + kk] = rb[j][3];
            }
        }
    };

    f32x16 acc[RM][RN];
#pragma unroll
    for (int r = 0; r < RM; ++r)
#pragma unroll
        for (int c = 0; c < RN; ++c)
#pragma unroll
            for (int v = 0; v < 16; ++v) acc[r][c][v] = 0.f;

    const int li = lane & 31, lh = lane >> 5;
    const int a_off = (wm * (BM / WM) + li) * LDT + lh * 4;
    const int b_offl = BM * LDT + (wn * (BN / WN) + li) * LDT + lh * 4;

    float *const stage0 = smem + ks * 2 * BUF;  // this group's two LDS stages
#pragma unroll
    for (int pc = 0; pc < NP; ++pc) load_piece(pc);
    advance_k();
#pragma unroll
    for (int pc = 0; pc < NP; ++pc) store_piece(pc, stage0);
#pragma unroll
    for (int pc = 0; pc < NP; ++pc) load_piece(pc);
    advance_k();
    __syncthreads();

    for (int kc = 0; kc < nit; ++kc) {
        const float *cur = stage0 + (kc & 1) * BUF;
        float *nxt = stage0 + ((kc + 1) & 1) * BUF;
        f32x4 af[RM], bf[RN];
#pragma unroll
        for (int r = 0; r < RM; ++r) af[r] = *reinterpret_cast<const f32x4 *>(cur + a_off + r * 32 * LDT);
#pragma unroll
        for (int c = 0; c < RN; ++c) bf[c] = *reinterpret_cast<const f32x4 *>(cur + b_offl + c * 32 * LDT);
#pragma unroll
        for (int kb = 0; kb < KC / 8; ++kb) {
            f32x4 afn[RM], bfn[RN];
#pragma unroll
            for (int jj = 0; jj < 4; ++jj) {
#pragma unroll
                for (int r = 0; r < RM; ++r)
#pragma unroll
                    for (int c = 0; c < RN; ++c)
                        acc[r][c] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[r][jj], bf[c][jj], acc[r][c], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
                const int gap = kb * 4 + jj;
                if (gap < NP) {
                    store_piece(gap, nxt);
                    load_piece(gap);
                } else if (gap == NP) {
                    advance_k();
                }
                if (jj == 2 && kb + 1 < KC / 8) {
#pragma unroll
                    for (int r = 0; r < RM; ++r) afn[r] = *reinterpret_cast<const f32x4 *>(cur + a_off + r * 32 * LDT + (kb + 1) * 8);
#pragma unroll
                    for (int c = 0; c < RN; ++c) bfn[c] = *reinterpret_cast<const f32x4 *>(cur + b_offl + c * 32 * LDT + (kb + 1) * 8);
                }
                __builtin_amdgcn_sched_barrier(0);
            }
            if (kb + 1 < KC / 8) {
#pragma unroll
                for (int r = 0; r < RM; ++r) af[r] = afn[r];
#pragma unroll
                for (int c = 0; c < RN; ++c) bf[c] = bfn[c];
            }
        }
        __syncthreads();
    }

    // ---- epilogue: D[row][col], col = lane&31, row = (v&3) + 8*(v>>2) + 4*(lane>>5) ----
    // Buffer addressed: the lane part of the offset ((4*lh)*ld + n) is computed once per column tile, the row part is a
    // scalar soffset, rows >= M fall outside the descriptor (dropped by the bounds check), columns >= N are pushed out
    // of bounds once.  fast_ok() guarantees rows_per_image % 8 == 0, so the two half-waves of one store (rows r0, r0+4)
    // belong to the same image and the per-image bias row is wave-uniform.
    if (KS > 1) {  // partial sums of groups 1..KS-1 -> LDS (lane-contiguous, conflict-free) -> group 0
        if (ks > 0) {
            float *dst = smem + (size_t)(ks - 1) * RM * RN * 16 * NT + tid;
#pragma unroll
            for (int r = 0; r < RM; ++r)
#pragma unroll
                for (int c = 0; c < RN; ++c)
#pragma unroll
                    for (int v = 0; v < 16; ++v) dst[((r * RN + c) * 16 + v) * NT] = acc[r][c][v];
        }
        __syncthreads();
        if (ks > 0) return;
#pragma unroll
        for (int g = 1; g < KS; ++g) {
            const float *src = smem + (size_t)(g - 1) * RM * RN * 16 * NT + tid;
#pragma unroll
            for (int r = 0; r < RM; ++r)
#pragma unroll
                for (int c = 0; c < RN; ++c)
#pragma unroll
                    for (int v = 0; v < 16; ++v) acc[r][c][v] += src[((r * RN + c) * 16 + v) * NT];
        }
    }
    {
        char *outc = reinterpret_cast<char *>(p.out + (size_t)z * p.sC);
        const __amdgpu_buffer_rsrc_t srd_o = __builtin_amdgcn_make_buffer_rsrc(outc, 0, p.M * p.ldc * 4, 0x00020000);
        const __amdgpu_buffer_rsrc_t srd_r =
            __builtin_amdgcn_make_buffer_rsrc(const_cast<char *>(reinterpret_cast<const char *>(p.res ? p.res + (size_t)z * p.sR : p.out)), 0,
                                              p.M * p.ldres * 4, 0x00020000);
        const unsigned rpi = p.rows_per_image;
#pragma unroll
        for (int c = 0; c < RN; ++c) {
            const int n = n0 + wn * (BN / WN) + c * 32 + li;
            const bool nok = n < p.N;
            const float bias = (p.bias && nok) ? p.bias[n] : 0.f;
            const int vo_o = nok ? (4 * lh * p.ldc + n) * 4 : OOB;
            const int vo_r = nok ? (4 * lh * p.ldres + n) * 4 : OOB;
#pragma unroll
            for (int r = 0; r < RM; ++r) {
                const int mrow = m0 + wm * (BM / WM) + r * 32;  // wave-uniform
                float addv[16];
#pragma unroll
                for (int v = 0; v < 16; ++v) addv[v] = bias;
                if (p.res) {
                    float t[16];
#pragma unroll
                    for (int v = 0; v < 16; ++v)
                        t[v] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(srd_r, vo_r, (mrow + (v & 3) + 8 * (v >> 2)) * p.ldres * 4, 0));
#pragma unroll
                    for (int v = 0; v < 16; ++v) addv[v] += t[v];
                }
                if (p.rowbias) {
                    // rows_per_image is a multiple of 8: the 8 rows (v & 3) + 4*lh of a row group v >> 2 belong to one image
                    // and that image index is wave-uniform -> four scalar divisions and four loads per tile
                    float t[4];
#pragma unroll
                    for (int g = 0; g < 4; ++g) {
                        const unsigned mr = (unsigned)(mrow + 8 * g);
                        const unsigned img = mr < (unsigned)p.M ? fast_div(mr, rpi, p.mg_rpi, p.sh_rpi) : 0u;
                        t[g] = nok ? p.rowbias[(size_t)img * p.ld_rowbias + n] : 0.f;
                    }
#pragma unroll
                    for (int v = 0; v < 16; ++v) addv[v] += t[v >> 2];
                }
                float vals[16];
#pragma unroll
                for (int v = 0; v < 16; ++v) {
                    const float val = (p.alpha * acc[r][c][v] + addv[v]) * p.out_scale;
                    vals[v] = val;
                    if (!p.egn_skip_raw)
                        __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, val), srd_o, vo_o,
                                                              (mrow + (v & 3) + 8 * (v >> 2)) * p.ldc * 4, 0);
                }
                if (p.egn_out) {
                    // GroupNorm (+ swish) of the output, here: rows_per_image == 16, so accumulator rows 0..15 (v < 8, both half-waves)
                    // are one image and rows 16..31 the next; a group is egn_cpg consecutive columns = consecutive lanes
                    float sa = 0.f, qa = 0.f, sb = 0.f, qb = 0.f;
#pragma unroll
                    for (int v = 0; v < 8; ++v) { sa += vals[v]; qa += vals[v] * vals[v]; }
#pragma unroll
                    for (int v = 8; v < 16; ++v) { sb += vals[v]; qb += vals[v] * vals[v]; }
                    sa += __shfl_xor(sa, 32); qa += __shfl_xor(qa, 32); sb += __shfl_xor(sb, 32); qb += __shfl_xor(qb, 32);
                    for (int o = 1; o < p.egn_cpg; o <<= 1) {
                        sa += __shfl_xor(sa, o); qa += __shfl_xor(qa, o); sb += __shfl_xor(sb, o); qb += __shfl_xor(qb, o);
                    }
                    const float inv_n = 1.0f / (16.0f * (float)p.egn_cpg);
                    const float ma = sa * inv_n, mb = sb * inv_n;
                    const float ra_ = rsqrtf(qa * inv_n - ma * ma + 1e-6f), rb_ = rsqrtf(qb * inv_n - mb * mb + 1e-6f);
                    const float ga = nok ? p.egn_gamma[n] : 0.f, be = nok ? p.egn_beta[n] : 0.f;
                    const float sca = ra_ * ga, sha = be - ma * ra_ * ga, scb = rb_ * ga, shb = be - mb * rb_ * ga;
                    const __amdgpu_buffer_rsrc_t srd_g =
                        __builtin_amdgcn_make_buffer_rsrc(reinterpret_cast<char *>(p.egn_out), 0, p.M * p.ldc * 4, 0x00020000);
#pragma unroll
                    for (int v = 0; v < 16; ++v) {
                        float y = v < 8 ? vals[v] * sca + sha : vals[v] * scb + shb;
                        if (p.egn_act) y = swish_e(y);
                        __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, y), srd_g, vo_o,
                                                              (mrow + (v & 3) + 8 * (v >> 2)) * p.ldc * 4, 0);
                    }
                }
                if (p.qstats) {
                    if (!nok || mrow + 32 > p.M) {  // edge tile: rows >= M / columns >= N contribute nothing
#pragma unroll
                        for (int v = 0; v < 16; ++v)
                            if (!nok || mrow + (v & 3) + 8 * (v >> 2) + 4 * lh >= p.M) vals[v] = 0.f;
                    }
                    emit_qstats(p, vals, mrow, n, nok, li, lh, p.qstats + (size_t)z * p.sQ);
                }
            }
        }
    }
}

// =====================================================================================================================
// Persistent variant of the fast path for SHORT K (the 1x1 convolutions, NIN layers and attention products: K = 256..512,
// 8..16 chunks).  With one workgroup per tile such a launch spends a third of its time outside the K loop (first loads,
// last stores: T ~ 14 us + 0.129 us x K per pair of tiles).  Here a workgroup walks tile ids blockIdx.x, blockIdx.x + gridDim.x,
// ... (flattened over the z slices of a batched product; the stride keeps the N tile, hence the weight rows and the XCD) and
// the software pipeline of the K loop runs ACROSS tiles: the last two iterations of a tile request the first two chunks of the
// next one (and the last one writes its chunk 0 to LDS), so only the first tile of a workgroup has a prologue and every epilogue
// runs with the next tile's loads in flight.  Same tiling, LDS layout, fragment order (hence bit-identical sums) and epilogue
// arithmetic as igemm_f32_fast_kernel.  Requirements: those of the fast path, n-major B, <= 32 taps, K >= 96.
template <int BM, int BN, int WM, int WN>
__global__ __launch_bounds__(64 * WM * WN) void igemm_f32_pers_kernel(const GemmParams p) {
    constexpr int NT = 64 * WM * WN;
    constexpr int RM = BM / (32 * WM), RN = BN / (32 * WN);
    constexpr int RPP = NT / 8;
    constexpr int AJ = BM / RPP, BJ = BN / RPP;
    constexpr int NP = AJ + BJ;
    constexpr int BUF = (BM + BN) * LDT;
    constexpr int OOB = (int)0x80000000u;
    static_assert(BM % RPP == 0 && BN % RPP == 0 && RM >= 1 && RN >= 1 && NP <= 14, "bad tile geometry");
    __shared__ __attribute__((aligned(16))) float smem[2 * BUF];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / WN, wn = wave % WN;

    const int nt = p.n_tiles;
    const int ids_per_z = p.flat_tiles ? p.m_tiles * nt : ((p.m_tiles + 7) / 8) * 8 * nt;
    const int total_ids = ids_per_z * p.z_count;
    auto tile_of = [&](int id_, int &z_, int &mtile_, int &ntile_) __attribute__((always_inline)) {
        z_ = id_ / ids_per_z;
        const int r_ = id_ - z_ * ids_per_z;
        if (p.flat_tiles) {
            mtile_ = r_ / nt;
            ntile_ = r_ - mtile_ * nt;
        } else {
            const int grp = r_ / (8 * nt);
            const int rem = r_ - grp * 8 * nt;
            ntile_ = rem >> 3;
            mtile_ = grp * 8 + (rem & 7);
        }
        return mtile_ < p.m_tiles;
    };
    int id = blockIdx.x, z_ld = 0, mt_ld = 0, ntile = 0;
    while (id < total_ids && !tile_of(id, z_ld, mt_ld, ntile)) id += (int)gridDim.x;
    if (id >= total_ids) return;
    const int n0 = ntile * BN;  // the same for every tile of this workgroup
#ifdef IGEMM_TRACE  // timing experiment (tools/pers_trace.py): thread 0 of every workgroup stamps the 100 MHz wall clock at entry and exit and
                    // counts its tiles, into the LAST 3 * gridDim.x floats of the RESIDUAL tensor (never written by the kernel; the last rows' results are wrong by construction)
    const unsigned long long tr_t0 = wall_clock64();
    int tr_tiles = 0;
#endif

    const int Cin = p.C0 + p.C1;
    const int taps = p.KH * p.KW;
    const int a_imgs = (p.M + p.OH * p.OW - 1) / (p.OH * p.OW);
    const int bias0 = (p.pad * p.W + p.pad) * p.lda0 * 4;
    const int bias1 = (p.pad * p.W + p.pad) * p.lda1 * 4;
    const int a0_bytes = a_imgs * p.H * p.W * p.lda0 * 4 + bias0;
    const int b_bytes = p.N * p.ldb * 4;
    auto make_a0 = [&](int z_) __attribute__((always_inline)) {
        const char *a0c = reinterpret_cast<const char *>(p.a0 + (size_t)z_ * p.sA);
        return __builtin_amdgcn_make_buffer_rsrc(const_cast<char *>(a0c - bias0), 0, a0_bytes, 0x00020000);
    };
    auto make_b = [&](int z_) __attribute__((always_inline)) {
        const char *bc = reinterpret_cast<const char *>(p.b + (size_t)z_ * p.sB);
        return __builtin_amdgcn_make_buffer_rsrc(const_cast<char *>(bc), 0, b_bytes, 0x00020000);
    };
    __amdgpu_buffer_rsrc_t srd_a0 = make_a0(z_ld);
    const char *a1c = reinterpret_cast<const char *>(p.C1 ? p.a1 : p.a0);
    const __amdgpu_buffer_rsrc_t srd_a1 =
        __builtin_amdgcn_make_buffer_rsrc(const_cast<char *>(a1c - bias1), 0, a_imgs * p.H * p.W * p.lda1 * 4 + bias1, 0x00020000);
    __amdgpu_buffer_rsrc_t srd_b = make_b(z_ld);

    const int arow = tid >> 3;
    const int kcol = (tid & 7) * 4;
    int rb0[AJ], rb1[AJ];
    unsigned inv[AJ];
    auto setup_rows = [&](int m0_) __attribute__((always_inline)) {  // row data of the tile whose loads are issued next
        const unsigned OHW = p.OH * p.OW;
        const unsigned rowbits = (1u << p.KW) - 1u;
#pragma unroll
        for (int j = 0; j < AJ; ++j) {
            const int m = m0_ + arow + RPP * j;
            rb0[j] = 0; rb1[j] = 0; inv[j] = 0xFFFFFFFFu;
            if (m < p.M) {
                const int img = (int)fast_div((unsigned)m, OHW, p.mg_ohw, p.sh_ohw);
                const int r = m - img * (int)OHW;
                const int oy = (int)fast_div((unsigned)r, (unsigned)p.OW, p.mg_ow, p.sh_ow);
                const int ox = r - oy * p.OW;
                const int iy0 = oy * p.stride - p.pad, ix0 = ox * p.stride - p.pad;
                const int pix = (img * p.H + iy0) * p.W + ix0;
                rb0[j] = pix * p.lda0 * 4 + bias0 + kcol * 4;
                rb1[j] = pix * p.lda1 * 4 + bias1 + kcol * 4;
                unsigned xb = 0;
                for (int tx = 0; tx < p.KW; ++tx) xb |= ((unsigned)(ix0 + tx) < (unsigned)p.W ? 0u : 1u) << tx;
                unsigned mk = 0;
                for (int ty = 0; ty < p.KH; ++ty) mk |= ((unsigned)(iy0 + ty) < (unsigned)p.H ? xb : rowbits) << (ty * p.KW);
                inv[j] = mk;
            }
        }
    };
    setup_rows(mt_ld * BM);
    int bvo[BJ];
#pragma unroll
    for (int j = 0; j < BJ; ++j) {
        const int n = n0 + arow + RPP * j;
        bvo[j] = n < p.N ? (n * p.ldb + kcol) * 4 : OOB;
    }

    // ---- wave-uniform state of the chunk whose loads are issued next; when it runs past the tile it rolls to the next tile ----
    int s_tap = 0, s_ty = 0, s_tx = 0, s_c0 = 0, s_kc = 0;
    const int nk = p.K / KC;  // >= 3 (launch_igemm)
    int z_nx = 0, m0_nx = 0;  // the tile the load side moved on to (valid when `more`)
    bool more = false;
    // gn_side_out (GemmParams): the chunk that sits in the staging registers -- requested before the last advance_k -- belongs to tile
    // rows st_m0.., channels st_c0..; st_m0 < 0 once the load side has run past the workgroup's last tile
    int ld_m0 = mt_ld * BM, st_m0 = 0, st_c0 = 0;
    auto roll = [&]() __attribute__((always_inline)) {
        int next = id + (int)gridDim.x, zz = 0, mt = 0, ntl = 0;
        while (next < total_ids && !tile_of(next, zz, mt, ntl)) next += (int)gridDim.x;
        more = next < total_ids;
        s_kc = 0; s_tap = 0; s_ty = 0; s_tx = 0; s_c0 = 0;
        ld_m0 = -1;
        if (more) {
            id = next;
            z_nx = zz; m0_nx = mt * BM;
            ld_m0 = m0_nx;
            if (p.z_count > 1) { srd_a0 = make_a0(zz); srd_b = make_b(zz); }
            setup_rows(m0_nx);
        } else {  // nothing left: the remaining requests of the pipeline fetch nothing (what they store is never read)
#pragma unroll
            for (int j = 0; j < AJ; ++j) inv[j] = 0xFFFFFFFFu;
#pragma unroll
            for (int j = 0; j < BJ; ++j) bvo[j] = OOB;
        }
    };
    auto advance_k = [&]() __attribute__((always_inline)) {
        st_m0 = ld_m0;
        st_c0 = s_c0;
        ++s_kc;
        ++s_tap;
        if (++s_tx == p.KW) { s_tx = 0; ++s_ty; }
        if (s_tap == taps) { s_tap = 0; s_ty = 0; s_tx = 0; s_c0 += KC; }
        if (s_kc == nk) roll();
    };

    f32x4 ra[AJ], rb[BJ];
    // gn_side_out: the affine pair of the thread's four channels travels WITH the chunk (requested behind its last A piece, consumed
    // when the chunk is staged) instead of being fetched at staging time, where the wave sat out its L2 round trip; a 128-row tile lies
    // inside one image when rows_per_image is a multiple of 128 (igemm_gn_side_ok), so one pair serves both of the thread's rows
    f32x4 tsc = {0.f, 0.f, 0.f, 0.f}, tsh = {0.f, 0.f, 0.f, 0.f};
    const bool side_any = p.gn_side_out != nullptr && ntile == 0;
    auto load_piece = [&](int pc) __attribute__((always_inline)) {
        if (pc < AJ) {
            const int j = pc;
            const int bad = __builtin_amdgcn_sbfe(inv[j], s_tap, 1);
            if (s_c0 < p.C0) {
                const int soff = ((s_ty * p.W + s_tx) * p.lda0 + s_c0) * 4;
                ra[j] = buf_load16(srd_a0, rb0[j] | (bad & OOB), soff);
            } else {
                const int soff = ((s_ty * p.W + s_tx) * p.lda1 + (s_c0 - p.C0)) * 4;
                ra[j] = buf_load16(srd_a1, rb1[j] | (bad & OOB), soff);
            }
            if (side_any && pc == AJ - 1 && ld_m0 >= 0) {
                const unsigned img = fast_div((unsigned)ld_m0, (unsigned)p.rows_per_image, p.mg_rpi, p.sh_rpi);
                const float *tb = p.gn_table + (size_t)img * 2 * Cin + s_c0 + kcol;
                tsc = *reinterpret_cast<const f32x4 *>(tb);
                tsh = *reinterpret_cast<const f32x4 *>(tb + Cin);
            }
        } else {
            const int j = pc - AJ;
            rb[j] = buf_load16(srd_b, bvo[j], (s_tap * Cin + s_c0) * 4);
        }
    };
    const bool side = p.gn_side_out != nullptr && ntile == 0;  // (every N tile stages the same A elements: the first one writes them)
    auto store_piece = [&](int pc, float *st) __attribute__((always_inline)) {
        if (pc < AJ) {
            *reinterpret_cast<f32x4 *>(&st[(arow + RPP * pc) * LDT + kcol]) = ra[pc];
            if (side) {  // the GroupNorm (+ swish) of the raw element, as gn_apply_kernel computes it
                const int m = st_m0 + arow + RPP * pc;
                if (st_m0 >= 0 && m < p.M) {
                    const int c = st_c0 + kcol;
                    const f32x4 sc = tsc, sh = tsh;
                    f32x4 y;
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        const float t = ra[pc][i] * sc[i] + sh[i];
                        y[i] = p.gn_act ? swish_e(t) : t;
                    }
                    if (IGEMM_OUT_AUX & 2) __builtin_nontemporal_store(y, reinterpret_cast<f32x4 *>(p.gn_side_out + (size_t)m * Cin + c));
                    else *reinterpret_cast<f32x4 *>(p.gn_side_out + (size_t)m * Cin + c) = y;
                }
            }
        } else {
            *reinterpret_cast<f32x4 *>(&st[BM * LDT + (arow + RPP * (pc - AJ)) * LDT + kcol]) = rb[pc - AJ];
        }
    };

    f32x16 acc[RM][RN];
    const int li = lane & 31, lh = lane >> 5;
    const int a_off = (wm * (BM / WM) + li) * LDT + lh * 4;
    const int b_offl = BM * LDT + (wn * (BN / WN) + li) * LDT + lh * 4;

    int z_ep = z_ld, m0_ep = mt_ld * BM;  // the tile being multiplied
#pragma unroll
    for (int pc = 0; pc < NP; ++pc) load_piece(pc);
    advance_k();
#pragma unroll
    for (int pc = 0; pc < NP; ++pc) store_piece(pc, smem);
#pragma unroll
    for (int pc = 0; pc < NP; ++pc) load_piece(pc);
    advance_k();
    __syncthreads();

    int it = 0;  // chunks multiplied so far by this workgroup: LDS stage it & 1 holds the current one
#ifndef IGEMM_PRIO
#define IGEMM_PRIO 0  // A/B (profiles/r05_notes.md): 1 the second-dispatched half of the grid takes wave priority 1 on its odd tiles, 2 on all of them
#endif
    int tile_no = 0;
    for (;;) {
        if (IGEMM_PRIO && blockIdx.x >= gridDim.x / 2) {
            if (IGEMM_PRIO == 2 || (tile_no & 1)) __builtin_amdgcn_s_setprio(1);
            else __builtin_amdgcn_s_setprio(0);
        }
        ++tile_no;
#pragma unroll
        for (int r = 0; r < RM; ++r)
#pragma unroll
            for (int c = 0; c < RN; ++c)
#pragma unroll
                for (int v = 0; v < 16; ++v) acc[r][c][v] = 0.f;
        for (int kc = 0; kc < nk; ++kc, ++it) {
            const float *cur = smem + (it & 1) * BUF;
            float *nxt = smem + ((it + 1) & 1) * BUF;
            f32x4 af[RM], bf[RN];
#pragma unroll
            for (int r = 0; r < RM; ++r) af[r] = *reinterpret_cast<const f32x4 *>(cur + a_off + r * 32 * LDT);
#pragma unroll
            for (int c = 0; c < RN; ++c) bf[c] = *reinterpret_cast<const f32x4 *>(cur + b_offl + c * 32 * LDT);
#pragma unroll
            for (int kb = 0; kb < KC / 8; ++kb) {
                f32x4 afn[RM], bfn[RN];
#pragma unroll
                for (int jj = 0; jj < 4; ++jj) {
#pragma unroll
                    for (int r = 0; r < RM; ++r)
#pragma unroll
                        for (int c = 0; c < RN; ++c)
                            acc[r][c] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[r][jj], bf[c][jj], acc[r][c], 0, 0, 0);
                    __builtin_amdgcn_sched_barrier(0);
                    const int gap = kb * 4 + jj;
                    if (gap < NP) {
                        store_piece(gap, nxt);
                        load_piece(gap);
                    } else if (gap == NP) {
                        advance_k();
                    }
                    if (jj == 2 && kb + 1 < KC / 8) {
#pragma unroll
                        for (int r = 0; r < RM; ++r) afn[r] = *reinterpret_cast<const f32x4 *>(cur + a_off + r * 32 * LDT + (kb + 1) * 8);
#pragma unroll
                        for (int c = 0; c < RN; ++c) bfn[c] = *reinterpret_cast<const f32x4 *>(cur + b_offl + c * 32 * LDT + (kb + 1) * 8);
                    }
                    __builtin_amdgcn_sched_barrier(0);
                }
                if (kb + 1 < KC / 8) {
#pragma unroll
                    for (int r = 0; r < RM; ++r) af[r] = afn[r];
#pragma unroll
                    for (int c = 0; c < RN; ++c) bf[c] = bfn[c];
                }
            }
            __syncthreads();
        }

        // ---- epilogue of tile (z_ep, m0_ep): as in igemm_f32_fast_kernel; the next tile's first chunks are in flight ----
        {
            char *outc = reinterpret_cast<char *>(p.out + (size_t)z_ep * p.sC);
            const __amdgpu_buffer_rsrc_t srd_o = __builtin_amdgcn_make_buffer_rsrc(outc, 0, p.M * p.ldc * 4, 0x00020000);
            const __amdgpu_buffer_rsrc_t srd_r =
                __builtin_amdgcn_make_buffer_rsrc(const_cast<char *>(reinterpret_cast<const char *>(p.res ? p.res + (size_t)z_ep * p.sR : p.out)), 0,
                                                  p.M * p.ldres * 4, 0x00020000);
            const unsigned rpi = p.rows_per_image;
#pragma unroll
            for (int c = 0; c < RN; ++c) {
                const int n = n0 + wn * (BN / WN) + c * 32 + li;
                const bool nok = n < p.N;
                const float bias = (p.bias && nok) ? p.bias[n] : 0.f;
                const int vo_o = nok ? (4 * lh * p.ldc + n) * 4 : OOB;
                const int vo_r = nok ? (4 * lh * p.ldres + n) * 4 : OOB;
#pragma unroll
                for (int r = 0; r < RM; ++r) {
                    const int mrow = m0_ep + wm * (BM / WM) + r * 32;  // wave-uniform
                    float addv[16];
#pragma unroll
                    for (int v = 0; v < 16; ++v) addv[v] = bias;
                    if (p.res) {
                        float t[16];
#pragma unroll
                        for (int v = 0; v < 16; ++v)
                            t[v] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(srd_r, vo_r, (mrow + (v & 3) + 8 * (v >> 2)) * p.ldres * 4, 0));
#pragma unroll
                        for (int v = 0; v < 16; ++v) addv[v] += t[v];
                    }
                    if (p.rowbias) {
                        float t[4];
#pragma unroll
                        for (int g = 0; g < 4; ++g) {
                            const unsigned mr = (unsigned)(mrow + 8 * g);
                            const unsigned img = mr < (unsigned)p.M ? fast_div(mr, rpi, p.mg_rpi, p.sh_rpi) : 0u;
                            t[g] = nok ? p.rowbias[(size_t)img * p.ld_rowbias + n] : 0.f;
                        }
#pragma unroll
                        for (int v = 0; v < 16; ++v) addv[v] += t[v >> 2];
                    }
                    float vals[16];
#pragma unroll
                    for (int v = 0; v < 16; ++v) {
                        const float val = (p.alpha * acc[r][c][v] + addv[v]) * p.out_scale;
                        vals[v] = val;
                        __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, val), srd_o, vo_o,
                                                              (mrow + (v & 3) + 8 * (v >> 2)) * p.ldc * 4, IGEMM_OUT_AUX);
                    }
                    if (p.qstats) {
                        if (!nok || mrow + 32 > p.M) {
#pragma unroll
                            for (int v = 0; v < 16; ++v)
                                if (!nok || mrow + (v & 3) + 8 * (v >> 2) + 4 * lh >= p.M) vals[v] = 0.f;
                        }
                        emit_qstats(p, vals, mrow, n, nok, li, lh, p.qstats + (size_t)z_ep * p.sQ);
                    }
                }
            }
        }
#ifdef IGEMM_TRACE
        ++tr_tiles;
#endif
        if (!more) break;
        z_ep = z_nx;
        m0_ep = m0_nx;
    }
#ifdef IGEMM_TRACE
    __syncthreads();
    if (threadIdx.x == 0 && p.res) {
        float *tr = const_cast<float *>(p.res) + (size_t)p.M * p.ldres - 3 * (size_t)gridDim.x;
        tr[3 * blockIdx.x] = (float)(tr_t0 & 0xFFFFFFull);
        tr[3 * blockIdx.x + 1] = (float)(wall_clock64() & 0xFFFFFFull);
        tr[3 * blockIdx.x + 2] = (float)tr_tiles;
    }
#endif
}

// ---- variant selection ------------------------------------------------------------------------------------------
// GDDIM_IGEMM_VARIANT (env, read once) overrides the large-tile kernel for A/B measurements:
//   0: 128x128, 8 waves (2x4)  [default]     1: 128x128, 4 waves (2x2)     2: 256x128, 8 waves (4x2) when M is large
static int big_variant() {
    int v = opts().igemm_variant;
    if (v < 0 || v > 2) v = 0;
    return v;
}

static void pick_tiles(const GemmParams &p, int batch_z, int &bm, int &bn) {
    auto blocks = [&](int BM_, int BN_) {
        return (long long)((p.M + BM_ - 1) / BM_) * ((p.N + BN_ - 1) / BN_) * batch_z;
    };
    // Largest tile that still gives every CU of the 256 at least ~1.5 blocks; the small spatial
    // levels (8x8, 4x4) of the U-Net otherwise leave most of the chip idle.
    const long long big_min = opts().igemm_bigmin, mid_min = opts().igemm_midmin;  // A/B only
    if (p.N > 64 && blocks(128, 128) >= big_min) { bm = 128; bn = 128; return; }
    if (blocks(128, 64) >= mid_min) { bm = 128; bn = 64; return; }
    bm = 64; bn = 64;
}

static bool fast_ok(const GemmParams &p);

// Intra-workgroup split-K factor of the 64x64 tile (igemm_f32_fast_kernel<..., KS>): when a launch has fewer than two tiles per
// CU and K is long, KS wave groups share a tile.  GDDIM_IGEMM_KS=1/2/4 overrides (A/B).
static int pick_ksplit(const GemmParams &p, int batch_z, int bm, int bn) {
    const int forced = opts().igemm_ks;
    if (bm != 64 || bn != 64 || !fast_ok(p) || p.b_kmajor || p.KH * p.KW > 32) return 1;
    const long long blocks = (long long)((p.M + 63) / 64) * ((p.N + 63) / 64) * batch_z;
    const int nk = p.K / KC;
    if (forced == 1 || forced == 2 || forced == 4) return nk >= 2 * forced ? forced : 1;
    if (blocks <= 512 && nk >= 32) return 4;
    if (blocks <= 1024 && nk >= 16) return 2;
    return 1;
}

// The persistent short-K kernel (igemm_f32_pers_kernel) takes a launch when it is eligible for the fast path with the 128x128
// tile, K is short (<= GDDIM_IGEMM_PERS_MAXK, default 1024) and there are more tiles than resident workgroups (2 per CU).
// GDDIM_IGEMM_NOPERSIST=1 routes everything back to one workgroup per tile (A/B).
static bool pers_ok(const GemmParams &p, int batch_z, int bm, int bn) {
    const int off = opts().igemm_nopersist, maxk = opts().igemm_pers_maxk;
    if (off || bm != 128 || bn != 128 || big_variant() != 0 || !fast_ok(p) || p.b_kmajor || p.KH * p.KW > 32) return false;
    if (p.K < 3 * KC || p.K > maxk || p.diag) return false;
    const long long tiles = (long long)((p.M + 127) / 128) * ((p.N + 127) / 128) * batch_z;
    return tiles > 2LL * device_cus() && tiles < (1LL << 24);
}

// The kernel symbol (as rocprofv3 prints it) that launch_igemm would pick for this problem.
const char *igemm_variant_name(const GemmParams &p, int batch_z) {
    int bm, bn;
    pick_tiles(p, batch_z, bm, bn);
    static thread_local char buf[96];
    int wmm = 2, wnn = 2;
    if (bm == 128 && bn == 128) {
        if (big_variant() == 2 && (long long)((p.M + 255) / 256) * ((p.N + bn - 1) / bn) * batch_z >= 512) { bm = 256; wmm = 4; wnn = 2; }
        else if (big_variant() != 1) { wmm = 2; wnn = 4; }
    }
    if (bm == 128 && bn == 64 && !opts().igemm_small4) { wmm = 4; wnn = 2; }
    const int ksp = pick_ksplit(p, batch_z, bm, bn);
    if (pers_ok(p, batch_z, bm, bn)) snprintf(buf, sizeof buf, "igemm_f32_pers_kernel<128,128,2,4>");
    else if (ksp > 1) snprintf(buf, sizeof buf, "igemm_f32_fast_kernel<64,64,2,2,false,false,%d>", ksp);
    else if (fast_ok(p)) snprintf(buf, sizeof buf, "igemm_f32_fast_kernel<%d,%d,%d,%d,%s>", bm, bn, wmm, wnn, p.b_kmajor ? "true" : "false");
    else snprintf(buf, sizeof buf, "igemm_f32_kernel<%d,%d,%d,%d,%s,%s>", bm, bn, wmm, wnn, p.b_kmajor ? "true" : "false", p.in_act ? "true" : "false");
    return buf;
}

static void fill_div(unsigned d, unsigned &magic, unsigned &shift) {
    unsigned s = 0;
    while ((1ull << s) < d) ++s;
    shift = s;
    magic = (unsigned)((((1ull << 32) * ((1ull << s) - d)) / d) + 1);
}

// Eligibility of the VALU-free fast path (see igemm_f32_fast_kernel).  GDDIM_IGEMM_NOFAST=1 forces the generic kernel.
static bool pers_ok(const GemmParams &p, int batch_z, int bm, int bn);
static void pick_tiles(const GemmParams &p, int batch_z, int &bm, int &bn);

bool igemm_gn_side_ok(const GemmParams &p, int batch_z) {
    if (batch_z != 1 || p.KH != 1 || p.KW != 1 || p.stride != 1 || p.pad != 0 || p.OH != p.H || p.OW != p.W) return false;
    if (!p.gn_table || !p.gn_side_out || p.rows_per_image <= 0 || p.rows_per_image % 128 != 0 || p.lda0 != p.C0 || (p.C1 > 0 && p.lda1 != p.C1)) return false;
    int bm, bn;
    pick_tiles(p, batch_z, bm, bn);
    return pers_ok(p, batch_z, bm, bn);
}

static bool fast_ok(const GemmParams &p) {
    if (opts().igemm_nofast || p.in_act) return false;
    const int Cin = p.C0 + p.C1;
    if (Cin % 32 != 0 || (p.C1 > 0 && p.C0 % 32 != 0) || p.KH * p.KW > 64 || (p.KH * p.KW > 32 && p.b_kmajor) || p.K % 32 != 0) return false;
    if (p.rowbias && p.rows_per_image % 8 != 0) return false;
    if ((long long)p.M * p.ldc * 4 + 64 >= (1LL << 31) || (p.res && (long long)p.M * p.ldres * 4 + 64 >= (1LL << 31))) return false;
    const long long ldmax = p.lda0 > p.lda1 ? p.lda0 : p.lda1;
    const long long imgs = (p.M + (long long)p.OH * p.OW - 1) / ((long long)p.OH * p.OW);
    const long long bias = ((long long)p.pad * p.W + p.pad) * ldmax * 4 + 64;
    const long long a_bytes = imgs * p.H * p.W * ldmax * 4 + bias;
    const long long b_bytes = (long long)(p.b_kmajor ? p.K : p.N) * p.ldb * 4;
    const long long soff_max = ((long long)(p.KH * p.W + p.KW) * ldmax + (p.C0 + p.C1)) * 4;
    return a_bytes + soff_max < (1LL << 31) && b_bytes + (long long)p.K * (p.b_kmajor ? p.ldb : 1) * 4 < (1LL << 31);
}

bool igemm_egn_ok(const GemmParams &p, int batch_z) {
    if (batch_z != 1 || p.rows_per_image != 16 || p.M % 16 != 0 || !fast_ok(p) || p.b_kmajor) return false;
    if (p.egn_cpg != 4 && p.egn_cpg != 8 && p.egn_cpg != 16 && p.egn_cpg != 32) return false;
    if (p.N % p.egn_cpg != 0) return false;
    int bm, bn;
    pick_tiles(p, batch_z, bm, bn);
    return !pers_ok(p, batch_z, bm, bn) && !(bm == 128 && bn == 128 && big_variant() == 2);
}

template <int BM, int BN, int WM, int WN>
static void launch_variant(const GemmParams &p, dim3 grid, hipStream_t s) {
    dim3 block(64 * WM * WN);
    if (fast_ok(p)) {
        if (p.b_kmajor) hipLaunchKernelGGL((igemm_f32_fast_kernel<BM, BN, WM, WN, true>), grid, block, 0, s, p);
        else if (p.KH * p.KW > 32) hipLaunchKernelGGL((igemm_f32_fast_kernel<BM, BN, WM, WN, false, true>), grid, block, 0, s, p);
        else hipLaunchKernelGGL((igemm_f32_fast_kernel<BM, BN, WM, WN, false>), grid, block, 0, s, p);
        return;
    }
    if (p.b_kmajor) hipLaunchKernelGGL((igemm_f32_kernel<BM, BN, WM, WN, true, false>), grid, block, 0, s, p);
    else if (p.in_act) hipLaunchKernelGGL((igemm_f32_kernel<BM, BN, WM, WN, false, true>), grid, block, 0, s, p);
    else hipLaunchKernelGGL((igemm_f32_kernel<BM, BN, WM, WN, false, false>), grid, block, 0, s, p);
}

// 64 B of device zeros per GPU (masked lanes of the staging loads read it); allocated once, never freed
const float *igemm_zero_buffer() {
    static const float *buf[64] = {nullptr};
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return nullptr;
    if (!buf[dev]) {
        float *p = nullptr;
        if (hipMalloc((void **)&p, 256) != hipSuccess) return nullptr;
        if (hipMemset(p, 0, 256) != hipSuccess) return nullptr;
        buf[dev] = p;
    }
    return buf[dev];
}

int launch_igemm(const GemmParams &p_in, int batch_z, hipStream_t s) {
    GemmParams p = p_in;
    GD_CHECK(p.C0 > 0 && p.M > 0 && p.N > 0, "igemm: empty problem (C0=%d M=%d N=%d)", p.C0, p.M, p.N);
    GD_CHECK(p.C0 % 4 == 0 && p.C1 % 4 == 0 && p.K % 4 == 0, "igemm: channel counts must be multiples of 4 (C0=%d C1=%d K=%d)",
             p.C0, p.C1, p.K);
    GD_CHECK(p.K == p.KH * p.KW * (p.C0 + p.C1), "igemm: K=%d != KH*KW*Cin=%d", p.K, p.KH * p.KW * (p.C0 + p.C1));
    GD_CHECK(!p.b_kmajor || p.N % 4 == 0, "igemm: k-major B needs N %% 4 == 0");
    GD_CHECK(p.ldb % 4 == 0 && p.lda0 % 4 == 0 && (p.C1 == 0 || p.lda1 % 4 == 0), "igemm: leading dims must be multiples of 4");
    GD_CHECK(!(p.b_kmajor && p.in_act), "igemm: in_act with k-major B is not instantiated");
    // staging offsets are 32-bit element counts
    {
        const long long imgs = (p.M + (long long)p.OH * p.OW - 1) / ((long long)p.OH * p.OW);
        const long long a_elems = imgs * p.H * p.W * (long long)(p.lda0 > p.lda1 ? p.lda0 : p.lda1);
        const long long b_elems = (long long)(p.b_kmajor ? p.K : p.N) * p.ldb;
        GD_CHECK(a_elems < (1LL << 31) && b_elems < (1LL << 31), "igemm: operand larger than 2^31 elements per z-slice");
    }
    p.diag = opts().igemm_diag;
    GD_CHECK(!p.egn_out || igemm_egn_ok(p, batch_z), "igemm: epilogue GroupNorm requested for a launch that cannot compute it");
    // (the engine decides these fusions once per batch size on the launch that covers the first image; a later image sub-range, or an
    // option table re-read in between, must not silently land on a kernel that ignores them: the stand-alone pass has been skipped)
    GD_CHECK(!p.gn_side_out || igemm_gn_side_ok(p, batch_z), "igemm: GroupNorm side output requested for a launch that cannot write it");
    fill_div((unsigned)(p.OH * p.OW), p.mg_ohw, p.sh_ohw);
    fill_div((unsigned)p.OW, p.mg_ow, p.sh_ow);
    fill_div((unsigned)(p.rows_per_image > 0 ? p.rows_per_image : 1), p.mg_rpi, p.sh_rpi);
    if (!p.zeros) p.zeros = igemm_zero_buffer();
    GD_CHECK(p.zeros != nullptr, "igemm: cannot allocate the zero buffer");
    int bm, bn;
    pick_tiles(p, batch_z, bm, bn);
    p.m_tiles = (p.M + bm - 1) / bm;
    p.n_tiles = (p.N + bn - 1) / bn;
    p.flat_tiles = p.m_tiles < 16;
    int gx = p.flat_tiles ? p.m_tiles * p.n_tiles : ((p.m_tiles + 7) / 8) * 8 * p.n_tiles;
    if (bm == 128 && bn == 128 && big_variant() == 2 && (long long)((p.M + 255) / 256) * p.n_tiles * batch_z >= 512) {
        p.m_tiles = (p.M + 255) / 256;
        p.flat_tiles = p.m_tiles < 16;
        gx = p.flat_tiles ? p.m_tiles * p.n_tiles : ((p.m_tiles + 7) / 8) * 8 * p.n_tiles;
        launch_variant<256, 128, 4, 2>(p, dim3(gx, 1, batch_z), s);
    } else if (pers_ok(p, batch_z, bm, bn)) {
        // persistent: 2 workgroups per CU walk the (z, tile) ids with a stride that keeps the N tile and the XCD
        p.z_count = batch_z;
        const int unit = p.flat_tiles ? p.n_tiles : 8 * p.n_tiles;
        const long long total = (long long)gx * batch_z;
        long long g = (2LL * device_cus() / unit) * unit;
        if (g <= 0 || g > total) g = total;
        hipLaunchKernelGGL((igemm_f32_pers_kernel<128, 128, 2, 4>), dim3((unsigned)g), dim3(512), 0, s, p);
    } else if (bm == 128 && bn == 128) {
        if (big_variant() == 1) launch_variant<128, 128, 2, 2>(p, dim3(gx, 1, batch_z), s);
        else launch_variant<128, 128, 2, 4>(p, dim3(gx, 1, batch_z), s);
    } else if (bm == 128) {
        if (opts().igemm_small4) launch_variant<128, 64, 2, 2>(p, dim3(gx, 1, batch_z), s);
        else launch_variant<128, 64, 4, 2>(p, dim3(gx, 1, batch_z), s);
    } else {
        const int ksp = pick_ksplit(p, batch_z, bm, bn);
        if (ksp == 4) hipLaunchKernelGGL((igemm_f32_fast_kernel<64, 64, 2, 2, false, false, 4>), dim3(gx, 1, batch_z), dim3(1024), 0, s, p);
        else if (ksp == 2) hipLaunchKernelGGL((igemm_f32_fast_kernel<64, 64, 2, 2, false, false, 2>), dim3(gx, 1, batch_z), dim3(512), 0, s, p);
        else launch_variant<64, 64, 2, 2>(p, dim3(gx, 1, batch_z), s);
    }
    GD_HIP(hipGetLastError());
    return 0;
}

}  // namespace gd
