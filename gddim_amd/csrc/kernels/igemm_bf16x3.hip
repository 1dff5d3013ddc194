// Split-precision implicit GEMM on the gfx950 bf16 matrix core ("bf16x3"), opt-in via gd_engine_set_precision.
//
// Every fp32 operand is written as hi + lo with hi = bf16_rne(x), lo = bf16_rne(x - hi) (|x - hi - lo| <= 2^-16 |x|) and the
// product is accumulated in fp32 as   a*b ~= a_hi*b_hi + a_hi*b_lo + a_lo*b_hi   -- three v_mfma_f32_32x32x16_bf16 per k16
// step; the dropped a_lo*b_lo term is ~2^-16 relative.  On the NCSN++ forward and through a 20-step sampler the result is as
// close to float64 as the plain fp32 path is (DESIGN.md "precision"), while the bf16 matrix core is 16x faster than the fp32
// MFMA, so the contraction costs 3/16 of the fp32 one and, unlike the fp32 MFMA, co-executes with VALU work.
//
// Same implicit-im2col addressing as igemm_f32_fast_kernel (per-row byte offsets + tap-validity bit masks, scalar tap/channel
// displacement in the buffer-load soffset, zero padding by the descriptor bounds check, 32-channel slab outermost / taps
// innermost), same double-buffered software pipeline (loads two chunks ahead, staging pieces between MFMA groups) and the same
// buffer-addressed epilogue.  Differences:
//   * A (activations, fp32 in HBM) is split into bf16 hi/lo when the staged registers are written to LDS (12 VALU per float4);
//   * B (weights) is split once on the host into two bf16 arrays [N][K];
//   * LDS holds four bf16 tiles per stage (A_hi, A_lo, B_hi, B_lo), rows padded from 64 to 80 bytes so that the 16-byte
//     fragment reads of a 16-lane group land on 16 distinct 16-byte slots (5*i mod 16) -- conflict free.
// Weight GEMMs only (convolutions, NIN, dense); the attention QK^T / PV products and the generic-path layers stay on the fp32
// kernels.
#include <cstdio>
#include <cstdlib>

#include "../common.h"
#include "../options.h"
#include "epilogue.h"

namespace gd {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));

#ifndef X3_ABL
#define X3_ABL 0  // compile-time ablation mask for timing experiments (tools/ablate_bf16x3.sh); 0 in the product build
#endif
constexpr int XKC = 32;   // K chunk (two k16 MFMA steps)
constexpr int XLD = 40;   // LDS row stride in bf16 elements: 64 B of data + 16 B pad = 80 B

__device__ __forceinline__ unsigned xfast_div(unsigned n, unsigned d, unsigned magic, unsigned shift) {
    if (d == 1) return n;
    const unsigned t = __umulhi(magic, n);
    return (t + ((n - t) >> 1)) >> (shift - 1);
}

// DIAG instantiations (GDDIM_IGEMM_DIAG, results wrong) switch parts of the kernel off for ablation timing:
//   1 no output stores, 2 one K chunk only, 8 no MFMAs, 16 no split / LDS stores, 32 no global loads, 64 no fragment reads
// PRE: the A operand is already split (rows of [lda0 bf16 hi | lda0 bf16 lo], written by gn_apply_kernel): staging is a plain
// 16-byte copy per thread -- no conversion work in the K loop.
// BACT: the B operand is an fp32 ACTIVATION matrix [N][ldb] (attention: K^T rows, V^T rows) instead of pre-split weights: it is
// split while staging exactly like A, and blockIdx.z walks the per-image slices (sA / sB / sC strides) of a batched product.
template <int BM, int BN, int WM, int WN, bool PRE = false, bool DIAG = false, bool BACT = false>
__global__ __launch_bounds__(64 * WM * WN) void igemm_bf16x3_kernel(const GemmParams p) {
    constexpr int NT = 64 * WM * WN;
    constexpr int RM = BM / (32 * WM), RN = BN / (32 * WN);
    constexpr int RPP = NT / 8;          // A rows staged per pass (8 threads x 16 B of fp32 per row)
    constexpr int AJ = BM / RPP;         // fp32 float4 staged per thread for A
    constexpr int BRP = BACT ? NT / 8 : NT / 4;  // B rows staged per pass (4 threads x 16 B of bf16, or 8 x 16 B of fp32, per row)
    constexpr int BJ = (BN + BRP - 1) / BRP;
    constexpr int NP = AJ + BJ;          // staging pieces per chunk (a B piece carries hi and lo)
    constexpr int NG = 6;                // MFMA groups per chunk: 2 k16 steps x 3 product terms
    constexpr int ALO = BM * XLD;        // A_lo tile offset
    constexpr int BBASE = 2 * BM * XLD;
    constexpr int STAGE = BBASE + 2 * BN * XLD;  // bf16 elements per LDS stage
    constexpr int OOB = (int)0x80000000u;
    int dg = DIAG ? p.diag : 0;
    if (DIAG) asm volatile("" : "+s"(dg));  // keep the switch word in an SGPR instead of re-reading the kernel arguments
    static_assert(BM % RPP == 0 && RM >= 1 && RN >= 1 && NP <= NG, "bad tile geometry");
    __shared__ __attribute__((aligned(16))) __bf16 smem[2 * STAGE];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);  // wave-uniform: keeps the epilogue row offsets in SGPRs
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

    const int Cin = p.C0 + p.C1;
    const int taps = p.KH * p.KW;
    const int a_imgs = (p.M + p.OH * p.OW - 1) / (p.OH * p.OW);
    const int bias0 = (p.pad * p.W + p.pad) * p.lda0 * 4;
    const int bias1 = (p.pad * p.W + p.pad) * p.lda1 * 4;
    const int z = BACT ? blockIdx.z : 0;
    const char *a0c = reinterpret_cast<const char *>(p.a0 + (size_t)z * p.sA);
    const __amdgpu_buffer_rsrc_t srd_a0 =
        __builtin_amdgcn_make_buffer_rsrc(const_cast<char *>(a0c - bias0), 0, a_imgs * p.H * p.W * p.lda0 * 4 + bias0, 0x00020000);
    const char *a1c = reinterpret_cast<const char *>(p.C1 ? p.a1 : p.a0);
    const __amdgpu_buffer_rsrc_t srd_a1 =
        __builtin_amdgcn_make_buffer_rsrc(const_cast<char *>(a1c - bias1), 0, a_imgs * p.H * p.W * p.lda1 * 4 + bias1, 0x00020000);
    const __amdgpu_buffer_rsrc_t srd_bh = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<char *>(reinterpret_cast<const char *>(p.b_hi)), 0, p.N * p.ldb * 2, 0x00020000);
    const __amdgpu_buffer_rsrc_t srd_bl = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<char *>(reinterpret_cast<const char *>(p.b_lo)), 0, p.N * p.ldb * 2, 0x00020000);
    const __amdgpu_buffer_rsrc_t srd_bf = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<char *>(reinterpret_cast<const char *>(BACT ? p.b + (size_t)z * p.sB : p.b)), 0, p.N * p.ldb * 4, 0x00020000);

    // ---- per-thread, loop-invariant row data (identical to igemm_f32_fast_kernel) ----
    // fp32 source: 8 threads x float4 per row.  Pre-split source: the first half of the workgroup copies the hi halves of the
    // rows, the second half the lo halves, 4 threads x 16 B per row (8 consecutive lanes = two rows of one tile, 80 B apart:
    // their b128 LDS stores are bank-conflict free)
    const int aplane = PRE ? tid / (NT / 2) : 0;
    const int arow = PRE ? (tid % (NT / 2)) >> 2 : tid >> 3;
    const int kcol = (tid & 7) * 4;           // fp32 source: float offset inside the 32-channel chunk
    const int aq = (tid & 3) * 8;             // pre-split source: bf16 offset inside the chunk (16 B)
    int rb0[AJ], rb1[AJ];
    unsigned inv[AJ];
    {
        const unsigned OHW = p.OH * p.OW;
        const unsigned rowbits = (1u << p.KW) - 1u;
#pragma unroll
        for (int j = 0; j < AJ; ++j) {
            const int m = m0 + arow + RPP * j;
            rb0[j] = 0; rb1[j] = 0; inv[j] = 0xFFFFFFFFu;
            if (m < p.M) {
                const int img = (int)xfast_div((unsigned)m, OHW, p.mg_ohw, p.sh_ohw);
                const int r = m - img * (int)OHW;
                const int oy = (int)xfast_div((unsigned)r, (unsigned)p.OW, p.mg_ow, p.sh_ow);
                const int ox = r - oy * p.OW;
                const int iy0 = oy * p.stride - p.pad, ix0 = ox * p.stride - p.pad;
                const int pix = (img * p.H + iy0) * p.W + ix0;
                rb0[j] = pix * p.lda0 * 4 + bias0 + (PRE ? (aplane * p.lda0 + aq) * 2 : kcol * 4);
                rb1[j] = pix * p.lda1 * 4 + bias1 + kcol * 4;
                unsigned xb = 0;
                for (int tx = 0; tx < p.KW; ++tx) xb |= ((unsigned)(ix0 + tx) < (unsigned)p.W ? 0u : 1u) << tx;
                unsigned mk = 0;
                for (int ty = 0; ty < p.KH; ++ty)
                    mk |= ((unsigned)(iy0 + ty) < (unsigned)p.H ? xb : rowbits) << (ty * p.KW);
                inv[j] = mk;
            }
        }
    }
    const int brow = BACT ? tid >> 3 : tid >> 2;  // B row inside a pass
    const int bq = (tid & 3) * 8;     // bf16 offset inside the 32-wide chunk (16 B)
    int bvo[BJ];
#pragma unroll
    for (int j = 0; j < BJ; ++j) {
        const int nl = brow + BRP * j;
        const int n = n0 + nl;
        bvo[j] = (nl < BN && n < p.N) ? (BACT ? (n * p.ldb + kcol) * 4 : (n * p.ldb + bq) * 2) : OOB;
    }

    int s_tap = 0, s_ty = 0, s_tx = 0, s_c0 = 0, s_kc = 0;
    const int nk = (DIAG && (dg & 2)) ? 1 : p.K / XKC;
    auto advance_k = [&]() {
        ++s_kc;
        ++s_tap;
        if (++s_tx == p.KW) { s_tx = 0; ++s_ty; }
        if (s_tap == taps) { s_tap = 0; s_ty = 0; s_tx = 0; s_c0 += XKC; }
    };

    f32x4 ra[AJ];
    u32x4 rbh[BJ], rbl[BJ];

    auto load_piece = [&](int pc) {
        if (DIAG && (dg & 32)) return;
        if (X3_ABL & 2) { if (pc < AJ) ra[pc] = f32x4{0.f, 0.f, 0.f, 0.f}; else { rbh[pc - AJ] = u32x4{0, 0, 0, 0}; rbl[pc - AJ] = u32x4{0, 0, 0, 0}; } return; }
        if (pc < AJ) {
            const int j = pc;
            const int bad = __builtin_amdgcn_sbfe(inv[j], s_tap < 32 ? s_tap : 31, 1) | (s_kc < nk ? 0 : -1);
            if (PRE) {
                const int soff = (s_ty * p.W + s_tx) * p.lda0 * 4 + s_c0 * 2;
                ra[j] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(srd_a0, rb0[j] | (bad & OOB), soff, 0));
            } else if (s_c0 < p.C0) {
                const int soff = ((s_ty * p.W + s_tx) * p.lda0 + s_c0) * 4;
                ra[j] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(srd_a0, rb0[j] | (bad & OOB), soff, 0));
            } else {
                const int soff = ((s_ty * p.W + s_tx) * p.lda1 + (s_c0 - p.C0)) * 4;
                ra[j] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(srd_a1, rb1[j] | (bad & OOB), soff, 0));
            }
        } else {
            const int j = pc - AJ;
            const int vo = s_kc < nk ? bvo[j] : OOB;
            if (BACT) {
                rbh[j] = __builtin_amdgcn_raw_buffer_load_b128(srd_bf, vo, (s_tap * Cin + s_c0) * 4, 0);  // four fp32 of row n
            } else {
                const int soff = (s_tap * Cin + s_c0) * 2;
                rbh[j] = __builtin_amdgcn_raw_buffer_load_b128(srd_bh, vo, soff, 0);
                rbl[j] = __builtin_amdgcn_raw_buffer_load_b128(srd_bl, vo, soff, 0);
            }
        }
    };
    // stage layout (bf16 elements): A_hi [BM][XLD] | A_lo [BM][XLD] | B_hi [BN][XLD] | B_lo [BN][XLD]
    auto store_piece = [&](int pc, __bf16 *st) {
        if (DIAG && (dg & 16)) return;
        if (X3_ABL & 4) return;
        if (pc < AJ && PRE) {
            *reinterpret_cast<f32x4 *>(st + aplane * ALO + (arow + RPP * pc) * XLD + aq) = ra[pc];
        } else if (pc < AJ) {
            const f32x4 v = ra[pc];
            bf16x4 h, l;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                h[i] = (__bf16)v[i];
                l[i] = (__bf16)(v[i] - (float)h[i]);
            }
            const int o = (arow + RPP * pc) * XLD + kcol;
            *reinterpret_cast<bf16x4 *>(st + o) = h;
            *reinterpret_cast<bf16x4 *>(st + ALO + o) = l;
        } else {
            const int j = pc - AJ;
            const int nl = brow + BRP * j;
            if (nl < BN && BACT) {
                const f32x4 v = __builtin_bit_cast(f32x4, rbh[j]);
                bf16x4 h, l;
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    h[i] = (__bf16)v[i];
                    l[i] = (__bf16)(v[i] - (float)h[i]);
                }
                const int o = BBASE + nl * XLD + kcol;
                *reinterpret_cast<bf16x4 *>(st + o) = h;
                *reinterpret_cast<bf16x4 *>(st + BN * XLD + o) = l;
            } else if (nl < BN) {
                const int o = BBASE + nl * XLD + bq;
                *reinterpret_cast<u32x4 *>(st + o) = rbh[j];
                *reinterpret_cast<u32x4 *>(st + BN * XLD + o) = rbl[j];
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
    const int a_off = (wm * (BM / WM) + li) * XLD + lh * 8;
    const int b_off = BBASE + (wn * (BN / WN) + li) * XLD + lh * 8;

#pragma unroll
    for (int pc = 0; pc < NP; ++pc) load_piece(pc);
    advance_k();
#pragma unroll
    for (int pc = 0; pc < NP; ++pc) store_piece(pc, smem);
#pragma unroll
    for (int pc = 0; pc < NP; ++pc) load_piece(pc);
    advance_k();
    __syncthreads();

    for (int kc = 0; kc < nk; ++kc) {
        const __bf16 *cur = smem + (kc & 1) * STAGE;
        __bf16 *nxt = smem + ((kc + 1) & 1) * STAGE;
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            bf16x8 ah[RM], al[RM], bh[RN], bl[RN];
            if ((X3_ABL & 8) || (DIAG && (dg & 64))) {
#pragma unroll
                for (int r = 0; r < RM; ++r) { ah[r] = bf16x8{}; al[r] = bf16x8{}; }
#pragma unroll
                for (int c = 0; c < RN; ++c) { bh[c] = bf16x8{}; bl[c] = bf16x8{}; }
            } else {
#pragma unroll
            for (int r = 0; r < RM; ++r) {
                ah[r] = *reinterpret_cast<const bf16x8 *>(cur + a_off + r * 32 * XLD + s * 16);
                al[r] = *reinterpret_cast<const bf16x8 *>(cur + ALO + a_off + r * 32 * XLD + s * 16);
            }
#pragma unroll
            for (int c = 0; c < RN; ++c) {
                bh[c] = *reinterpret_cast<const bf16x8 *>(cur + b_off + c * 32 * XLD + s * 16);
                bl[c] = *reinterpret_cast<const bf16x8 *>(cur + BN * XLD + b_off + c * 32 * XLD + s * 16);
            }
            }
#pragma unroll
            for (int term = 0; term < 3; ++term) {
#pragma unroll
                for (int r = 0; r < RM; ++r)
#pragma unroll
                    for (int c = 0; c < RN; ++c) {
                        const bf16x8 a = term == 2 ? al[r] : ah[r];
                        const bf16x8 b = term == 1 ? bl[c] : bh[c];
                        if ((X3_ABL & 16) || (DIAG && (dg & 8))) {
                            asm volatile("" ::"v"(a), "v"(b));
                        } else {
                            acc[r][c] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[r][c], 0, 0, 0);
                        }
                    }
                __builtin_amdgcn_sched_barrier(0);
                const int gap = s * 3 + term;
                if (gap < NP) {
                    store_piece(gap, nxt);
                    load_piece(gap);
                } else if (gap == NP) {
                    advance_k();
                }
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        if (NP == NG) advance_k();
        __syncthreads();
    }

    // ---- epilogue (kernels/epilogue.h) ----
    if (X3_ABL & 1) {
        float sink = 0.f;
#pragma unroll
        for (int r = 0; r < RM; ++r)
#pragma unroll
            for (int c = 0; c < RN; ++c)
#pragma unroll
                for (int v = 0; v < 16; ++v) sink += acc[r][c][v];
        if (sink == 1.2345e-30f) p.out[0] = sink;
    } else {
        epilogue_rows<RM, RN>(p, acc, m0 + wm * (BM / WM), n0 + wn * (BN / WN), li, lh, DIAG && (dg & 1),
                              BACT ? p.out + (size_t)z * p.sC : nullptr, BACT ? z : 0);
    }
}

// ---- host side ---------------------------------------------------------------------------------------------------------
bool igemm_bf16x3_ok(const GemmParams &p, int batch_z) {
    if (batch_z != 1 || p.b_kmajor || p.in_act || !p.b_hi || !p.b_lo) return false;
    const int Cin = p.C0 + p.C1;
    if (Cin % 32 != 0 || (p.C1 > 0 && p.C0 % 32 != 0) || p.KH * p.KW > 32 || p.K % 32 != 0 || p.ldb % 8 != 0) return false;
    if (p.a_presplit && (p.C1 != 0 || p.lda0 % 8 != 0)) return false;
    if (p.rowbias && p.rows_per_image % 8 != 0) return false;
    const long long ldmax = p.lda0 > p.lda1 ? p.lda0 : p.lda1;
    const long long imgs = (p.M + (long long)p.OH * p.OW - 1) / ((long long)p.OH * p.OW);
    const long long bias = ((long long)p.pad * p.W + p.pad) * ldmax * 4 + 64;
    const long long a_bytes = imgs * p.H * p.W * ldmax * 4 + bias;
    const long long soff_max = ((long long)(p.KH * p.W + p.KW) * ldmax + Cin) * 4;
    if (a_bytes + soff_max >= (1LL << 31) || (long long)p.N * p.ldb * 2 + (long long)p.K * 2 >= (1LL << 31)) return false;
    if ((long long)p.M * p.ldc * 4 + 64 >= (1LL << 31) || (p.res && (long long)p.M * p.ldres * 4 + 64 >= (1LL << 31))) return false;
    return true;
}

// batched activation x activation products (attention QK^T, PV, V^T): both operands fp32, split in the kernel
bool igemm_bf16x3_act_ok(const GemmParams &p, int batch_z) {
    if (opts().x3_noattn || batch_z < 1 || p.b_kmajor || p.in_act || !p.b || p.a_presplit || p.C1 != 0 || p.KH != 1 || p.KW != 1) return false;
    if (p.C0 % 32 != 0 || p.K != p.C0 || p.ldb % 4 != 0 || p.lda0 % 4 != 0 || p.rowbias) return false;
    if (p.res && (long long)p.M * p.ldres * 4 + 64 >= (1LL << 31)) return false;
    if ((long long)((p.M + 255) / 256) * ((p.N + 127) / 128) * batch_z < 256) return false;  // only the 256x128 tile is instantiated
    const long long a_bytes = (long long)p.M * p.lda0 * 4 + 64, b_bytes = (long long)p.N * p.ldb * 4 + 64;
    return a_bytes + (long long)p.K * 4 < (1LL << 31) && b_bytes + (long long)p.K * 4 < (1LL << 31) &&
           (long long)p.M * p.ldc * 4 + 64 < (1LL << 31);
}

int launch_igemm_bf16x3_act(const GemmParams &p_in, int batch_z, hipStream_t s) {
    GemmParams p = p_in;
    GD_CHECK(igemm_bf16x3_act_ok(p, batch_z), "igemm_bf16x3 (activation B): problem not eligible");
    unsigned mg, sh;
    auto fill = [](unsigned d, unsigned &magic, unsigned &shift) {
        unsigned q = 0;
        while ((1ull << q) < d) ++q;
        shift = q;
        magic = (unsigned)((((1ull << 32) * ((1ull << q) - d)) / d) + 1);
    };
    fill((unsigned)(p.OH * p.OW), p.mg_ohw, p.sh_ohw);
    fill((unsigned)p.OW, p.mg_ow, p.sh_ow);
    fill((unsigned)(p.rows_per_image > 0 ? p.rows_per_image : 1), p.mg_rpi, p.sh_rpi);
    (void)mg; (void)sh;
    p.m_tiles = (p.M + 255) / 256;
    p.n_tiles = (p.N + 127) / 128;
    p.flat_tiles = 1;
    hipLaunchKernelGGL((igemm_bf16x3_kernel<256, 128, 4, 2, false, false, true>), dim3(p.m_tiles * p.n_tiles, 1, batch_z), dim3(512), 0, s, p);
    GD_HIP(hipGetLastError());
    return 0;
}

static void xpick(const GemmParams &p, int &bm, int &bn) {
    auto blocks = [&](int BM_, int BN_) { return (long long)((p.M + BM_ - 1) / BM_) * ((p.N + BN_ - 1) / BN_); };
    if (p.N > 64 && blocks(256, 128) >= 256) { bm = 256; bn = 128; return; }
    if (p.N > 64 && blocks(128, 128) >= 192) { bm = 128; bn = 128; return; }
    bm = 64; bn = 64;
}

const char *igemm_bf16x3_name(const GemmParams &p) {
    int bm, bn;
    xpick(p, bm, bn);
    if (bm == 256) return p.a_presplit ? "igemm_bf16x3_kernel<256,128,4,2,pre>" : "igemm_bf16x3_kernel<256,128,4,2>";
    if (bm == 128) return p.a_presplit ? "igemm_bf16x3_kernel<128,128,2,4,pre>" : "igemm_bf16x3_kernel<128,128,2,4>";
    return p.a_presplit ? "igemm_bf16x3_kernel<64,64,2,2,pre>" : "igemm_bf16x3_kernel<64,64,2,2>";
}

static void xfill_div(unsigned d, unsigned &magic, unsigned &shift) {
    unsigned s = 0;
    while ((1ull << s) < d) ++s;
    shift = s;
    magic = (unsigned)((((1ull << 32) * ((1ull << s) - d)) / d) + 1);
}

int launch_igemm_bf16x3(const GemmParams &p_in, hipStream_t s) {
    GemmParams p = p_in;
    GD_CHECK(igemm_bf16x3_ok(p, 1), "igemm_bf16x3: problem not eligible");
    xfill_div((unsigned)(p.OH * p.OW), p.mg_ohw, p.sh_ohw);
    xfill_div((unsigned)p.OW, p.mg_ow, p.sh_ow);
    xfill_div((unsigned)(p.rows_per_image > 0 ? p.rows_per_image : 1), p.mg_rpi, p.sh_rpi);
    int bm, bn;
    xpick(p, bm, bn);
    p.m_tiles = (p.M + bm - 1) / bm;
    p.n_tiles = (p.N + bn - 1) / bn;
    p.flat_tiles = p.m_tiles < 16;
    const int gx = p.flat_tiles ? p.m_tiles * p.n_tiles : ((p.m_tiles + 7) / 8) * 8 * p.n_tiles;
    p.diag = opts().igemm_diag;
    const int variant = opts().x3_variant;  // A/B measurements only
    if (variant == 1 && bm == 256 && p.a_presplit) {  // 128x128 tiles, 4 waves, 80 KB LDS: two workgroups per CU
        p.m_tiles = (p.M + 127) / 128;
        p.flat_tiles = p.m_tiles < 16;
        const int gx1 = p.flat_tiles ? p.m_tiles * p.n_tiles : ((p.m_tiles + 7) / 8) * 8 * p.n_tiles;
        hipLaunchKernelGGL((igemm_bf16x3_kernel<128, 128, 2, 2, true>), dim3(gx1), dim3(256), 0, s, p);
    } else
    if (bm == 256 && p.diag && p.a_presplit) hipLaunchKernelGGL((igemm_bf16x3_kernel<256, 128, 4, 2, true, true>), dim3(gx), dim3(512), 0, s, p);
    else if (bm == 256 && p.diag) hipLaunchKernelGGL((igemm_bf16x3_kernel<256, 128, 4, 2, false, true>), dim3(gx), dim3(512), 0, s, p);
    else if (bm == 256 && p.a_presplit) hipLaunchKernelGGL((igemm_bf16x3_kernel<256, 128, 4, 2, true>), dim3(gx), dim3(512), 0, s, p);
    else if (bm == 256) hipLaunchKernelGGL((igemm_bf16x3_kernel<256, 128, 4, 2>), dim3(gx), dim3(512), 0, s, p);
    else if (bm == 128 && p.a_presplit) hipLaunchKernelGGL((igemm_bf16x3_kernel<128, 128, 2, 4, true>), dim3(gx), dim3(512), 0, s, p);
    else if (bm == 128) hipLaunchKernelGGL((igemm_bf16x3_kernel<128, 128, 2, 4>), dim3(gx), dim3(512), 0, s, p);
    else if (p.a_presplit) hipLaunchKernelGGL((igemm_bf16x3_kernel<64, 64, 2, 2, true>), dim3(gx), dim3(256), 0, s, p);
    else hipLaunchKernelGGL((igemm_bf16x3_kernel<64, 64, 2, 2>), dim3(gx), dim3(256), 0, s, p);
    GD_HIP(hipGetLastError());
    return 0;
}

}  // namespace gd
