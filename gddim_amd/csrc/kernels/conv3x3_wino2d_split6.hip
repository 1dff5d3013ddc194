// 3x3 / stride 1 / pad 1 convolution through the two-dimensional Winograd transform F(2x2, 3x3) with every component GEMM in the EXACT
// three-piece bf16 split on the bf16 matrix core (v_mfma_f32_32x32x16_bf16), fp32 accumulation -- the prototype VERDICT r5 item 2 asks for
// (profiles/r05_split6_error_study.md: the one split form at fp32-class error; reference operator: nn.Conv in cld_jax/models/layers.py:95-107).
//
//     V = B^T d B,  U = G g G^T,  M_ij[tile][n] = sum_c V_ij[tile][c] U_ij[n][c],  Y = A^T M A          (conv3x3_wino2d_f32.hip)
//     x = x0 + x1 + x2 with x0 = trunc_bf16(x), x1 = trunc_bf16(x - x0), x2 = x - x0 - x1: EXACT for a 24-bit significand (8 + 8 + 8 bits,
//     every remainder is formed without rounding); the weights are split once at load with round-to-nearest pieces.
//     v u ~= v1 u1 + v2 u0 + v0 u2 + v1 u0 + v0 u1 + v0 u0   (every dropped term <= 2^-24 |v u|; a bf16 x bf16 product is exact in fp32)
// Six bf16 MFMAs per component and 16 channels instead of eight fp32 ones: 24 / 16 = 1.5 fp32-MFMA-equivalents of matrix time per output
// against 3 for the production hybrid kernel.  What it costs is everything around the matrix core, which all doubles per unit time
// (profiles/r05_notes.md): the activation-side transform AND the split are plain fp32 / integer VALU (free beside bf16 MFMAs as long as they
// are not packed instructions: tools/valu_share.hip), 96 KB of V pieces through LDS and 96 KB of U pieces from L2 per 16-channel chunk.
//
// Workgroup = 256 threads = 4 wave64 (one per SIMD, the 512-register budget), one per CU.  Tile = 64 Winograd tiles (an 8 x 8 block of
// 2 x 2-output tiles = 16 x 16 output pixels of ONE image) x 64 outputs x 16 components.  Wave w owns component COLUMN j = w, i.e. the four
// components (i, w), i = 0..3, of both 32-tile blocks and both 32-output blocks: 16 accumulators of 32 x 32 (256 registers).
// K = Cin is walked in chunks of 16 channels, each in two PHASES h = 0, 1 = component rows {2h, 2h + 1}: the V stage of a phase (8 components
// x 3 pieces x 64 tiles x 16 channels of bf16 = 48 KB) is read while the other stage is written, 48 MFMAs per wave and ONE barrier per phase.
//   LDS   V[h][i & 1][j][piece][k half][tile][8 bf16]: an A fragment (32 tiles x 16 channels) is one contiguous KB, ds_read_b128, conflict-free;
//         the producing thread (tile, channel quad) writes 8 bytes per (component, piece): ds_write_b64, 16 lanes = 8 tiles x 2 quads = 128
//         contiguous bytes.
//   U     straight from global / L2 into the B registers, fragment order [n tile][chunk][h][wave][i & 1][n block][piece][lane][8 bf16]
//         (split6_weights_kernel): a wave's 12 fragments of a phase are 12 contiguous KB; a ring of four 3-fragment slots, refilled as soon
//         as a slot's 12 MFMAs are issued (three groups = 1.1k cycles ahead).
//   d     thread (tile, quad) loads the three patch rows its component-row half needs (12 x 16 B, zero padding by the buffer bounds check),
//         one phase ahead; B^T rows in registers, then per component: column combination (4 adds), split (8 and / 8 sub / 6 perm), 3 stores.
//   loop  the last 12-MFMA group of a phase runs AFTER the barrier, under the first fragment reads of the next phase (its operands are
//         registers): the matrix pipe never waits for LDS.
// Epilogue: the row transform A^T . is in-lane (a wave holds all four i of its column), the column transform . A crosses the four waves
// through LDS (128 KB, the idle stages); wave (tile block, n block) finishes 32 tiles x 32 outputs: bias, residual, scale, store.
// Shapes: H % 16 == 0, W % 16 == 0, Cin % 16 == 0, N % 64 == 0.
#include <cstdio>
#include <cstdlib>

#include "../common.h"

namespace gd {

typedef float s6f32x16 __attribute__((ext_vector_type(16)));
typedef float s6f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int s6u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int s6u32x2 __attribute__((ext_vector_type(2)));
typedef __bf16 s6bf16x8 __attribute__((ext_vector_type(8)));

#ifndef S6_ABL
#define S6_ABL 0  // timing ablations (results wrong): 1 no patch loads in the loop, 2 no V stores, 4 no A fragment reads, 8 no barrier in the loop,
                  // 16 no transform / split arithmetic, 32 no U loads in the loop, 64 no MFMAs
#endif

constexpr int S6_STAGE = 8 * 3 * 2 * 64 * 16;  // bytes of one V stage (8 components x 3 pieces x 2 k halves x 64 tiles x 16 B)
constexpr int S6_SMEM = 131072;                // two stages (96 KB) during the K loop, the 128 KB exchange area in the epilogue
constexpr int S6_OOB = (int)0x80000000u;

struct Split6Params {
    const float *x;
    const unsigned short *u;  // split6_weights_kernel's fragments
    const float *bias, *res;
    float *out;
    int B, H, W, C, N;
    int bx_tiles, tiles_per_image, m_tiles, n_tiles, nk;  // 16x16-pixel blocks per row / per image, in the batch; 64-output tiles; chunks
    float out_scale;
};

// scalar fp32 VALU on purpose: beside bf16 MFMAs a v_pk_add_f32 costs issue time, v_add_f32 does not (MI355X_MICROARCH.md), and the
// compiler SLP-packs adjacent fp32 adds by itself
__device__ __forceinline__ float s6_add(float a, float b) {
    float r;
    asm("v_add_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
}
__device__ __forceinline__ float s6_sub(float a, float b) {
    float r;
    asm("v_sub_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
}
__device__ __forceinline__ s6f32x4 s6_add4(s6f32x4 a, s6f32x4 b) { return s6f32x4{s6_add(a[0], b[0]), s6_add(a[1], b[1]), s6_add(a[2], b[2]), s6_add(a[3], b[3])}; }
__device__ __forceinline__ s6f32x4 s6_sub4(s6f32x4 a, s6f32x4 b) { return s6f32x4{s6_sub(a[0], b[0]), s6_sub(a[1], b[1]), s6_sub(a[2], b[2]), s6_sub(a[3], b[3])}; }

// (x0, x1) -> one dword holding the top halves: bf16(x0) in bits 15:0, bf16(x1) in 31:16 (truncation)
__device__ __forceinline__ unsigned s6_pack_hi(float x0, float x1) {
    return __builtin_amdgcn_perm(__builtin_bit_cast(unsigned, x1), __builtin_bit_cast(unsigned, x0), 0x07060302u);
}
__device__ __forceinline__ float s6_trunc(float x) { return __builtin_bit_cast(float, __builtin_bit_cast(unsigned, x) & 0xFFFF0000u); }

// the three bf16 pieces of four channels: out[piece] = 8 bytes in channel order
__device__ __forceinline__ void s6_split4(s6f32x4 v, s6u32x2 out[3]) {
    float r1[4], r2[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) r1[e] = s6_sub(v[e], s6_trunc(v[e]));
#pragma unroll
    for (int e = 0; e < 4; ++e) r2[e] = s6_sub(r1[e], s6_trunc(r1[e]));
    out[0] = s6u32x2{s6_pack_hi(v[0], v[1]), s6_pack_hi(v[2], v[3])};
    out[1] = s6u32x2{s6_pack_hi(r1[0], r1[1]), s6_pack_hi(r1[2], r1[3])};
    out[2] = s6u32x2{s6_pack_hi(r2[0], r2[1]), s6_pack_hi(r2[2], r2[3])};
}

__device__ __forceinline__ s6f32x4 s6_load16(__amdgpu_buffer_rsrc_t rsrc, int voffset, int soffset) {
    return __builtin_bit_cast(s6f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsrc, voffset, soffset, 0));
}

__global__ void __launch_bounds__(256) conv3x3_wino2d_split6_kernel(const Split6Params p) {
    __shared__ __attribute__((aligned(16))) unsigned char smem[S6_SMEM];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);

    // ---- block -> (m tile, n tile).  Consecutive workgroup ids go round the 8 XCDs: the workgroups of one XCD share ONE n tile, i.e. one
    // slice of the weight fragments in that XCD's L2 (the whole U of a 256 -> 256 layer is 6.3 MB, a slice 1.6 MB) ----
    int mtile, ntile;
    const int total = p.m_tiles * p.n_tiles;
    if ((8 % p.n_tiles) == 0 && total % 8 == 0) {
        const int per = 8 / p.n_tiles;  // XCDs per n tile
        const int xcd = blockIdx.x & 7, grp = blockIdx.x >> 3;
        ntile = xcd % p.n_tiles;
        mtile = grp * per + xcd / p.n_tiles;
    } else {
        mtile = blockIdx.x / p.n_tiles;
        ntile = blockIdx.x - mtile * p.n_tiles;
    }
    if (mtile >= p.m_tiles) return;
    const int img = mtile / p.tiles_per_image;
    const int brem = mtile - img * p.tiles_per_image;
    const int by = brem / p.bx_tiles, bx = brem - by * p.bx_tiles;
    const int Y0 = by * 16, X0 = bx * 16;
    const int H = p.H, W = p.W, C = p.C;

    // ---- producer identity: (tile, channel quad).  lane bits: 0 = quad bit 0, 1..3 = tile bits 0..2, 4 = quad bit 1, 5 = tile bit 3; the
    // wave is tile bits 4..5 (two tile rows of the 8 x 8 block) ----
    const int tile = (wave << 4) | (((lane >> 5) & 1) << 3) | ((lane >> 1) & 7);
    const int q = (lane & 1) | (((lane >> 4) & 1) << 1);
    const int ty = tile >> 3, tx = tile & 7;
    int voff[4][4];
#pragma unroll
    for (int r = 0; r < 4; ++r)
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            const int y = Y0 + 2 * ty + r - 1, x = X0 + 2 * tx + s - 1;
            const bool ok = (unsigned)y < (unsigned)H && (unsigned)x < (unsigned)W;
            voff[r][s] = ok ? (((img * H + y) * W + x) * C + q * 4) * 4 : S6_OOB;
        }
    const __amdgpu_buffer_rsrc_t srd_x = __builtin_amdgcn_make_buffer_rsrc(const_cast<char *>(reinterpret_cast<const char *>(p.x)), 0,
                                                                          p.B * H * W * C * 4, 0x00020000);
    // the wave's weight fragments: [n tile][phase][wave][12 fragments of 1 KB]
    const size_t u_tile_bytes = (size_t)p.nk * 2 * 4 * 12 * 1024;
    const __amdgpu_buffer_rsrc_t srd_u = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<char *>(reinterpret_cast<const char *>(p.u)) + (size_t)ntile * u_tile_bytes + (size_t)wave * 12 * 1024, 0,
        (int)(u_tile_bytes - (size_t)wave * 12 * 1024), 0x00020000);
    const int u_vo = lane * 16;

    // LDS byte offsets: the thread's 8-byte store slot / the lane's 16-byte fragment slot inside a (component, piece) block of 2 KB
    const int vw_off = ((q >> 1) * 64 + tile) * 16 + (q & 1) * 8;
    const int ar_off = ((lane >> 5) * 64 + (lane & 31)) * 16;
    constexpr int CS_BYTES = 3 * 2048;  // one component: 3 pieces x 2 k halves x 64 tiles x 16 B

    s6f32x16 acc[4][2][2];  // [component row i][tile block][n block]
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
            for (int b = 0; b < 2; ++b)
#pragma unroll
                for (int v = 0; v < 16; ++v) acc[i][a][b][v] = 0.f;

    const int NP = 2 * p.nk;  // phases
    s6f32x4 pr[3][4];         // the three patch rows of the half to produce next
    s6f32x4 rc[2][4];         // B^T rows (2 component rows x 4 patch columns)
    s6bf16x8 Areg[2][2][3];   // [i & 1][tile block][piece]
    s6bf16x8 Ureg[4][3];      // ring: slot = (i & 1) * 2 + n block

    // patch rows of half hp of chunk kc -> pr
    auto load_patch = [&](int kc, int hp) __attribute__((always_inline)) {
        const int so = __builtin_amdgcn_readfirstlane(kc * 64);
#pragma unroll
        for (int r = 0; r < 3; ++r)
#pragma unroll
            for (int s = 0; s < 4; ++s) pr[r][s] = hp == 0 ? s6_load16(srd_x, voff[r][s], so) : s6_load16(srd_x, voff[r + 1][s], so);
    };
    // B^T rows of half hp: hp = 0: d0 - d2, d1 + d2 (pr = rows 0, 1, 2); hp = 1: d2 - d1, d1 - d3 (pr = rows 1, 2, 3)
    auto row_combine = [&](int hp) __attribute__((always_inline)) {
        if (S6_ABL & 16) {
#pragma unroll
            for (int s = 0; s < 4; ++s) { rc[0][s] = pr[0][s]; rc[1][s] = pr[1][s]; }
            return;
        }
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            if (hp == 0) { rc[0][s] = s6_sub4(pr[0][s], pr[2][s]); rc[1][s] = s6_add4(pr[1][s], pr[2][s]); }
            else { rc[0][s] = s6_sub4(pr[1][s], pr[0][s]); rc[1][s] = s6_sub4(pr[0][s], pr[2][s]); }
        }
    };
    // component (i2, j) of the half in rc -> three 8-byte stores into stage `st`
    auto produce = [&](int i2, int j, unsigned char *st) __attribute__((always_inline)) {
        s6f32x4 v;
        if (S6_ABL & 16) v = rc[i2][j];
        else if (j == 0) v = s6_sub4(rc[i2][0], rc[i2][2]);
        else if (j == 1) v = s6_add4(rc[i2][1], rc[i2][2]);
        else if (j == 2) v = s6_sub4(rc[i2][2], rc[i2][1]);
        else v = s6_sub4(rc[i2][1], rc[i2][3]);
        s6u32x2 pc[3];
        if (S6_ABL & 16) { pc[0] = s6u32x2{__builtin_bit_cast(unsigned, v[0]), __builtin_bit_cast(unsigned, v[1])}; pc[1] = pc[0]; pc[2] = pc[0]; }
        else s6_split4(v, pc);
        if (S6_ABL & 2) { asm volatile("" ::"v"(pc[0]), "v"(pc[1]), "v"(pc[2])); return; }
#pragma unroll
        for (int pi = 0; pi < 3; ++pi) *reinterpret_cast<s6u32x2 *>(st + (i2 * 4 + j) * CS_BYTES + pi * 2048 + vw_off) = pc[pi];
    };
    auto read_a = [&](int i2, const unsigned char *st) __attribute__((always_inline)) {
        if (S6_ABL & 4) return;
#pragma unroll
        for (int tb = 0; tb < 2; ++tb)
#pragma unroll
            for (int pi = 0; pi < 3; ++pi)
                Areg[i2][tb][pi] = *reinterpret_cast<const s6bf16x8 *>(st + (i2 * 4 + wave) * CS_BYTES + pi * 2048 + tb * 512 + ar_off);
    };
    // slot (i2, nb) of the ring <- phase ph
    auto load_u = [&](int ph, int slot) __attribute__((always_inline)) {
        if ((S6_ABL & 32) && ph >= 2) return;
        const int so = __builtin_amdgcn_readfirstlane(ph * (4 * 12 * 1024) + slot * 3 * 1024);
#pragma unroll
        for (int pi = 0; pi < 3; ++pi)
            Ureg[slot][pi] = __builtin_bit_cast(s6bf16x8, __builtin_amdgcn_raw_buffer_load_b128(srd_u, u_vo, so + pi * 1024, 0));
    };

    // six products per (tile block): a1 u1, a2 u0, a0 u2, a1 u0, a0 u1, a0 u0 -- the small ones first; the two tile blocks alternate so that
    // consecutive MFMAs never share an accumulator
    constexpr int PA[6] = {1, 2, 0, 1, 0, 0}, PB[6] = {1, 0, 2, 0, 1, 0};
#define S6_MFMA(I, I2, NB, T, TB)                                                                                                      \
    if (!(S6_ABL & 64)) acc[I][TB][NB] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(Areg[I2][TB][PA[T]], Ureg[(I2) * 2 + (NB)][PB[T]], acc[I][TB][NB], 0, 0, 0)
#define S6_HALF_GROUP(I, I2, NB, T0)                                                                    \
    S6_MFMA(I, I2, NB, T0, 0); S6_MFMA(I, I2, NB, T0, 1); S6_MFMA(I, I2, NB, T0 + 1, 0); S6_MFMA(I, I2, NB, T0 + 1, 1); \
    S6_MFMA(I, I2, NB, T0 + 2, 0); S6_MFMA(I, I2, NB, T0 + 2, 1)

    unsigned char *stage0 = smem, *stage1 = smem + S6_STAGE;

    // ---- prologue: V of phase 0 -> stage 0; the patch rows phase 0 will turn into phase 1's V; the weight slots 0..2 of phase 0 ----
    load_patch(0, 0);
    load_u(0, 0);
    load_u(0, 1);
    load_u(0, 2);
    row_combine(0);
#pragma unroll
    for (int i2 = 0; i2 < 2; ++i2)
#pragma unroll
        for (int j = 0; j < 4; ++j) produce(i2, j, stage0);
    load_patch(0, 1);
    __syncthreads();

    // One phase.  HH: component-row half being multiplied (stage HH is read, stage 1 - HH is written with the next phase's V).
#define S6_PHASE(HH, PHASE_INDEX, FIRST)                                                                                               \
    {                                                                                                                                  \
        const int ph = (PHASE_INDEX);                                                                                                  \
        unsigned char *rd = (HH) ? stage1 : stage0, *wr = (HH) ? stage0 : stage1;                                                      \
        const bool more1 = ph + 1 < NP, more2 = ph + 2 < NP;                                                                           \
        read_a(0, rd);                                                                                                                 \
        if (!(FIRST)) { S6_HALF_GROUP(2 * (1 - (HH)) + 1, 1, 1, 0); S6_HALF_GROUP(2 * (1 - (HH)) + 1, 1, 1, 3); }                      \
        __builtin_amdgcn_sched_barrier(0);                                                                                             \
        load_u(ph, 3);                                                                                                                 \
        if (more1) row_combine(1 - (HH));                                                                                              \
        if (more2 && !((S6_ABL & 1) && ph >= 2)) load_patch((ph + 2) >> 1, (HH));                                                      \
        __builtin_amdgcn_sched_barrier(0);                                                                                             \
        S6_HALF_GROUP(2 * (HH), 0, 0, 0);                                                                                              \
        if (more1) { produce(0, 0, wr); }                                                                                              \
        __builtin_amdgcn_sched_barrier(0);                                                                                             \
        S6_HALF_GROUP(2 * (HH), 0, 0, 3);                                                                                              \
        if (more1) { produce(0, 1, wr); }                                                                                              \
        if (more1) load_u(ph + 1, 0);                                                                                                  \
        read_a(1, rd);                                                                                                                 \
        __builtin_amdgcn_sched_barrier(0);                                                                                             \
        S6_HALF_GROUP(2 * (HH), 0, 1, 0);                                                                                              \
        if (more1) { produce(0, 2, wr); produce(0, 3, wr); }                                                                           \
        __builtin_amdgcn_sched_barrier(0);                                                                                             \
        S6_HALF_GROUP(2 * (HH), 0, 1, 3);                                                                                              \
        if (more1) { produce(1, 0, wr); }                                                                                              \
        if (more1) load_u(ph + 1, 1);                                                                                                  \
        __builtin_amdgcn_sched_barrier(0);                                                                                             \
        S6_HALF_GROUP(2 * (HH) + 1, 1, 0, 0);                                                                                          \
        if (more1) { produce(1, 1, wr); produce(1, 2, wr); }                                                                           \
        __builtin_amdgcn_sched_barrier(0);                                                                                             \
        S6_HALF_GROUP(2 * (HH) + 1, 1, 0, 3);                                                                                          \
        if (more1) { produce(1, 3, wr); }                                                                                              \
        if (more1) load_u(ph + 1, 2);                                                                                                  \
        __builtin_amdgcn_sched_barrier(0);                                                                                             \
        if (!(S6_ABL & 8)) __syncthreads();                                                                                            \
    }

    S6_PHASE(0, 0, true);
    S6_PHASE(1, 1, false);
    for (int kc = 1; kc < p.nk; ++kc) {
        S6_PHASE(0, 2 * kc, false);
        S6_PHASE(1, 2 * kc + 1, false);
    }
    // the deferred group of the last phase (component row 3, n block 1)
    S6_HALF_GROUP(3, 1, 1, 0);
    S6_HALF_GROUP(3, 1, 1, 3);
#undef S6_PHASE
#undef S6_HALF_GROUP
#undef S6_MFMA

    // ---- epilogue.  In-lane: z_a = sum_i A^T[a][i] M_i (a = 0: m0 + m1 + m2, a = 1: m1 - m2 - m3).  Exchange through LDS:
    // zx[wave j][a][tile block][n block][register quad][lane][4] (lane-contiguous 16-byte slots: conflict-free both ways) ----
    float *zx = reinterpret_cast<float *>(smem);
#pragma unroll
    for (int tb = 0; tb < 2; ++tb)
#pragma unroll
        for (int nb = 0; nb < 2; ++nb)
#pragma unroll
            for (int rq = 0; rq < 4; ++rq) {
                s6f32x4 z0, z1;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const int v = rq * 4 + e;
                    z0[e] = (acc[0][tb][nb][v] + acc[1][tb][nb][v]) + acc[2][tb][nb][v];
                    z1[e] = (acc[1][tb][nb][v] - acc[2][tb][nb][v]) - acc[3][tb][nb][v];
                }
                *reinterpret_cast<s6f32x4 *>(zx + ((((wave * 2 + 0) * 2 + tb) * 2 + nb) * 4 + rq) * 256 + lane * 4) = z0;
                *reinterpret_cast<s6f32x4 *>(zx + ((((wave * 2 + 1) * 2 + tb) * 2 + nb) * 4 + rq) * 256 + lane * 4) = z1;
            }
    __syncthreads();
    // wave (tb, nb) finishes 32 tiles x 32 outputs: y_a0 = z_a[0] + z_a[1] + z_a[2], y_a1 = z_a[1] - z_a[2] - z_a[3]
    {
        const int tb = wave >> 1, nb = wave & 1;
        const int n = ntile * 64 + nb * 32 + (lane & 31);
        const float bias = p.bias ? p.bias[n] : 0.f;
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
            for (int rq = 0; rq < 4; ++rq) {
                s6f32x4 zj[4];
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    zj[j] = *reinterpret_cast<const s6f32x4 *>(zx + ((((j * 2 + a) * 2 + tb) * 2 + nb) * 4 + rq) * 256 + lane * 4);
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const int v = rq * 4 + e;
                    const int t = tb * 32 + (v & 3) + 8 * (v >> 2) + 4 * (lane >> 5);  // tile of accumulator register v (MFMA C/D layout)
                    const int oy = Y0 + 2 * (t >> 3) + a, ox = X0 + 2 * (t & 7);
                    const float y0 = (zj[0][e] + zj[1][e]) + zj[2][e];
                    const float y1 = (zj[1][e] - zj[2][e]) - zj[3][e];
                    const size_t o = ((size_t)(img * H + oy) * W + ox) * p.N + n;
                    float r0 = y0 + bias, r1 = y1 + bias;
                    if (p.res) { r0 += p.res[o]; r1 += p.res[o + p.N]; }
                    __builtin_nontemporal_store(r0 * p.out_scale, p.out + o);
                    __builtin_nontemporal_store(r1 * p.out_scale, p.out + o + p.N);
                }
            }
    }
}

// U = G g G^T in double, rounded once to fp32, then three round-to-nearest bf16 pieces, in the fragment order of the kernel above.
// w: [N][9 Cin] packed as everywhere else ([n][(ky * 3 + kx) * Cin + c]).  One thread per (n, c).
__device__ __forceinline__ unsigned short s6_bf16_rne(float x) {
    unsigned u = __builtin_bit_cast(unsigned, x);
    u += 0x7FFFu + ((u >> 16) & 1u);
    return (unsigned short)(u >> 16);
}

__global__ void split6_weights_kernel(const float *__restrict__ w, unsigned short *__restrict__ u, int N, int C) {
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= N * C) return;
    const int n = idx / C, c = idx - n * C;
    const double G[4][3] = {{1, 0, 0}, {0.5, 0.5, 0.5}, {0.5, -0.5, 0.5}, {0, 0, 1}};
    double g[3][3];
    for (int ky = 0; ky < 3; ++ky)
        for (int kx = 0; kx < 3; ++kx) g[ky][kx] = (double)w[(size_t)n * 9 * C + (ky * 3 + kx) * C + c];
    const int nk = C / 16;
    const int ntile = n >> 6, nb = (n >> 5) & 1, nl = n & 31;
    const int chunk = c >> 4, khalf = (c >> 3) & 1, e = c & 7;
    for (int i = 0; i < 4; ++i)
        for (int j = 0; j < 4; ++j) {
            double s = 0;
            for (int ky = 0; ky < 3; ++ky)
                for (int kx = 0; kx < 3; ++kx) s += G[i][ky] * g[ky][kx] * G[j][kx];
            const float f = (float)s;
            const unsigned short p0 = s6_bf16_rne(f);
            const float r1 = f - __builtin_bit_cast(float, (unsigned)p0 << 16);
            const unsigned short p1 = s6_bf16_rne(r1);
            const float r2 = r1 - __builtin_bit_cast(float, (unsigned)p1 << 16);
            const unsigned short pcs[3] = {p0, p1, s6_bf16_rne(r2)};
            const int h = i >> 1, i2 = i & 1;
            for (int pi = 0; pi < 3; ++pi) {
                const size_t frag = ((((((size_t)ntile * nk + chunk) * 2 + h) * 4 + j) * 2 + i2) * 2 + nb) * 3 + pi;
                u[frag * 512 + (khalf * 32 + nl) * 8 + e] = pcs[pi];
            }
        }
}

// ---- host side ---------------------------------------------------------------------------------------------------------------------
bool conv3x3_wino2d_split6_shape_ok(int batch, int H, int W, int Cin, int Cout) {
    return batch >= 1 && H % 16 == 0 && W % 16 == 0 && Cin % 16 == 0 && Cin >= 16 && Cout % 64 == 0 &&
           (long long)batch * H * W * Cin * 4 < (1LL << 31) && (long long)batch * H * W * Cout * 4 < (1LL << 31);
}

long long conv3x3_wino2d_split6_weight_bytes(int Cin, int Cout) { return 96LL * Cin * Cout; }  // 16 components x 3 pieces x 2 B

int launch_wino2d_split6_weights(const float *w, unsigned short *u, int Cout, int Cin, hipStream_t s) {
    const int n = Cout * Cin;
    hipLaunchKernelGGL(split6_weights_kernel, dim3((n + 255) / 256), dim3(256), 0, s, w, u, Cout, Cin);
    GD_HIP(hipGetLastError());
    return 0;
}

int launch_conv3x3_wino2d_split6(const float *x, const unsigned short *u, const float *bias, const float *res, float *out, int batch, int H,
                                 int W, int Cin, int Cout, float out_scale, hipStream_t s) {
    GD_CHECK(conv3x3_wino2d_split6_shape_ok(batch, H, W, Cin, Cout), "conv3x3_winograd2d_split6: shape not eligible (B=%d H=%d W=%d Cin=%d Cout=%d)",
             batch, H, W, Cin, Cout);
    Split6Params p{};
    p.x = x; p.u = u; p.bias = bias; p.res = res; p.out = out;
    p.B = batch; p.H = H; p.W = W; p.C = Cin; p.N = Cout;
    p.bx_tiles = W / 16;
    p.tiles_per_image = (H / 16) * (W / 16);
    p.m_tiles = batch * p.tiles_per_image;
    p.n_tiles = Cout / 64;
    p.nk = Cin / 16;
    p.out_scale = out_scale;
    const int grid = p.m_tiles * p.n_tiles;
    hipLaunchKernelGGL(conv3x3_wino2d_split6_kernel, dim3(grid), dim3(256), 0, s, p);
    GD_HIP(hipGetLastError());
    return 0;
}

}  // namespace gd
