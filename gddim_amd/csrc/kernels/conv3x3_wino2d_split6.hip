// 3x3 / stride 1 / pad 1 convolution through the two-dimensional Winograd transform F(2x2, 3x3) with every component GEMM in the EXACT
// three-piece bf16 split on the bf16 matrix core (v_mfma_f32_32x32x16_bf16), fp32 accumulation -- the operator VERDICT r5 item 2 asks for
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
// Workgroup = 512 threads = 8 wave64 (two per SIMD, 256 registers each), one per CU (147 KB of LDS).  Tile = 64 Winograd tiles (an 8 x 8
// block of 2 x 2-output tiles = 16 x 16 output pixels of ONE image) x 64 outputs x 16 components.  Wave w = (column j = w & 3, row parity
// ih = w >> 2) owns components (ih, j) and (ih + 2, j) of both 32-tile blocks and both 32-output blocks: 8 accumulators of 32 x 32 = 128
// registers.  K = Cin is walked in chunks of 16 channels, each in two PHASES h = 0, 1 = component rows {2h, 2h + 1}: the V stage of a phase
// (8 components x 3 pieces x 64 tiles x 16 channels of bf16 = 48 KB) is read while the other stage is written; 24 MFMAs per wave and ONE
// barrier per phase.  The two waves of a SIMD (w and w + 4) take the phase in opposite order -- ih = 0: multiply, then produce the next
// phase's V; ih = 1: produce, then multiply -- so that one wave's MFMAs run under the other's VALU / LDS / load instructions (a wave issues
// in order: in the first, one-wave-per-SIMD form every LDS or global instruction cost the wave 15-60 cycles of issue that nothing covered,
// and the loop ran at 3 100-4 000 cycles per phase against 1 536 of MFMA issue; profiles/r06_notes.md).
//   LDS   V[h][i & 1][j][piece][k half][tile][8 bf16]: an A fragment (32 tiles x 16 channels) is one contiguous KB, ds_read_b128, conflict-free;
//         the producing thread (tile, channel quad, row parity) writes 8 bytes per (component, piece): ds_write_b64, 16 lanes = 8 tiles x 2
//         quads = 128 contiguous bytes.
//   U     straight from global / L2 into the B registers, fragment order [n tile][chunk][h][column j][i & 1][n block][piece][lane][8 bf16]
//         (split6_weights_kernel): a wave's 6 fragments of a phase are 6 contiguous KB, requested right after the previous phase's MFMAs.
//   d     the tile's 18 x 18-pixel input region, 16 channels at a time, ONCE per chunk: 1 296 16-byte pieces, three coalesced loads per
//         thread (zero padding by the buffer bounds check) two chunks ahead, parked in registers for a phase and a half, then stored into a
//         two-stage raw ring in LDS laid out [row][column parity][column / 2][channel quad] so that the (tile, quad) threads' ds_read_b128 of
//         one patch position are conflict-free.  (First form: every thread loaded its own patch pixels from global memory, 12 loads of 16 B
//         per phase touching 32 cache lines each -- 196 KB per phase through a 64 B/clk L1 for 48 KB of payload.)  Thread (tile, quad, ih)
//         reads the two patch rows its component row combines (B^T row: one add per element), then per component: column combination
//         (4 adds), split (8 and / 8 sub / 6 perm), 3 stores.
// Epilogue: the row transform A^T . takes the rows of the two waves of a column (the ih = 0 wave stores its part into the exchange area,
// the ih = 1 wave adds its own after a barrier), the column transform . A crosses the four columns: 128 KB over the idle stages, [tile][n] so
// that a reader holds four consecutive outputs; thread (tile, n quad) finishes 2 x 2 pixels x 4 outputs: bias, residual, scale, 16-byte stores.
// Shapes: H % 16 == 0, W % 16 == 0, Cin % 16 == 0, N % 64 == 0.
// Epilogue GroupNorm (EGN variant, 16 x 16 images: a tile is a whole image x 64 outputs = whole groups): the lane keeps its 32 outputs,
// the group sums go lanes -> waves through 1 KB of LDS, then normalise (+ swish) and store -- what the fp32 Winograd kernels do there.
// Where it stands (MI355X, B = 256; tools/bench_split6.py, tools/split6_layers.py, tools/power_modes.sh, profiles/r06_notes.md): rms error
// 0.33 of the production hybrid fp32 kernel's; stand-alone 1.16-1.23x (16x16, 256 -> 256) and 1.09-1.18x (32x32, 128 -> 128) its speed over
// 20-launch bursts, 1.12x / 1.08x sustained; planned in as precision mode 2 (GDDIM_PRECISION=split6).  End to end the mode is a wash
// (136.5-140.6 images/s against 137.6-141.1 in fp32 mode, 0.98-1.01 box by box): the modes that keep the bf16 matrix core busy run at a 12-14 % lower shader clock.
#include <cstdio>
#include <cstdlib>

#include <type_traits>

#include "../common.h"
#include "act.h"

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
constexpr int S6_RAW = 6 * 256 * 16;           // bytes of one raw stage: 18 x 18 pixels x 64 B = 1 296 pieces, padded to 6 per thread
constexpr int S6_SMEM = 2 * S6_STAGE + 2 * S6_RAW;  // 147 456 B during the K loop; the epilogue's exchange area (128 KB) lies over it
constexpr int S6_OOB = (int)0x80000000u;

struct Split6Params {
    const float *x;
    const unsigned short *u;  // split6_weights_kernel's fragments
    const float *bias, *res;
    const float *rowbias;  // per-image bias row (the time-embedding projection), rowbias[img * ld_rowbias + n]; null: none
    float *qstats;         // fused GroupNorm statistics of the output, float2 [img][slice][n / 4] = (sum, sum of squares); null: none
    int ld_rowbias, qs_slices;
    // GroupNorm (+ swish) of the OUTPUT in the epilogue (common.h: GemmParams::egn_out; 16 x 16 images only: a tile is a whole image x 64
    // outputs = whole groups of egn_cpg = 4 / 8 / 16 channels); egn_skip_raw: the raw output has no other reader and is not stored
    float *egn_out;
    const float *egn_gamma, *egn_beta;
    int egn_cpg, egn_act, egn_skip_raw;
    float *out;
    int B, H, W, C, N;
    int bx_tiles, tiles_per_image, m_tiles, n_tiles, nk;  // 16x16-pixel blocks per row / per image, in the batch; 64-output tiles; chunks
    float out_scale;
};

// Scalar fp32 VALU on purpose: beside bf16 MFMAs a v_pk_add_f32 costs issue time, v_add_f32 does not (MI355X_MICROARCH.md).  The
// compiler packs adjacent fp32 adds by itself (SLP vectoriser): this file is built with -fno-slp-vectorize (build.py) and keeps its
// arithmetic on scalars, so that the instruction scheduler still sees plain VALU instructions it can place between the MFMAs
// (inline assembly would be opaque to the sched_group_barrier pipeline below).
struct s6v4 { float e[4]; };
__device__ __forceinline__ s6v4 s6_add4(s6v4 a, s6v4 b) { return s6v4{{a.e[0] + b.e[0], a.e[1] + b.e[1], a.e[2] + b.e[2], a.e[3] + b.e[3]}}; }
__device__ __forceinline__ s6v4 s6_sub4(s6v4 a, s6v4 b) { return s6v4{{a.e[0] - b.e[0], a.e[1] - b.e[1], a.e[2] - b.e[2], a.e[3] - b.e[3]}}; }

// (x0, x1) -> one dword holding the top halves: bf16(x0) in bits 15:0, bf16(x1) in 31:16 (truncation)
__device__ __forceinline__ unsigned s6_pack_hi(float x0, float x1) {
    return __builtin_amdgcn_perm(__builtin_bit_cast(unsigned, x1), __builtin_bit_cast(unsigned, x0), 0x07060302u);
}
__device__ __forceinline__ float s6_trunc(float x) { return __builtin_bit_cast(float, __builtin_bit_cast(unsigned, x) & 0xFFFF0000u); }

// the three bf16 pieces of four channels: out[piece] = 8 bytes in channel order
__device__ __forceinline__ void s6_split4(s6v4 vv, s6u32x2 out[3]) {
    const float *v = vv.e;
    float r1[4], r2[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) r1[e] = v[e] - s6_trunc(v[e]);
#pragma unroll
    for (int e = 0; e < 4; ++e) r2[e] = r1[e] - s6_trunc(r1[e]);
    out[0] = s6u32x2{s6_pack_hi(v[0], v[1]), s6_pack_hi(v[2], v[3])};
    out[1] = s6u32x2{s6_pack_hi(r1[0], r1[1]), s6_pack_hi(r1[2], r1[3])};
    out[2] = s6u32x2{s6_pack_hi(r2[0], r2[1]), s6_pack_hi(r2[2], r2[3])};
}

__device__ __forceinline__ s6v4 s6_load16(__amdgpu_buffer_rsrc_t rsrc, int voffset, int soffset) {
    const s6f32x4 t = __builtin_bit_cast(s6f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsrc, voffset, soffset, 0));
    return s6v4{{t[0], t[1], t[2], t[3]}};
}

// The workgroup's program for one role (IH = 0: multiply first, then produce; IH = 1: produce first, then multiply), so that the two waves
// of a SIMD -- waves w and w + 4 of the workgroup -- alternate between the matrix pipe and the vector / LDS / load instructions.
// PERSISTENT: one workgroup per CU walks its tiles; the first loads of the next tile (two region chunks, the first U fragments) are
// requested before the epilogue of the current one, and nothing is dispatched between tiles (the one-tile-per-workgroup form lost 34-50 us
// per launch between workgroups: profiles/r06_notes.md).
template <int IH, bool EGN>
__device__ __forceinline__ void s6_body(const Split6Params &p, unsigned char *smem, int tid, int lane, int wave) {
    constexpr int CS_BYTES = 3 * 2048;  // one component: 3 pieces x 2 k halves x 64 tiles x 16 B
    const int jw = wave & 3;            // component column
    unsigned char *stage0 = smem, *stage1 = smem + S6_STAGE;
    unsigned char *raw0 = smem + 2 * S6_STAGE, *raw1 = raw0 + S6_RAW;
    const int H = p.H, W = p.W, C = p.C;
    const int u_vo = lane * 16;
    const int raw_wr = tid * 16;
    // ---- producer identity: (tile, channel quad), component row parity = IH.  lane bits: 0 = quad bit 0, 1..3 = tile bits 0..2, 4 = quad
    // bit 1, 5 = tile bit 3; jw is tile bits 4..5 (two tile rows of the 8 x 8 block) ----
    const int tile = (jw << 4) | (((lane >> 5) & 1) << 3) | ((lane >> 1) & 7);
    const int q = (lane & 1) | (((lane >> 4) & 1) << 1);
    const int ty = tile >> 3, tx = tile & 7;
    // the thread's patch position (r, s): region row 2 ty + r, column 2 tx + s -> parity s & 1, column / 2 = tx + (s >> 1)
    const int raw_rd = ((2 * ty * 2) * 9 + tx) * 64 + q * 16;  // + (r * 2 + (s & 1)) * 576 + (s >> 1) * 64
    // LDS byte offsets: the thread's 8-byte store slot / the lane's 16-byte fragment slot inside a (component, piece) block of 2 KB
    const int vw_off = ((q >> 1) * 64 + tile) * 16 + (q & 1) * 8;
    const int ar_off = ((lane >> 5) * 64 + (lane & 31)) * 16;
    const __amdgpu_buffer_rsrc_t srd_x = __builtin_amdgcn_make_buffer_rsrc(const_cast<char *>(reinterpret_cast<const char *>(p.x)), 0,
                                                                          p.B * H * W * C * 4, 0x00020000);
    const __amdgpu_buffer_rsrc_t srd_o = __builtin_amdgcn_make_buffer_rsrc(reinterpret_cast<char *>(p.out), 0, p.B * H * W * p.N * 4, 0x00020000);
    const __amdgpu_buffer_rsrc_t srd_r = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<char *>(reinterpret_cast<const char *>(p.res ? p.res : p.out)), 0, p.B * H * W * p.N * 4, 0x00020000);
    const size_t u_tile_bytes = (size_t)p.nk * 2 * 4 * 12 * 1024;
    const size_t u_wave = (size_t)jw * 12 * 1024 + (size_t)IH * 6 * 1024;  // [n tile][phase][column jw][row parity][n block][piece] x 1 KB

    s6v4 pr[2][4];          // the two patch rows the thread's component row combines
    s6v4 rc[4];             // its B^T row (4 patch columns)
    s6f32x4 rawreg[2][3];   // region pieces between their loads and their LDS stores ([1]: the second chunk of a tile's first two)
    s6bf16x8 Areg[2][3];    // [tile block][piece]
    s6bf16x8 Ureg[2][3];    // [n block][piece]
    s6f32x16 acc[2][2][2];  // [component row i >> 1][tile block][n block]: rows IH and IH + 2
    int rvo[3];             // the thread's three region pieces of the current tile (byte offsets into x; out of the image: out of bounds)

#ifdef S6_TRACE  // in-kernel timeline (tools/split6_trace.py): lane 0 of waves 0 and 4 (the two waves of one SIMD) of two workgroups stamps the
                 // 100 MHz wall clock and the shader clock into the tail of the residual tensor: slot = 80 stamps x (wall, cycles)
    const int trslot = blockIdx.x == 0 ? 0 : blockIdx.x == 201 ? 1 : -1;
    float *trbuf = const_cast<float *>(p.res) + (size_t)p.B * p.H * p.W * p.N - 1024 + ((trslot < 0 ? 0 : trslot) * 2 + (wave == 4 ? 1 : 0)) * 160;
    const bool trw = trslot >= 0 && lane == 0 && (wave == 0 || wave == 4) && p.res;
    bool tr_first = true;
    auto trace = [&](int idx) __attribute__((always_inline)) {
        if (trw && tr_first && idx < 80) {
            trbuf[idx] = (float)(wall_clock64() & 0xFFFFFFull);
            trbuf[80 + idx] = (float)(__builtin_readcyclecounter() & 0xFFFFFFull);
        }
    };
#define S6_TR(i) trace(i)
#else
#define S6_TR(i)
#endif

    // ---- the workgroup's tiles.  Consecutive workgroup ids go round the 8 XCDs: the workgroups of one XCD share ONE n tile, i.e. one slice
    // of the weight fragments in that XCD's L2 (the whole U of a 256 -> 256 layer is 6.3 MB, a slice 1.6 MB), and walk its m tiles ----
    const int total = p.m_tiles * p.n_tiles;
    const bool xcd_map = (8 % p.n_tiles) == 0 && (int)gridDim.x % 8 == 0;
    const int per = xcd_map ? 8 / p.n_tiles : 1;
    const int xcd = blockIdx.x & 7;
    const int m_step = xcd_map ? (int)gridDim.x / p.n_tiles : 0;       // m tiles between two of this workgroup's
    int mtile = xcd_map ? (blockIdx.x >> 3) * per + xcd / p.n_tiles : 0, ntile = xcd_map ? xcd % p.n_tiles : 0;
    int flat = blockIdx.x;                                             // (the plain walk when the n tiles do not divide the XCDs)
    if (!xcd_map) { mtile = flat / p.n_tiles; ntile = flat - mtile * p.n_tiles; }
    auto have_tile = [&]() { return xcd_map ? mtile < p.m_tiles : flat < total; };
    auto next_tile = [&]() {
        if (xcd_map) mtile += m_step;
        else { flat += gridDim.x; mtile = flat / p.n_tiles; ntile = flat - mtile * p.n_tiles; }
    };
    int img = 0, Y0 = 0, X0 = 0;
    __amdgpu_buffer_rsrc_t srd_u = srd_x;
    auto enter_tile = [&]() __attribute__((always_inline)) {  // coordinates, region offsets and weight slice of (mtile, ntile)
        img = mtile / p.tiles_per_image;
        const int brem = mtile - img * p.tiles_per_image;
        const int by = brem / p.bx_tiles, bx = brem - by * p.bx_tiles;
        Y0 = by * 16; X0 = bx * 16;
        // raw ring: piece L = ((row * 2 + column parity) * 9 + column / 2) * 4 + quad of the 18 x 18 region; thread t moves pieces t + 512 i
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            const int L = i * 512 + tid;
            const int row = L / 72, rem = L - row * 72, par = rem / 36, r2 = rem - par * 36;
            const int y = Y0 - 1 + row, x = X0 - 1 + 2 * (r2 >> 2) + par;
            const bool ok = L < 1296 && (unsigned)y < (unsigned)H && (unsigned)x < (unsigned)W;
            rvo[i] = ok ? (((img * H + y) * W + x) * C + (r2 & 3) * 4) * 4 : S6_OOB;
        }
        srd_u = __builtin_amdgcn_make_buffer_rsrc(const_cast<char *>(reinterpret_cast<const char *>(p.u)) + (size_t)ntile * u_tile_bytes + u_wave, 0,
                                                  (int)(u_tile_bytes - u_wave), 0x00020000);
    };

    auto load_raw = [&](int kc, int set) __attribute__((always_inline)) {
        if ((S6_ABL & 1) && kc >= 2) return;
        const int so = __builtin_amdgcn_readfirstlane(kc * 64);
#pragma unroll
        for (int i = 0; i < 3; ++i) rawreg[set][i] = __builtin_bit_cast(s6f32x4, __builtin_amdgcn_raw_buffer_load_b128(srd_x, rvo[i], so, 0));
    };
    auto store_raw = [&](unsigned char *st, int set) __attribute__((always_inline)) {
#pragma unroll
        for (int i = 0; i < 3; ++i) *reinterpret_cast<s6f32x4 *>(st + i * 8192 + raw_wr) = rawreg[set][i];
    };
    // Component row i = 2 hp + IH of the thread's (tile, quad): B^T row i = d0 - d2, d1 + d2, d2 - d1, d1 - d3 (patch rows), then the four
    // components (i, j): column combination, split, three 8-byte stores each
    auto produce_row = [&](const unsigned char *rawst, int hp, unsigned char *st) __attribute__((always_inline)) {
        const int i = 2 * hp + IH;
        const int ra = i == 0 ? 0 : i == 1 ? 1 : i == 2 ? 2 : 1, rb = i == 0 ? 2 : i == 1 ? 2 : i == 2 ? 1 : 3;
        // (two patch columns at a time: eight 16-byte reads in flight at once would hold 32 registers next to 128 accumulators)
#pragma unroll
        for (int half = 0; half < 2; ++half) {
#pragma unroll
            for (int sx = 2 * half; sx < 2 * half + 2; ++sx) {
                const s6f32x4 ta = *reinterpret_cast<const s6f32x4 *>(rawst + raw_rd + (ra * 2 + (sx & 1)) * 576 + (sx >> 1) * 64);
                const s6f32x4 tb_ = *reinterpret_cast<const s6f32x4 *>(rawst + raw_rd + (rb * 2 + (sx & 1)) * 576 + (sx >> 1) * 64);
                pr[0][sx] = s6v4{{ta[0], ta[1], ta[2], ta[3]}};
                pr[1][sx] = s6v4{{tb_[0], tb_[1], tb_[2], tb_[3]}};
                rc[sx] = (S6_ABL & 16) ? pr[0][sx] : i == 1 ? s6_add4(pr[0][sx], pr[1][sx]) : s6_sub4(pr[0][sx], pr[1][sx]);
            }
            __builtin_amdgcn_sched_barrier(0);
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            s6v4 v;
            if (S6_ABL & 16) v = rc[j];
            else if (j == 0) v = s6_sub4(rc[0], rc[2]);
            else if (j == 1) v = s6_add4(rc[1], rc[2]);
            else if (j == 2) v = s6_sub4(rc[2], rc[1]);
            else v = s6_sub4(rc[1], rc[3]);
            s6u32x2 pc[3];
            if (S6_ABL & 16) { pc[0] = s6u32x2{__builtin_bit_cast(unsigned, v.e[0]), __builtin_bit_cast(unsigned, v.e[1])}; pc[1] = pc[0]; pc[2] = pc[0]; }
            else s6_split4(v, pc);
            if (S6_ABL & 2) { asm volatile("" ::"v"(pc[0]), "v"(pc[1]), "v"(pc[2])); continue; }
#pragma unroll
            for (int pi = 0; pi < 3; ++pi) *reinterpret_cast<s6u32x2 *>(st + (IH * 4 + j) * CS_BYTES + pi * 2048 + vw_off) = pc[pi];
            __builtin_amdgcn_sched_barrier(0);  // (one component at a time: interleaved components keep four sets of temporaries alive)
        }
    };
    auto load_u = [&](int ph) __attribute__((always_inline)) {
        if ((S6_ABL & 32) && ph >= 2) return;
        const int so = __builtin_amdgcn_readfirstlane(ph * (4 * 12 * 1024));
#pragma unroll
        for (int nb = 0; nb < 2; ++nb)
#pragma unroll
            for (int pi = 0; pi < 3; ++pi)
                Ureg[nb][pi] = __builtin_bit_cast(s6bf16x8, __builtin_amdgcn_raw_buffer_load_b128(srd_u, u_vo, so + (nb * 3 + pi) * 1024, 0));
    };
    // the wave's 24 MFMAs of a phase: component (2 hh + IH, jw), both tile blocks x both n blocks x six products -- a1 u1, a2 u0, a0 u2, a1 u0,
    // a0 u1, a0 u0, the small ones first; consecutive MFMAs go round the four accumulators
    auto multiply = [&](int hh, const unsigned char *rd) __attribute__((always_inline)) {
        constexpr int PA[6] = {1, 2, 0, 1, 0, 0}, PB[6] = {1, 0, 2, 0, 1, 0};
        if (!(S6_ABL & 4)) {
#pragma unroll
            for (int tb = 0; tb < 2; ++tb)
#pragma unroll
                for (int pi = 0; pi < 3; ++pi)
                    Areg[tb][pi] = *reinterpret_cast<const s6bf16x8 *>(rd + (IH * 4 + jw) * CS_BYTES + pi * 2048 + tb * 512 + ar_off);
        }
        if (S6_ABL & 64) return;
#pragma unroll
        for (int t = 0; t < 6; ++t)
#pragma unroll
            for (int nb = 0; nb < 2; ++nb)
#pragma unroll
                for (int tb = 0; tb < 2; ++tb)
                    acc[hh][tb][nb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(Areg[tb][PA[t]], Ureg[nb][PB[t]], acc[hh][tb][nb], 0, 0, 0);
    };
    // One phase, everything but its chunk index known at compile time: HH = component-row half being multiplied (V of the phase lies in
    // stage HH; the next phase is (chunk kc + HH, half 1 - HH) and its V goes to stage 1 - HH), MORE = there is a next phase.
    auto phase = [&](auto HHc, auto MOREc, int kc) __attribute__((always_inline)) {
        constexpr int HH = decltype(HHc)::value;
        constexpr bool more1 = decltype(MOREc)::value;
        const int ph = 2 * kc + HH;
        unsigned char *rd = HH ? stage1 : stage0, *wr = HH ? stage0 : stage1;
        const unsigned char *rawn = ((kc + HH) & 1) ? raw1 : raw0;
        // the region of chunk kc + 2: requested at the top of this chunk's first phase (a phase and a half before its stores: the first form
        // requested it at the end of the phase and its stores waited 250-400 cycles), stored in the second -- its stage was last read in the first
        if (more1 && HH == 0) load_raw(kc + 2, 0);
        if (IH == 0) {
            multiply(HH, rd);
            __builtin_amdgcn_sched_barrier(0);
            if (more1) load_u(ph + 1);
            if (more1) produce_row(rawn, 1 - HH, wr);
        } else {
            if (more1) produce_row(rawn, 1 - HH, wr);
            __builtin_amdgcn_sched_barrier(0);
            multiply(HH, rd);
            __builtin_amdgcn_sched_barrier(0);
            if (more1) load_u(ph + 1);
        }
        if (more1 && HH == 1) store_raw((kc & 1) ? raw1 : raw0, 0);
        __builtin_amdgcn_sched_barrier(0);
        if (!(S6_ABL & 8)) __syncthreads();
        S6_TR(2 + ph);
    };
    typedef std::integral_constant<int, 0> H0;
    typedef std::integral_constant<int, 1> H1;

    if (!have_tile()) return;
    enter_tile();
    S6_TR(0);
    load_raw(0, 0);
    load_raw(1, 1);
    load_u(0);
    while (true) {
        // ---- prologue of the tile: its first two region chunks (already requested) into the raw ring, V of phase 0 into stage 0 ----
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int a = 0; a < 2; ++a)
#pragma unroll
                for (int b = 0; b < 2; ++b)
#pragma unroll
                    for (int v = 0; v < 16; ++v) acc[i][a][b][v] = 0.f;
        store_raw(raw0, 0);
        store_raw(raw1, 1);
        __syncthreads();
        produce_row(raw0, 0, stage0);
        __syncthreads();
        S6_TR(1);
        for (int kc = 0; kc < p.nk - 1; ++kc) {
            phase(H0{}, std::true_type{}, kc);
            phase(H1{}, std::true_type{}, kc);
        }
        phase(H0{}, std::true_type{}, p.nk - 1);
        phase(H1{}, std::false_type{}, p.nk - 1);
        S6_TR(39);

        // ---- the next tile's first loads go out before this tile's epilogue (their registers are idle until the next prologue) ----
        // (EGN variant: after the accumulators' last use -- its epilogue keeps the tile's 32 outputs per lane in registers until the
        // group statistics are complete)
        const int e_img = img, e_Y0 = Y0, e_X0 = X0, e_ntile = ntile, e_mtile = mtile;
        next_tile();
        const bool more_tiles = have_tile();
        auto prefetch_next = [&]() __attribute__((always_inline)) {
            if (more_tiles) {
                enter_tile();
                load_raw(0, 0);
                load_raw(1, 1);
                load_u(0);
            }
        };
        if (!EGN) prefetch_next();

        // ---- epilogue, one output row parity a at a time.  z_a = sum_i A^T[a][i] M_i (a = 0: m0 + m1 + m2, a = 1: m1 - m2 - m3) takes the
        // rows of the two waves of a column: EVERY wave stores its part (IH = 0: m0 + m2 | -m2, IH = 1: m1 | m1 - m3) into
        // zx[column j][IH][tile][n] (floats, n contiguous: 8 planes of 16 KB = the 128 KB over the idle stages) and the readers add the two
        // parts.  (First 8-wave form: both parities at once, the IH = 1 waves adding to the IH = 0 waves' planes in a second pass while those
        // idled: 3.2 us of exchange per tile.)  The MFMA layout has a lane = one output n, so the stores are single floats (lanes =
        // consecutive n: conflict-free) and the readers get FOUR consecutive n per lane: a store instruction costs the CU's store path about
        // the same whatever its width, and 64 dword stores per lane made the first form's epilogue 9-11 us of an 80 us tile.
        // Reader = thread (tile, n quad): y_ab = sum_j z_a[j] A^T[b][j] (b = 0: z0 + z1 + z2, b = 1: z1 - z2 - z3), 16-byte stores; lane bits
        // 0..3 = n quad (a wave-level store covers whole 256-byte rows of four pixels); tile = 4 (wave * 2 + k) + (lane >> 4) ----
        float *zx = reinterpret_cast<float *>(smem);
        const int nq = lane & 15;
        const int n = e_ntile * 64 + nq * 4;
        s6f32x4 bias = {0.f, 0.f, 0.f, 0.f};
        if (p.bias) bias = *reinterpret_cast<const s6f32x4 *>(p.bias + n);
        if (p.rowbias) bias += *reinterpret_cast<const s6f32x4 *>(p.rowbias + (size_t)e_img * p.ld_rowbias + n);  // (a tile is one image)
        const bool has_res = p.res != nullptr;
        float qs_s = 0.f, qs_q = 0.f;  // the lane's share of the statistics entry (tile row = wave, channel quad nq)
        s6f32x4 keep[2][2][2];         // EGN: the lane's outputs [a][k][pixel column]
#pragma unroll
        for (int a = 0; a < 2; ++a) {
#pragma unroll
            for (int tb = 0; tb < 2; ++tb)
#pragma unroll
                for (int nb = 0; nb < 2; ++nb)
#pragma unroll
                    for (int v = 0; v < 16; ++v) {
                        const float lo = acc[0][tb][nb][v], hi = acc[1][tb][nb][v];  // component rows IH, IH + 2
                        const float z = a == 0 ? (IH == 0 ? lo + hi : lo) : (IH == 0 ? -hi : lo - hi);
                        const int t = tb * 32 + (v & 3) + 8 * (v >> 2);  // + 4 (lane >> 5): tile of accumulator register v (MFMA C / D layout)
                        zx[(jw * 2 + IH) * 4096 + (t + 4 * (lane >> 5)) * 64 + nb * 32 + (lane & 31)] = z;
                    }
            if (EGN && a == 1) prefetch_next();
            __syncthreads();
            if (a == 0) S6_TR(41);
#pragma unroll
            for (int k = 0; k < 2; ++k) {
                const int t = 4 * (wave * 2 + k) + (lane >> 4);
                const int oy = e_Y0 + 2 * (t >> 3) + a, ox = e_X0 + 2 * (t & 7);
                const int vo = (((e_img * H + oy) * W + ox) * p.N + n) * 4;
                s6f32x4 zj[4];
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    zj[j] = *reinterpret_cast<const s6f32x4 *>(zx + (j * 2 + 0) * 4096 + t * 64 + nq * 4) +
                            *reinterpret_cast<const s6f32x4 *>(zx + (j * 2 + 1) * 4096 + t * 64 + nq * 4);
                s6f32x4 r0 = (zj[0] + zj[1]) + zj[2] + bias;
                s6f32x4 r1 = (zj[1] - zj[2]) - zj[3] + bias;
                if (has_res) {  // (requested where it is consumed: a prefetch of 8 or 16 registers per lane spills next to the accumulators and the
                                // next tile's loads in flight, and measured no faster in the U-Net)
                    r0 += __builtin_bit_cast(s6f32x4, __builtin_amdgcn_raw_buffer_load_b128(srd_r, vo, 0, 0));
                    r1 += __builtin_bit_cast(s6f32x4, __builtin_amdgcn_raw_buffer_load_b128(srd_r, vo, p.N * 4, 0));
                }
                r0 *= p.out_scale;
                r1 *= p.out_scale;
#pragma unroll
                for (int e4 = 0; e4 < 4; ++e4) {
                    qs_s += r0[e4] + r1[e4];
                    qs_q += r0[e4] * r0[e4] + r1[e4] * r1[e4];
                }
                if (EGN) { keep[a][k][0] = r0; keep[a][k][1] = r1; }
                if (EGN && p.egn_skip_raw) continue;
                const s6u32x4 d0 = __builtin_bit_cast(s6u32x4, r0), d1 = __builtin_bit_cast(s6u32x4, r1);
                __builtin_amdgcn_raw_buffer_store_b128(d0, srd_o, vo, 0, 2);  // (aux 2: nt)
                __builtin_amdgcn_raw_buffer_store_b128(d1, srd_o, vo, p.N * 4, 2);
                // A 16-byte store reads its last data register for the last lanes many cycles after issue: with the registers rewritten two
                // instructions later, element 3 of lanes 12..15 (mod 16) of the FIRST store came out as the second store's (measured, every
                // shape, deterministic).  The hazard recogniser does not cover it; keep the data registers alone for a while.
                asm volatile("s_nop 15\n\ts_nop 15" ::"v"(d0), "v"(d1));
            }
            if (a == 0) __syncthreads();  // (the planes are rewritten with the other parity)
        }
        if (EGN) {
            // GroupNorm (+ swish) of the output (plan.h: Op::egn_peer), as the fp32 Winograd kernels do it on 16 x 16 images: the tile is
            // the whole image for 64 channels.  Lane sums (32 values: 8 pixels x 4 channels, already in qs_s / qs_q) -> the lanes of the
            // group's other channel quads (egn_cpg / 4 neighbours in n) and the wave's four tile columns -> the eight waves through the
            // 1 KB behind the exchange area (free until the next prologue) -> (mean, rstd) as flax: var = E[y^2] - E[y]^2, eps 1e-6.
            float gsum = qs_s, gsq = qs_q;
            for (int o = 4; o < p.egn_cpg; o <<= 1) { gsum += __shfl_xor(gsum, o >> 2); gsq += __shfl_xor(gsq, o >> 2); }
            gsum += __shfl_xor(gsum, 16); gsq += __shfl_xor(gsq, 16);
            gsum += __shfl_xor(gsum, 32); gsq += __shfl_xor(gsq, 32);
            float2 *red = reinterpret_cast<float2 *>(smem + 131072);  // [wave][n quad]
            if (lane < 16) red[wave * 16 + lane] = make_float2(gsum, gsq);
            __syncthreads();
            gsum = 0.f; gsq = 0.f;
#pragma unroll
            for (int w8 = 0; w8 < 8; ++w8) {
                const float2 v = red[w8 * 16 + nq];
                gsum += v.x; gsq += v.y;
            }
            const float inv_n = 1.0f / (256.0f * (float)p.egn_cpg);
            const float mean = gsum * inv_n;
            const float rstd = rsqrtf(gsq * inv_n - mean * mean + 1e-6f);
            const s6f32x4 ga = *reinterpret_cast<const s6f32x4 *>(p.egn_gamma + n), be = *reinterpret_cast<const s6f32x4 *>(p.egn_beta + n);
            const s6f32x4 sc = rstd * ga, sh = be - mean * rstd * ga;
            const __amdgpu_buffer_rsrc_t srd_g = __builtin_amdgcn_make_buffer_rsrc(reinterpret_cast<char *>(p.egn_out), 0, p.B * H * W * p.N * 4, 0x00020000);
#pragma unroll
            for (int a = 0; a < 2; ++a)
#pragma unroll
                for (int k = 0; k < 2; ++k) {
                    const int t = 4 * (wave * 2 + k) + (lane >> 4);
                    const int oy = e_Y0 + 2 * (t >> 3) + a, ox = e_X0 + 2 * (t & 7);
                    const int vo = (((e_img * H + oy) * W + ox) * p.N + n) * 4;
                    s6f32x4 t0 = keep[a][k][0] * sc + sh, t1 = keep[a][k][1] * sc + sh;
                    if (p.egn_act) {
#pragma unroll
                        for (int e4 = 0; e4 < 4; ++e4) { t0[e4] = swish_e(t0[e4]); t1[e4] = swish_e(t1[e4]); }
                    }
                    const s6u32x4 d0 = __builtin_bit_cast(s6u32x4, t0), d1 = __builtin_bit_cast(s6u32x4, t1);
                    __builtin_amdgcn_raw_buffer_store_b128(d0, srd_g, vo, 0, 2);
                    __builtin_amdgcn_raw_buffer_store_b128(d1, srd_g, vo, p.N * 4, 2);
                    asm volatile("s_nop 15\n\ts_nop 15" ::"v"(d0), "v"(d1));  // (the store-data hazard above)
                }
        }
        // Fused GroupNorm statistics (kernels/qstats.h: one writer per entry, no atomics): wave w finished tile row w of the block = 32
        // output pixels = one "slice"; its lanes hold 2 x 2 tiles x 2 x 2 pixels x 4 channels each, the four 16-lane groups the four
        // tiles of a k.  Entry (image, slice = block-in-image * 8 + w, channel quad) = (sum, sum of squares) over those 32 pixels x 4 channels.
        if (p.qstats) {
            qs_s += __shfl_xor(qs_s, 16); qs_q += __shfl_xor(qs_q, 16);
            qs_s += __shfl_xor(qs_s, 32); qs_q += __shfl_xor(qs_q, 32);
            if (lane < 16) {
                const int brem_e = e_mtile - e_img * p.tiles_per_image;
                float2 *o = reinterpret_cast<float2 *>(p.qstats) + ((size_t)e_img * p.qs_slices + brem_e * 8 + wave) * (p.N >> 2) + (n >> 2);
                *o = make_float2(qs_s, qs_q);
            }
        }
        S6_TR(42);
#ifdef S6_TRACE
        tr_first = false;
#endif
        if (!more_tiles) break;
        __syncthreads();  // (the exchange area lies over the stages and the raw ring the next prologue writes)
    }
#undef S6_TR
}

template <bool EGN>
__global__ void __launch_bounds__(512) conv3x3_wino2d_split6_kernel(const Split6Params p) {
    __shared__ __attribute__((aligned(16))) unsigned char smem[S6_SMEM];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    // component row parity ih = wave >> 2: waves w and w + 4 share a SIMD and take a phase in opposite order
    if ((wave >> 2) == 0) s6_body<0, EGN>(p, smem, tid, lane, wave);
    else s6_body<1, EGN>(p, smem, tid, lane, wave);
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

// The engine's GEMM op on this kernel (precision mode 2): a single-source 3x3 / stride 1 / pad 1 convolution with dense rows, optional bias,
// per-image bias row, residual, output scale and fused GroupNorm statistics (one slice per 32 output pixels); no epilogue GroupNorm.
bool conv3x3_wino2d_split6_gemm_ok(const GemmParams &g) {
    if (g.KH != 3 || g.KW != 3 || g.stride != 1 || g.pad != 1 || g.C1 != 0 || g.lda0 != g.C0 || g.ldc != g.N || g.K != 9 * g.C0) return false;
    if (g.alpha != 1.f || g.in_act || g.b_kmajor || g.gn_side_out || g.a_presplit || g.wk_partial) return false;
    // epilogue GroupNorm: whole images per tile (16 x 16) and whole groups per lane neighbourhood
    if (g.egn_out && (g.H != 16 || g.W != 16 || (g.egn_cpg != 4 && g.egn_cpg != 8 && g.egn_cpg != 16) || !g.egn_gamma || !g.egn_beta)) return false;
    if (g.res && g.ldres != g.N) return false;
    const int HW = g.H * g.W;
    if (HW <= 0 || g.M % HW != 0 || g.OH != g.H || g.OW != g.W) return false;
    if (g.qstats && g.qs_slices != HW / 32) return false;
    return conv3x3_wino2d_split6_shape_ok(g.M / HW, g.H, g.W, g.C0, g.N);
}

static int launch_split6_params(Split6Params &p, hipStream_t s);

int launch_conv3x3_wino2d_split6_gemm(const GemmParams &g, const unsigned short *u, hipStream_t s) {
    GD_CHECK(conv3x3_wino2d_split6_gemm_ok(g), "conv3x3_wino2d_split6: launch not eligible (M=%d H=%d W=%d Cin=%d N=%d)", g.M, g.H, g.W, g.C0, g.N);
    Split6Params p{};
    p.x = g.a0; p.u = u; p.bias = g.bias; p.res = g.res; p.out = g.out;
    p.rowbias = g.rowbias; p.ld_rowbias = g.ld_rowbias; p.qstats = g.qstats; p.qs_slices = g.qs_slices;
    p.egn_out = g.egn_out; p.egn_gamma = g.egn_gamma; p.egn_beta = g.egn_beta; p.egn_cpg = g.egn_cpg; p.egn_act = g.egn_act;
    p.egn_skip_raw = g.egn_skip_raw;
    p.B = g.M / (g.H * g.W); p.H = g.H; p.W = g.W; p.C = g.C0; p.N = g.N;
    p.out_scale = g.out_scale;
    return launch_split6_params(p, s);
}

int launch_conv3x3_wino2d_split6(const float *x, const unsigned short *u, const float *bias, const float *res, float *out, int batch, int H,
                                 int W, int Cin, int Cout, float out_scale, hipStream_t s) {
    GD_CHECK(conv3x3_wino2d_split6_shape_ok(batch, H, W, Cin, Cout), "conv3x3_winograd2d_split6: shape not eligible (B=%d H=%d W=%d Cin=%d Cout=%d)",
             batch, H, W, Cin, Cout);
    Split6Params p{};
    p.x = x; p.u = u; p.bias = bias; p.res = res; p.out = out;
    p.B = batch; p.H = H; p.W = W; p.C = Cin; p.N = Cout;
    p.out_scale = out_scale;
    return launch_split6_params(p, s);
}

static int launch_split6_params(Split6Params &p, hipStream_t s) {
    p.bx_tiles = p.W / 16;
    p.tiles_per_image = (p.H / 16) * (p.W / 16);
    p.m_tiles = p.B * p.tiles_per_image;
    p.n_tiles = p.N / 64;
    p.nk = p.C / 16;
    // persistent: one workgroup per CU walks the tiles (a multiple of 8 workgroups so that a workgroup's XCD fixes its n tile)
    const int total = p.m_tiles * p.n_tiles;
    const int cus = device_cus() / 8 * 8;
    const int grid = total < cus ? total : cus;
    if (p.egn_out) hipLaunchKernelGGL(conv3x3_wino2d_split6_kernel<true>, dim3(grid), dim3(512), 0, s, p);
    else hipLaunchKernelGGL(conv3x3_wino2d_split6_kernel<false>, dim3(grid), dim3(512), 0, s, p);
    GD_HIP(hipGetLastError());
    return 0;
}

}  // namespace gd
