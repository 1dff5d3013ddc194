// 3x3 / stride 1 / pad 1 convolution in fp32 with the HYBRID Winograd transform F(2,3) along y x F(4,3) along x: 24 products per 2 x 4
// output tile = 3 per output pixel, against 4 for F(2x2,3x3) (conv3x3_wino2d_f32.hip), 6 for F(2,3) and 9 for the direct sum.
//
// Round 2 left the 3x3 convolutions on F(2x2,3x3) at 0.72 of the fp32 matrix peak in ISSUED products -- within 12-18 % of what the pipe
// can issue at the clock it sustains -- so the remaining lever is, again, fewer products.  F(4,3) has transform constants (4, 5, 8)
// that cost fp32 accuracy; tools/wino_error_study.py measures it (profiles/r03_winograd_error_study.md): one F(4,3) direction is at
// 1.4x the direct fp32 kernel's rms error (3.7e-6 worst element at Cin = 512), F(4x4,3x3) at 3.7x; the end-to-end budget has 65x.
// For the tile with top-left output pixel (2 ty, 4 tx), its 4 x 6 input patch d[r][s] = in[2 ty - 1 + r][4 tx - 1 + s], kernel g[ky][kx]:
//     V = By^T d Bx,   U = Gy g Gx^T,   M_ij[tile][n] = sum_c V_ij[tile][c] U_ij[n][c]   (i < 4, j < 6),   Y = Ay^T M Ax   (2 x 4)
//     y: the F(2,3) of the other Winograd kernels (By^T = [1 0 -1 0; 0 1 1 0; 0 -1 1 0; 0 1 0 -1], row 3 NEGATED on both sides)
//     x: F(4,3) on the points 0, +-1, +-2, inf (tools/winograd_matrices.py, exact check):
//        Bx^T = [4 0 -5 0 1 0; 0 -4 -4 1 1 0; 0 4 -4 -1 1 0; 0 -2 -1 2 1 0; 0 2 -1 -2 1 0; 0 4 0 -5 0 1]
//        Gx   = [1/4 0 0; -1/6 -1/6 -1/6; -1/6 1/6 -1/6; 1/24 1/12 1/6; 1/24 -1/12 1/6; 0 0 1]
//        Ax^T = [1 1 1 1 1 0; 0 1 -1 2 -2 0; 0 1 1 4 4 0; 0 1 -1 8 -8 1]
// The weight transform is computed in double and rounded once; the activation-side transforms are fp32 adds and fmas by 2, 4, 8.
//
// Workgroup: 512 threads = 8 wave64; tile 32 Winograd tiles (256 pixels) x 64 outputs x 24 components.  Wave (i, wn) owns component row
// i (its six components j) of the 32 tiles x output block wn (32 columns): six 32x32x2 fp32 accumulators, 48 MFMAs per 16-channel chunk.
// What differs from the F(2x2,3x3) kernel besides the transform:
//  * the WEIGHTS NEVER TOUCH LDS.  With the components split over the waves a wave's B fragments are read by no other wave of the
//    workgroup, so staging them through LDS was a copy without reuse: here U is stored in HBM in fragment order
//    U[n / 32][Cin / 8][i][j][channel half][n % 32][4] and every wave loads its 1 KB fragments straight into the registers the MFMA
//    reads (twelve buffer_load_dwordx4 per chunk, one chunk ahead, into the registers the previous chunk's MFMAs have just consumed).
//    LDS holds V only (48.5 KB per 16-channel stage).
//  * K loop in 16-channel chunks: 48 MFMAs per wave between barriers in 24 gaps of two -- two components side by side, so that no MFMA
//    waits for its predecessor's accumulator.  Staging role: thread (tile, channel quad, patch row r) loads the six pixels of ONE patch row
//    as 16-byte pieces; the y transform comes in through a DPP operand from the neighbouring lane exactly as in the F(2x2,3x3) kernel
//    (24 v_fmac_f32_dpp, in place), the x transform F(4,3) is 12 packed-fp32 instructions per channel pair, twelve ds_write_b64 (slot
//    order [j][i]).
//  * wino42_persistent (Cin >= 64): one workgroup per CU walks over tiles and the K-loop pipeline runs ACROSS tiles (the loading state
//    rolls over to the workgroup's next tile; only its first tile has a prologue).  wino42_tile<true> is the one-tile form for Cin = 32.
// Epilogue: x-direction output transform in-lane (six components -> four columns), y-direction across the four component-row waves
// through LDS, then wave f of an output block finishes 8 tiles x 32 columns: bias, per-image bias row, residual, scale, store, GroupNorm
// quad statistics, and the epilogue GroupNorm on 8x8 / 16x16 images -- the F(2x2,3x3) kernel's epilogue on 2 x 4 tiles.
// Measured (profiles/r03_notes.md): 324-330 TFLOP/s algorithmic in the sampler = 0.69-0.70 of the nominal fp32 MFMA peak in issued products
// at a power-limited 1.7-2.0 GHz; the loop lives at 256 VGPRs (a second pixel register set spills), the weight-fragment stream and the LDS
// stores cost clock rather than cycles.
// Shapes: W in {8, 16, 32, 64}, H even, H W / 8 a multiple of 8, Cin % 32 == 0 (two-source concat: C0 % 32 == 0); N % 4 == 0 and
// N >= 32 when the epilogue emits GroupNorm statistics (any N otherwise).
#include <cstdio>
#include <cstdlib>
#include <type_traits>

#include "../common.h"
#include "../options.h"
#include "act.h"
#include "epilogue.h"

namespace gd {

typedef float w4f32x16 __attribute__((ext_vector_type(16)));
typedef float w4f32x4 __attribute__((ext_vector_type(4)));
typedef float w4f32x2 __attribute__((ext_vector_type(2)));
typedef unsigned int w4u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int w4u32x2 __attribute__((ext_vector_type(2)));

#ifndef W42_ABL
#define W42_ABL 0  // timing ablations (results wrong): 1 no pixel loads in the K loop, 2 no LDS stores, 4 no weight-fragment loads in the K loop,
                   // 8 no barrier, 16 no transform arithmetic, 64 (persistent form) every pixel load from one L1-resident line: the instruction
                   // count stays, the L2 / HBM traffic and latency go, 256 (persistent form) half of them, 128 (persistent form) the pixel
                   // addresses of a channel-blocked activation layout [C/16][pixel][16]: whole 128-byte lines per request, the same bytes from HBM
#endif

constexpr int W4_LD = 388;               // LDS row stride (floats): 24 components x 16 channels + 4 pad = 97 16-byte units (odd)
constexpr int W4_STAGE = 32 * W4_LD;     // floats per V stage (48.5 KB)
constexpr int W4_SMEM = 8 * 4096 + 1024; // floats: the epilogue's exchange area (8 waves x 64 registers x 64 lanes) + the GroupNorm scratch

__device__ __forceinline__ w4f32x2 w4_load8(__amdgpu_buffer_rsrc_t rsrc, int voffset, int soffset) {
    w4u32x2 v = __builtin_amdgcn_raw_buffer_load_b64(rsrc, voffset, soffset, 0);
    return __builtin_bit_cast(w4f32x2, v);
}
template <int AUX = 0>
__device__ __forceinline__ w4f32x4 w4_load16(__amdgpu_buffer_rsrc_t rsrc, int voffset, int soffset) {
    w4u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(rsrc, voffset, soffset, AUX);
    return __builtin_bit_cast(w4f32x4, v);
}
#ifndef W42_U_AUX
#define W42_U_AUX 0   // cache-policy bits of the weight-fragment loads (A/B: 1 sc0, 2 nt -> streaming, 3 both)
#endif
#ifndef W42_PX_AUX
#define W42_PX_AUX 0  // ... of the pixel loads
#endif
#ifndef W42_OUT_AUX
#define W42_OUT_AUX 2  // ... of the persistent form's output stores and residual loads: nt (streaming).  Round 5, three alternating repetitions in
                       // one box: 139.95 -> 140.46 images/s (+0.35 %), this kernel 228.4 -> 227.3 us per launch and the gn_apply passes that
                       // read its output 94.7 -> 92.2 ms per step -- the 64 KB a tile writes and the residual it reads once no longer push
                       // weight fragments and pixels out of the XCD's L2 (profiles/r05_notes.md).  0 = the default policy (A/B)
#endif
#ifndef W42_PRIO
#define W42_PRIO 0    // 1: s_setprio 1 for the second-dispatched half of the workgroup (waves 4-7) before the main loop
#endif

// packed fp32 on a channel pair (on this chip every fp32 VALU cycle is taken from the fp32 matrix pipe: one instruction for two channels)
__device__ __forceinline__ w4f32x2 pk_add(w4f32x2 a, w4f32x2 b) {
    w4f32x2 r;
    asm("v_pk_add_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
}
__device__ __forceinline__ w4f32x2 pk_sub(w4f32x2 a, w4f32x2 b) {
    w4f32x2 r;
    asm("v_pk_add_f32 %0, %1, %2 neg_lo:[0,1] neg_hi:[0,1]" : "=v"(r) : "v"(a), "v"(b));
    return r;
}
// a * K + b with K one of the inline constants 2, -2, 4, -4 (op_sel_hi:[1,0,1]: both halves multiply by the one 32-bit constant)
#define W4_PK_FMA(NAME, KSTR)                                                                            \
    __device__ __forceinline__ w4f32x2 NAME(w4f32x2 a, w4f32x2 b) {                                      \
        w4f32x2 r;                                                                                       \
        asm("v_pk_fma_f32 %0, %1, " KSTR ", %2 op_sel_hi:[1,0,1]" : "=v"(r) : "v"(a), "v"(b));           \
        return r;                                                                                        \
    }
#ifdef W42_XSCALAR  // A/B (profiles/r05_notes.md): the x transform on single floats (two VALU instructions where the packed form has one)
#undef W4_PK_FMA
__device__ __forceinline__ w4f32x2 pk_add_s(w4f32x2 a, w4f32x2 b) {
    w4f32x2 r;
    asm("v_add_f32 %0, %2, %4\n\tv_add_f32 %1, %3, %5" : "=&v"(r.x), "=&v"(r.y) : "v"(a.x), "v"(a.y), "v"(b.x), "v"(b.y));
    return r;
}
__device__ __forceinline__ w4f32x2 pk_sub_s(w4f32x2 a, w4f32x2 b) {
    w4f32x2 r;
    asm("v_sub_f32 %0, %2, %4\n\tv_sub_f32 %1, %3, %5" : "=&v"(r.x), "=&v"(r.y) : "v"(a.x), "v"(a.y), "v"(b.x), "v"(b.y));
    return r;
}
#define pk_add pk_add_s
#define pk_sub pk_sub_s
#define W4_PK_FMA(NAME, KSTR)                                                                            \
    __device__ __forceinline__ w4f32x2 NAME(w4f32x2 a, w4f32x2 b) {                                      \
        w4f32x2 r;                                                                                       \
        asm("v_fma_f32 %0, %2, " KSTR ", %4\n\tv_fma_f32 %1, %3, " KSTR ", %5"                           \
            : "=&v"(r.x), "=&v"(r.y) : "v"(a.x), "v"(a.y), "v"(b.x), "v"(b.y));                           \
        return r;                                                                                        \
    }
#endif
W4_PK_FMA(pk_fma_p2, "2.0")
W4_PK_FMA(pk_fma_m2, "-2.0")
W4_PK_FMA(pk_fma_p4, "4.0")
W4_PK_FMA(pk_fma_m4, "-4.0")
#undef W4_PK_FMA

// The tile program.  SHORTK: Cin = 32 (two chunks: no steady state), a kernel symbol of its own -- as a run-time branch inside one
// kernel the two chunk sequences made the register allocator keep the prologue's weight fragments alive across the main loop (41 spills).
template <bool SHORTK>
__device__ __forceinline__ void wino42_tile(const GemmParams &p, float *smem) {
    constexpr int OOB = (int)0x80000000u;

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wi = wave & 3, wn = wave >> 2;  // component row, output block (32 columns)
    const int li = lane & 31, lh = lane >> 5;

    // ---- XCD-aware block -> (mtile, ntile): the N tiles of an M tile get ids congruent mod 8 (one XCD, one L2 for the shared pixels) ----
    const int nt = p.n_tiles;
    int mtile, ntile;
    if (p.flat_tiles) {
        mtile = blockIdx.x / nt;
        ntile = blockIdx.x - mtile * nt;
    } else {
        const int grp = blockIdx.x / (8 * nt);
        const int rem = blockIdx.x - grp * 8 * nt;
        ntile = rem >> 3;
        mtile = grp * 8 + (rem & 7);
    }
    if (mtile >= p.m_tiles) return;
    const int t0 = mtile * 32, n0 = ntile * 64;

    const int W = p.W, H = p.H, HW = H * W;
    const int Wt = W >> 2;        // tiles per tile row: 2, 4, 8 or 16
    const int wts = p.sh_ow;      // log2(Wt) (launch_conv3x3_wino42)
    const int TI = HW >> 3;       // tiles per image (a multiple of 8)
    const int T = p.M >> 3;       // tiles in the batch
    const int Cin = p.C0 + p.C1;
    const int NK8 = Cin >> 3;     // 8-channel groups (the granularity of the weight layout)
    const int NK = Cin >> 4;      // 16-channel chunks of the K loop (even)
    const int a_imgs = p.M / HW;
    const int bias0 = (W + 1) * p.lda0 * 4, bias1 = (W + 1) * p.lda1 * 4;
    const __amdgpu_buffer_rsrc_t srd_a0 = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<char *>(reinterpret_cast<const char *>(p.a0) - bias0), 0, a_imgs * HW * p.lda0 * 4 + bias0, 0x00020000);
    const __amdgpu_buffer_rsrc_t srd_a1 = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<char *>(reinterpret_cast<const char *>(p.C1 ? p.a1 : p.a0) - bias1), 0, a_imgs * HW * p.lda1 * 4 + bias1, 0x00020000);
    const int nblk = (p.N + 31) >> 5;
    const __amdgpu_buffer_rsrc_t srd_u = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<char *>(reinterpret_cast<const char *>(p.b_wino)), 0, nblk * NK8 * (24 * 1024), 0x00020000);

    // ---- staging role: (tile tl, channel quad chq, patch row r).  Thread r loads row r of the tile's 4 x 6 patch (six pixels x 16 B: four
    // of the chunk's 16 channels).  By^T row r combines two patch rows -- 0: r0 - r2, 1: r1 + r2, 2: r2 - r1, 3: r1 - r3 -- and the
    // partner row sits in a neighbouring lane of the same 4-lane group (lanes 2, 2, 1, 1): ONE v_fmac_f32 with a DPP quad_perm operand
    // per element, mine + sb * partner.  Row 3 comes out NEGATED that way (r3 - r1) and so is row 3 of the transformed weights. ----
    const int tl = tid >> 4, chq = (tid >> 2) & 3, sr = tid & 3;
    const float sb = sr == 1 ? 1.f : -1.f;
    int pix = 0;           // pixel index of the patch origin (2 ty - 1, 4 tx - 1); may be "negative" by at most W + 1: the bias covers it
    unsigned bad = 0x3Fu;  // bit s: pixel s of the thread's patch row lies outside the image (or the tile is past the batch)
    {
        const int t = t0 + tl;
        if (t < T) {
            const int img = (int)epi_div((unsigned)t, (unsigned)TI, p.mg_ohw, p.sh_ohw);
            const int tr = t - img * TI;
            const int ty = tr >> wts;
            const int tx = tr - (ty << wts);
            const int y0 = 2 * ty - 1, x0 = 4 * tx - 1;
            pix = (img * H + y0) * W + x0;
            unsigned xb = 0;
#pragma unroll
            for (int s = 0; s < 6; ++s) xb |= ((unsigned)(x0 + s) < (unsigned)W ? 0u : 1u) << s;
            bad = (unsigned)(y0 + sr) < (unsigned)H ? xb : 0x3Fu;
        }
    }
    int vo[6];  // voffsets of the thread's 6 pixels in the current source (out-of-image ones already out of bounds)
    auto set_vo = [&](int lda, int biasb) __attribute__((always_inline)) {
#pragma unroll
        for (int k = 0; k < 6; ++k) {
            const int off = (pix + sr * W + k) * lda * 4 + biasb + chq * 16;
            vo[k] = ((bad >> k) & 1u) ? OOB : off;
        }
    };
    // V row of a tile: component (i, j) at (j * 4 + i) * 16 floats, the chunk's 16 channels in order.
    const int v_lds = tl * W4_LD + sr * 16 + chq * 4;  // + j * 64 (+ 2 for the quad's second channel pair)

    // ---- wave-uniform state of the chunk whose PIXEL loads are issued next (two chunks ahead of the multiplication) ----
    int s_c = 0;
    bool s_first = true;  // s_c < C0: the chunk comes from source 0
    auto advance = [&]() __attribute__((always_inline)) {
        s_c += 16;
        if (s_first && s_c >= p.C0 && p.C1 > 0) {  // once per tile: on to the second source of a channel concat
            s_first = false;
            set_vo(p.lda1, bias1);
        }
    };
    set_vo(p.lda0, bias0);

    w4f32x4 rr[6];
    w4f32x2 vv6[6];
    auto load_raw_to = [&](w4f32x4 *dst, int k) __attribute__((always_inline)) {
        if ((W42_ABL & 1) && s_c >= 32) return;
        if (s_first) dst[k] = w4_load16(srd_a0, vo[k], s_c * 4);
        else dst[k] = w4_load16(srd_a1, vo[k], (s_c - p.C0) * 4);
    };
    auto load_raw = [&](int k) __attribute__((always_inline)) { load_raw_to(rr, k); };
    // y transform IN PLACE (the DPP operand reads the partner lane's old value in the same instruction): x[s] <- By^T row sr of the patch
    auto combine_in = [&](w4f32x4 *x) __attribute__((always_inline)) {
        if (W42_ABL & 16) return;
#pragma unroll
        for (int s = 0; s < 6; ++s) {
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                float m = x[s][e];
                // (s_nop: a VALU write of an operand right in front of a DPP read needs two wait states, and inline asm is opaque to the
                // hazard recogniser)
                if (s == 0 && e == 0)
                    asm volatile("s_nop 1\n\tv_fmac_f32_dpp %0, %0, %1 quad_perm:[2,2,1,1] row_mask:0xf bank_mask:0xf" : "+v"(m) : "v"(sb));
                else
                    asm volatile("v_fmac_f32_dpp %0, %0, %1 quad_perm:[2,2,1,1] row_mask:0xf bank_mask:0xf" : "+v"(m) : "v"(sb));
                x[s][e] = m;
            }
        }
    };
    // x transform F(4,3) of channel pair h of the combined row cm[0..5] -> vv6[0..5] (Bx^T above, factored: 12 packed instructions)
    auto xform = [&](const w4f32x4 *cm, int h) __attribute__((always_inline)) {
        w4f32x2 d[6];
#pragma unroll
        for (int s = 0; s < 6; ++s) d[s] = w4f32x2{cm[s][2 * h], cm[s][2 * h + 1]};
        if (W42_ABL & 16) {
#pragma unroll
            for (int j = 0; j < 6; ++j) vv6[j] = d[j];
            return;
        }
        const w4f32x2 t3 = pk_sub(d[4], d[2]);          // d4 - d2
        const w4f32x2 t4 = pk_sub(d[3], d[1]);          // d3 - d1
        const w4f32x2 u0 = pk_sub(d[0], d[2]);          // d0 - d2
        vv6[0] = pk_fma_p4(u0, t3);                     // 4 d0 - 5 d2 + d4
        const w4f32x2 t1 = pk_fma_m4(d[2], d[4]);       // d4 - 4 d2
        const w4f32x2 t2 = pk_fma_m4(d[1], d[3]);       // d3 - 4 d1
        vv6[1] = pk_add(t1, t2);
        vv6[2] = pk_sub(t1, t2);
        vv6[3] = pk_fma_p2(t4, t3);                     // (d4 - d2) + 2 (d3 - d1)
        vv6[4] = pk_fma_m2(t4, t3);
        const w4f32x2 u5 = pk_sub(d[5], d[3]);          // d5 - d3
        vv6[5] = pk_fma_m4(t4, u5);                     // 4 d1 - 5 d3 + d5
    };
    auto store_v = [&](int j, int par, int h) __attribute__((always_inline)) {
        if (W42_ABL & 2) { asm volatile("" ::"v"(vv6[j])); return; }
        *reinterpret_cast<w4f32x2 *>(&smem[v_lds + par * W4_STAGE + j * 64 + 2 * h]) = vv6[j];
    };

    // ---- weight fragments: wave (wi, wn), component (wi, j), channel group g (8 channels): 1 KB at
    // ((nb * NK8 + 2 kc + g) * 24 + wi * 6 + j) * 1024 bytes, lane * 16 ----
    w4f32x4 bf[6][2];
    const int nb = ntile * 2 + wn;
    int s_ub = ((nb * NK8) * 24 + wi * 6) * 1024;  // byte offset of component (wi, 0), group 0 of the chunk whose fragments are loaded next
    const int u_vo = nb < nblk ? lane * 16 : OOB;
    auto load_u = [&](int j, int g) __attribute__((always_inline)) { bf[j][g] = w4_load16(srd_u, u_vo, s_ub + g * (24 * 1024) + j * 1024); };

    w4f32x16 acc[6];  // (never zeroed: the first MFMA of every accumulator, in the tile's first chunk, takes the constant 0 as its C operand)
    const int a_off = li * W4_LD + wi * 16 + lh * 4;  // + j * 64 + g * 8: fragment of component (wi, j), channel group g

    // ---- prologue: weight fragments of chunk 0, pixels of chunks 0 and 1 requested up front; chunk 0 -> LDS stage 0 ----
    {
        w4f32x4 r0[6];
#pragma unroll
        for (int j = 0; j < 6; ++j) { load_u(j, 0); load_u(j, 1); }
        s_ub += 48 * 1024;
#pragma unroll
        for (int k = 0; k < 6; ++k) load_raw_to(r0, k);
        advance();
#pragma unroll
        for (int k = 0; k < 6; ++k) load_raw(k);
        combine_in(r0);
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            xform(r0, h);
#pragma unroll
            for (int j = 0; j < 6; ++j) store_v(j, 0, h);
        }
    }
    advance();
    __syncthreads();

    // ---- epilogue geometry: wave f = wi of output block wn finishes the unit of 8 tiles t0 + 8 f .. + 7 x 32 columns.  Set right before the
    // tile's LAST chunk, which requests the residual tile (nothing here is live across the K loop: the loop is register-bound) ----
    const int f = wi;
    const int tu = t0 + 8 * f;  // first tile of the unit (wave-uniform)
    int img = 0, mu = 0, n = 0, lane_px = 0, vo_r = OOB;
    bool nok = false;
    __amdgpu_buffer_rsrc_t srd_r = srd_u;
    auto set_geometry = [&]() __attribute__((always_inline)) {
        const int tuc = tu < T ? tu : 0;
        img = (int)epi_div((unsigned)tuc, (unsigned)TI, p.mg_ohw, p.sh_ohw);
        const int tru = tuc - img * TI;
        const int tyu = tru >> wts, txu = tru - (tyu << wts);
        mu = img * HW + 2 * tyu * W + 4 * txu;  // pixel of the unit's first tile (top-left output)
        n = n0 + wn * 32 + li;
        nok = n < p.N && tu < T;
        srd_r = __builtin_amdgcn_make_buffer_rsrc(const_cast<char *>(reinterpret_cast<const char *>(p.res ? p.res : p.out)), 0,
                                                  p.M * p.ldres * 4, 0x00020000);
        // tile k = 4 lh + vv of the unit.  Wt >= 8: eight tiles of one tile row (the half-wave adds 16 pixels along x); Wt = 4: two tile
        // rows of four (the half-wave adds a tile row = 2 W pixels); Wt = 2: four tile rows of two (the half-wave adds two tile rows)
        lane_px = Wt >= 8 ? lh * 16 : Wt == 4 ? lh * 2 * W : lh * 4 * W;
        vo_r = nok ? (lane_px * p.ldres + n) * 4 : OOB;
    };
    auto pix_off = [&](int vv, int a, int b) __attribute__((always_inline)) {
        const int base = Wt == 2 ? (vv >> 1) * 2 * W + (vv & 1) * 4 : vv * 4;
        return mu + base + a * W + b;
    };
    float rs[2][4][4];
    const bool has_res = p.res != nullptr;
    auto load_res = [&](int idx) __attribute__((always_inline)) {  // idx = (a * 4 + b) * 4 + vv
        const int a = idx >> 4, b = (idx >> 2) & 3, vv = idx & 3;
        rs[a][b][vv] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(srd_r, vo_r, pix_off(vv, a, b) * p.ldres * 4, 0));
    };

    // ---- K loop over 16-channel chunks: 48 MFMAs per wave between barriers, in 24 gaps of two (two components side by side: no
    // MFMA waits for its predecessor's accumulator).  MODE 0: steady state (chunk kc multiplied; its successor's pixels registers -> the
    // other LDS stage and weight fragments requested into the registers just consumed; chunk kc+2's pixels requested); MODE 1: the
    // second-to-last chunk (no pixels left to request); MODE 2: the last chunk (nothing staged; the residual tile is requested) ----
    w4f32x4 af[2][2];  // [buffer][component of the pair]
    auto chunk = [&](auto mode_tag, auto par_tag, auto first_tag) __attribute__((always_inline)) {
        constexpr int MODE = decltype(mode_tag)::value;
        constexpr int PAR = decltype(par_tag)::value;  // stage the chunk is multiplied from (compile time: immediate offsets)
        constexpr bool FIRST = decltype(first_tag)::value != 0;
        const float *cur = smem + PAR * W4_STAGE;
        constexpr int nxt = PAR ^ 1;
        af[0][0] = *reinterpret_cast<const w4f32x4 *>(cur + a_off);
        af[0][1] = *reinterpret_cast<const w4f32x4 *>(cur + a_off + 64);
#pragma unroll
        for (int jp = 0; jp < 3; ++jp) {
#pragma unroll
            for (int g = 0; g < 2; ++g) {
                const int grp = jp * 2 + g, buf = grp & 1;
#pragma unroll
                for (int jj = 0; jj < 4; ++jj) {
                    if (FIRST && g == 0 && jj == 0) {
                        const w4f32x16 zero = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
                        acc[2 * jp] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[buf][0][jj], bf[2 * jp][g][jj], zero, 0, 0, 0);
                        acc[2 * jp + 1] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[buf][1][jj], bf[2 * jp + 1][g][jj], zero, 0, 0, 0);
                    } else {
                        acc[2 * jp] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[buf][0][jj], bf[2 * jp][g][jj], acc[2 * jp], 0, 0, 0);
                        acc[2 * jp + 1] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[buf][1][jj], bf[2 * jp + 1][g][jj], acc[2 * jp + 1], 0, 0, 0);
                    }
                    __builtin_amdgcn_sched_barrier(0);
                    const int gap = grp * 4 + jj;  // 24 gaps of two MFMAs
                    if (jj == 0 && grp < 5) {  // the next group's fragments, one group ahead of their use
                        const int nj = (grp + 1) >> 1, ng = (grp + 1) & 1;
                        af[buf ^ 1][0] = *reinterpret_cast<const w4f32x4 *>(cur + a_off + (2 * nj) * 64 + ng * 8);
                        af[buf ^ 1][1] = *reinterpret_cast<const w4f32x4 *>(cur + a_off + (2 * nj + 1) * 64 + ng * 8);
                    }
                    if (jj == 3 && MODE < 2 && !((W42_ABL & 4) && !FIRST)) {  // the next chunk's fragments of this group, into the registers just consumed
                        load_u(2 * jp, g);
                        load_u(2 * jp + 1, g);
                    }
                    if (MODE < 2) {
                        if (gap == 6) combine_in(rr);
                        if (gap == 7) xform(rr, 0);
                        if (gap >= 8 && gap <= 10) { store_v(2 * (gap - 8), nxt, 0); store_v(2 * (gap - 8) + 1, nxt, 0); }
                        if (gap == 11) xform(rr, 1);
                        if (gap >= 12 && gap <= 14) { store_v(2 * (gap - 12), nxt, 1); store_v(2 * (gap - 12) + 1, nxt, 1); }
                    }
                    if (MODE == 0) {
                        if (gap >= 12 && gap <= 17) load_raw(gap - 12);  // (rr is free once its second channel pair is transformed)
                        if (gap == 18) advance();
                    }
                    if (MODE == 2 && has_res) {
                        if (gap < 16) { load_res(2 * gap); load_res(2 * gap + 1); }
                    }
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
        }
        if (MODE < 2) s_ub += 48 * 1024;
        if (!(W42_ABL & 8)) __syncthreads();
    };
    using I0 = std::integral_constant<int, 0>;
    using I1 = std::integral_constant<int, 1>;
    using I2 = std::integral_constant<int, 2>;
    if (SHORTK) {  // Cin = 32
        chunk(I1{}, I0{}, I1{});
        set_geometry();
        chunk(I2{}, I1{}, I0{});
    } else {       // (NK is even and >= 4: Cin % 32 == 0)
        chunk(I0{}, I0{}, I1{});
        chunk(I0{}, I1{}, I0{});
        for (int kc = 2; kc < NK - 2; kc += 2) {
            chunk(I0{}, I0{}, I0{});
            chunk(I0{}, I1{}, I0{});
        }
        chunk(I1{}, I0{}, I0{});
        set_geometry();
        chunk(I2{}, I1{}, I0{});
    }

    // ---- output transform along x (in-lane): z[b] = sum_j Ax^T[b][j] M_ij; then along y across the four component-row waves of an
    // output block, through LDS (the stages are idle after the last barrier) ----
    float *mine = smem + (wn * 4 + wi) * 4096 + lane;
#pragma unroll
    for (int v = 0; v < 16; ++v) {
        const float m0 = acc[0][v], m1 = acc[1][v], m2 = acc[2][v], m3 = acc[3][v], m4 = acc[4][v], m5 = acc[5][v];
        const float s1 = m1 + m2, d1 = m1 - m2, s2 = m3 + m4, d2 = m3 - m4;
        mine[(0 * 16 + v) * 64] = (m0 + s1) + s2;
        mine[(1 * 16 + v) * 64] = d1 + 2.f * d2;
        mine[(2 * 16 + v) * 64] = s1 + 4.f * s2;
        mine[(3 * 16 + v) * 64] = (d1 + 8.f * d2) + m5;
    }
    __syncthreads();
    if (tu >= T) return;
    float y[2][4][4];  // [a][b][vv]: output pixel (2 ty + a, 4 tx + b) of tile 4 lh + vv of the unit
    {
        const float *src = smem + (wn * 4) * 4096 + lane;
#pragma unroll
        for (int b = 0; b < 4; ++b)
#pragma unroll
            for (int vv = 0; vv < 4; ++vv) {
                const int o = (b * 16 + 4 * f + vv) * 64;
                const float z0 = src[o], z1 = src[4096 + o], z2 = src[2 * 4096 + o], z3 = src[3 * 4096 + o];
                y[0][b][vv] = z0 + z1 + z2;
                y[1][b][vv] = z1 - z2 - z3;
            }
    }

    // ---- epilogue of the unit ----
    const __amdgpu_buffer_rsrc_t srd_o =
        __builtin_amdgcn_make_buffer_rsrc(reinterpret_cast<char *>(p.out), 0, p.M * p.ldc * 4, 0x00020000);
    const int vo_o = nok ? (lane_px * p.ldc + n) * 4 : OOB;
    float addv = (p.bias && nok) ? p.bias[n] : 0.f;
    if (p.rowbias && nok) addv += p.rowbias[(size_t)img * p.ld_rowbias + n];
    float ps[2] = {0.f, 0.f}, pq[2] = {0.f, 0.f};  // partial statistics by output row a
    float vals[2][4][4];                           // finished outputs (kept for the epilogue GroupNorm)
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 4; ++b)
#pragma unroll
            for (int vv = 0; vv < 4; ++vv) {
                float add = addv;
                if (has_res) add += rs[a][b][vv];
                const float val = (p.alpha * y[a][b][vv] + add) * p.out_scale;
                vals[a][b][vv] = val;
                if (!p.egn_skip_raw)
                    __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, val), srd_o, vo_o, pix_off(vv, a, b) * p.ldc * 4, 0);
                ps[a] += val;
                pq[a] += val * val;
            }
    if (p.egn_out) {
        // GroupNorm (+ swish) of the output, here (plan.h: Op::egn_peer).  rows_per_image == 64 (8x8 images): this unit is exactly one
        // image x 32 columns.  rows_per_image == 256 (16x16 images): the image is the whole workgroup tile, and the four units f of an
        // output block add their column sums through the 4 KB of LDS behind the exchange area.  A group is egn_cpg consecutive
        // columns = consecutive lanes.
        float gsum = ps[0] + ps[1], gsq = pq[0] + pq[1];
        gsum += __shfl_xor(gsum, 32); gsq += __shfl_xor(gsq, 32);
        float rows = 64.0f;
        if (p.rows_per_image == 256) {
            float2 *red = reinterpret_cast<float2 *>(smem + 8 * 4096);  // [wn][unit][32 columns]
            if (lh == 0) red[(wn * 4 + f) * 32 + li] = make_float2(gsum, gsq);
            __syncthreads();  // (every wave of the workgroup is here: whole images, so no unit is past the batch)
            gsum = 0.f; gsq = 0.f;
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const float2 v = red[(wn * 4 + u) * 32 + li];
                gsum += v.x; gsq += v.y;
            }
            rows = 256.0f;
        }
        for (int o = 1; o < p.egn_cpg; o <<= 1) { gsum += __shfl_xor(gsum, o); gsq += __shfl_xor(gsq, o); }
        const float inv_n = 1.0f / (rows * (float)p.egn_cpg);
        const float mean = gsum * inv_n;
        const float rstd = rsqrtf(gsq * inv_n - mean * mean + 1e-6f);
        const float ga = nok ? p.egn_gamma[n] : 0.f, be = nok ? p.egn_beta[n] : 0.f;
        const float sc = rstd * ga, sh = be - mean * rstd * ga;
        const __amdgpu_buffer_rsrc_t srd_g =
            __builtin_amdgcn_make_buffer_rsrc(reinterpret_cast<char *>(p.egn_out), 0, p.M * p.ldc * 4, 0x00020000);
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
            for (int b = 0; b < 4; ++b)
#pragma unroll
                for (int vv = 0; vv < 4; ++vv) {
                    float t = vals[a][b][vv] * sc + sh;
                    if (p.egn_act) t = swish_e(t);
                    __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, t), srd_g, vo_o, pix_off(vv, a, b) * p.ldc * 4, 0);
                }
    }
    if (p.qstats) {
        // the unit's 64 pixels are two 32-row slices of the output matrix.  W = 32 / 64: its two pixel rows a (a whole row / an aligned
        // half of one), both half-waves; W = 8 / 16: the half-wave's four tiles (two tile rows of 2 / one tile row of 4, both pixel rows)
        float s0, q0, s1, q1;
        if (Wt >= 8) {
            s0 = ps[0] + __shfl_xor(ps[0], 32); q0 = pq[0] + __shfl_xor(pq[0], 32);
            s1 = ps[1] + __shfl_xor(ps[1], 32); q1 = pq[1] + __shfl_xor(pq[1], 32);
        } else {
            const float ts = ps[0] + ps[1], tq = pq[0] + pq[1];
            const float os = __shfl_xor(ts, 32), oq = __shfl_xor(tq, 32);
            s0 = lh == 0 ? ts : os; q0 = lh == 0 ? tq : oq;
            s1 = lh == 0 ? os : ts; q1 = lh == 0 ? oq : tq;
        }
#pragma unroll
        for (int g = 0; g < 2; ++g) {
            float s = g == 0 ? s0 : s1, q = g == 0 ? q0 : q1;
            if (!nok) { s = 0.f; q = 0.f; }
            s += __shfl_xor(s, 1); q += __shfl_xor(q, 1);
            s += __shfl_xor(s, 2); q += __shfl_xor(q, 2);
            if ((li & 3) == 0 && lh == 0 && nok) {
                const int first = (mu - img * HW) + (Wt >= 8 ? g * W : g * 32);  // first pixel of the slice inside the image
                float2 *o = reinterpret_cast<float2 *>(p.qstats) + ((size_t)img * p.qs_slices + (first >> 5)) * (p.N >> 2) + (n >> 2);
                o[0] = make_float2(s, q);
            }
        }
    }
}

// ---- the persistent form (Cin >= 64) -------------------------------------------------------------------------------------------------
// One workgroup per CU walks over tiles vb = blockIdx.x + k gridDim.x, and the K-loop pipeline RUNS ACROSS TILES: the loading state of
// the pixels (two chunks ahead of the multiplication) and of the weight fragments (one chunk ahead) simply rolls over to the next tile
// when the current one has no channels left, so the last two chunks of a tile stage the first two of its successor and only the first
// tile of a workgroup has a prologue.  The epilogue then runs with the successor's first loads in flight.  Per tile this removes the
// workgroup launch, the exposed first-load latency and the kernel-argument set-up: measured ~11 us of every tile's 33-100 us with one
// workgroup per tile (profiles/r03_notes.md).  LDS: V stage 0 | V stage 1, the epilogue's exchange area over stage 1 onwards (stage 0
// already holds the successor's first chunk); a wave keeps the z values of its own unit in registers, so the area is 96 KB, not 128.
constexpr int W4P_EXB = W4_STAGE;                       // exchange area base (floats)
constexpr int W4P_EXW = 48 * 64;                        // floats per wave in it (48 registers x 64 lanes)
constexpr int W4P_RED = W4P_EXB + 8 * W4P_EXW;          // GroupNorm scratch: [wn][unit][32 columns] (sum, sum of squares)
constexpr int W4P_SMEM = W4P_RED + 512;                 // 149 504 bytes

// NW: output blocks (32 columns) per workgroup tile.  2: 512 threads, 32 tiles x 64 outputs -- the production tile.  1: 256 threads = the four
// component-row waves of ONE output block, 32 tiles x 32 outputs, for launches that would leave CUs idle with the larger tile (small
// batches, the 8x8 level at the reference's per-device batch of 128): twice the workgroups, and every thread stages TWO tiles' patch rows
// (slots u = 0, 1: tiles tl and tl + 16) -- one wave per SIMD may hold 512 registers, so the second register set costs nothing here.
template <int NW>
__device__ __forceinline__ void wino42_persistent(const GemmParams &p, float *smem) {
    constexpr int S = 2 / NW;  // staging slots per thread
    constexpr int OOB = (int)0x80000000u;

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wi = wave & 3, wn = wave >> 2;  // component row, output block (32 columns)
    const int li = lane & 31, lh = lane >> 5;

    const int W = p.W, H = p.H, HW = H * W;
    const int Wt = W >> 2;        // tiles per tile row: 2, 4, 8 or 16
    const int wts = p.sh_ow;      // log2(Wt) (launch_conv3x3_wino42)
    const int TI = HW >> 3;       // tiles per image (a multiple of 8)
    const int T = p.M >> 3;       // tiles in the batch
    const int Cin = p.C0 + p.C1;
    const int NK8 = Cin >> 3;     // 8-channel groups (the granularity of the weight layout)
    const int NK = Cin >> 4;      // 16-channel chunks of the K loop (even, >= 4)
    const int a_imgs = p.M / HW;
    const int bias0 = (W + 1) * p.lda0 * 4, bias1 = (W + 1) * p.lda1 * 4;
    const __amdgpu_buffer_rsrc_t srd_a0 = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<char *>(reinterpret_cast<const char *>(p.a0) - bias0), 0, a_imgs * HW * p.lda0 * 4 + bias0, 0x00020000);
    const __amdgpu_buffer_rsrc_t srd_a1 = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<char *>(reinterpret_cast<const char *>(p.C1 ? p.a1 : p.a0) - bias1), 0, a_imgs * HW * p.lda1 * 4 + bias1, 0x00020000);
    const int nblk = (p.N + 31) >> 5;
    const __amdgpu_buffer_rsrc_t srd_u = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<char *>(reinterpret_cast<const char *>(p.b_wino)), 0, nblk * NK8 * (24 * 1024), 0x00020000);

    // ---- virtual block id -> (mtile, ntile), XCD-aware as in the one-tile form (gridDim.x is a multiple of 8 or the whole tile count) ----
    const int nt = p.n_tiles;
#ifdef W42_XCDMAP  // A/B (profiles/r05_notes.md): FOUR consecutive M tiles share an XCD (ids congruent mod 8) -- the M tiles of a 32x32 image, whose
                   // neighbours read each other's halo rows (2 of the 10 pixel rows a tile reads) -- instead of one; groups of 32 M tiles
    const int nvb = p.flat_tiles ? p.m_tiles * nt : ((p.m_tiles + 31) / 32) * 32 * nt;
    auto decode = [&](int vb, int &mt, int &ntl) __attribute__((always_inline)) {
        if (p.flat_tiles) {
            mt = vb / nt;
            ntl = vb - mt * nt;
        } else {
            const int grp = vb / (32 * nt);
            const int rem = vb - grp * 32 * nt;
            ntl = rem >> 5;
            const int l = rem & 31;                      // id inside the group: XCD = l & 7
            mt = grp * 32 + (l & 7) * 4 + (l >> 3);      // M tiles 4 x .. 4 x + 3 get ids x, x + 8, x + 16, x + 24
        }
    };
#else
    const int nvb = p.flat_tiles ? p.m_tiles * nt : ((p.m_tiles + 7) / 8) * 8 * nt;
    auto decode = [&](int vb, int &mt, int &ntl) __attribute__((always_inline)) {
        if (p.flat_tiles) {
            mt = vb / nt;
            ntl = vb - mt * nt;
        } else {
            const int grp = vb / (8 * nt);
            const int rem = vb - grp * 8 * nt;
            ntl = rem >> 3;
            mt = grp * 8 + (rem & 7);
        }
    };
#endif
    // the first valid tile at or after vb (stride gridDim.x); vb >= nvb: none
    auto next_valid = [&](int vb, int &mt, int &ntl) __attribute__((always_inline)) {
        while (vb < nvb) {
            decode(vb, mt, ntl);
            if (mt < p.m_tiles) break;
            vb += (int)gridDim.x;
        }
        return vb;
    };
    int cur_vb, cur_mt = 0, cur_nt = 0;
    cur_vb = next_valid((int)blockIdx.x, cur_mt, cur_nt);
    if (cur_vb >= nvb) return;
#ifdef W42_TRACE  // in-kernel timeline (tools/wino42_trace.py): thread 0 of workgroups 0 and 100 stamps the 100 MHz wall clock and the shader
                  // clock at six (-DW42_TRACE=2: eight) points of its first tiles into the tail of the residual tensor
    int trn = 0;
    const int trslot = blockIdx.x == 0 ? 0 : blockIdx.x == 100 ? 1 : -1;
    float *trbuf = const_cast<float *>(p.res) + (size_t)p.M * p.ldres - 128 + (trslot < 0 ? 0 : trslot) * 64;
    const bool trw = trslot >= 0 && tid == 0 && p.res;
    auto trace = [&]() __attribute__((always_inline)) {
        if (trw && trn < 24) {
            trbuf[trn] = (float)(wall_clock64() & 0xFFFFFFull);
            trbuf[32 + trn] = (float)(__builtin_readcyclecounter() & 0xFFFFFFull);
        }
        ++trn;
    };
#endif
    // the workgroup's next tile (nx_vb >= nvb: none)
    int nx_vb, nx_mt = 0, nx_nt = 0;
    nx_vb = next_valid(cur_vb + (int)gridDim.x, nx_mt, nx_nt);

    // ---- staging role: (tile tl, channel quad chq, patch row r), see wino42_tile; NW = 1: slot u stages tile tl + 16 u ----
    const int tl = tid >> 4, chq = (tid >> 2) & 3, sr = tid & 3;
    const float sb = sr == 1 ? 1.f : -1.f;
    int pix[S];
    unsigned bad[S];
    int vo[S][6];
    auto set_vo = [&](int lda, int biasb) __attribute__((always_inline)) {
#pragma unroll
        for (int u = 0; u < S; ++u)
#pragma unroll
            for (int k = 0; k < 6; ++k) {
                int off = (pix[u] + sr * W + k) * lda * 4 + biasb + chq * 16;
                if (W42_ABL & 128) off = (pix[u] + sr * W + k) * 64 + biasb + chq * 16;  // timing ablation: a channel-blocked layout [C/16][pixel][16]
                vo[u][k] = ((bad[u] >> k) & 1u) ? OOB : off;
            }
    };
    // pixel-loading state -> M tile mt (mt < 0: no tile, every load out of bounds)
    auto setup_pix = [&](int mt) __attribute__((always_inline)) {
#pragma unroll
        for (int u = 0; u < S; ++u) {
            pix[u] = 0;
            bad[u] = 0x3Fu;
            const int t = mt * 32 + tl + 16 * u;
            if (mt >= 0 && t < T) {
                const int img = (int)epi_div((unsigned)t, (unsigned)TI, p.mg_ohw, p.sh_ohw);
                const int tr = t - img * TI;
                const int ty = tr >> wts;
                const int tx = tr - (ty << wts);
                const int y0 = 2 * ty - 1, x0 = 4 * tx - 1;
                pix[u] = (img * H + y0) * W + x0;
                unsigned xb = 0;
#pragma unroll
                for (int s = 0; s < 6; ++s) xb |= ((unsigned)(x0 + s) < (unsigned)W ? 0u : 1u) << s;
                bad[u] = (unsigned)(y0 + sr) < (unsigned)H ? xb : 0x3Fu;
            }
        }
        set_vo(p.lda0, bias0);
    };
    const int v_lds = tl * W4_LD + sr * 16 + chq * 4;  // + u * 16 tiles + j * 64 (+ 2 for the quad's second channel pair)

    int s_c = 0;
    bool s_first = true;
    auto px_advance = [&]() __attribute__((always_inline)) {
        s_c += 16;
        if (s_c >= Cin) {  // the tile has no channels left: on to the next tile of this workgroup (or to nothing)
            s_c = 0;
            s_first = true;
            setup_pix(nx_vb < nvb ? nx_mt : -1);
        } else if (s_first && s_c >= p.C0 && p.C1 > 0) {  // on to the second source of a channel concat
            s_first = false;
            set_vo(p.lda1, bias1);
        }
    };

    w4f32x4 rr[S][6];
    w4f32x2 vv6[S][6];  // (a slot of its own each: the second slot's last stores run beside the first slot's first transform)
    bool steady = false;  // (timing ablations only: set after the prologue)
    auto load_raw_to = [&](w4f32x4 *dst, int k, int u) __attribute__((always_inline)) {
        if ((W42_ABL & 1) && steady) return;
        if ((W42_ABL & 64) && steady) { dst[k] = w4_load16<W42_PX_AUX>(srd_a0, bias0 + (lane & 3) * 16, 0); return; }  // every pixel load from ONE L1-resident line
        if ((W42_ABL & 256) && steady && k >= 3) { dst[k] = w4_load16<W42_PX_AUX>(srd_a0, bias0 + (lane & 3) * 16, 0); return; }  // ... half of them
        if (W42_ABL & 128) {  // (whole 128-byte lines per request: pixels x 64 B contiguous inside a 16-channel block of a_imgs HW pixels)
            if (s_first) dst[k] = w4_load16<W42_PX_AUX>(srd_a0, vo[u][k], (s_c >> 4) * a_imgs * HW * 64);
            else dst[k] = w4_load16<W42_PX_AUX>(srd_a1, vo[u][k], ((s_c - p.C0) >> 4) * a_imgs * HW * 64);
            return;
        }
        if (s_first) dst[k] = w4_load16<W42_PX_AUX>(srd_a0, vo[u][k], s_c * 4);
        else dst[k] = w4_load16<W42_PX_AUX>(srd_a1, vo[u][k], (s_c - p.C0) * 4);
    };
    auto load_raw = [&](int k, int u) __attribute__((always_inline)) { load_raw_to(rr[u], k, u); };
    auto combine_in = [&](w4f32x4 *x) __attribute__((always_inline)) {
        if (W42_ABL & 16) return;
#pragma unroll
        for (int s = 0; s < 6; ++s) {
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                float m = x[s][e];
                if (s == 0 && e == 0)
                    asm volatile("s_nop 1\n\tv_fmac_f32_dpp %0, %0, %1 quad_perm:[2,2,1,1] row_mask:0xf bank_mask:0xf" : "+v"(m) : "v"(sb));
                else
                    asm volatile("v_fmac_f32_dpp %0, %0, %1 quad_perm:[2,2,1,1] row_mask:0xf bank_mask:0xf" : "+v"(m) : "v"(sb));
                x[s][e] = m;
            }
        }
    };
    auto xform = [&](const w4f32x4 *cm, int h, int u) __attribute__((always_inline)) {
        w4f32x2 d[6];
#pragma unroll
        for (int s = 0; s < 6; ++s) d[s] = w4f32x2{cm[s][2 * h], cm[s][2 * h + 1]};
        if (W42_ABL & 16) {
#pragma unroll
            for (int j = 0; j < 6; ++j) vv6[u][j] = d[j];
            return;
        }
        const w4f32x2 t3 = pk_sub(d[4], d[2]);
        const w4f32x2 t4 = pk_sub(d[3], d[1]);
        const w4f32x2 u0 = pk_sub(d[0], d[2]);
        vv6[u][0] = pk_fma_p4(u0, t3);
        const w4f32x2 t1 = pk_fma_m4(d[2], d[4]);
        const w4f32x2 t2 = pk_fma_m4(d[1], d[3]);
        vv6[u][1] = pk_add(t1, t2);
        vv6[u][2] = pk_sub(t1, t2);
        vv6[u][3] = pk_fma_p2(t4, t3);
        vv6[u][4] = pk_fma_m2(t4, t3);
        const w4f32x2 u5 = pk_sub(d[5], d[3]);
        vv6[u][5] = pk_fma_m4(t4, u5);
    };
    auto store_v = [&](int j, int par, int h, int u) __attribute__((always_inline)) {
        if (W42_ABL & 2) { asm volatile("" ::"v"(vv6[u][j])); return; }
        *reinterpret_cast<w4f32x2 *>(&smem[v_lds + u * 16 * W4_LD + par * W4_STAGE + j * 64 + 2 * h]) = vv6[u][j];
    };

    // ---- weight-fragment loading state (one chunk ahead) ----
    w4f32x4 bf[6][2];
    int s_ub = 0, u_vo = OOB, ukc = 0;
    auto setup_u = [&](int ntl) __attribute__((always_inline)) {  // ntl < 0: no tile
        const int nb = ntl * NW + wn;
        s_ub = ((nb * NK8) * 24 + wi * 6) * 1024;
        u_vo = (ntl >= 0 && nb < nblk) ? lane * 16 : OOB;
        ukc = 0;
    };
    auto u_advance = [&]() __attribute__((always_inline)) {
        s_ub += 48 * 1024;
        if (++ukc == NK) setup_u(nx_vb < nvb ? nx_nt : -1);
    };
    auto load_u = [&](int j, int g) __attribute__((always_inline)) {
        if ((W42_ABL & 4) && steady) return;
        bf[j][g] = w4_load16<W42_U_AUX>(srd_u, u_vo, s_ub + g * (24 * 1024) + j * 1024);
    };

    w4f32x16 acc[6];
    const int a_off = li * W4_LD + wi * 16 + lh * 4;  // + j * 64 + g * 8

    // ---- prologue (first tile of the workgroup only) ----
    setup_pix(cur_mt);
    setup_u(cur_nt);
    {
        w4f32x4 r0[S][6];
#pragma unroll
        for (int j = 0; j < 6; ++j) { load_u(j, 0); load_u(j, 1); }
        u_advance();
#pragma unroll
        for (int u = 0; u < S; ++u)
#pragma unroll
            for (int k = 0; k < 6; ++k) load_raw_to(r0[u], k, u);
        px_advance();
#pragma unroll
        for (int u = 0; u < S; ++u)
#pragma unroll
            for (int k = 0; k < 6; ++k) load_raw(k, u);
        px_advance();
#pragma unroll
        for (int u = 0; u < S; ++u) {
            combine_in(r0[u]);
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                xform(r0[u], h, u);
#pragma unroll
                for (int j = 0; j < 6; ++j) store_v(j, 0, h, u);
            }
        }
    }
    __syncthreads();

    // ---- epilogue geometry of the CURRENT tile (set before its last chunk, which requests the residual) ----
    const int f = wi;
    int tu = 0, img = 0, mu = 0, n = 0, lane_px = 0, vo_r = OOB;
    bool nok = false;
    const __amdgpu_buffer_rsrc_t srd_r = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<char *>(reinterpret_cast<const char *>(p.res ? p.res : p.out)), 0, p.M * p.ldres * 4, 0x00020000);
    auto set_geometry = [&]() __attribute__((always_inline)) {
        tu = cur_mt * 32 + 8 * f;
        const int tuc = tu < T ? tu : 0;
        img = (int)epi_div((unsigned)tuc, (unsigned)TI, p.mg_ohw, p.sh_ohw);
        const int tru = tuc - img * TI;
        const int tyu = tru >> wts, txu = tru - (tyu << wts);
        mu = img * HW + 2 * tyu * W + 4 * txu;
        n = cur_nt * (32 * NW) + wn * 32 + li;
        nok = n < p.N && tu < T;
        lane_px = Wt >= 8 ? lh * 16 : Wt == 4 ? lh * 2 * W : lh * 4 * W;
        vo_r = nok ? (lane_px * p.ldres + n) * 4 : OOB;
    };
    auto pix_off = [&](int vv, int a, int b) __attribute__((always_inline)) {
        const int base = Wt == 2 ? (vv >> 1) * 2 * W + (vv & 1) * 4 : vv * 4;
        return mu + base + a * W + b;
    };
    float rs[2][4][4];
    const bool has_res = p.res != nullptr;
    auto load_res = [&](int idx) __attribute__((always_inline)) {  // idx = (a * 4 + b) * 4 + vv
        const int a = idx >> 4, b = (idx >> 2) & 3, vv = idx & 3;
        rs[a][b][vv] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(srd_r, vo_r, pix_off(vv, a, b) * p.ldres * 4, W42_OUT_AUX));
    };

    // ---- one 16-channel chunk: 48 MFMAs per wave between barriers, in 24 gaps of two.  Every chunk stages the chunk after it (registers
    // -> the other LDS stage), requests the pixels two chunks ahead and the weight fragments one chunk ahead -- of this tile or of the next.
    // FIRST: the tile's first chunk (accumulators start from the constant 0); LAST: its last one (the residual tile is requested) ----
    w4f32x4 af[2][2];
    auto chunk = [&](auto par_tag, auto first_tag, auto last_tag) __attribute__((always_inline)) {
        constexpr int PAR = decltype(par_tag)::value;
        constexpr bool FIRST = decltype(first_tag)::value != 0;
        constexpr bool LAST = decltype(last_tag)::value != 0;
        const float *cur = smem + PAR * W4_STAGE;
        constexpr int nxt = PAR ^ 1;
        af[0][0] = *reinterpret_cast<const w4f32x4 *>(cur + a_off);
        af[0][1] = *reinterpret_cast<const w4f32x4 *>(cur + a_off + 64);
#pragma unroll
        for (int jp = 0; jp < 3; ++jp) {
#pragma unroll
            for (int g = 0; g < 2; ++g) {
                const int grp = jp * 2 + g, buf = grp & 1;
#pragma unroll
                for (int jj = 0; jj < 4; ++jj) {
                    if (FIRST && g == 0 && jj == 0) {
                        const w4f32x16 zero = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
                        acc[2 * jp] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[buf][0][jj], bf[2 * jp][g][jj], zero, 0, 0, 0);
                        acc[2 * jp + 1] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[buf][1][jj], bf[2 * jp + 1][g][jj], zero, 0, 0, 0);
                    } else {
                        acc[2 * jp] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[buf][0][jj], bf[2 * jp][g][jj], acc[2 * jp], 0, 0, 0);
                        acc[2 * jp + 1] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[buf][1][jj], bf[2 * jp + 1][g][jj], acc[2 * jp + 1], 0, 0, 0);
                    }
                    __builtin_amdgcn_sched_barrier(0);
                    const int gap = grp * 4 + jj;
                    if (jj == 0 && grp < 5) {
                        const int nj = (grp + 1) >> 1, ng = (grp + 1) & 1;
                        af[buf ^ 1][0] = *reinterpret_cast<const w4f32x4 *>(cur + a_off + (2 * nj) * 64 + ng * 8);
                        af[buf ^ 1][1] = *reinterpret_cast<const w4f32x4 *>(cur + a_off + (2 * nj + 1) * 64 + ng * 8);
                    }
                    if (jj == 3) {  // the next chunk's fragments of this group, into the registers just consumed
                        load_u(2 * jp, g);
                        load_u(2 * jp + 1, g);
                    }
                    if (S == 2) {  // the second slot, six gaps ahead of the first: transformed in gaps 0-8, its next pixels requested in 6-11
                        if (gap == 0) combine_in(rr[S - 1]);
                        if (gap == 1) xform(rr[S - 1], 0, S - 1);
                        if (gap >= 2 && gap <= 4) { store_v(2 * (gap - 2), nxt, 0, S - 1); store_v(2 * (gap - 2) + 1, nxt, 0, S - 1); }
                        if (gap == 5) xform(rr[S - 1], 1, S - 1);
                        if (gap >= 6 && gap <= 8) { store_v(2 * (gap - 6), nxt, 1, S - 1); store_v(2 * (gap - 6) + 1, nxt, 1, S - 1); }
                        if (gap >= 6 && gap <= 11) load_raw(gap - 6, S - 1);
                    }
                    if (gap == 6) combine_in(rr[0]);
                    if (gap == 7) xform(rr[0], 0, 0);
                    if (gap >= 8 && gap <= 10) { store_v(2 * (gap - 8), nxt, 0, 0); store_v(2 * (gap - 8) + 1, nxt, 0, 0); }
                    if (gap == 11) xform(rr[0], 1, 0);
                    if (gap >= 12 && gap <= 14) { store_v(2 * (gap - 12), nxt, 1, 0); store_v(2 * (gap - 12) + 1, nxt, 1, 0); }
#ifdef W42_LATE  // timing experiment: the pixel requests W42_LATE gaps later (how much of the time hangs on their lead?)
                    if (gap >= 12 + W42_LATE && gap <= 17 + W42_LATE) load_raw(gap - 12 - W42_LATE, 0);
                    if (gap == 18 + W42_LATE) px_advance();
#else
                    if (gap >= 12 && gap <= 17) load_raw(gap - 12, 0);
                    if (gap == 18) px_advance();
#endif
                    if (LAST && has_res && gap < 16) { load_res(2 * gap); load_res(2 * gap + 1); }
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
        }
        u_advance();
        if (!(W42_ABL & 8)) __syncthreads();
    };
    using I0 = std::integral_constant<int, 0>;
    using I1 = std::integral_constant<int, 1>;

    if (W42_PRIO && wave >= 4) __builtin_amdgcn_s_setprio(1);
    steady = true;
    for (;;) {
#ifdef W42_TRACE
        trace();
#endif
        chunk(I0{}, I1{}, I0{});
#ifdef W42_TRACE
        trace();  // (after the tile's first chunk: six stamps per tile, eight with -DW42_TRACE=2)
#endif
        chunk(I1{}, I0{}, I0{});
        for (int kc = 2; kc < NK - 2; kc += 2) {
            chunk(I0{}, I0{}, I0{});
            chunk(I1{}, I0{}, I0{});
        }
        chunk(I0{}, I0{}, I0{});
        set_geometry();
        chunk(I1{}, I0{}, I1{});
#ifdef W42_TRACE
        trace();
#endif

        // ---- output transform along x (in-lane): z[b][v] = sum_j Ax^T[b][j] M_ij for tile 4 lh + (v & 3) + 8 (v >> 2); then along y across
        // the four component-row waves of an output block through LDS: wave i keeps the z of its own unit (v >> 2 == i) and writes the
        // three other units' to its 48-register slab ----
        float zo[4][4];  // [b][vv]: own unit
        {
            float *mine = smem + W4P_EXB + (wn * 4 + wi) * W4P_EXW + lane;
            auto emit = [&](auto f_tag) __attribute__((always_inline)) {
                constexpr int F = decltype(f_tag)::value;  // == wi
#pragma unroll
                for (int v = 0; v < 16; ++v) {
                    const float m0 = acc[0][v], m1 = acc[1][v], m2 = acc[2][v], m3 = acc[3][v], m4 = acc[4][v], m5 = acc[5][v];
                    const float s1 = m1 + m2, d1 = m1 - m2, s2 = m3 + m4, d2 = m3 - m4;
                    const float z0 = (m0 + s1) + s2, z1 = d1 + 2.f * d2, z2 = s1 + 4.f * s2, z3 = (d1 + 8.f * d2) + m5;
                    const int fu = v >> 2, vv = v & 3;
                    if (fu == F) {
                        zo[0][vv] = z0; zo[1][vv] = z1; zo[2][vv] = z2; zo[3][vv] = z3;
                    } else {
                        const int r = fu < F ? fu : fu - 1;  // rank of unit fu among the three foreign ones
                        mine[((0 * 3 + r) * 4 + vv) * 64] = z0;
                        mine[((1 * 3 + r) * 4 + vv) * 64] = z1;
                        mine[((2 * 3 + r) * 4 + vv) * 64] = z2;
                        mine[((3 * 3 + r) * 4 + vv) * 64] = z3;
                    }
                }
            };
            if (wi == 0) emit(std::integral_constant<int, 0>{});
            else if (wi == 1) emit(std::integral_constant<int, 1>{});
            else if (wi == 2) emit(std::integral_constant<int, 2>{});
            else emit(std::integral_constant<int, 3>{});
        }
        __syncthreads();
#ifdef W42_TRACE
        trace();
#endif
        float y[2][4][4];  // [a][b][vv]: output pixel (2 ty + a, 4 tx + b) of tile 4 lh + vv of the unit
        {
            // component row k of my unit: from wave k's slab at rank (f < k ? f : f - 1), or my own registers (k == f)
            const float *slab = smem + W4P_EXB + (wn * 4) * W4P_EXW + lane;
            auto gather = [&](auto f_tag) __attribute__((always_inline)) {
                constexpr int F = decltype(f_tag)::value;  // == f == wi
#pragma unroll
                for (int b = 0; b < 4; ++b)
#pragma unroll
                    for (int vv = 0; vv < 4; ++vv) {
                        float z[4];
#pragma unroll
                        for (int k = 0; k < 4; ++k) {
                            if (k == F) z[k] = zo[b][vv];
                            else {
                                const int r = F < k ? F : F - 1;
                                z[k] = slab[k * W4P_EXW + ((b * 3 + r) * 4 + vv) * 64];
                            }
                        }
                        y[0][b][vv] = z[0] + z[1] + z[2];
                        y[1][b][vv] = z[1] - z[2] - z[3];
                    }
            };
            if (wi == 0) gather(std::integral_constant<int, 0>{});
            else if (wi == 1) gather(std::integral_constant<int, 1>{});
            else if (wi == 2) gather(std::integral_constant<int, 2>{});
            else gather(std::integral_constant<int, 3>{});
        }
        __syncthreads();  // (the successor's first chunk writes stage 1, which the exchange area covers)
#ifdef W42_TRACE
        trace();
#endif

        // ---- epilogue of the unit ----
        if (tu < T) {
            const __amdgpu_buffer_rsrc_t srd_o =
                __builtin_amdgcn_make_buffer_rsrc(reinterpret_cast<char *>(p.out), 0, p.M * p.ldc * 4, 0x00020000);
            const int vo_o = nok ? (lane_px * p.ldc + n) * 4 : OOB;
#if defined(W42_TRACE) && W42_TRACE >= 2  // fine mode: what the epilogue waits for (everything in flight; then the bias)
            __builtin_amdgcn_s_waitcnt(0x0F70);
            trace();
#endif
            float addv = (p.bias && nok) ? p.bias[n] : 0.f;
            if (p.rowbias && nok) addv += p.rowbias[(size_t)img * p.ld_rowbias + n];
#if defined(W42_TRACE) && W42_TRACE >= 2
            asm volatile("" ::"v"(addv));
            __builtin_amdgcn_s_waitcnt(0x0F70);
            trace();
#endif
            float ps[2] = {0.f, 0.f}, pq[2] = {0.f, 0.f};
            float vals[2][4][4];
#pragma unroll
            for (int a = 0; a < 2; ++a)
#pragma unroll
                for (int b = 0; b < 4; ++b)
#pragma unroll
                    for (int vv = 0; vv < 4; ++vv) {
                        float add = addv;
                        if (has_res) add += rs[a][b][vv];
                        const float val = (p.alpha * y[a][b][vv] + add) * p.out_scale;
                        vals[a][b][vv] = val;
                        if (!p.egn_skip_raw)
                            __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, val), srd_o, vo_o, pix_off(vv, a, b) * p.ldc * 4, W42_OUT_AUX);
                        ps[a] += val;
                        pq[a] += val * val;
                    }
            if (p.egn_out) {
                // GroupNorm (+ swish) of the output (plan.h: Op::egn_peer), see wino42_tile.  rows_per_image == 256: every wave of the
                // workgroup is here (whole images: no unit is past the batch), the scratch is rewritten only after the next tile's barriers
                float gsum = ps[0] + ps[1], gsq = pq[0] + pq[1];
                gsum += __shfl_xor(gsum, 32); gsq += __shfl_xor(gsq, 32);
                float rows = 64.0f;
                if (p.rows_per_image == 256) {
                    float2 *red = reinterpret_cast<float2 *>(smem + W4P_RED);  // [wn][unit][32 columns]
                    if (lh == 0) red[(wn * 4 + f) * 32 + li] = make_float2(gsum, gsq);
                    __syncthreads();
                    gsum = 0.f; gsq = 0.f;
#pragma unroll
                    for (int u = 0; u < 4; ++u) {
                        const float2 v = red[(wn * 4 + u) * 32 + li];
                        gsum += v.x; gsq += v.y;
                    }
                    rows = 256.0f;
                }
                for (int o = 1; o < p.egn_cpg; o <<= 1) { gsum += __shfl_xor(gsum, o); gsq += __shfl_xor(gsq, o); }
                const float inv_n = 1.0f / (rows * (float)p.egn_cpg);
                const float mean = gsum * inv_n;
                const float rstd = rsqrtf(gsq * inv_n - mean * mean + 1e-6f);
                const float ga = nok ? p.egn_gamma[n] : 0.f, be = nok ? p.egn_beta[n] : 0.f;
                const float sc = rstd * ga, sh = be - mean * rstd * ga;
                const __amdgpu_buffer_rsrc_t srd_g =
                    __builtin_amdgcn_make_buffer_rsrc(reinterpret_cast<char *>(p.egn_out), 0, p.M * p.ldc * 4, 0x00020000);
#pragma unroll
                for (int a = 0; a < 2; ++a)
#pragma unroll
                    for (int b = 0; b < 4; ++b)
#pragma unroll
                        for (int vv = 0; vv < 4; ++vv) {
                            float t = vals[a][b][vv] * sc + sh;
                            if (p.egn_act) t = swish_e(t);
                            __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, t), srd_g, vo_o, pix_off(vv, a, b) * p.ldc * 4, W42_OUT_AUX);
                        }
            }
            if (p.qstats) {
                float s0, q0, s1, q1;
                if (Wt >= 8) {
                    s0 = ps[0] + __shfl_xor(ps[0], 32); q0 = pq[0] + __shfl_xor(pq[0], 32);
                    s1 = ps[1] + __shfl_xor(ps[1], 32); q1 = pq[1] + __shfl_xor(pq[1], 32);
                } else {
                    const float ts = ps[0] + ps[1], tq = pq[0] + pq[1];
                    const float os = __shfl_xor(ts, 32), oq = __shfl_xor(tq, 32);
                    s0 = lh == 0 ? ts : os; q0 = lh == 0 ? tq : oq;
                    s1 = lh == 0 ? os : ts; q1 = lh == 0 ? oq : tq;
                }
#pragma unroll
                for (int g = 0; g < 2; ++g) {
                    float s = g == 0 ? s0 : s1, q = g == 0 ? q0 : q1;
                    if (!nok) { s = 0.f; q = 0.f; }
                    s += __shfl_xor(s, 1); q += __shfl_xor(q, 1);
                    s += __shfl_xor(s, 2); q += __shfl_xor(q, 2);
                    if ((li & 3) == 0 && lh == 0 && nok) {
                        const int first = (mu - img * HW) + (Wt >= 8 ? g * W : g * 32);
                        float2 *o = reinterpret_cast<float2 *>(p.qstats) + ((size_t)img * p.qs_slices + (first >> 5)) * (p.N >> 2) + (n >> 2);
                        o[0] = make_float2(s, q);
                    }
                }
            }
        }
#ifdef W42_TRACE
        trace();
#endif
        if (nx_vb >= nvb) break;
        cur_vb = nx_vb; cur_mt = nx_mt; cur_nt = nx_nt;
        nx_vb = next_valid(cur_vb + (int)gridDim.x, nx_mt, nx_nt);
    }
}

__global__ __launch_bounds__(512, 2) void conv3x3_wino42_f32_kernel(const GemmParams p) {
    __shared__ __attribute__((aligned(16))) float smem[W4P_SMEM];
    wino42_persistent<2>(p, smem);
}

// ... 32 tiles x 32 outputs, four waves (one per SIMD, up to 512 registers each): the tile of launches the 64-output tile cannot fill the chip with
__global__ __launch_bounds__(256, 1) void conv3x3_wino42_n32_f32_kernel(const GemmParams p) {
    __shared__ __attribute__((aligned(16))) float smem[W4P_SMEM];
    wino42_persistent<1>(p, smem);
}

__global__ __launch_bounds__(512, 2) void conv3x3_wino42_k32_f32_kernel(const GemmParams p) {
    __shared__ __attribute__((aligned(16))) float smem[W4_SMEM];
    wino42_tile<true>(p, smem);
}

// U[n / 32][c / 8][i][j][(c >> 2) & 1][n % 32][c & 3] = (Gy g Gx^T)_ij from the packed direct weights w[n][(ky*3 + kx)*Cin + c]; sums in
// double, one rounding; row i = 3 negated (see the kernel); columns N .. 32 ceil(N / 32) - 1 are zeros
__global__ void wino42_weights_kernel(const float *__restrict__ w, float *__restrict__ u, int N, int Npad, int Cin) {
    const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= (long long)Npad * Cin) return;
    const int c = (int)(idx % Cin);
    const int n = (int)(idx / Cin);
    double g[3][3];
    for (int ky = 0; ky < 3; ++ky)
        for (int kx = 0; kx < 3; ++kx) g[ky][kx] = n < N ? (double)w[(long long)n * 9 * Cin + (ky * 3 + kx) * Cin + c] : 0.0;
    const double Gy[4][3] = {{1, 0, 0}, {0.5, 0.5, 0.5}, {0.5, -0.5, 0.5}, {0, 0, -1}};  // (row 3 negated)
    const double Gx[6][3] = {{0.25, 0, 0},
                             {-1.0 / 6, -1.0 / 6, -1.0 / 6},
                             {-1.0 / 6, 1.0 / 6, -1.0 / 6},
                             {1.0 / 24, 1.0 / 12, 1.0 / 6},
                             {1.0 / 24, -1.0 / 12, 1.0 / 6},
                             {0, 0, 1}};
    const int NK = Cin >> 3;
    float *ub = u + (((long long)(n >> 5) * NK + (c >> 3)) * 24) * 256 + ((c >> 2) & 1) * 128 + (n & 31) * 4 + (c & 3);
    for (int i = 0; i < 4; ++i)
        for (int j = 0; j < 6; ++j) {
            double acc = 0;
            for (int ky = 0; ky < 3; ++ky)
                for (int kx = 0; kx < 3; ++kx) acc += Gy[i][ky] * g[ky][kx] * Gx[j][kx];
            ub[(i * 6 + j) * 256] = (float)acc;
        }
}

// ---- host side ---------------------------------------------------------------------------------------------------------------
long long wino42_weight_floats(int N, int Cin) { return 24LL * (((long long)N + 31) / 32 * 32) * Cin; }

int launch_wino42_weights(const float *w, float *u, int N, int Cin, hipStream_t s) {
    GD_CHECK(w && u && N > 0 && Cin > 0 && Cin % 8 == 0, "wino42_weights: bad argument (N=%d Cin=%d)", N, Cin);
    const int Npad = (N + 31) / 32 * 32;
    const long long total = (long long)Npad * Cin;
    hipLaunchKernelGGL(wino42_weights_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, w, u, N, Npad, Cin);
    GD_HIP(hipGetLastError());
    return 0;
}

static void w4fill_div(unsigned d, unsigned &magic, unsigned &shift) {
    unsigned s = 0;
    while ((1ull << s) < d) ++s;
    shift = s;
    magic = (unsigned)((((1ull << 32) * ((1ull << s) - d)) / d) + 1);
}

// what the kernel can compute at all (p.b_wino is not looked at)
bool conv3x3_wino42_shape_ok(const GemmParams &p) {
    if (p.in_act || p.b_kmajor || p.a_presplit || p.gn_table || p.wk_split > 1) return false;
    // epilogue GroupNorm: images that are one unit (8x8) or one workgroup tile (16x16).  (A 32x32 image spans four M tiles; normalising it
    // in this kernel was built and measured in round 3 -- it costs the convolution what the stand-alone pass costs, profiles/r03_notes.md --
    // and removed in round 4.)
    if (p.egn_out && ((p.rows_per_image != 64 && p.rows_per_image != 256) || p.N % p.egn_cpg != 0 ||
                      (p.egn_cpg != 4 && p.egn_cpg != 8 && p.egn_cpg != 16 && p.egn_cpg != 32))) return false;
    if (p.KH != 3 || p.KW != 3 || p.stride != 1 || p.pad != 1 || p.OH != p.H || p.OW != p.W) return false;
    const int Cin = p.C0 + p.C1;
    if (Cin % 32 != 0 || (p.C1 > 0 && (p.C0 % 32 != 0 || !p.a1)) || p.K != 9 * Cin) return false;
    if ((p.W != 8 && p.W != 16 && p.W != 32 && p.W != 64) || p.H % 2 != 0 || (p.H * p.W / 8) % 8 != 0) return false;
    if (p.rows_per_image != p.H * p.W || p.M % (p.H * p.W) != 0 || p.N < 1) return false;
    if ((p.N % 4 != 0 || p.N < 32) && (p.qstats || p.egn_out)) return false;  // (a narrow or ragged N only without the column-quad statistics)
    if (p.qstats && p.qs_slices != p.rows_per_image / 32) return false;
    if (p.rowbias && p.ld_rowbias < 0) return false;  // (0: one row for every image)
    if ((long long)p.M * p.ldc * 4 + 64 >= (1LL << 31) || (p.res && (long long)p.M * p.ldres * 4 + 64 >= (1LL << 31))) return false;
    const long long ldmax = p.lda0 > p.lda1 ? p.lda0 : p.lda1;
    const long long a_bytes = ((long long)p.M + 3 * p.W + 8) * ldmax * 4 + Cin * 4 + 64;
    return a_bytes < (1LL << 31) && wino42_weight_floats(p.N, Cin) * 4 + 64 < (1LL << 31);
}

// The workgroup tile of a launch: 2 = 32 tiles x 64 outputs (eight waves), 1 = 32 tiles x 32 outputs (four waves), 0 = not this kernel.
// A persistent launch takes ceil(tiles / CUs) rounds; a 32-output tile costs W42_N32_COST of a 64-output one (measured:
// profiles/r04_notes.md -- half the products, all of the pixel staging), so the narrow tile wins exactly where the wide one leaves CUs
// idle: the 8x8 level below a batch of 256, the 16x16 level below 64, every level of a small batch.  GDDIM_WINO42_NW=1 / 2 forces a tile,
// GDDIM_WINO42_MINBLOCKS is the least number of wide tiles a Cin = 32 launch (the two-chunk form: no narrow alternative) must have; it is not
// consulted for any other shape.
constexpr double W42_N32_COST = 0.6;
static int wino42_nw(const GemmParams &p) {
    if (opts().no_wino42 || !conv3x3_wino42_shape_ok(p)) return 0;
    const long long mt = (p.M / 8 + 31) / 32, t64 = mt * ((p.N + 63) / 64), t32 = mt * ((p.N + 31) / 32);
    if (p.C0 + p.C1 == 32) return t64 >= opts().wino42_min_blocks ? 2 : 0;  // (the two-chunk form has the wide tile only)
    if (opts().wino42_nw == 1 || opts().wino42_nw == 2) return opts().wino42_nw;
    const long long cus = device_cus();
    const double c64 = (double)((t64 + cus - 1) / cus), c32 = W42_N32_COST * (double)((t32 + cus - 1) / cus);
    return c32 < c64 ? 1 : 2;
}

// what the engine sends to it: an eligible shape (GDDIM_NOWINO42=1 routes everything back to the F(2x2,3x3) kernel)
bool conv3x3_wino42_ok(const GemmParams &p) { return wino42_nw(p) != 0; }

const char *conv3x3_wino42_name() { return "conv3x3_wino42_f32_kernel"; }
// the kernel symbol launch_conv3x3_wino42 would pick for this problem (as rocprofv3 prints it)
const char *conv3x3_wino42_name_for(const GemmParams &p) {
    if (p.C0 + p.C1 == 32) return "conv3x3_wino42_k32_f32_kernel";
    return wino42_nw(p) == 1 ? "conv3x3_wino42_n32_f32_kernel" : "conv3x3_wino42_f32_kernel";
}

int launch_conv3x3_wino42(const GemmParams &p_in, hipStream_t s) {
    GemmParams p = p_in;
    GD_CHECK(conv3x3_wino42_shape_ok(p) && p.b_wino, "conv3x3_wino42: problem not eligible");
    if (p.C1 == 0) p.lda1 = 4;
    w4fill_div((unsigned)(p.H * p.W / 8), p.mg_ohw, p.sh_ohw);  // tiles per image
    p.sh_ow = p.W == 8 ? 1 : p.W == 16 ? 2 : p.W == 32 ? 3 : 4;  // log2(W / 4)
    const int nw = p.C0 + p.C1 == 32 ? 2 : (wino42_nw(p) == 1 ? 1 : 2);  // (a direct call with a shape the engine would not send: the wide tile)
    p.m_tiles = (p.M / 8 + 31) / 32;
    p.n_tiles = (p.N + 32 * nw - 1) / (32 * nw);
    p.flat_tiles = p.m_tiles < 16;
#ifdef W42_XCDMAP
    const int gx = (p.flat_tiles || p.C0 + p.C1 == 32) ? (p.flat_tiles ? p.m_tiles * p.n_tiles : ((p.m_tiles + 7) / 8) * 8 * p.n_tiles)
                                                        : ((p.m_tiles + 31) / 32) * 32 * p.n_tiles;
#else
    const int gx = p.flat_tiles ? p.m_tiles * p.n_tiles : ((p.m_tiles + 7) / 8) * 8 * p.n_tiles;
#endif
    if (p.C0 + p.C1 == 32) {
        hipLaunchKernelGGL(conv3x3_wino42_k32_f32_kernel, dim3(gx), dim3(512), 0, s, p);
    } else {
        // persistent: one workgroup per CU (a multiple of 8, so that a workgroup's tiles keep their XCD); GDDIM_WINO42_GRID overrides the
        // workgroup count (tests: a small grid makes every workgroup walk over several tiles)
        const int cus = device_cus() / 8 * 8 > 0 ? device_cus() / 8 * 8 : 8;
        int grid = gx < cus ? gx : cus;
        if (const int g = opts().wino42_grid) {
            if (g > 0 && g < grid) grid = g;
        }
        if (nw == 1) hipLaunchKernelGGL(conv3x3_wino42_n32_f32_kernel, dim3(grid), dim3(256), 0, s, p);
        else hipLaunchKernelGGL(conv3x3_wino42_f32_kernel, dim3(grid), dim3(512), 0, s, p);
    }
    GD_HIP(hipGetLastError());
    return 0;
}

}  // namespace gd
