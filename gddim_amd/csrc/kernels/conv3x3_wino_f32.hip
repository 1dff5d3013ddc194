// 3x3 / stride 1 / pad 1 convolution in fp32 with the one-dimensional Winograd transform F(2,3) along x.
//
// The fp32 matrix core is the roofline of this path (157 TFLOP/s; igemm_f32_fast_kernel sits at 86 % of it), so the only
// way past it is to issue fewer multiplications.  For an output pixel pair (x = 2p, 2p+1) of one row, the three taps
// g0 g1 g2 of a kernel row ky and the four inputs d0..d3 = in[y+ky-1][2p-1 .. 2p+2]:
//     M0 = (d0 - d2) g0,   M1 = (d1 + d2)(g0 + g1 + g2)/2,   M2 = (d2 - d1)(g0 - g1 + g2)/2,   M3 = (d1 - d3) g2
//     out[2p] = M0 + M1 + M2,   out[2p+1] = M1 - M2 - M3
// The sums over (ky, c) commute with the output transform, so each component t is a plain GEMM
//     M_t[pair][n] = sum_{ky,c} V_t[pair][ky,c] * U_t[n][ky,c]          (4 components x (M/2) x N x 3 Cin)
// = 12 M N Cin flops instead of 18 M N Cin: x 1.5 fewer matrix-core cycles.  Both transforms are additions only (the 1/2
// lives in the pre-transformed weights, which are computed in double and rounded once), so the result differs from the
// direct fp32 sum by a few more roundings, not by a conditioning problem (F(2,3) is the benign Winograd size).
//
// Kernel: 256 threads = 4 wave64 as 2 x 2; workgroup tile BP pixel pairs x BN output channels x 4 components, each wave
// (BP/2) x (BN/2) x 4 as v_mfma_f32_32x32x2_f32 accumulators (64 x 128: 8 accumulators = 128 VGPRs); two workgroups per CU
// (55 KB of LDS each) so one computes while the other runs its output transform and epilogue.  K is consumed in chunks of
// 8 channels of one kernel row: one ds_read_b128 per (fragment, component) feeds four MFMAs, 32 MFMAs per wave between
// barriers.  The input transform runs in the staging path (raw float4 loads of the four pixels of a pair -> 16 VALU -> four
// float4 LDS stores) on half of the waves, the other half stage more of the (pre-transformed) weights, so every wave issues
// six global loads and six LDS stores per chunk.  Addressing follows igemm_f32_fast_kernel: chunk state in SGPRs, the row
// part of an address precomputed per thread, out-of-image pixels pushed out of the buffer bounds (hardware zero fill).
// K order: 32-channel slab outermost, then kernel row, then the four 8-channel chunks of the slab, so the 128-byte lines of
// a pixel are touched by consecutive chunks.  Weights arrive as U[n][ky][Cin/8][component][8] (launch_wino_weights).
#include <cstdio>
#include <cstdlib>
#include <type_traits>

#include "../common.h"
#include "../options.h"
#include "act.h"
#include "epilogue.h"

namespace gd {

typedef float wf32x16 __attribute__((ext_vector_type(16)));
typedef float wf32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int wu32x4 __attribute__((ext_vector_type(4)));

constexpr int WKC = 8;    // channels per chunk
constexpr int WLD = 36;   // LDS row stride (floats): a row = 4 components x 8 channels + 4 pad -> 144 B, an odd number of 16-byte
                          // units, so the ds_read_b128 of 8 consecutive rows and the 8 stores of one weight row cover all banks

#ifndef WINO_B_AUX
#define WINO_B_AUX 0  // cache policy of the weight loads (A/B builds: 2 = nt)
#endif
#ifndef WINO_ABL
#define WINO_ABL 0  // timing ablations (results wrong): 1 no global loads in the K loop, 2 no LDS stores, 4 no fragment reads,
                    // 8 no barrier, 16 no transform arithmetic
#endif
#ifndef WINO_A_MAP
#define WINO_A_MAP 0  // A/B builds: 1 = activation lanes (pair = tid & 63, half = tid >> 6) instead of (tid >> 1, tid & 1)
#endif

template <int AUX = 0>
__device__ __forceinline__ wf32x4 wbuf_load16(__amdgpu_buffer_rsrc_t rsrc, int voffset, int soffset) {
    wu32x4 v = __builtin_amdgcn_raw_buffer_load_b128(rsrc, voffset, soffset, AUX);
    return __builtin_bit_cast(wf32x4, v);
}

template <int BP, int BN>
__global__ __launch_bounds__(BP * 4, BP == 128 ? 2 : BN >= 128 ? 2 : 3) void conv3x3_wino_f32_kernel(const GemmParams p) {
    constexpr int NT = BP * 4;                       // BP/32 x 2 waves: every wave owns 32 pairs x BN/2 outputs x 4 components
    constexpr int RM = 1, RN = BN / 64;              // 32x32 accumulators per wave and component
    constexpr int STAGE = (BP + BN) * WLD;           // floats per LDS stage: V[BP][4][8] | U[BN][4][8] (rows padded to 36)
    constexpr int NA = BP * 2;                       // threads that stage activations: (pair, 4-channel half of the chunk)
    constexpr int NBP = BN * 8;                      // 16-byte weight pieces per chunk: (n, component, half)
    constexpr int BA = NA == NT ? NBP / NT : (NBP / NA - 4) / 2;  // weight pieces of an activation thread ...
    constexpr int BO = NA == NT ? 0 : BA + 4;                       // ... and of the other threads
    constexpr int NG = 4 + BA > BO ? 4 + BA : BO;    // MFMA gaps that carry staging work
    constexpr int OOB = (int)0x80000000u;
    static_assert(NA % 64 == 0 && NA <= NT && NA * BA + (NT - NA) * BO == NBP && NG <= 16 && RM >= 1 && RN >= 1, "bad tile geometry");
    __shared__ __attribute__((aligned(16))) float smem[2 * STAGE];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1;
    const bool is_a = wave < NA / 64;  // wave-uniform role

    // PERSIST (the 128 x 128 tile, one workgroup per CU): the workgroup walks tile ids blockIdx.x, blockIdx.x + gridDim.x, ...;
    // gridDim.x is a multiple of 8 * n_tiles, so the N tile (and with it the weight addresses) and the XCD of a workgroup never
    // change.  The software pipeline of the K loop runs ACROSS tiles: the last three iterations of a tile already request chunks
    // 0..2 of the next one, so a tile has no prologue of its own and its epilogue runs with the next tile's first loads in flight.
    constexpr bool PERSIST = BP == 128;
    const int nt = p.n_tiles;
    const int total_ids = p.flat_tiles ? p.m_tiles * nt : ((p.m_tiles + 7) / 8) * 8 * nt;
    auto tile_of = [&](int tid_, int &mtile_, int &ntile_) __attribute__((always_inline)) {
        if (p.flat_tiles) {
            mtile_ = tid_ / nt;
            ntile_ = tid_ - mtile_ * nt;
        } else {
            const int grp = tid_ / (8 * nt);
            const int rem = tid_ - grp * 8 * nt;
            ntile_ = rem >> 3;
            mtile_ = grp * 8 + (rem & 7);
        }
        return mtile_ < p.m_tiles;
    };
    int id = blockIdx.x;
    int mtile, ntile;
    if (!tile_of(id, mtile, ntile)) return;
    int pair0 = mtile * BP;  // first pixel pair of the tile whose loads are being issued
    const int n0 = ntile * BN;

    const int W = p.W, H = p.H, HW = H * W;
    const int Cin = p.C0 + p.C1;
    const int a_imgs = (p.M + HW - 1) / HW;
    const int bias0 = (W + 1) * p.lda0 * 4, bias1 = (W + 1) * p.lda1 * 4;
    const char *a0c = reinterpret_cast<const char *>(p.a0);
    const __amdgpu_buffer_rsrc_t srd_a0 =
        __builtin_amdgcn_make_buffer_rsrc(const_cast<char *>(a0c - bias0), 0, a_imgs * HW * p.lda0 * 4 + bias0, 0x00020000);
    const char *a1c = reinterpret_cast<const char *>(p.C1 ? p.a1 : p.a0);
    const __amdgpu_buffer_rsrc_t srd_a1 =
        __builtin_amdgcn_make_buffer_rsrc(const_cast<char *>(a1c - bias1), 0, a_imgs * HW * p.lda1 * 4 + bias1, 0x00020000);
    const __amdgpu_buffer_rsrc_t srd_b = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<char *>(reinterpret_cast<const char *>(p.b_wino)), 0, p.N * p.ldb_wino * 4, 0x00020000);

    // ---- loop-invariant per-thread staging data.  Every thread owns NS = 6 staging slots (one float4 register each, one MFMA gap
    // each).  Activation thread: slots 0..3 are the raw pixels d0..d3 of its pair, slots 4.. weight pieces; weight thread: all
    // slots are weight pieces.  sv[] / sl[] hold the global voffset / LDS offset of a weight slot; an activation thread keeps
    // (rb0, rb1, inv) in sv[0..2] and its LDS offset in sl[0]. ----
    constexpr int NS = NG;
    int sv[NS], sl[NS];
#pragma unroll
    for (int i = 0; i < NS; ++i) {
        // weight piece of slot i: activation threads use slots 4..4+BA-1, weight threads slots 0..BO-1
        const bool used = is_a ? (i >= 4 && i < 4 + BA) : i < BO;
        const int q = is_a ? tid + NA * (i - 4) : NA * BA + (tid - NA) + (NT - NA) * i;
        const int nl = (q >> 3) & (BN - 1), j = q & 7;
        const int n = n0 + nl;
        sv[i] = (used && n < p.N) ? (n * p.ldb_wino + j * 4) * 4 : OOB;
        sl[i] = (BP + nl) * WLD + j * 4;
    }
    const int a_pl = WINO_A_MAP ? (tid & (BP - 1)) : tid >> 1, a_cq = WINO_A_MAP ? tid / BP : tid & 1;
    if (is_a) sl[0] = a_pl * WLD + a_cq * 4;
    auto setup_a = [&]() __attribute__((always_inline)) {  // activation thread: addresses of its pair in the tile starting at pair0
        const int pair = pair0 + a_pl;
        sv[0] = 0; sv[1] = 0; sv[2] = -1;  // rb0, rb1, inv (all taps invalid)
        if (2 * pair < p.M) {
            const int m = 2 * pair;
            const int img = (int)epi_div((unsigned)m, (unsigned)HW, p.mg_ohw, p.sh_ohw);
            const int r = m - img * HW;
            const int y = (int)epi_div((unsigned)r, (unsigned)W, p.mg_ow, p.sh_ow);
            const int x0 = r - y * W;
            const int pix = (img * H + y - 1) * W + x0 - 1;
            sv[0] = pix * p.lda0 * 4 + bias0 + a_cq * 16;
            sv[1] = pix * p.lda1 * 4 + bias1 + a_cq * 16;
            unsigned xb = 0;
#pragma unroll
            for (int j = 0; j < 4; ++j) xb |= ((unsigned)(x0 - 1 + j) < (unsigned)W ? 0u : 1u) << j;
            unsigned mk = 0;
#pragma unroll
            for (int ky = 0; ky < 3; ++ky) mk |= ((unsigned)(y + ky - 1) < (unsigned)H ? xb : 15u) << (4 * ky);
            sv[2] = (int)mk;
        }
    };
    if (is_a) setup_a();

    // ---- wave-uniform state of the chunk whose loads are issued next ----
    int s_kc = 0, s_sub = 0, s_ky = 0, s_c = 0;
    const int nk = 3 * (Cin / WKC);
    const int c8n = Cin / WKC;
    // Activation thread: voffsets of its four pixels for the current (source, kernel row), out-of-image pixels already pushed out
    // of bounds.  They change every fourth chunk only, so the K loop spends 3 VALU per chunk on addresses instead of 16.
    int vo[4] = {OOB, OOB, OOB, OOB};
    auto refresh_vo = [&]() __attribute__((always_inline)) {
        const int rb = s_c < p.C0 ? sv[0] : sv[1];
#pragma unroll
        for (int j = 0; j < 4; ++j) vo[j] = rb | (__builtin_amdgcn_sbfe(sv[2], s_ky * 4 + j, 1) & OOB);
    };
    auto advance = [&](auto role_a) __attribute__((always_inline)) {
        ++s_kc;
        ++s_sub;
        s_c += WKC;
        if (s_sub == 4) {
            s_sub = 0;
            ++s_ky;
            s_c -= 32;
            if (s_ky == 3) { s_ky = 0; s_c += 32; }
            if constexpr (decltype(role_a)::value) refresh_vo();
        }
    };

    // Two sets of staging registers: the loads issued in iteration kc fetch chunk kc+3 and are written to LDS in iteration kc+2,
    // so a global load has two chunks (4-8 thousand cycles) to return instead of one.  Set (kc+1)&1 is the one iteration kc
    // stores and re-requests; the set index is a compile-time constant (the K loop is unrolled by two).
    wf32x4 rs2[2][NS];
    auto load_a = [&](auto set, int j) __attribute__((always_inline)) {  // raw pixel j of the pair
        constexpr int S = decltype(set)::value;
        if (s_c < p.C0) {
            const int soff = ((s_ky * W + j) * p.lda0 + s_c) * 4;
            rs2[S][j] = wbuf_load16(srd_a0, vo[j], soff);
        } else {
            const int soff = ((s_ky * W + j) * p.lda1 + (s_c - p.C0)) * 4;
            rs2[S][j] = wbuf_load16(srd_a1, vo[j], soff);
        }
    };
    auto load_b = [&](auto set, int i) __attribute__((always_inline)) {
        constexpr int S = decltype(set)::value;
        const int soff = ((s_ky * c8n + (s_c >> 3)) * 32) * 4;
        rs2[S][i] = wbuf_load16<WINO_B_AUX>(srd_b, sv[i], soff);
    };
    auto store_x = [&](wf32x4 v, float *dst) __attribute__((always_inline)) {
        if (!(WINO_ABL & 2)) *reinterpret_cast<wf32x4 *>(dst) = v;
        else asm volatile("" ::"v"(v));
    };
    auto store_b = [&](auto set, int i, float *st) __attribute__((always_inline)) { store_x(rs2[decltype(set)::value][i], &st[sl[i]]); };
    auto store_v = [&](int t, wf32x4 v, float *st) __attribute__((always_inline)) { store_x(v, &st[t * 8 + sl[0]]); };
    // Staging work of MFMA gap g: write slot g's piece of chunk kc+1 to the other LDS stage, then request slot g for chunk
    // kc+3.  The input transform is arranged so that an activation thread frees its raw pixels in slot order:
    //   gap 0: V0 = d0 - d2 stored;                       gap 1: V1 = d1 + d2 stored, d2 <- d2 - d1 (= V2), d3 <- d1 - d3 (= V3);
    //   gap 2: V2 stored;                                 gap 3: V3 stored.
    // ROLE / LOADS are compile-time tags (std::integral_constant): the two roles run two separate copies of the K loop (no
    // control-flow joins between them inside the loop -- joins made the compiler copy the staging registers and wait for
    // vmcnt(0) in every gap), and the last three iterations, which have nothing left to request, are peeled without loads.
    auto stage_gap = [&](int g, float *st, auto set, auto role_a, auto loads) __attribute__((always_inline)) {
        constexpr int S = decltype(set)::value;
        wf32x4 (&rs)[NS] = rs2[S];
        if constexpr (decltype(role_a)::value) {
            if (g == 0) store_v(0, (WINO_ABL & 16) ? rs[0] : rs[0] - rs[2], st);
            else if (g == 1) {
                store_v(1, (WINO_ABL & 16) ? rs[1] : rs[1] + rs[2], st);
                if (!(WINO_ABL & 16)) {
                    rs[2] = rs[2] - rs[1];
                    rs[3] = rs[1] - rs[3];
                }
            } else if (g == 2) store_v(2, rs[2], st);
            else if (g == 3) store_v(3, rs[3], st);
            else if (g < 4 + BA) store_b(set, g, st);
            if constexpr (decltype(loads)::value && !(WINO_ABL & 1)) {
                if (g < 4) load_a(set, g);
                else if (g < 4 + BA) load_b(set, g);
            }
        } else {
            if (g < BO) {
                store_b(set, g, st);
                if constexpr (decltype(loads)::value && !(WINO_ABL & 1)) load_b(set, g);
            }
        }
    };
    auto load_all = [&](auto set, auto role_a) __attribute__((always_inline)) {
#pragma unroll
        for (int g = 0; g < NS; ++g) {
            if constexpr (decltype(role_a)::value) {
                if (g < 4) load_a(set, g);
                else if (g < 4 + BA) load_b(set, g);
            } else if (g < BO) load_b(set, g);
        }
    };

    wf32x16 acc[4][RM][RN];
    auto zero_acc = [&]() __attribute__((always_inline)) {
#pragma unroll
        for (int t = 0; t < 4; ++t)
#pragma unroll
            for (int r = 0; r < RM; ++r)
#pragma unroll
                for (int c = 0; c < RN; ++c)
#pragma unroll
                    for (int v = 0; v < 16; ++v) acc[t][r][c][v] = 0.f;
    };

    const int li = lane & 31, lh = lane >> 5;
    const int a_off = (wm * 32 + li) * WLD + lh * 4;
    const int b_off = (BP + wn * (BN / 2) + li) * WLD + lh * 4;

    wf32x4 af[2][RM], bf[2][RN];  // fragments of the component being multiplied and of the next one
    auto load_frags = [&](int t, const float *st, int set) __attribute__((always_inline)) {
#pragma unroll
        for (int r = 0; r < RM; ++r) af[set][r] = *reinterpret_cast<const wf32x4 *>(st + t * 8 + a_off + r * 32 * WLD);
#pragma unroll
        for (int c = 0; c < RN; ++c) bf[set][c] = *reinterpret_cast<const wf32x4 *>(st + t * 8 + b_off + c * 32 * WLD);
    };
    auto chunk = [&](int kc, auto set, auto role_a, auto loads) __attribute__((always_inline)) {
        const float *cur = smem + (kc & 1) * STAGE;
        float *nxt = smem + ((kc + 1) & 1) * STAGE;
        if (!(WINO_ABL & 4) || kc == 0) load_frags(0, cur, 0);
#pragma unroll
        for (int t = 0; t < 4; ++t) {
#pragma unroll
            for (int jj = 0; jj < 4; ++jj) {
#pragma unroll
                for (int r = 0; r < RM; ++r)
#pragma unroll
                    for (int c = 0; c < RN; ++c)
                        acc[t][r][c] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[t & 1][r][jj], bf[t & 1][c][jj], acc[t][r][c], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
                const int gap = t * 4 + jj;
                if (jj == 0 && t < 3 && !(WINO_ABL & 4)) load_frags(t + 1, cur, (t + 1) & 1);
                if (gap < NG) stage_gap(gap, nxt, set, role_a, loads);
                else if (gap == NG) advance(role_a);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        if (NG == 16) advance(role_a);
        if (!(WINO_ABL & 8)) __syncthreads();
    };
    // ---- output transform + epilogue.  Accumulator row (v&3) + 8 (v>>2) + 4 lh of tile r is pixel pair
    // prow + that; its two pixels are rows 2*pair and 2*pair + 1 of the output matrix.  Column = lane & 31. ----
    auto epilogue = [&](int pair0_done) __attribute__((always_inline)) {
        char *outc = reinterpret_cast<char *>(p.out);
        const __amdgpu_buffer_rsrc_t srd_o = __builtin_amdgcn_make_buffer_rsrc(outc, 0, p.M * p.ldc * 4, 0x00020000);
        const __amdgpu_buffer_rsrc_t srd_r =
            __builtin_amdgcn_make_buffer_rsrc(const_cast<char *>(reinterpret_cast<const char *>(p.res ? p.res : p.out)), 0,
                                              p.M * p.ldres * 4, 0x00020000);
        const unsigned rpi = p.rows_per_image;
#pragma unroll
        for (int c = 0; c < RN; ++c) {
            const int n = n0 + wn * (BN / 2) + c * 32 + li;
            const bool nok = n < p.N;
            const float bias = (p.bias && nok) ? p.bias[n] : 0.f;
            const int vo_o = nok ? (8 * lh * p.ldc + n) * 4 : OOB;
            const int vo_r = nok ? (8 * lh * p.ldres + n) * 4 : OOB;
#pragma unroll
            for (int r = 0; r < RM; ++r) {
                const int prow = pair0_done + wm * 32 + r * 32;  // wave-uniform
                const int mrow = 2 * prow;
                float res_e[16], res_o[16];
                if (p.res && !(WINO_ABL & 64)) {
#pragma unroll
                    for (int v = 0; v < 16; ++v) {
                        const int so = (mrow + 2 * ((v & 3) + 8 * (v >> 2))) * p.ldres * 4;
                        res_e[v] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(srd_r, vo_r, so, 0));
                        res_o[v] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(srd_r, vo_r, so + p.ldres * 4, 0));
                    }
                }
                float addv[4];
#pragma unroll
                for (int g = 0; g < 4; ++g) addv[g] = bias;
                if (p.rowbias) {
                    // rows_per_image is a multiple of 16: the 8 pairs (v & 3) + 4 lh of a group v >> 2 belong to one image
#pragma unroll
                    for (int g = 0; g < 4; ++g) {
                        const unsigned mr = (unsigned)(mrow + 16 * g);
                        const unsigned img = mr < (unsigned)p.M ? epi_div(mr, rpi, p.mg_rpi, p.sh_rpi) : 0u;
                        addv[g] += nok ? p.rowbias[(size_t)img * p.ld_rowbias + n] : 0.f;
                    }
                }
                float s0 = 0.f, q0 = 0.f, s1 = 0.f, q1 = 0.f;
                float yev[16], yov[16];  // kept for the epilogue GroupNorm (dead otherwise)
#pragma unroll
                for (int v = 0; v < 16; ++v) {
                    const float m0 = acc[0][r][c][v], m1 = acc[1][r][c][v], m2 = acc[2][r][c][v], m3 = acc[3][r][c][v];
                    float ye = m0 + m1 + m2, yo = m1 - m2 - m3;
                    float ae = addv[v >> 2], ao = addv[v >> 2];
                    if (p.res && !(WINO_ABL & 64)) { ae += res_e[v]; ao += res_o[v]; }
                    ye = (p.alpha * ye + ae) * p.out_scale;
                    yo = (p.alpha * yo + ao) * p.out_scale;
                    if (BP == 64) { yev[v] = ye; yov[v] = yo; }
                    const int so = (mrow + 2 * ((v & 3) + 8 * (v >> 2))) * p.ldc * 4;
                    if (WINO_ABL & 32) { asm volatile("" ::"v"(ye), "v"(yo)); }
                    else if (!(BP == 64 && p.egn_skip_raw)) {
                        __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, ye), srd_o, vo_o, so, 0);
                        __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, yo), srd_o, vo_o, so + p.ldc * 4, 0);
                    }
                    if (v < 8) { s0 += ye + yo; q0 += ye * ye + yo * yo; }
                    else { s1 += ye + yo; q1 += ye * ye + yo * yo; }
                }
                if (BP == 64 && p.egn_out) {  // (the 128-pair tile has no registers to spare and never sees 8x8 images at sane batches)
                    // GroupNorm (+ swish) of the output, here: rows_per_image == 64, so the 32 pairs of this accumulator tile (both
                    // half-waves) are exactly one image; a group is egn_cpg consecutive columns = consecutive lanes
                    float gs = s0 + s1, gq = q0 + q1;
                    gs += __shfl_xor(gs, 32); gq += __shfl_xor(gq, 32);
                    for (int o = 1; o < p.egn_cpg; o <<= 1) { gs += __shfl_xor(gs, o); gq += __shfl_xor(gq, o); }
                    const float inv_n = 1.0f / (64.0f * (float)p.egn_cpg);
                    const float mean = gs * inv_n;
                    const float rstd = rsqrtf(gq * inv_n - mean * mean + 1e-6f);
                    const float ga = nok ? p.egn_gamma[n] : 0.f, be = nok ? p.egn_beta[n] : 0.f;
                    const float sc = rstd * ga, sh = be - mean * rstd * ga;
                    const __amdgpu_buffer_rsrc_t srd_g =
                        __builtin_amdgcn_make_buffer_rsrc(reinterpret_cast<char *>(p.egn_out), 0, p.M * p.ldc * 4, 0x00020000);
#pragma unroll
                    for (int v = 0; v < 16; ++v) {
                        float ne = yev[v] * sc + sh, no = yov[v] * sc + sh;
                        if (p.egn_act) { ne = swish_e(ne); no = swish_e(no); }
                        const int so = (mrow + 2 * ((v & 3) + 8 * (v >> 2))) * p.ldc * 4;
                        __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, ne), srd_g, vo_o, so, 0);
                        __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, no), srd_g, vo_o, so + p.ldc * 4, 0);
                    }
                }
                if (p.qstats) {
                    // pairs 0..15 of the tile (v < 8) are output rows mrow .. mrow+31 = one 32-row slice, pairs 16..31 the next one
                    // (rows_per_image % 64 == 0 and M % 64 == 0: a tile is inside one image and either whole or absent)
                    if (!nok) { s0 = q0 = s1 = q1 = 0.f; }
                    s0 += __shfl_xor(s0, 1); q0 += __shfl_xor(q0, 1); s1 += __shfl_xor(s1, 1); q1 += __shfl_xor(q1, 1);
                    s0 += __shfl_xor(s0, 2); q0 += __shfl_xor(q0, 2); s1 += __shfl_xor(s1, 2); q1 += __shfl_xor(q1, 2);
                    s0 += __shfl_xor(s0, 32); q0 += __shfl_xor(q0, 32); s1 += __shfl_xor(s1, 32); q1 += __shfl_xor(q1, 32);
                    if ((li & 3) == 0 && lh == 0 && nok && mrow < p.M) {
                        const int Q = p.N >> 2;
                        const int img = mrow / (int)rpi;
                        const int sl = (mrow - img * (int)rpi) >> 5;
                        float2 *o = reinterpret_cast<float2 *>(p.qstats) + ((size_t)img * p.qs_slices + sl) * Q + (n >> 2);
                        o[0] = make_float2(s0, q0);
                        o[Q] = make_float2(s1, q1);
                    }
                }
            }
        }
    };

    // Prologue of the first tile and the tile loop of one role.  At the top of iteration kc: LDS stage kc&1 holds chunk kc,
    // register set (kc+1)&1 chunk kc+1 (raw), set kc&1 chunk kc+2 (possibly still in flight).  nk is a multiple of 12.
    auto run = [&](auto role_a) __attribute__((always_inline)) {
        constexpr std::integral_constant<int, 0> S0{};
        constexpr std::integral_constant<int, 1> S1{};
        if constexpr (decltype(role_a)::value) refresh_vo();
#ifdef WINO_TRACE
        const unsigned long long tr0 = wall_clock64();
        int trn = 0;
        float *trbuf = const_cast<float *>(p.res) + (size_t)p.M * p.ldres - 256 + (blockIdx.x == 0 ? 0 : 128);
        const bool trw = (blockIdx.x == 0 || blockIdx.x == 100) && tid == 0 && p.res;
        const unsigned long long cy0 = __builtin_readcyclecounter();
        auto trace = [&]() __attribute__((always_inline)) {
            if (trw && trn < 60) { trbuf[2 * trn] = (float)(wall_clock64() - tr0); trbuf[2 * trn + 1] = (float)(__builtin_readcyclecounter() - cy0); }
            ++trn;
        };
#else
        auto trace = [&]() __attribute__((always_inline)) {};
#endif
        load_all(S0, role_a);  // chunk 0
        advance(role_a);
        load_all(S1, role_a);  // chunk 1
        advance(role_a);
        if constexpr (decltype(role_a)::value) {
            wf32x4 (&rs)[NS] = rs2[0];
            store_v(0, rs[0] - rs[2], smem);
            store_v(1, rs[1] + rs[2], smem);
            store_v(2, rs[2] - rs[1], smem);
            store_v(3, rs[1] - rs[3], smem);
#pragma unroll
            for (int i = 4; i < 4 + BA; ++i) store_b(S0, i, smem);
        } else {
#pragma unroll
            for (int i = 0; i < BO; ++i) store_b(S0, i, smem);
        }
        load_all(S0, role_a);  // chunk 2
        advance(role_a);
        __syncthreads();
        for (;;) {
            trace();  // tile start
            zero_acc();
            int kc = 0;
            for (; kc < nk - 4; kc += 2) {  // iterations 0 .. nk-5 request chunks 3 .. nk-2
                chunk(kc, S1, role_a, std::true_type{});
                chunk(kc + 1, S0, role_a, std::true_type{});
#ifdef WINO_TRACE
                trace();
#endif
            }
            chunk(nk - 4, S1, role_a, std::true_type{});  // requests chunk nk-1; the state now points past the tile
            trace();  // main loop done
            if constexpr (!PERSIST) {
                chunk(nk - 3, S0, role_a, std::false_type{});
                chunk(nk - 2, S1, role_a, std::false_type{});
                chunk(nk - 1, S0, role_a, std::false_type{});  // (what it stores is never read)
                epilogue(pair0);
                break;
            } else {
                // roll over to the next tile of this workgroup: iterations nk-3 .. nk-1 request ITS chunks 0 .. 2 (and the last one
                // writes its chunk 0 to stage 0), so the loop simply continues after the epilogue.  Without a next tile the
                // addresses are pushed out of bounds: the loads fetch nothing and what is stored is never read.
                const int pair0_done = pair0;
                int next = id + (int)gridDim.x, mt = 0, ntl = 0;
                while (next < total_ids && !tile_of(next, mt, ntl)) next += (int)gridDim.x;
                const bool more = next < total_ids;
                s_kc = 0; s_sub = 0; s_ky = 0; s_c = 0;
                if (more) {
                    id = next;
                    pair0 = mt * BP;
                    if constexpr (decltype(role_a)::value) setup_a();
                } else {
#pragma unroll
                    for (int i = 0; i < NS; ++i) sv[i] = OOB;
                    if constexpr (decltype(role_a)::value) { sv[0] = 0; sv[1] = 0; sv[2] = -1; }
                }
                if constexpr (decltype(role_a)::value) refresh_vo();
                chunk(nk - 3, S0, role_a, std::true_type{});
                chunk(nk - 2, S1, role_a, std::true_type{});
                chunk(nk - 1, S0, role_a, std::true_type{});
                trace();  // K loop done
                epilogue(pair0_done);
                trace();  // epilogue issued
                if (!more) break;
            }
        }
    };
    if (is_a) run(std::true_type{});
    else run(std::false_type{});

}

// U[n][ky][c/8][t][c%8] from the packed direct weights w[n][(ky*3 + kx)*Cin + c]; sums in double, one rounding
__global__ void wino_weights_kernel(const float *__restrict__ w, float *__restrict__ u, int N, int Cin) {
    const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    const long long total = (long long)N * 3 * Cin;
    if (idx >= total) return;
    const int c = (int)(idx % Cin);
    const int ky = (int)((idx / Cin) % 3);
    const long long n = idx / (3LL * Cin);
    const float *wr = w + n * 9 * Cin + (long long)ky * 3 * Cin + c;
    const double g0 = wr[0], g1 = wr[Cin], g2 = wr[2 * Cin];
    float *ur = u + n * 12 * Cin + ((long long)(ky * (Cin / 8) + c / 8) * 4) * 8 + (c & 7);
    ur[0] = (float)g0;
    ur[8] = (float)(0.5 * (g0 + g1 + g2));
    ur[16] = (float)(0.5 * (g0 - g1 + g2));
    ur[24] = (float)g2;
}

// ---- host side ---------------------------------------------------------------------------------------------------------------
int launch_wino_weights(const float *w, float *u, int N, int Cin, hipStream_t s) {
    GD_CHECK(w && u && N > 0 && Cin > 0 && Cin % 8 == 0, "wino_weights: bad argument (N=%d Cin=%d)", N, Cin);
    const long long total = (long long)N * 3 * Cin;
    hipLaunchKernelGGL(wino_weights_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, w, u, N, Cin);
    GD_HIP(hipGetLastError());
    return 0;
}

static void wfill_div(unsigned d, unsigned &magic, unsigned &shift) {
    unsigned s = 0;
    while ((1ull << s) < d) ++s;
    shift = s;
    magic = (unsigned)((((1ull << 32) * ((1ull << s) - d)) / d) + 1);
}

// Tile choice: 64 pairs x 128 outputs when that gives every CU two workgroups, 64 x 64 (half the accumulators, up to four
// workgroups per CU) for the 8x8 level; 0 = neither fills the chip (the direct kernel's smaller tiles take the problem).
static int wino_bn(const GemmParams &p) {
    const int min_blocks = opts().wino_min_blocks, no_small = opts().wino_nosmall;
    const long long mt = (p.M / 2 + 63) / 64;
    if (mt * ((p.N + 127) / 128) >= min_blocks) return 128;
    if (!no_small && mt * ((p.N + 63) / 64) >= min_blocks) return 64;
    return 0;
}

// what the kernel can compute at all (p.b_wino is not looked at)
bool conv3x3_wino_shape_ok(const GemmParams &p) {
    if (p.in_act || p.b_kmajor || p.a_presplit || p.gn_table) return false;
    if (p.KH != 3 || p.KW != 3 || p.stride != 1 || p.pad != 1 || p.OH != p.H || p.OW != p.W) return false;
    const int Cin = p.C0 + p.C1;
    if (Cin % 32 != 0 || (p.C1 > 0 && (p.C0 % 32 != 0 || !p.a1)) || p.K != 9 * Cin) return false;
    if (p.W % 2 != 0 || (p.H * p.W) % 64 != 0 || p.M % 64 != 0 || p.rows_per_image != p.H * p.W || p.N % 4 != 0 || p.N < 64) return false;
    if (p.qstats && p.qs_slices != p.rows_per_image / 32) return false;
    if (p.egn_out && (p.rows_per_image != 64 || p.N % p.egn_cpg != 0 ||
                      (p.egn_cpg != 4 && p.egn_cpg != 8 && p.egn_cpg != 16 && p.egn_cpg != 32))) return false;
    if ((long long)p.M * p.ldc * 4 + 64 >= (1LL << 31) || (p.res && (long long)p.M * p.ldres * 4 + 64 >= (1LL << 31))) return false;
    const long long ldmax = p.lda0 > p.lda1 ? p.lda0 : p.lda1;
    const long long a_bytes = ((long long)p.M + 2 * p.W + 8) * ldmax * 4 + Cin * 4 + 64;
    return a_bytes < (1LL << 31) && (long long)p.N * 12 * Cin * 4 + 64 < (1LL << 31);
}

// what the engine sends to it: an eligible shape that fills the chip; GDDIM_NOWINO=1 routes everything back to the direct kernel
bool conv3x3_wino_ok(const GemmParams &p) {
    return !opts().no_wino && conv3x3_wino_shape_ok(p) && wino_bn(p) != 0;
}

// 128 pairs x 128 outputs, eight waves, ONE workgroup per CU: a third less staging traffic per MAC (the weight chunk is shared by
// twice as many pairs); taken when it still gives every CU two tiles.  GDDIM_WINO_NOBIG=1 (A/B) keeps the 64-pair tiles.
static bool wino_big(const GemmParams &p) {
    return !opts().wino_nobig && wino_bn(p) == 128 && (long long)((p.M / 2 + 127) / 128) * ((p.N + 127) / 128) >= 512;
}

// the engine would send this problem to a 64-pair Winograd tile and that tile can normalise its output (p.egn_* set)
bool conv3x3_wino_egn_ok(const GemmParams &p) {
    return p.egn_out != nullptr && conv3x3_wino_ok(p) && !wino_big(p);
}

const char *conv3x3_wino_name(const GemmParams &p) {
    if (wino_big(p)) return "conv3x3_wino_f32_kernel<128,128>";
    return wino_bn(p) == 64 ? "conv3x3_wino_f32_kernel<64,64>" : "conv3x3_wino_f32_kernel<64,128>";
}

int launch_conv3x3_wino(const GemmParams &p_in, hipStream_t s) {
    GemmParams p = p_in;
    GD_CHECK(conv3x3_wino_shape_ok(p) && p.b_wino, "conv3x3_wino: problem not eligible");
    p.ldb_wino = 12 * (p.C0 + p.C1);
    if (p.C1 == 0) p.lda1 = 4;
    wfill_div((unsigned)(p.H * p.W), p.mg_ohw, p.sh_ohw);
    wfill_div((unsigned)p.W, p.mg_ow, p.sh_ow);
    wfill_div((unsigned)(p.rows_per_image > 0 ? p.rows_per_image : 1), p.mg_rpi, p.sh_rpi);
    int bn = wino_bn(p);
    if (!bn) bn = p.N > 64 ? 128 : 64;  // small stand-alone problems (gd_conv3x3_winograd)
    const bool use_big = wino_big(p);
    GD_CHECK(!(use_big && p.egn_out), "conv3x3_wino: epilogue GroupNorm is not built into the 128-pair tile");
    const int BP = use_big ? 128 : 64;
    p.m_tiles = (p.M / 2 + BP - 1) / BP;
    p.n_tiles = (p.N + bn - 1) / bn;
    p.flat_tiles = p.m_tiles < 16;
    const int gx = p.flat_tiles ? p.m_tiles * p.n_tiles : ((p.m_tiles + 7) / 8) * 8 * p.n_tiles;
    if (use_big) {
        // persistent: one workgroup per CU walks tile ids with a stride that is a multiple of 8 * n_tiles (same N tile, same XCD)
        const int one_per_tile = opts().wino_nopersist;  // A/B only
        static int cus = 0;
        if (!cus) {
            int dev = 0;
            (void)hipGetDevice(&dev);
            if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus <= 0) cus = 256;
        }
        const int unit = p.flat_tiles ? p.n_tiles : 8 * p.n_tiles;
        int g = (cus / unit) * unit;
        if (g <= 0 || g > gx || one_per_tile) g = gx;
        hipLaunchKernelGGL((conv3x3_wino_f32_kernel<128, 128>), dim3(g), dim3(512), 0, s, p);
    }
    else if (bn == 128) hipLaunchKernelGGL((conv3x3_wino_f32_kernel<64, 128>), dim3(gx), dim3(256), 0, s, p);
    else hipLaunchKernelGGL((conv3x3_wino_f32_kernel<64, 64>), dim3(gx), dim3(256), 0, s, p);
    GD_HIP(hipGetLastError());
    return 0;
}

}  // namespace gd
