// 3x3 / stride 1 / pad 1 convolution in fp32 with the TWO-dimensional Winograd transform F(2x2, 3x3).
//
// Round 1's kernel transformed along x only, F(2,3): 4 products per output pair and kernel row, 12 M N Cin flops instead of 18 (removed
// in round 4: the hybrid kernel's 32-output tile covers the small launches it was kept for).  With the matrix pipe saturated at the clock
// the chip sustains the only way further is fewer products again: transforming along y as well needs 16 products
// per 2 x 2 output quad instead of 36, 8 M N Cin flops -- x 2.25 fewer than the direct sum.  Since round 3 the production path of the
// 64x64 .. 8x8 levels is the hybrid F(2,3)y x F(4,3)x kernel (conv3x3_wino42_f32.hip); this one runs the 4x4 level (channel-split launch)
// and is the A/B fallback behind GDDIM_NOWINO42=1.  For the quad with top-left output pixel (2 qy, 2 qx), its 4 x 4 input patch
// d[r][s] = in[2 qy - 1 + r][2 qx - 1 + s] and the kernel g[ky][kx]:
//     V = B^T d B,   U = G g G^T,   M_ij[quad][n] = sum_c V_ij[quad][c] U_ij[n][c],   Y = A^T M A   (2 x 2 outputs)
//     B^T = [1 0 -1 0; 0 1 1 0; 0 -1 1 0; 0 1 0 -1],  G = [1 0 0; 1/2 1/2 1/2; 1/2 -1/2 1/2; 0 0 1],  A^T = [1 1 1 0; 0 1 -1 -1]
// i.e. the one-dimensional F(2,3) applied to rows and to columns.  Both activation-side transforms are
// additions only; the 1/2 and 1/4 live in the pre-transformed weights (computed in double, rounded once): the benign Winograd size.
//
// Sixteen components x 32 x 32 accumulators do not fit one wave, so the components are split over the waves of a workgroup:
// 512 threads = 8 wave64; workgroup tile 64 quads (256 pixels) x 64 outputs x 16 components; wave (wi, wq) owns component row i = wi
// (its four components j = 0..3) of quad block wq (32 quads) x 64 outputs = eight 32x32x2 fp32 accumulators.  K = Cin is consumed in chunks of 8 channels: LDS rows [quad or n][16 components x 8
// channels + 4 pad] (132 floats = an odd number of 16-byte units: the ds_read_b128 of 16 consecutive rows cover all banks), two stages
// of 66 KB (V stage 0 | V stage 1 | U stage 0 | U stage 1: the stage is an immediate offset, the K loop is unrolled by two), one
// workgroup per CU.  One ds_read_b128 per (fragment, component) feeds four MFMAs, 32 MFMAs per wave between barriers.
// Staging: thread (quad, 4-channel half, patch row i) loads ONE row of the quad's 4 x 4 patch (4 pixels x 16 B, out-of-image pixels
// pushed out of the buffer bounds -> zero padding); the row B^T row i combines it with lives in a neighbouring lane and comes in
// through a DPP operand (16 v_fmac_f32_dpp); the x-direction transform is 8 packed-fp32 additions; four component float4s go to
// LDS, their slots rotated by the component row so that the ds_write_b128 lane groups are conflict-free.  Every thread also copies
// four 16-byte pieces of the weight chunk.  Chunk state in SGPRs, per-thread pixel offsets precomputed (they do not depend on the
// channel), distance-one register prefetch as in igemm_f32_fast_kernel; the last two chunks of a tile are peeled (nothing left to
// request) and the residual tile is requested behind the second-to-last one.  On this chip every fp32 VALU cycle is taken from the
// fp32 matrix pipe, and under this kernel the shader clock sits near 2.0 GHz (power): VALU work, LDS traffic and global requests per
// MFMA are what the K loop is tuned for (profiles/r02_notes.md: timeline, ablations, what did not help).
// Epilogue: the x-direction output transform is in-lane; the y-direction one combines the four component rows, which live in four
// waves: they exchange their 64 partial registers through the (now idle) LDS stages, then wave (h, c) of a quad block finishes quad
// half h x output tile c: bias, per-image bias row, residual, scale, store, fused GroupNorm quad statistics.
// Shapes: W in {8, 16, 32, 64} (W/2 a power of two, so a 16-quad unit is whole quad rows or an aligned piece of one and the 32-row
// statistics slices stay inside one wave), H even, H W / 4 a multiple of 16, Cin % 32 == 0 (two-source concat: C0 % 32 == 0); N % 4 == 0
// and N >= 32 when the epilogue emits GroupNorm statistics (any N otherwise: the 6-channel head convolution).
// Weights arrive as U[n][Cin/8][i][j][8] (launch_wino2d_weights; component row i = 3 negated), 16/9 of the direct size.
#include <cstdio>
#include <cstdlib>
#include <type_traits>

#include "../common.h"
#include "../options.h"
#include "act.h"
#include "epilogue.h"

namespace gd {

typedef float w2f32x16 __attribute__((ext_vector_type(16)));
typedef float w2f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int w2u32x4 __attribute__((ext_vector_type(4)));

#ifndef W2_ABL
#define W2_ABL 0  // timing ablations (results wrong): 1 no global loads in the K loop, 2 no LDS stores, 4 no fragment reads, 8 no barrier,
                  // 16 no transform arithmetic, 64 raw loads from L1-resident lines (no L2 -> L1 traffic for the activations), 128 raw loads at the addresses of a
                  // channel-blocked activation layout [image][C / 8][H][W][8] (single-source problems)
#endif

constexpr int W2_LD = 132;             // LDS row stride (floats): 16 components x 8 channels + 4 pad
constexpr int W2_HALF = 64 * W2_LD;    // floats per operand stage.  LDS image: V stage 0 | V stage 1 | U stage 0 | U stage 1, so that the
                                       // stage of an access is an immediate offset (< 64 KB) on a per-thread base address
constexpr int W2_STAGE = 2 * W2_HALF;

__device__ __forceinline__ w2f32x4 w2_load16(__amdgpu_buffer_rsrc_t rsrc, int voffset, int soffset) {
    w2u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(rsrc, voffset, soffset, 0);
    return __builtin_bit_cast(w2f32x4, v);
}

// a - b on the packed-fp32 path (two v_pk_add_f32 with the negate modifiers; the compiler packs additions by itself but scalarises
// subtractions, and on this chip every fp32 VALU cycle is taken from the fp32 matrix pipe)
__device__ __forceinline__ w2f32x4 w2_sub4(w2f32x4 a, w2f32x4 b) {
    typedef float f2 __attribute__((ext_vector_type(2)));
    const f2 alo = {a[0], a[1]}, ahi = {a[2], a[3]}, blo = {b[0], b[1]}, bhi = {b[2], b[3]};
    f2 rlo, rhi;
    asm("v_pk_add_f32 %0, %1, %2 neg_lo:[0,1] neg_hi:[0,1]" : "=v"(rlo) : "v"(alo), "v"(blo));
    asm("v_pk_add_f32 %0, %1, %2 neg_lo:[0,1] neg_hi:[0,1]" : "=v"(rhi) : "v"(ahi), "v"(bhi));
    return w2f32x4{rlo[0], rlo[1], rhi[0], rhi[1]};
}

__device__ __forceinline__ w2f32x4 w2_add4(w2f32x4 a, w2f32x4 b) {
    typedef float f2 __attribute__((ext_vector_type(2)));
    const f2 alo = {a[0], a[1]}, ahi = {a[2], a[3]}, blo = {b[0], b[1]}, bhi = {b[2], b[3]};
    f2 rlo, rhi;
    asm("v_pk_add_f32 %0, %1, %2" : "=v"(rlo) : "v"(alo), "v"(blo));
    asm("v_pk_add_f32 %0, %1, %2" : "=v"(rhi) : "v"(ahi), "v"(bhi));
    return w2f32x4{rlo[0], rlo[1], rhi[0], rhi[1]};
}

// The tile program.  SPLIT: the channel-split launch of the 4x4 level (GemmParams::wk_split), a kernel symbol of its own so that
// profiles keep the two apart.
template <bool SPLIT>
__device__ __forceinline__ void wino2d_tile(const GemmParams &p, float *smem) {
    constexpr int OOB = (int)0x80000000u;

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wi = wave & 3, wq = wave >> 2;  // component row, quad block
#ifdef W2_TRACE  // in-kernel timeline (tools/wino2d_trace.py): thread 0 of a few workgroups stamps the 100 MHz wall clock into the tail of the residual
    int trn = 0;
    const int trslot = blockIdx.x == 0 ? 0 : blockIdx.x == 700 ? 1 : blockIdx.x == 1500 ? 2 : blockIdx.x == 2040 ? 3 : -1;
    float *trbuf = const_cast<float *>(p.res) + (size_t)p.M * p.ldres - 64 + (trslot < 0 ? 0 : trslot) * 16;
    const bool trw = trslot >= 0 && tid == 0 && p.res;
    auto trace = [&]() __attribute__((always_inline)) {
        if (trw) {
            trbuf[trn] = (float)(wall_clock64() & 0xFFFFFFull);
            trbuf[8 + trn] = (float)(__builtin_readcyclecounter() & 0xFFFFFFull);  // shader clock
        }
        ++trn;
    };
    trace();
#endif
    const int li = lane & 31, lh = lane >> 5;

    // ---- XCD-aware block -> (mtile, ntile), as in the other GEMM kernels ----
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
    const int q0 = mtile * 64, n0 = ntile * 64;

    const int W = p.W, H = p.H, HW = H * W;
    const int Wq = W >> 1;        // quads per quad row: 4, 8, 16 or 32
    const int wqs = p.sh_ow;      // log2(Wq) (launch_conv3x3_wino2d)
    const int QI = HW >> 2;       // quads per image (a multiple of 16)
    const int Q = p.M >> 2;       // quads in the batch
    const int Cin = p.C0 + p.C1;
    const int a_imgs = p.M / HW;
    const int bias0 = (W + 1) * p.lda0 * 4, bias1 = (W + 1) * p.lda1 * 4;
    const __amdgpu_buffer_rsrc_t srd_a0 = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<char *>(reinterpret_cast<const char *>(p.a0) - bias0), 0, a_imgs * HW * p.lda0 * 4 + bias0, 0x00020000);
    const __amdgpu_buffer_rsrc_t srd_a1 = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<char *>(reinterpret_cast<const char *>(p.C1 ? p.a1 : p.a0) - bias1), 0, a_imgs * HW * p.lda1 * 4 + bias1, 0x00020000);
    const __amdgpu_buffer_rsrc_t srd_b = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<char *>(reinterpret_cast<const char *>(p.b_wino)), 0, p.N * p.ldb_wino * 4, 0x00020000);

    // ---- staging role: (quad ql, channel half hq, patch row si).  Thread si loads row si of the quad's 4 x 4 patch (four pixels x 16 B).
    // B^T row si combines two patch rows -- 0: r0 - r2, 1: r1 + r2, 2: r2 - r1, 3: r1 - r3 -- and the partner row sits in a neighbouring
    // lane of the same 4-lane group (lanes 2, 2, 1, 1): ONE v_fmac_f32 with a DPP quad_perm operand per element, mine + sb * partner, and
    // every input line is requested once per quad instead of twice.  Row 3 comes out NEGATED that way (r3 - r1) and so is row 3 of the
    // transformed weights (wino2d_weights_kernel): the products are unchanged. ----
    const int ql = tid >> 3, hq = (tid >> 2) & 1, si = tid & 3;
    const float sb = si == 1 ? 1.f : -1.f;
    int img_t = 0;       // the thread's image (W2_ABL & 128)
    int pix = 0;         // pixel index of the patch origin (2 qy - 1, 2 qx - 1); may be "negative" by at most W + 1: the bias covers it
    unsigned bad = 0xFu; // bit s: pixel s of the thread's patch row lies outside the image (or the quad is past the batch)
    {
        const int q = q0 + ql;
        if (q < Q) {
            const int img = (int)epi_div((unsigned)q, (unsigned)QI, p.mg_ohw, p.sh_ohw);
            const int qr = q - img * QI;
            const int qy = qr >> wqs;
            const int qx = qr - (qy << wqs);
            const int y0 = 2 * qy - 1, x0 = 2 * qx - 1;
            pix = (img * H + y0) * W + x0;
            img_t = img;
            unsigned xb = 0;
#pragma unroll
            for (int s = 0; s < 4; ++s) xb |= ((unsigned)(x0 + s) < (unsigned)W ? 0u : 1u) << s;
            bad = (unsigned)(y0 + si) < (unsigned)H ? xb : 15u;
        }
    }
    int vo[4];  // voffsets of the thread's 4 pixels in the current source (out-of-image ones already out of bounds)
    auto set_vo = [&](int lda, int biasb) __attribute__((always_inline)) {
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            int off = (pix + si * W + k) * lda * 4 + biasb + hq * 16;
            // timing ablation: addresses of a channel-blocked layout [image][C / 8][H][W][8] (the bytes read are the same tensor's, permuted)
            if (W2_ABL & 128) off = ((img_t * (lda / 8 - 1)) * HW + (pix + si * W + k)) * 32 + biasb + hq * 16;
            vo[k] = ((bad >> k) & 1u) ? OOB : off;
        }
    };
    // (set_vo below, once the chunk state is known)
    // LDS slot (16 B) of component (i, j), channel half hq in a V row: i * 8 + ((2 j + hq + 2 i) & 7) -- the j / hq slots of component
    // row i ROTATED by 2 i.  A ds_write_b128 is serviced in groups of 8 consecutive lanes = the (si, hq) of one quad, on 32 banks: without
    // the rotation the four si of a group hit the same slot modulo 8 (4-way conflict, 44 % of the LDS cycles of the kernel).
    int v_lds[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) v_lds[j] = ql * W2_LD + si * 32 + ((2 * j + hq + 2 * si) & 7) * 4;
    // weight pieces: row brow of the tile, 16-byte pieces bp + 8 k (k = 0..3) of the 128-float row chunk
    const int brow = tid >> 3, bp = tid & 7;
    const int bvo = (n0 + brow < p.N) ? ((n0 + brow) * p.ldb_wino + bp * 4) * 4 : OOB;
    const int b_lds = 2 * W2_HALF + brow * W2_LD + bp * 4;  // + k * 32

    // ---- wave-uniform state of the chunk whose loads are issued next.  With wk_split > 1 (GemmParams: 4x4 images) this workgroup is
    // split blockIdx.y of its tile and multiplies the channels [c_begin, c_begin + 8 nk) only ----
    const int ksplit = SPLIT ? p.wk_split : 1;
    const int nk = (Cin >> 3) / ksplit;  // even, >= 4
    const int c_begin = (int)blockIdx.y * nk * 8;
    int s_kc = c_begin >> 3, s_c = c_begin;
    bool s_first = p.C1 == 0 || c_begin < p.C0;  // s_c < C0: the chunk comes from source 0
    auto advance = [&]() __attribute__((always_inline)) {
        ++s_kc;
        s_c += 8;
        if (s_first && s_c >= p.C0 && p.C1 > 0) {  // once per tile: on to the second source of a channel concat
            s_first = false;
            set_vo(p.lda1, bias1);
        }
    };

    if (s_first) set_vo(p.lda0, bias0);
    else set_vo(p.lda1, bias1);

    w2f32x4 rr[4], wb[4], cm[4];
    // (chunks past the tile's last one are never requested: the last two iterations of the K loop are peeled)
    auto load_raw_to = [&](w2f32x4 *dst, int k) __attribute__((always_inline)) {
        if ((W2_ABL & 1) && s_kc >= (c_begin >> 3) + 2) return;
        if (W2_ABL & 64) { dst[k] = w2_load16(srd_a0, vo[0], 0); return; }  // every raw load of a thread hits one L1-resident line
        if (W2_ABL & 128) { dst[k] = w2_load16(srd_a0, vo[k], (s_c >> 3) * HW * 32); return; }
        if (s_first) dst[k] = w2_load16(srd_a0, vo[k], s_c * 4);
        else dst[k] = w2_load16(srd_a1, vo[k], (s_c - p.C0) * 4);
    };
    auto load_raw = [&](int k) __attribute__((always_inline)) { load_raw_to(rr, k); };
    auto load_w_to = [&](w2f32x4 *dst, int k) __attribute__((always_inline)) {
        if ((W2_ABL & 1) && s_kc >= (c_begin >> 3) + 2) return;
        dst[k] = w2_load16(srd_b, bvo, (s_kc * 128 + k * 32) * 4);
    };
    auto load_w = [&](int k) __attribute__((always_inline)) { load_w_to(wb, k); };
    auto combine_from = [&](const w2f32x4 *src) __attribute__((always_inline)) {
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            cm[s] = src[s];
            if (W2_ABL & 16) continue;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                float m = src[s][e];
                // (s_nop: a VALU write of an operand right in front of a DPP read needs two wait states, and inline asm is opaque to the
                // hazard recogniser)
                if (s == 0 && e == 0)
                    asm volatile("s_nop 1\n\tv_fmac_f32_dpp %0, %1, %2 quad_perm:[2,2,1,1] row_mask:0xf bank_mask:0xf" : "+v"(m) : "v"(src[s][e]), "v"(sb));
                else
                    asm volatile("v_fmac_f32_dpp %0, %1, %2 quad_perm:[2,2,1,1] row_mask:0xf bank_mask:0xf" : "+v"(m) : "v"(src[s][e]), "v"(sb));
                cm[s][e] = m;
            }
        }
    };
    auto combine = [&]() __attribute__((always_inline)) { combine_from(rr); };
    auto store_v = [&](int j, int par) __attribute__((always_inline)) {
        w2f32x4 v;
        if (j == 0) v = w2_sub4(cm[0], cm[2]);
        else if (j == 1) v = w2_add4(cm[1], cm[2]);
        else if (j == 2) v = w2_sub4(cm[2], cm[1]);
        else v = w2_sub4(cm[1], cm[3]);
        if (W2_ABL & 16) v = cm[j];
        if (W2_ABL & 2) { asm volatile("" ::"v"(v)); return; }
        *reinterpret_cast<w2f32x4 *>(&smem[v_lds[j] + par * W2_HALF]) = v;
    };
    auto store_w_from = [&](const w2f32x4 *src, int k, int par) __attribute__((always_inline)) {
        if (W2_ABL & 2) { asm volatile("" ::"v"(src[k])); return; }
        *reinterpret_cast<w2f32x4 *>(&smem[b_lds + par * W2_HALF + k * 32]) = src[k];
    };
    auto store_w = [&](int k, int par) __attribute__((always_inline)) { store_w_from(wb, k, par); };

    w2f32x16 acc[4][2];  // (never zeroed: the first MFMA of every accumulator, in the tile's first chunk, takes the constant 0 as its C operand)
    int a_off[4];  // fragment of component (wi, j): rotated slot, see v_lds
#pragma unroll
    for (int j = 0; j < 4; ++j) a_off[j] = (wq * 32 + li) * W2_LD + wi * 32 + ((j + wi) & 3) * 8 + lh * 4;
    constexpr int B_ROWS32 = 32 * W2_LD;
    int b_off[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) b_off[j] = 2 * W2_HALF + li * W2_LD + wi * 32 + lh * 4 + j * 8;

    // ---- prologue: chunk 0 -> LDS stage 0, chunk 1 -> registers.  Both chunks are requested up front (chunk 0 into a second register
    // set that only lives here, before the accumulators do): one memory latency per tile instead of two. ----
    {
        w2f32x4 r0[4], w0[4];
        // (the weight pieces first: their addresses need none of the pixel arithmetic)
#pragma unroll
        for (int k = 0; k < 4; ++k) load_w_to(w0, k);
#pragma unroll
        for (int k = 0; k < 4; ++k) load_raw_to(r0, k);
        advance();
#pragma unroll
        for (int k = 0; k < 4; ++k) load_w(k);
#pragma unroll
        for (int k = 0; k < 4; ++k) load_raw(k);
        combine_from(r0);
#pragma unroll
        for (int j = 0; j < 4; ++j) store_v(j, 0);
#pragma unroll
        for (int k = 0; k < 4; ++k) store_w_from(w0, k, 0);
    }
    advance();
    __syncthreads();
#ifdef W2_TRACE
    trace();
#endif

    // ---- epilogue geometry, needed already inside the K loop (the residual is requested one chunk before the loop ends): wave (h, c) of
    // a quad block finishes quad half h (16 quads) x output tile c (32 columns) ----
    const int h = wi & 1, c = wi >> 1;
    const int qu = q0 + wq * 32 + 16 * h;  // first quad of the unit (wave-uniform)
    const int img = (int)epi_div((unsigned)(qu < Q ? qu : 0), (unsigned)QI, p.mg_ohw, p.sh_ohw);
    const int qr = (qu < Q ? qu : 0) - img * QI;
    const int qy = qr >> wqs, qxu = qr - (qy << wqs);
    const int mu = img * HW + 2 * qy * W + 2 * qxu;  // pixel of the unit's first quad (top-left output)
    const int n = n0 + c * 32 + li;
    const bool nok = n < p.N && qu < Q;
    const __amdgpu_buffer_rsrc_t srd_r = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<char *>(reinterpret_cast<const char *>(p.res ? p.res : p.out)), 0, p.M * p.ldres * 4, 0x00020000);
    // quad (vv & 3) + 8 (vv >> 2) + 4 lh of the unit: the half-wave adds 4 quads = 8 pixels along x, or, on 8-wide images (4 quads
    // per quad row), one quad row = 2 W pixels
    // (4-wide images, 4 quads each: one whole image = 16 pixels)
    const int lane_px = Wq == 2 ? lh * 16 : Wq == 4 ? lh * 2 * W : 8 * lh;
    const int vo_r = nok ? (lane_px * p.ldres + n) * 4 : OOB;
    // pixel offset (from mu) of quad vv of the unit, lane-independent part: W = 8: four quad rows of 4 (vv >> 2 selects a pair of them),
    // W = 16: two quad rows of 8; else one aligned run of 16 quads
    auto pix_off = [&](int vv, int a, int b) __attribute__((always_inline)) {
        const int dq = (vv & 3) + 8 * (vv >> 2);
        const int base = Wq == 2   ? (vv >> 2) * 32 + ((vv & 3) >> 1) * 2 * W + (vv & 1) * 2  // W = 4: quads 8 apart are two images apart
                         : Wq == 4 ? (vv >> 2) * 4 * W + (vv & 3) * 2
                         : Wq == 8 ? (vv >> 2) * 2 * W + (vv & 3) * 2
                                   : dq * 2;
        return mu + base + a * W + b;
    };
    float rs[2][2][8];
    const bool has_res = p.res != nullptr && ksplit == 1;
    auto load_res = [&](int idx) __attribute__((always_inline)) {  // idx = (a * 2 + b) * 8 + vv
        const int a = idx >> 4, b = (idx >> 3) & 1, vv = idx & 7;
        rs[a][b][vv] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(srd_r, vo_r, pix_off(vv, a, b) * p.ldres * 4, 0));
    };

    // ---- K loop.  MODE 0: steady state (chunk kc multiplied, chunk kc+1 registers -> the other LDS stage, chunk kc+2 requested);
    // MODE 1: the tile's second-to-last chunk (nothing left to request: the residual tile is requested instead, its latency hides
    // behind the last chunk); MODE 2: the last chunk (no staging at all) ----
    w2f32x4 af[2], bf[2][2];
    auto chunk = [&](auto mode_tag, auto par_tag, auto first_tag) __attribute__((always_inline)) {
        constexpr int MODE = decltype(mode_tag)::value;
        constexpr int PAR = decltype(par_tag)::value;  // stage the chunk is multiplied from (compile time: immediate offsets)
        constexpr bool FIRST = decltype(first_tag)::value != 0;  // the tile's first chunk: accumulators start from the constant 0
        const float *cur = smem + PAR * W2_HALF;
        constexpr int nxt = PAR ^ 1;
        af[0] = *reinterpret_cast<const w2f32x4 *>(cur + a_off[0]);
        bf[0][0] = *reinterpret_cast<const w2f32x4 *>(cur + b_off[0]);
        bf[0][1] = *reinterpret_cast<const w2f32x4 *>(cur + b_off[0] + B_ROWS32);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
#pragma unroll
            for (int jj = 0; jj < 4; ++jj) {
                if (FIRST && jj == 0) {
                    const w2f32x16 zero = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
                    acc[j][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[j & 1][jj], bf[j & 1][0][jj], zero, 0, 0, 0);
                    acc[j][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[j & 1][jj], bf[j & 1][1][jj], zero, 0, 0, 0);
                } else {
                    acc[j][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[j & 1][jj], bf[j & 1][0][jj], acc[j][0], 0, 0, 0);
                    acc[j][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[j & 1][jj], bf[j & 1][1][jj], acc[j][1], 0, 0, 0);
                }
                __builtin_amdgcn_sched_barrier(0);
                const int gap = j * 4 + jj;
                if (jj == 0 && j < 3 && !(W2_ABL & 4)) {  // fragments of the next component, one component ahead of their use
                    af[(j + 1) & 1] = *reinterpret_cast<const w2f32x4 *>(cur + a_off[j + 1]);
                    bf[(j + 1) & 1][0] = *reinterpret_cast<const w2f32x4 *>(cur + b_off[j + 1]);
                    bf[(j + 1) & 1][1] = *reinterpret_cast<const w2f32x4 *>(cur + b_off[j + 1] + B_ROWS32);
                }
                // staging work in the shadow of the MFMA pair
                // (the weights first: loads return in order, and theirs are the short ones)
                if (MODE < 2) {
                    if (gap < 4) store_w(gap, nxt);
                    if (gap == 4) combine();
                    if (gap >= 4 && gap < 8) store_v(gap - 4, nxt);
                }
                if (MODE == 0) {
                    if (gap < 4) load_w(gap);
                    else if (gap < 8) load_raw(gap - 4);
                    else if (gap == 8) advance();
                }
                if (MODE == 1 && has_res) {
                    load_res(2 * gap);
                    load_res(2 * gap + 1);
                }
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        if (!(W2_ABL & 8)) __syncthreads();
    };
    using I0 = std::integral_constant<int, 0>;
    using I1 = std::integral_constant<int, 1>;
    using I2 = std::integral_constant<int, 2>;
    chunk(I0{}, I0{}, I1{});  // (nk is even and >= 4: the first pair of chunks always has something to request)
    chunk(I0{}, I1{}, I0{});
    for (int kc = 2; kc < nk - 2; kc += 2) {
        chunk(I0{}, I0{}, I0{});
        chunk(I0{}, I1{}, I0{});
    }
    chunk(I1{}, I0{}, I0{});
    chunk(I2{}, I1{}, I0{});
#ifdef W2_TRACE
    trace();
#endif

    // ---- output transform along x (in-lane): z[b] = sum_j A^T[b][j] M_ij ----
    // then along y across the four component-row waves of a quad block, through LDS (the stages are idle after the last barrier)
    float *mine = smem + (wq * 4 + wi) * 4096 + lane;
#pragma unroll
    for (int c = 0; c < 2; ++c)
#pragma unroll
        for (int v = 0; v < 16; ++v) {
            const float m0 = acc[0][c][v], m1 = acc[1][c][v], m2 = acc[2][c][v], m3 = acc[3][c][v];
            mine[((0 * 2 + c) * 16 + v) * 64] = m0 + m1 + m2;
            mine[((1 * 2 + c) * 16 + v) * 64] = m1 - m2 - m3;
        }
    __syncthreads();
    if (qu >= Q) return;
    float y[2][2][8];  // [a][b][vv]: output pixel (2 qy + a, 2 qx + b) of quad row (vv & 3) + 8 (vv >> 2) + 4 lh of the unit
    {
        const float *src = smem + (wq * 4) * 4096 + lane;
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int vv = 0; vv < 8; ++vv) {
                const int o = ((b * 2 + c) * 16 + 8 * h + vv) * 64;
                const float z0 = src[o], z1 = src[4096 + o], z2 = src[2 * 4096 + o], z3 = src[3 * 4096 + o];
                y[0][b][vv] = z0 + z1 + z2;
                y[1][b][vv] = z1 - z2 - z3;
            }
    }

#ifdef W2_TRACE
    trace();
#endif
    if (SPLIT) {
        // one split of the channel range: the partial outputs go to wk_partial[split][m][n] as they are; bias, residual, scale,
        // statistics and the GroupNorm belong to wino2d_splitk_finish_kernel
        float *part = p.wk_partial + (size_t)blockIdx.y * p.M * p.N;
        const __amdgpu_buffer_rsrc_t srd_p = __builtin_amdgcn_make_buffer_rsrc(reinterpret_cast<char *>(part), 0, p.M * p.N * 4, 0x00020000);
        const int vo_p = nok ? (lane_px * p.N + n) * 4 : OOB;
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
            for (int b = 0; b < 2; ++b)
#pragma unroll
                for (int vv = 0; vv < 8; ++vv)
                    __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, y[a][b][vv]), srd_p, vo_p, pix_off(vv, a, b) * p.N * 4, 0);
        return;
    }
    // ---- epilogue of the unit ----
    const __amdgpu_buffer_rsrc_t srd_o =
        __builtin_amdgcn_make_buffer_rsrc(reinterpret_cast<char *>(p.out), 0, p.M * p.ldc * 4, 0x00020000);
    const int vo_o = nok ? (lane_px * p.ldc + n) * 4 : OOB;
    float addv = (p.bias && nok) ? p.bias[n] : 0.f;
    if (p.rowbias && nok) addv += p.rowbias[(size_t)img * p.ld_rowbias + n];
    float ps[2][2] = {{0.f, 0.f}, {0.f, 0.f}}, pq[2][2] = {{0.f, 0.f}, {0.f, 0.f}};  // partial statistics by (a, vv >> 2)
    float vals[2][2][8];  // finished outputs (kept for the epilogue GroupNorm)
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int vv = 0; vv < 8; ++vv) {
                float add = addv;
                if (has_res) add += rs[a][b][vv];
                const float val = (p.alpha * y[a][b][vv] + add) * p.out_scale;
                vals[a][b][vv] = val;
                if (!p.egn_skip_raw)
                    __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, val), srd_o, vo_o, pix_off(vv, a, b) * p.ldc * 4, 0);
                ps[a][vv >> 2] += val;
                pq[a][vv >> 2] += val * val;
            }
#ifdef W2_TRACE
    trace();
    __builtin_amdgcn_s_waitcnt(0);
    trace();
#endif
    if (p.egn_out) {
        // GroupNorm (+ swish) of the output, here (plan.h: Op::egn_peer).  rows_per_image == 64 (8x8 images): this unit is exactly one
        // image x 32 columns.  rows_per_image == 256 (16x16 images): the image is the whole workgroup tile, and the four units (wq, h) of
        // a column block add their column sums through the 4 KB of LDS behind the exchange area.  A group is egn_cpg consecutive
        // columns = consecutive lanes.
        float gsum = (ps[0][0] + ps[0][1]) + (ps[1][0] + ps[1][1]), gsq = (pq[0][0] + pq[0][1]) + (pq[1][0] + pq[1][1]);
        gsum += __shfl_xor(gsum, 32); gsq += __shfl_xor(gsq, 32);
        float rows = 64.0f;
        if (p.rows_per_image == 256) {
            float2 *red = reinterpret_cast<float2 *>(smem + 8 * 4096);  // [c][unit][32 columns]
            if (lh == 0) red[(c * 4 + wq * 2 + h) * 32 + li] = make_float2(gsum, gsq);
            __syncthreads();  // (every wave of the workgroup is here: whole images, so no unit is past the batch)
            gsum = 0.f; gsq = 0.f;
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const float2 v = red[(c * 4 + u) * 32 + li];
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
            for (int b = 0; b < 2; ++b)
#pragma unroll
                for (int vv = 0; vv < 8; ++vv) {
                    float t = vals[a][b][vv] * sc + sh;
                    if (p.egn_act) t = swish_e(t);
                    __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, t), srd_g, vo_o, pix_off(vv, a, b) * p.ldc * 4, 0);
                }
    }
    if (p.qstats) {
        // the unit's 64 pixels are two 32-row slices of the output matrix: W = 8 / 16: its quads with vv >> 2 = 0 / 1 (two quad rows / one
        // quad row, both pixel rows of each); W = 32 / 64: its two pixel rows (a whole row / an aligned half of one)
#pragma unroll
        for (int g = 0; g < 2; ++g) {
            float s = Wq <= 8 ? ps[0][g] + ps[1][g] : ps[g][0] + ps[g][1];
            float q = Wq <= 8 ? pq[0][g] + pq[1][g] : pq[g][0] + pq[g][1];
            if (!nok) { s = 0.f; q = 0.f; }
            s += __shfl_xor(s, 1); q += __shfl_xor(q, 1);
            s += __shfl_xor(s, 2); q += __shfl_xor(q, 2);
            s += __shfl_xor(s, 32); q += __shfl_xor(q, 32);
            if ((li & 3) == 0 && lh == 0 && nok) {
                const int first = (mu - img * HW) + (Wq <= 8 ? g * 32 : g * W);  // first pixel of the slice inside the image
                float2 *o = reinterpret_cast<float2 *>(p.qstats) + ((size_t)img * p.qs_slices + (first >> 5)) * (p.N >> 2) + (n >> 2);
                o[0] = make_float2(s, q);
            }
        }
    }
}

__global__ __launch_bounds__(512, 2) void conv3x3_wino2d_f32_kernel(const GemmParams p) {
    __shared__ __attribute__((aligned(16))) float smem[2 * W2_STAGE];
    wino2d_tile<false>(p, smem);
}

__global__ __launch_bounds__(512, 2) void conv3x3_wino2d_splitk_f32_kernel(const GemmParams p) {
    __shared__ __attribute__((aligned(16))) float smem[2 * W2_STAGE];
    wino2d_tile<true>(p, smem);
}

// U[n][c/8][i][j][c%8] = (G g G^T)_ij from the packed direct weights w[n][(ky*3 + kx)*Cin + c]; sums in double, one rounding
__global__ void wino2d_weights_kernel(const float *__restrict__ w, float *__restrict__ u, int N, int Cin) {
    const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= (long long)N * Cin) return;
    const int c = (int)(idx % Cin);
    const long long n = idx / Cin;
    const float *wr = w + n * 9 * Cin + c;
    double g[3][3];
    for (int ky = 0; ky < 3; ++ky)
        for (int kx = 0; kx < 3; ++kx) g[ky][kx] = wr[(ky * 3 + kx) * Cin];
    const double G[4][3] = {{1, 0, 0}, {0.5, 0.5, 0.5}, {0.5, -0.5, 0.5}, {0, 0, 1}};
    float *ur = u + n * 16 * Cin + (long long)(c / 8) * 128 + (c & 7);
    for (int i = 0; i < 4; ++i)
        for (int j = 0; j < 4; ++j) {
            double acc = 0;
            for (int ky = 0; ky < 3; ++ky)
                for (int kx = 0; kx < 3; ++kx) acc += G[i][ky] * g[ky][kx] * G[j][kx];
            ur[(i * 4 + j) * 8] = (float)(i == 3 ? -acc : acc);  // row 3 negated, like the staged activations (see the kernel)
        }
}

// ---- host side ---------------------------------------------------------------------------------------------------------------
int launch_wino2d_weights(const float *w, float *u, int N, int Cin, hipStream_t s) {
    GD_CHECK(w && u && N > 0 && Cin > 0 && Cin % 8 == 0, "wino2d_weights: bad argument (N=%d Cin=%d)", N, Cin);
    const long long total = (long long)N * Cin;
    hipLaunchKernelGGL(wino2d_weights_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, w, u, N, Cin);
    GD_HIP(hipGetLastError());
    return 0;
}

static void w2fill_div(unsigned d, unsigned &magic, unsigned &shift) {
    unsigned s = 0;
    while ((1ull << s) < d) ++s;
    shift = s;
    magic = (unsigned)((((1ull << 32) * ((1ull << s) - d)) / d) + 1);
}

// what the kernel can compute at all (p.b_wino is not looked at)
bool conv3x3_wino2d_shape_ok(const GemmParams &p) {
    if (p.in_act || p.b_kmajor || p.a_presplit || p.gn_table) return false;
    if (p.egn_out && ((p.rows_per_image != 64 && p.rows_per_image != 256) || p.N % p.egn_cpg != 0 ||
                      (p.egn_cpg != 4 && p.egn_cpg != 8 && p.egn_cpg != 16 && p.egn_cpg != 32))) return false;
    if (p.KH != 3 || p.KW != 3 || p.stride != 1 || p.pad != 1 || p.OH != p.H || p.OW != p.W) return false;
    const int Cin = p.C0 + p.C1;
    if (Cin % 32 != 0 || (p.C1 > 0 && (p.C0 % 32 != 0 || !p.a1)) || p.K != 9 * Cin) return false;
    const bool img4 = p.W == 4 && p.H == 4 && p.wk_split > 1;  // 4x4 images: only with the channel range split (launch_conv3x3_wino2d_splitk)
    if (!img4 && ((p.W != 8 && p.W != 16 && p.W != 32 && p.W != 64) || p.H % 2 != 0 || (p.H * p.W / 4) % 16 != 0)) return false;
    // (a split's chunk count: even and >= 4, the K loop's unrolling; a split may start in either source of a channel concat)
    if (p.wk_split > 1 && (((p.C0 + p.C1) / 8) % p.wk_split != 0 || ((p.C0 + p.C1) / 8 / p.wk_split) % 2 != 0 || (p.C0 + p.C1) / 8 / p.wk_split < 4 || !p.wk_partial))
        return false;
    if (p.rows_per_image != p.H * p.W || p.M % (p.H * p.W) != 0 || p.N < 1) return false;
    if ((p.N % 4 != 0 || p.N < 32) && (p.qstats || p.egn_out)) return false;  // (a narrow or ragged N only without the column-quad statistics)
    if (p.qstats && p.qs_slices != p.rows_per_image / 32) return false;
    if (p.rowbias && p.ld_rowbias < 0) return false;  // (0: one row for every image)
    if ((long long)p.M * p.ldc * 4 + 64 >= (1LL << 31) || (p.res && (long long)p.M * p.ldres * 4 + 64 >= (1LL << 31))) return false;
    const long long ldmax = p.lda0 > p.lda1 ? p.lda0 : p.lda1;
    const long long a_bytes = ((long long)p.M + 3 * p.W + 8) * ldmax * 4 + Cin * 4 + 64;
    return a_bytes < (1LL << 31) && (long long)p.N * 16 * Cin * 4 + 64 < (1LL << 31);
}

// what the engine sends to it: an eligible shape with at least one workgroup per CU.  GDDIM_NOWINO2=1 routes everything back to the
// direct implicit-GEMM kernel; GDDIM_WINO2_MINBLOCKS overrides the fill criterion (A/B).
bool conv3x3_wino2d_ok(const GemmParams &p) {
    if (opts().no_wino2 || !conv3x3_wino2d_shape_ok(p)) return false;
    return (long long)((p.M / 4 + 63) / 64) * ((p.N + 63) / 64) >= opts().wino2_min_blocks;
}

const char *conv3x3_wino2d_name() { return "conv3x3_wino2d_f32_kernel"; }

int launch_conv3x3_wino2d(const GemmParams &p_in, hipStream_t s) {
    GemmParams p = p_in;
    {   // the kernel addresses with 32-bit byte offsets into buffer descriptors (voffset + soffset): every operand, including the
        // (W + 1)-pixel lead-in of the inputs, has to stay below 2^31 bytes (B = 512 of 64x64x128: 1.07 GB per tensor)
        const long long ldmax = p.lda0 > p.lda1 ? p.lda0 : p.lda1;
        GD_CHECK(((long long)p.M + 3 * p.W + 8) * ldmax * 4 + (long long)(p.C0 + p.C1) * 4 + 64 < (1LL << 31) &&
                     (long long)p.M * p.ldc * 4 + 64 < (1LL << 31) && (!p.res || (long long)p.M * p.ldres * 4 + 64 < (1LL << 31)),
                 "conv3x3_wino2d: operand beyond the 2^31-byte range of 32-bit buffer offsets (M=%d lda=%lld ldc=%d)", p.M, ldmax, p.ldc);
    }
    GD_CHECK(conv3x3_wino2d_shape_ok(p) && p.b_wino, "conv3x3_wino2d: problem not eligible");
    p.ldb_wino = 16 * (p.C0 + p.C1);
    if (p.C1 == 0) p.lda1 = 4;
    w2fill_div((unsigned)(p.H * p.W / 4), p.mg_ohw, p.sh_ohw);  // quads per image
    p.sh_ow = p.W == 4 ? 1 : p.W == 8 ? 2 : p.W == 16 ? 3 : p.W == 32 ? 4 : 5;  // log2(W / 2)
    p.m_tiles = (p.M / 4 + 63) / 64;
    p.n_tiles = (p.N + 63) / 64;
    p.flat_tiles = p.m_tiles < 16;
    const int gx = p.flat_tiles ? p.m_tiles * p.n_tiles : ((p.m_tiles + 7) / 8) * 8 * p.n_tiles;
    if (p.wk_split > 1) hipLaunchKernelGGL(conv3x3_wino2d_splitk_f32_kernel, dim3(gx, p.wk_split), dim3(512), 0, s, p);
    else hipLaunchKernelGGL(conv3x3_wino2d_f32_kernel, dim3(gx), dim3(512), 0, s, p);
    GD_HIP(hipGetLastError());
    return 0;
}

// ---- 4x4 images: channel range split over workgroups + a finishing pass ------------------------------------------------------
// The 4x4 level of the U-Net has M = 16 B rows: 64 workgroup tiles at B = 256, a quarter of the CUs.  Every tile is therefore given
// wk_split workgroups, each with 1 / wk_split of the channels; they store their partial outputs (after the output transform) and this
// kernel adds them IN INDEX ORDER (deterministic) and runs the shared epilogue: alpha, bias, per-image bias row, residual, scale,
// store, GroupNorm quad statistics (one slice per 16-row image) and the epilogue GroupNorm (plan.h: Op::egn_peer).
// A wave takes (image b, block of 16 column quads): lane = (pixel group pxg, column quad f_lo), four pixels per lane, so that a load
// instruction covers four 256-byte row pieces and every partial is requested up front; the 16-pixel sums come from two shuffles.
__global__ __launch_bounds__(256) void wino2d_splitk_finish_kernel(const GemmParams p, int images) {
    const int NQ = p.N >> 2, FB = (NQ + 15) >> 4;
    const int lane = threadIdx.x & 63;
    const long long gw = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (gw >= (long long)images * FB) return;  // (whole waves leave: the shuffles below are among live lanes of one wave)
    const int b = (int)(gw / FB), fb = (int)(gw - (long long)b * FB);
    const int f = fb * 16 + (lane & 15), pxg = lane >> 4;
    const bool live = f < NQ;
    const int n = live ? f * 4 : 0;
    typedef float f4 __attribute__((ext_vector_type(4)));
    f4 add = {0.f, 0.f, 0.f, 0.f};
    if (p.bias) add = *reinterpret_cast<const f4 *>(p.bias + n);
    if (p.rowbias) add += *reinterpret_cast<const f4 *>(p.rowbias + (size_t)b * p.ld_rowbias + n);
    const size_t slice = (size_t)p.M * p.N;
    f4 part[4][8], rs[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const size_t m = (size_t)b * 16 + pxg * 4 + i;
#pragma unroll
        for (int sp = 0; sp < 8; ++sp)
            if (sp < p.wk_split) part[i][sp] = __builtin_nontemporal_load(reinterpret_cast<const f4 *>(p.wk_partial + sp * slice + m * p.N + n));
        if (p.res) rs[i] = *reinterpret_cast<const f4 *>(p.res + m * p.ldres + n);
    }
    f4 vals[4];
    f4 s4 = {0.f, 0.f, 0.f, 0.f}, q4 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const size_t m = (size_t)b * 16 + pxg * 4 + i;
        f4 acc = part[i][0];
#pragma unroll
        for (int sp = 1; sp < 8; ++sp)
            if (sp < p.wk_split) acc += part[i][sp];  // index order: deterministic
        f4 v = p.alpha * acc + add;
        if (p.res) v += rs[i];
        v *= p.out_scale;
        if (!live) v = f4{0.f, 0.f, 0.f, 0.f};
        vals[i] = v;
        if (live && !p.egn_skip_raw) *reinterpret_cast<f4 *>(p.out + m * p.ldc + n) = v;
        s4 += v;
        q4 += v * v;
    }
    float gs = (s4[0] + s4[1]) + (s4[2] + s4[3]), gq = (q4[0] + q4[1]) + (q4[2] + q4[3]);
    gs += __shfl_xor(gs, 16); gq += __shfl_xor(gq, 16);
    gs += __shfl_xor(gs, 32); gq += __shfl_xor(gq, 32);
    if (p.qstats && live && pxg == 0) reinterpret_cast<float2 *>(p.qstats)[((size_t)b * p.qs_slices) * NQ + f] = make_float2(gs, gq);
    if (p.egn_out) {
        // a group is egn_cpg consecutive columns = egn_cpg / 4 consecutive quads, inside this wave's 16-quad block
        for (int o = 1; o < (p.egn_cpg >> 2); o <<= 1) { gs += __shfl_xor(gs, o); gq += __shfl_xor(gq, o); }
        const float inv_n = 1.0f / (16.0f * (float)p.egn_cpg);
        const float mean = gs * inv_n;
        const float rstd = rsqrtf(gq * inv_n - mean * mean + 1e-6f);
        if (live) {
            const f4 ga = *reinterpret_cast<const f4 *>(p.egn_gamma + n), be = *reinterpret_cast<const f4 *>(p.egn_beta + n);
            const f4 sc = rstd * ga, sh = be - mean * rstd * ga;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                f4 y = vals[i] * sc + sh;
                if (p.egn_act) {
#pragma unroll
                    for (int k = 0; k < 4; ++k) y[k] = swish_e(y[k]);
                }
                *reinterpret_cast<f4 *>(p.egn_out + ((size_t)b * 16 + pxg * 4 + i) * p.ldc + n) = y;
            }
        }
    }
}

// The number of splits for this problem, 0 when the split path does not take it.  GDDIM_NOWINO2_SPLITK=1 switches it off (A/B).
int conv3x3_wino2d_splitk_factor(const GemmParams &p) {
    if (opts().no_wino2_splitk || p.H != 4 || p.W != 4 || p.rows_per_image != 16 || p.M % 16 != 0 || p.N % 4 != 0 || p.N < 32 || p.ldc % 4 != 0) return 0;
    if (p.res && p.ldres % 4 != 0) return 0;
    if (p.rowbias && p.ld_rowbias % 4 != 0) return 0;
    if (p.qstats && p.qs_slices != 1) return 0;
    if (p.egn_out && (p.N % p.egn_cpg != 0 || (p.egn_cpg != 4 && p.egn_cpg != 8 && p.egn_cpg != 16 && p.egn_cpg != 32) || (p.N / 4) % (p.egn_cpg / 4) != 0)) return 0;
    const int Cin = p.C0 + p.C1;
    if (Cin % 32 != 0) return 0;
    const long long tiles = (long long)((p.M / 4 + 63) / 64) * ((p.N + 63) / 64);
    const int cus = device_cus();
    // at most one workgroup per CU where the tiles allow it (GDDIM_WINO2_KS forces a factor: A/B); a launch with as many tiles as CUs or
    // more (B >= 1024 on 256 CUs: the unsplit F(2x2,3x3) kernel does not take 4-wide images, and the direct kernel ran these layers at
    // 124 TFLOP/s in round 5's batch sweep) still splits in two and walks its 2 x tiles workgroups in rounds
    const auto eligible = [&](int ks) {
        GemmParams q = p;
        q.wk_split = ks;
        q.wk_partial = reinterpret_cast<float *>(16);  // (only tested for null)
        q.egn_out = nullptr; q.qstats = nullptr;        // (the main kernel emits neither)
        return conv3x3_wino2d_shape_ok(q);
    };
    for (int ks = 8; ks >= 2; ks >>= 1) {
        if (opts().wino2_ks ? ks != opts().wino2_ks : tiles * ks > cus) continue;
        if (eligible(ks)) return ks;
    }
    if (!opts().wino2_ks && tiles * 2 > cus && tiles <= 4LL * cus && eligible(2)) return 2;
    return 0;
}

const char *conv3x3_wino2d_splitk_name() { return "conv3x3_wino2d_splitk_f32_kernel"; }

int launch_conv3x3_wino2d_splitk(const GemmParams &p_in, hipStream_t s) {
    GemmParams p = p_in;
    p.wk_split = conv3x3_wino2d_splitk_factor(p_in);
    GD_CHECK(p.wk_split > 1 && p.wk_partial && p.b_wino, "conv3x3_wino2d_splitk: problem not eligible");
    GemmParams m = p;  // the main kernel: partial outputs only
    m.bias = nullptr; m.rowbias = nullptr; m.res = nullptr; m.qstats = nullptr; m.egn_out = nullptr; m.egn_skip_raw = 0;
    GD_TRY(launch_conv3x3_wino2d(m, s));
    const long long waves = (long long)(p.M / 16) * ((p.N / 4 + 15) / 16);
    hipLaunchKernelGGL(wino2d_splitk_finish_kernel, dim3((unsigned)((waves + 3) / 4)), dim3(256), 0, s, p, p.M / 16);
    GD_HIP(hipGetLastError());
    return 0;
}

}  // namespace gd
