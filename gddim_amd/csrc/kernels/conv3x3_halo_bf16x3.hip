// 3x3 / stride 1 / pad 1 convolution in split precision (bf16x3) with the input staged ONCE per 32-channel slab.
//
// igemm_bf16x3_kernel fetches every activation element once per tap (nine times) from L2 into LDS; that traffic, not the matrix
// pipe, bounds it (profiles/r01_igemm_investigation.md).  Here a workgroup's 256 output pixels are `rows` complete image rows of one
// image (W = 32: 8 rows, W = 16: the whole 16x16 image), so their receptive field is a (rows + 2) x (W + 2) pixel halo; the halo of a
// 32-channel slab (pre-split bf16 hi and lo planes, written by gn_apply_kernel) is copied to LDS once and the nine taps read their A
// fragments from it at shifted addresses: tap (ty, tx) of output pixel (y, x) is halo pixel (y + ty, x + tx).  Activation traffic
// L2 -> LDS drops from 9 x 256 to (rows + 2)(W + 2) pixels per slab (x 6.8 less at 32x32); the weight chunk of a (tap, slab) is
// staged per tap as before.  Image borders are zeros in the halo (buffer bounds check / out-of-image mask), so the K loop has no
// predication at all.
//
// Tile 256 x 128, 8 waves (4 x 2), each 2 x 2 accumulators of v_mfma_f32_32x32x16_bf16, three MFMAs per product term
// (a_hi b_hi + a_hi b_lo + a_lo b_hi).  LDS: halo [2 stages][hi | lo][HP pixels][80 B] + weights [2 stages][hi | lo][128][80 B]
// (80-byte records: the 16-byte fragment reads of consecutive pixels fall on distinct bank groups).
// K order: slab outermost, taps innermost -- the same order as the packed weights' scalar addressing assumes ([n][tap][c]).
#include <cstdio>
#include <cstdlib>

#include "../common.h"
#include "../options.h"
#include "act.h"
#include "epilogue.h"

namespace gd {

typedef float hf32x16 __attribute__((ext_vector_type(16)));
typedef float hf32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int hu32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 hbf16x8 __attribute__((ext_vector_type(8)));

constexpr int HBM_ = 256, HBN_ = 128;      // tile
constexpr int HREC = 40;                   // LDS record stride in bf16 elements: 64 B of data + 16 B pad
constexpr int HP_MAX = 10 * 34;            // halo pixels: max over (W = 32: 10 x 34 = 340, W = 16: 18 x 18 = 324)
constexpr int HA_PLANE = HP_MAX * HREC;    // one plane (hi or lo) of the halo, elements
constexpr int HB_PLANE = HBN_ * HREC;      // one plane of the weight chunk
constexpr int HSTAGE = 2 * HA_PLANE + 2 * HB_PLANE;  // elements per stage: A_hi | A_lo | B_hi | B_lo
constexpr int HAJ = (HP_MAX * 8 + 511) / 512;        // 16-byte halo pieces per thread per slab (8 per pixel: 4 hi + 4 lo)

// FUSE: the input is the RAW fp32 tensor of a GroupNorm(+swish) whose per-(image, channel) affine map comes in p.gn_table; the
// normalisation, the activation and the bf16 hi/lo split are applied to the halo values in registers between the global load and
// the LDS store, once per halo element (x 1.33 of the tile) -- the separate gn_apply pass over the tensor (4 B read + 4 B write per
// element) disappears.  A thread's pieces are float4 columns (4 channels) of one fixed 4-channel slot of every slab.
template <bool FUSE>
__global__ __launch_bounds__(512) void conv3x3_halo_bf16x3_kernel(const GemmParams p) {
    constexpr int NT = 512, WN = 2, RM = 2, RN = 2;
    constexpr int OOB = (int)0x80000000u;
    __shared__ __attribute__((aligned(16))) __bf16 smem[2 * HSTAGE];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / WN, wn = wave % WN;

    const int nt = p.n_tiles;
    const int W = p.W, H = p.H, C = p.C0 + p.C1;  // C1 > 0 only with FUSE: the GroupNorm ran on a channel concat of two tensors
    const int HW2 = W + 2;
    const int rows = HBM_ / W;                 // complete image rows per tile (H*W is a multiple of 256)
    const int HP = (rows + 2) * HW2;           // halo pixels
    // The kernel is PERSISTENT: a workgroup walks tile ids blockIdx.x, blockIdx.x + gridDim.x, ... (gridDim.x a multiple of 8, so
    // the XCD of an id is preserved) and requests the first loads of its next tile BEFORE the epilogue of the current one, so that
    // their latency hides behind the epilogue instead of opening every tile.  p.diag == 1 (A/B) launches one workgroup per tile.
    const int total_ids = p.flat_tiles ? p.m_tiles * nt : ((p.m_tiles + 7) / 8) * 8 * nt;
    auto tile_of = [&](int id, int &mtile, int &ntile) {
        if (p.flat_tiles) {
            mtile = id / nt;
            ntile = id - mtile * nt;
        } else {
            const int grp = id / (8 * nt);
            const int rem = id - grp * 8 * nt;
            ntile = rem >> 3;
            mtile = grp * 8 + (rem & 7);
        }
        return mtile < p.m_tiles;
    };
    int m0 = 0, n0 = 0, img = 0, y0 = 0;       // coordinates of the tile whose loads are being issued

    const char *ac = reinterpret_cast<const char *>(p.a0);
    const int a_imgs = (p.M + H * W - 1) / (H * W);
    const __amdgpu_buffer_rsrc_t srd_a =
        __builtin_amdgcn_make_buffer_rsrc(const_cast<char *>(ac), 0, a_imgs * H * W * p.lda0 * 4, 0x00020000);
    const char *a1c = reinterpret_cast<const char *>((FUSE && p.C1) ? p.a1 : p.a0);
    const __amdgpu_buffer_rsrc_t srd_a1 =
        __builtin_amdgcn_make_buffer_rsrc(const_cast<char *>(a1c), 0, a_imgs * H * W * p.lda1 * 4, 0x00020000);
    const __amdgpu_buffer_rsrc_t srd_bh = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<char *>(reinterpret_cast<const char *>(p.b_hi)), 0, p.N * p.ldb * 2, 0x00020000);
    const __amdgpu_buffer_rsrc_t srd_bl = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<char *>(reinterpret_cast<const char *>(p.b_lo)), 0, p.N * p.ldb * 2, 0x00020000);

    // ---- halo pieces of this thread: piece q = tid + 512 j -> (plane, pixel, 16-byte slot).  The LDS side (alds) does not depend
    // on the tile; the global side (avo, avo1, bvo, tab_base) is recomputed by setup() for every tile ----
    int avo[HAJ];   // global voffset (bytes, without the channel slab), OOB for out-of-image / unused pieces
    int avo1[FUSE ? HAJ : 1];  // the same for the second source of a fused concat GroupNorm
    int alds[HAJ];  // LDS element offset inside a stage
#pragma unroll
    for (int j = 0; j < HAJ; ++j) {
        const int q = tid + NT * j;
        // pre-split source: piece -> (plane, pixel, 16-byte slot of 8 bf16); raw fp32 source: piece -> (pixel, float4 slot), the
        // slot (q & 7 = tid & 7) is the same for all pieces of a thread
        const int plane = FUSE ? 0 : (q >= HP * 4 ? 1 : 0);
        const int r = q - plane * HP * 4;
        const int pix = FUSE ? q >> 3 : r >> 2, slot = FUSE ? q & 7 : r & 3;
        alds[j] = q < HP * 8 ? (FUSE ? pix * HREC + slot * 4 : plane * HA_PLANE + pix * HREC + slot * 8) : 0;
    }
    const int brow = tid >> 2;        // weight row inside the chunk (128 rows x 4 threads)
    const int bq = (tid & 3) * 8;
    int bvo = OOB;
    const float *tab_base = nullptr;  // fused GroupNorm: [C scale | C shift] of the tile's image, at the thread's 4-channel slot
    auto setup = [&](int id) {
        int mtile, ntile;
        tile_of(id, mtile, ntile);
        m0 = mtile * HBM_;
        n0 = ntile * HBN_;
        img = m0 / (H * W);
        y0 = (m0 - img * H * W) / W;  // first output row of the tile inside its image
#pragma unroll
        for (int j = 0; j < HAJ; ++j) {
            const int q = tid + NT * j;
            const int plane = FUSE ? 0 : (q >= HP * 4 ? 1 : 0);
            const int r = q - plane * HP * 4;
            const int pix = FUSE ? q >> 3 : r >> 2, slot = FUSE ? q & 7 : r & 3;
            avo[j] = OOB;  // outside the image (the load returns zero and the zero is stored) or an unused piece
            if (FUSE) avo1[j] = OOB;
            const int hy = pix / HW2, hx = pix - hy * HW2;
            const int y = y0 + hy - 1, x = hx - 1;
            if (q < HP * 8 && (unsigned)y < (unsigned)H && (unsigned)x < (unsigned)W && img < a_imgs) {
                avo[j] = FUSE ? ((img * H + y) * W + x) * p.lda0 * 4 + slot * 16
                              : ((img * H + y) * W + x) * p.lda0 * 4 + plane * p.lda0 * 2 + slot * 16;
                if (FUSE) avo1[j] = ((img * H + y) * W + x) * p.lda1 * 4 + slot * 16;
            }
        }
        const int bn = n0 + brow;
        bvo = (brow < HBN_ && bn < p.N) ? (bn * p.ldb + bq) * 2 : OOB;
        if (FUSE) tab_base = p.gn_table + (size_t)(img < a_imgs ? img : 0) * 2 * C + (tid & 7) * 4;
    };

    hu32x4 ra[HAJ], rbh, rbl;
    const int nslab = C / 32;
    // fused GroupNorm: scale / shift of the thread's four channels in the slab whose pieces are stored next
    hf32x4 sc4 = {0.f, 0.f, 0.f, 0.f}, sh4 = {0.f, 0.f, 0.f, 0.f};
    auto load_tab = [&](int slab) {
        if (FUSE && slab < nslab) {
            sc4 = *reinterpret_cast<const hf32x4 *>(tab_base + slab * 32);
            sh4 = *reinterpret_cast<const hf32x4 *>(tab_base + C + slab * 32);
        }
    };
    auto load_halo = [&](int j, int slab) {
        const bool ok = slab < nslab;
        if (FUSE && slab * 32 >= p.C0)  // second source of the concat (wave-uniform branch)
            ra[j] = __builtin_amdgcn_raw_buffer_load_b128(srd_a1, ok ? avo1[j] : OOB, (slab * 32 - p.C0) * 4, 0);
        else
            ra[j] = __builtin_amdgcn_raw_buffer_load_b128(srd_a, ok ? avo[j] : OOB, slab * (FUSE ? 128 : 64), 0);
    };
    auto store_halo = [&](int j, __bf16 *st) {
        if (tid + NT * j >= HP * 8) return;
        if (!FUSE) {
            *reinterpret_cast<hu32x4 *>(st + alds[j]) = ra[j];
            return;
        }
        typedef __bf16 hbf16x4 __attribute__((ext_vector_type(4)));
        const hf32x4 v = __builtin_bit_cast(hf32x4, ra[j]);
        hbf16x4 h, l;
        const bool inside = avo[j] != OOB;  // zero padding applies to the ACTIVATED tensor: outside pixels stay exactly zero
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            float y = v[i] * sc4[i] + sh4[i];
            y = p.gn_act ? swish_e(y) : y;
            y = inside ? y : 0.f;
            h[i] = (__bf16)y;
            l[i] = (__bf16)(y - (float)h[i]);
        }
        *reinterpret_cast<hbf16x4 *>(st + alds[j]) = h;
        *reinterpret_cast<hbf16x4 *>(st + HA_PLANE + alds[j]) = l;
    };
    auto load_b = [&](int tap, int slab) {
        const int soff = (tap * C + slab * 32) * 2;
        const int vo = slab < nslab ? bvo : OOB;
        rbh = __builtin_amdgcn_raw_buffer_load_b128(srd_bh, vo, soff, 0);
        rbl = __builtin_amdgcn_raw_buffer_load_b128(srd_bl, vo, soff, 0);
    };
    auto store_b = [&](__bf16 *st) {
        if (brow < HBN_) {
            const int o = 2 * HA_PLANE + brow * HREC + bq;
            *reinterpret_cast<hu32x4 *>(st + o) = rbh;
            *reinterpret_cast<hu32x4 *>(st + HB_PLANE + o) = rbl;
        }
    };

    hf32x16 acc[RM][RN];
    const int li = lane & 31, lh = lane >> 5;
    // A fragment base per r tile: output pixel m_local = wm*64 + r*32 + li -> (ly, x) -> halo pixel (ly, x) for tap (0, 0)
    int a_off[RM];
#pragma unroll
    for (int r = 0; r < RM; ++r) {
        const int ml = wm * 64 + r * 32 + li;
        const int ly = ml / W, x = ml - ly * W;
        a_off[r] = (ly * HW2 + x) * HREC + lh * 8;
    }
    const int b_off = 2 * HA_PLANE + (wn * 64 + li) * HREC + lh * 8;

    auto issue_prologue = [&]() {  // first loads of the tile set up by setup(): affine map, halo of slab 0, weight chunk (0, 0)
        load_tab(0);
#pragma unroll
        for (int j = 0; j < HAJ; ++j) load_halo(j, 0);
        load_b(0, 0);
    };

    int id = blockIdx.x;
    {
        int mt, ntl;
        if (id >= total_ids || !tile_of(id, mt, ntl)) return;  // (padded ids of the XCD-aware mapping)
    }
    setup(id);
    issue_prologue();
    for (;;) {
        // ---- halo of slab 0 and weight chunk (tap 0, slab 0) into stage 0; request chunk (tap 1, slab 0) ----
#pragma unroll
        for (int j = 0; j < HAJ; ++j) store_halo(j, smem);
        store_b(smem);
        load_b(1, 0);
        __syncthreads();
#pragma unroll
        for (int r = 0; r < RM; ++r)
#pragma unroll
            for (int c = 0; c < RN; ++c)
#pragma unroll
                for (int v = 0; v < 16; ++v) acc[r][c][v] = 0.f;

        // Iteration (slab, tap) reads A from halo stage (slab & 1) and B from weight stage ((slab + tap) & 1) (9 taps: the parity of
        // slab*9 + tap); during it the weight chunk of the next iteration (already in registers) is stored to the other B stage and the
        // one after that is requested; the halo of slab + 1 is requested during taps 0..HAJ-1 (one piece per tap) and stored into the
        // other halo stage during taps 9-HAJ..8.  The tap loop is unrolled so that every register-array index is a constant.
        for (int slab = 0; slab < nslab; ++slab) {
            const __bf16 *curA = smem + (slab & 1) * HSTAGE;
            __bf16 *nxtA = smem + ((slab + 1) & 1) * HSTAGE;
    #pragma unroll
            for (int tap = 0; tap < 9; ++tap) {
                const __bf16 *curB = smem + ((slab + tap) & 1) * HSTAGE;
                __bf16 *nxtB = smem + ((slab + tap + 1) & 1) * HSTAGE;
                const int ty = tap / 3, tx = tap - ty * 3;
                const int tap_off = (ty * HW2 + tx) * HREC;
                const int tap2 = (tap + 2) % 9, slab2 = slab + (tap + 2) / 9;  // the chunk two iterations ahead
    #pragma unroll
                for (int s = 0; s < 2; ++s) {
                    hbf16x8 ah[RM], al[RM], bh[RN], bl[RN];
    #pragma unroll
                    for (int r = 0; r < RM; ++r) {
                        ah[r] = *reinterpret_cast<const hbf16x8 *>(curA + a_off[r] + tap_off + s * 16);
                        al[r] = *reinterpret_cast<const hbf16x8 *>(curA + HA_PLANE + a_off[r] + tap_off + s * 16);
                    }
    #pragma unroll
                    for (int c = 0; c < RN; ++c) {
                        bh[c] = *reinterpret_cast<const hbf16x8 *>(curB + b_off + c * 32 * HREC + s * 16);
                        bl[c] = *reinterpret_cast<const hbf16x8 *>(curB + HB_PLANE + b_off + c * 32 * HREC + s * 16);
                    }
    #pragma unroll
                    for (int term = 0; term < 3; ++term) {
    #pragma unroll
                        for (int r = 0; r < RM; ++r)
    #pragma unroll
                            for (int c = 0; c < RN; ++c) {
                                const hbf16x8 a = term == 2 ? al[r] : ah[r];
                                const hbf16x8 b = term == 1 ? bl[c] : bh[c];
                                acc[r][c] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[r][c], 0, 0, 0);
                            }
                        __builtin_amdgcn_sched_barrier(0);
                        const int gap = s * 3 + term;
                        if (gap == 0) {
                            store_b(nxtB);
                            load_b(tap2, slab2);
                        } else if (gap == 1) {
                            if (tap == 0) load_tab(slab + 1);  // first in the queue: the map of the slab whose pieces follow
                            if (tap < HAJ) load_halo(tap, slab + 1);
                        } else if (gap == 2) {
                            if (tap >= 9 - HAJ) store_halo(tap - (9 - HAJ), nxtA);
                        }
                        __builtin_amdgcn_sched_barrier(0);
                    }
                }
                __syncthreads();
            }
        }


        // ---- next tile: its first loads go out now, the epilogue below covers their latency ----
        const int done_m0 = m0, done_n0 = n0;
        int next = id + (int)gridDim.x;
        bool more = false;
        if (p.diag != 1) {
            int mt, ntl;
            while (next < total_ids && !tile_of(next, mt, ntl)) next += (int)gridDim.x;
            more = next < total_ids;
        }
        if (more) {
            setup(next);
            issue_prologue();
        }
        epilogue_rows<RM, RN>(p, acc, done_m0 + wm * 64, done_n0 + wn * 64, li, lh, false);
        if (!more) break;
        id = next;
    }
}

// ---- host side ---------------------------------------------------------------------------------------------------------------
static bool halo_shape_ok(const GemmParams &p);

// fused-GroupNorm variant: same shapes, raw fp32 input (GDDIM_X3_NOFUSE=1 keeps the separate gn_apply pass)
bool conv3x3_halo_fused_ok(const GemmParams &p) {
    const int off = opts().x3_nohalo | opts().x3_nofuse;
    // every N tile of a row block stages (and therefore normalises) the halo again; measured (tools/variants_x3.sh): fusing the
    // 256-output layers too still wins (+3.2 % all fused, +2.7 % only N <= 128), so there is no limit by default
    const int max_n = opts().x3_fuse_maxn;
    if (off || !p.b_hi || !p.b_lo || p.N > max_n) return false;
    GemmParams q = p;
    q.a_presplit = 0;  // igemm_bf16x3_ok then applies the concat-aware addressing limits of a 4-byte-per-element source
    if (!q.gn_table) q.gn_table = reinterpret_cast<const float *>(&q);  // shape test only: any non-null marker
    return halo_shape_ok(q);
}

bool conv3x3_halo_ok(const GemmParams &p) {
    if (opts().x3_nohalo || !p.a_presplit || !p.b_hi || !p.b_lo) return false;
    return halo_shape_ok(p);
}

static bool halo_shape_ok(const GemmParams &p) {
    const bool two = p.gn_table && p.C1 > 0;  // the fused variant may read a concat of two raw tensors
    if (p.KH != 3 || p.KW != 3 || p.stride != 1 || p.pad != 1 || p.C0 % 32 != 0 || p.lda0 != p.C0) return false;
    if (two ? (p.C1 % 32 != 0 || p.lda1 != p.C1 || !p.a1) : p.C1 != 0) return false;
    if (p.H != p.W || p.OH != p.H || p.OW != p.W || (p.W != 32 && p.W != 16)) return false;
    if ((p.H * p.W) % HBM_ != 0 || p.M % HBM_ != 0 || p.N <= 64) return false;
    const long long blocks = (long long)(p.M / HBM_) * ((p.N + HBN_ - 1) / HBN_);
    return blocks >= 256 && igemm_bf16x3_ok(p, 1);
}

const char *conv3x3_halo_name() { return "conv3x3_halo_bf16x3_kernel"; }
const char *conv3x3_halo_fused_name() { return "conv3x3_halo_bf16x3_kernel<gn>"; }

static void hfill_div(unsigned d, unsigned &magic, unsigned &shift) {
    unsigned s = 0;
    while ((1ull << s) < d) ++s;
    shift = s;
    magic = (unsigned)((((1ull << 32) * ((1ull << s) - d)) / d) + 1);
}

int launch_conv3x3_halo(const GemmParams &p_in, hipStream_t s) {
    GemmParams p = p_in;
    GD_CHECK(p.gn_table ? conv3x3_halo_fused_ok(p) : conv3x3_halo_ok(p), "conv3x3_halo: problem not eligible");
    hfill_div((unsigned)(p.rows_per_image > 0 ? p.rows_per_image : 1), p.mg_rpi, p.sh_rpi);
    p.m_tiles = p.M / HBM_;
    p.n_tiles = (p.N + HBN_ - 1) / HBN_;
    p.flat_tiles = p.m_tiles < 16;
    const int total = p.flat_tiles ? p.m_tiles * p.n_tiles : ((p.m_tiles + 7) / 8) * 8 * p.n_tiles;
    const int one_per_tile = opts().x3_nopersist;  // A/B only
    static int cus = 0;
    if (!cus) {
        int dev = 0;
        (void)hipGetDevice(&dev);
        if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus <= 0) cus = 256;
        cus = (cus / 8) * 8;  // one workgroup per CU (150 KB of LDS each); a multiple of 8 keeps a tile id on its XCD
    }
    p.diag = one_per_tile ? 1 : 0;
    const int gx = one_per_tile ? total : (total < cus ? total : cus);
    if (p.gn_table) hipLaunchKernelGGL(conv3x3_halo_bf16x3_kernel<true>, dim3(gx), dim3(512), 0, s, p);
    else hipLaunchKernelGGL(conv3x3_halo_bf16x3_kernel<false>, dim3(gx), dim3(512), 0, s, p);
    GD_HIP(hipGetLastError());
    return 0;
}

}  // namespace gd
