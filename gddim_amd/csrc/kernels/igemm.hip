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
// The next chunk's global loads are issued before the current chunk's MFMAs (register staging) and are
// branch-free (masked lanes read the tensor base; zeros are substituted when the registers go to LDS), so
// HBM/L2 latency hides under ~64 MFMAs.  LDS is double buffered: one barrier per K chunk.
//
// Block -> tile mapping is XCD-aware: consecutive block ids round-robin over the 8 XCDs, so ids
// are grouped in runs of 8*n_tiles with the M tile varying fastest inside a run; the n_tiles
// blocks that share one A tile then carry ids congruent mod 8 and run on the same XCD (shared L2).
#include "../common.h"

namespace gd {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int KC = 32;   // K chunk
constexpr int LDT = 36;  // LDS row stride (floats): 32 + 4 pad -> 144 B, odd multiple of 16 B

__device__ __forceinline__ float swish_f(float x) { return x * __frcp_rn(1.0f + __expf(-x)); }

template <int BM, int BN, bool B_KMAJOR, bool IN_ACT>
__global__ __launch_bounds__(256, 2) void igemm_f32_kernel(const GemmParams p) {
    constexpr int RM = BM / 64, RN = BN / 64;  // 32x32 MFMA tiles per wave
    constexpr int AJ = BM / 32, BJ = BN / 32;  // float4 staged per thread (A / B)
    constexpr int BUF = (BM + BN) * LDT;       // floats per LDS stage
    __shared__ __attribute__((aligned(16))) float smem[2 * BUF];  // double buffered: one barrier per K chunk

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;

    // ---- XCD-aware block -> (mtile, ntile) ----
    const int nt = p.n_tiles;
    const int id = blockIdx.x;
    const int grp = id / (8 * nt);
    const int rem = id - grp * 8 * nt;
    const int ntile = rem >> 3;
    const int mtile = grp * 8 + (rem & 7);
    if (mtile >= p.m_tiles) return;
    const int m0 = mtile * BM, n0 = ntile * BN;
    const int z = blockIdx.z;

    const float *a0 = p.a0 + (size_t)z * p.sA;
    const float *a1 = p.a1;
    const float *bptr = p.b + (size_t)z * p.sB;

    // ---- per-thread A row bookkeeping (fixed over the K loop) ----
    const int arow = tid >> 3;        // 0..31, +32 per j
    const int kcol = (tid & 7) * 4;   // float offset inside the chunk
    int a_pix[AJ], a_iy0[AJ], a_ix0[AJ];
    {
        const int OHW = p.OH * p.OW;
#pragma unroll
        for (int j = 0; j < AJ; ++j) {
            const int m = m0 + arow + 32 * j;
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
                a_iy0[j] = -(1 << 28);  // always out of the image: masked
                a_ix0[j] = 0;
            }
        }
    }
    const int Cin = p.C0 + p.C1;
    // running decomposition of this thread's k = kc*32 + kcol into (dy, dx, c)
    int c_run = kcol, dy = 0, dx = 0;
    while (c_run >= Cin) {
        c_run -= Cin;
        if (++dx == p.KW) { dx = 0; ++dy; }
    }

    // Staging registers.  Loads are unconditional (masked lanes read the tensor base) so the K loop has no
    // divergent branches and no wait sits between the global loads and the MFMAs; the zero fill (image border,
    // ragged M / N / K) is applied from the mask bits when the registers are written to LDS.
    f32x4 ra[AJ], rb[BJ];
    unsigned amask = 0, bmask = 0;

    auto load_tiles = [&](int kc) {
        const int kabs = kc * KC + kcol;
        {
            const bool kvalid = kabs < p.K;
            const bool first = c_run < p.C0;
            const float *src = first ? a0 : a1;
            const int ld = first ? p.lda0 : p.lda1;
            const int cc = first ? c_run : c_run - p.C0;
            amask = 0;
#pragma unroll
            for (int j = 0; j < AJ; ++j) {
                const int iy = a_iy0[j] + dy, ix = a_ix0[j] + dx;
                const bool ok = kvalid && (unsigned)iy < (unsigned)p.H && (unsigned)ix < (unsigned)p.W;
                const float *ptr = ok ? src + (size_t)(a_pix[j] + iy * p.W + ix) * ld + cc : a0;
                ra[j] = *reinterpret_cast<const f32x4 *>(ptr);
                amask |= (ok ? 1u : 0u) << j;
            }
        }
        bmask = 0;
        if (!B_KMAJOR) {
#pragma unroll
            for (int j = 0; j < BJ; ++j) {
                const int n = n0 + arow + 32 * j;
                const bool ok = n < p.N && kabs < p.K;
                const float *ptr = ok ? bptr + (size_t)n * p.ldb + kabs : bptr;
                rb[j] = *reinterpret_cast<const f32x4 *>(ptr);
                bmask |= (ok ? 1u : 0u) << j;
            }
        } else {
            const int kk = kc * KC + (tid & 31);
#pragma unroll
            for (int j = 0; j < BJ; ++j) {
                const int n = n0 + ((tid >> 5) + 8 * j) * 4;
                const bool ok = n < p.N && kk < p.K;
                const float *ptr = ok ? bptr + (size_t)kk * p.ldb + n : bptr;
                rb[j] = *reinterpret_cast<const f32x4 *>(ptr);
                bmask |= (ok ? 1u : 0u) << j;
            }
        }
        // advance (dy, dx, c) to the next chunk
        c_run += KC;
        while (c_run >= Cin) {
            c_run -= Cin;
            if (++dx == p.KW) { dx = 0; ++dy; }
        }
    };

    auto store_tiles = [&](float *As, float *Bs) {
        const f32x4 zero = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int j = 0; j < AJ; ++j) {
            f32x4 v = ((amask >> j) & 1u) ? ra[j] : zero;
            if (IN_ACT) { v[0] = swish_f(v[0]); v[1] = swish_f(v[1]); v[2] = swish_f(v[2]); v[3] = swish_f(v[3]); }
            *reinterpret_cast<f32x4 *>(&As[(arow + 32 * j) * LDT + kcol]) = v;
        }
        if (!B_KMAJOR) {
#pragma unroll
            for (int j = 0; j < BJ; ++j)
                *reinterpret_cast<f32x4 *>(&Bs[(arow + 32 * j) * LDT + kcol]) = ((bmask >> j) & 1u) ? rb[j] : zero;
        } else {
            const int kk = tid & 31;
#pragma unroll
            for (int j = 0; j < BJ; ++j) {
                const f32x4 v = ((bmask >> j) & 1u) ? rb[j] : zero;
                const int nl = ((tid >> 5) + 8 * j) * 4;
                Bs[(nl + 0) * LDT + kk] = v[0];
                Bs[(nl + 1) * LDT + kk] = v[1];
                Bs[(nl + 2) * LDT + kk] = v[2];
                Bs[(nl + 3) * LDT + kk] = v[3];
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
    const int a_off = (wm * (BM / 2) + li) * LDT + lh * 4;
    const int b_off = BM * LDT + (wn * (BN / 2) + li) * LDT + lh * 4;

    load_tiles(0);
    store_tiles(smem, smem + BM * LDT);
    __syncthreads();
    for (int kc = 0; kc < nk; ++kc) {
        const float *cur = smem + (kc & 1) * BUF;
        float *nxt = smem + ((kc + 1) & 1) * BUF;
        const bool more = kc + 1 < nk;
        if (more) load_tiles(kc + 1);  // global loads stay in flight under the MFMAs below
#pragma unroll
        for (int kb = 0; kb < KC / 8; ++kb) {
            f32x4 af[RM], bf[RN];
#pragma unroll
            for (int r = 0; r < RM; ++r) af[r] = *reinterpret_cast<const f32x4 *>(cur + a_off + r * 32 * LDT + kb * 8);
#pragma unroll
            for (int c = 0; c < RN; ++c) bf[c] = *reinterpret_cast<const f32x4 *>(cur + b_off + c * 32 * LDT + kb * 8);
#pragma unroll
            for (int jj = 0; jj < 4; ++jj)
#pragma unroll
                for (int r = 0; r < RM; ++r)
#pragma unroll
                    for (int c = 0; c < RN; ++c)
                        acc[r][c] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[r][jj], bf[c][jj], acc[r][c], 0, 0, 0);
        }
        // the other stage was last read in iteration kc-1, and every wave has passed that iteration's barrier
        if (more) store_tiles(nxt, nxt + BM * LDT);
        __syncthreads();
    }

    // ---- epilogue: D[row][col], col = lane&31, row = (v&3) + 8*(v>>2) + 4*(lane>>5) ----
    // Residual / per-image bias loads are issued 16 at a time (one 32x32 tile) before any is consumed, so the
    // epilogue pays one memory round trip per tile instead of one per element.
    float *outp = p.out + (size_t)z * p.sC;
    const int rpi = p.rows_per_image;
#pragma unroll
    for (int c = 0; c < RN; ++c) {
        const int n = n0 + wn * (BN / 2) + c * 32 + li;
        const bool nok = n < p.N;
        const float bias = (p.bias && nok) ? p.bias[n] : 0.f;
#pragma unroll
        for (int r = 0; r < RM; ++r) {
            const int mb = m0 + wm * (BM / 2) + r * 32 + 4 * lh;
            float addv[16];
#pragma unroll
            for (int v = 0; v < 16; ++v) addv[v] = bias;
            if (p.res) {
                float t[16];
#pragma unroll
                for (int v = 0; v < 16; ++v) {
                    const int m = mb + (v & 3) + 8 * (v >> 2);
                    const float *ptr = (nok && m < p.M) ? p.res + (size_t)m * p.ldres + n : p.res;
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
#pragma unroll
            for (int v = 0; v < 16; ++v) {
                const int m = mb + (v & 3) + 8 * (v >> 2);
                if (nok && m < p.M) outp[(size_t)m * p.ldc + n] = (p.alpha * acc[r][c][v] + addv[v]) * p.out_scale;
            }
        }
    }
}

static void pick_tiles(const GemmParams &p, int batch_z, int &bm, int &bn) {
    auto blocks = [&](int BM_, int BN_) {
        return (long long)((p.M + BM_ - 1) / BM_) * ((p.N + BN_ - 1) / BN_) * batch_z;
    };
    // Largest tile that still gives every CU of the 256 at least ~1.5 blocks; the small spatial
    // levels (8x8, 4x4) of the U-Net otherwise leave most of the chip idle.
    if (p.N > 64 && blocks(128, 128) >= 384) { bm = 128; bn = 128; return; }
    if (blocks(128, 64) >= 384) { bm = 128; bn = 64; return; }
    bm = 64; bn = 64;
}

const char *igemm_variant_name(const GemmParams &p, int batch_z) {
    int bm, bn;
    pick_tiles(p, batch_z, bm, bn);
    if (bm == 128 && bn == 128) return "igemm_f32<128,128>";
    if (bm == 128) return "igemm_f32<128,64>";
    return "igemm_f32<64,64>";
}

int launch_igemm(const GemmParams &p_in, int batch_z, hipStream_t s) {
    GemmParams p = p_in;
    GD_CHECK(p.C0 > 0 && p.M > 0 && p.N > 0, "igemm: empty problem (C0=%d M=%d N=%d)", p.C0, p.M, p.N);
    GD_CHECK(p.C0 % 4 == 0 && p.C1 % 4 == 0 && p.K % 4 == 0, "igemm: channel counts must be multiples of 4 (C0=%d C1=%d K=%d)",
             p.C0, p.C1, p.K);
    GD_CHECK(p.K == p.KH * p.KW * (p.C0 + p.C1), "igemm: K=%d != KH*KW*Cin=%d", p.K, p.KH * p.KW * (p.C0 + p.C1));
    GD_CHECK(!p.b_kmajor || p.N % 4 == 0, "igemm: k-major B needs N %% 4 == 0");
    GD_CHECK(p.ldb % 4 == 0 && p.lda0 % 4 == 0 && (p.C1 == 0 || p.lda1 % 4 == 0), "igemm: leading dims must be multiples of 4");
    int bm, bn;
    pick_tiles(p, batch_z, bm, bn);
    p.m_tiles = (p.M + bm - 1) / bm;
    p.n_tiles = (p.N + bn - 1) / bn;
    const int gx = ((p.m_tiles + 7) / 8) * 8 * p.n_tiles;
    dim3 grid(gx, 1, batch_z), block(256);
#define GD_LAUNCH(BM_, BN_)                                                                                         \
    do {                                                                                                            \
        if (p.b_kmajor) hipLaunchKernelGGL((igemm_f32_kernel<BM_, BN_, true, false>), grid, block, 0, s, p);        \
        else if (p.in_act) hipLaunchKernelGGL((igemm_f32_kernel<BM_, BN_, false, true>), grid, block, 0, s, p);     \
        else hipLaunchKernelGGL((igemm_f32_kernel<BM_, BN_, false, false>), grid, block, 0, s, p);                  \
    } while (0)
    GD_CHECK(!(p.b_kmajor && p.in_act), "igemm: in_act with k-major B is not instantiated");
    if (bm == 128 && bn == 128) GD_LAUNCH(128, 128);
    else if (bm == 128) GD_LAUNCH(128, 64);
    else GD_LAUNCH(64, 64);
#undef GD_LAUNCH
    GD_HIP(hipGetLastError());
    return 0;
}

}  // namespace gd
