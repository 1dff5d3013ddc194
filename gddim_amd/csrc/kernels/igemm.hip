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
// The next chunk's global loads are issued before the current chunk's MFMAs (register staging),
// so HBM/L2 latency hides under ~64 MFMAs.
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

template <int BM, int BN>
__global__ __launch_bounds__(256, 2) void igemm_f32_kernel(const GemmParams p) {
    constexpr int RM = BM / 64, RN = BN / 64;  // 32x32 MFMA tiles per wave
    constexpr int AJ = BM / 32, BJ = BN / 32;  // float4 staged per thread (A / B)
    __shared__ __attribute__((aligned(16))) float smem[(BM + BN) * LDT];
    float *As = smem;
    float *Bs = smem + BM * LDT;

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
                a_pix[j] = -1;
                a_iy0[j] = 0;
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

    f32x4 ra[AJ], rb[BJ];

    auto load_tiles = [&](int kc) {
        const int kabs = kc * KC + kcol;
        // ---- A ----
        {
            const bool kvalid = kabs < p.K;
            const float *src;
            int ld, cc;
            if (c_run < p.C0) { src = a0; ld = p.lda0; cc = c_run; }
            else              { src = a1; ld = p.lda1; cc = c_run - p.C0; }
#pragma unroll
            for (int j = 0; j < AJ; ++j) {
                const int iy = a_iy0[j] + dy, ix = a_ix0[j] + dx;
                const bool ok = kvalid && a_pix[j] >= 0 && (unsigned)iy < (unsigned)p.H && (unsigned)ix < (unsigned)p.W;
                f32x4 v = {0.f, 0.f, 0.f, 0.f};
                if (ok) v = *reinterpret_cast<const f32x4 *>(src + (size_t)(a_pix[j] + iy * p.W + ix) * ld + cc);
                if (p.in_act) { v[0] = swish_f(v[0]); v[1] = swish_f(v[1]); v[2] = swish_f(v[2]); v[3] = swish_f(v[3]); }
                ra[j] = v;
            }
        }
        // ---- B ----
        if (!p.b_kmajor) {
#pragma unroll
            for (int j = 0; j < BJ; ++j) {
                const int n = n0 + arow + 32 * j;
                f32x4 v = {0.f, 0.f, 0.f, 0.f};
                if (n < p.N && kabs < p.K) v = *reinterpret_cast<const f32x4 *>(bptr + (size_t)n * p.ldb + kabs);
                rb[j] = v;
            }
        } else {
            const int kk = kc * KC + (tid & 31);
#pragma unroll
            for (int j = 0; j < BJ; ++j) {
                const int n = n0 + ((tid >> 5) + 8 * j) * 4;
                f32x4 v = {0.f, 0.f, 0.f, 0.f};
                if (n < p.N && kk < p.K) v = *reinterpret_cast<const f32x4 *>(bptr + (size_t)kk * p.ldb + n);
                rb[j] = v;
            }
        }
        // advance (dy, dx, c) to the next chunk
        c_run += KC;
        while (c_run >= Cin) {
            c_run -= Cin;
            if (++dx == p.KW) { dx = 0; ++dy; }
        }
    };

    auto store_tiles = [&]() {
#pragma unroll
        for (int j = 0; j < AJ; ++j) *reinterpret_cast<f32x4 *>(&As[(arow + 32 * j) * LDT + kcol]) = ra[j];
        if (!p.b_kmajor) {
#pragma unroll
            for (int j = 0; j < BJ; ++j) *reinterpret_cast<f32x4 *>(&Bs[(arow + 32 * j) * LDT + kcol]) = rb[j];
        } else {
            const int kk = tid & 31;
#pragma unroll
            for (int j = 0; j < BJ; ++j) {
                const int nl = ((tid >> 5) + 8 * j) * 4;
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
    const float *a_frag = As + (wm * (BM / 2) + li) * LDT + lh * 4;
    const float *b_frag = Bs + (wn * (BN / 2) + li) * LDT + lh * 4;

    load_tiles(0);
    store_tiles();
    __syncthreads();
    for (int kc = 0; kc < nk; ++kc) {
        if (kc + 1 < nk) load_tiles(kc + 1);  // global loads in flight under the MFMAs below
#pragma unroll
        for (int kb = 0; kb < KC / 8; ++kb) {
            f32x4 af[RM], bf[RN];
#pragma unroll
            for (int r = 0; r < RM; ++r) af[r] = *reinterpret_cast<const f32x4 *>(a_frag + r * 32 * LDT + kb * 8);
#pragma unroll
            for (int c = 0; c < RN; ++c) bf[c] = *reinterpret_cast<const f32x4 *>(b_frag + c * 32 * LDT + kb * 8);
#pragma unroll
            for (int jj = 0; jj < 4; ++jj)
#pragma unroll
                for (int r = 0; r < RM; ++r)
#pragma unroll
                    for (int c = 0; c < RN; ++c)
                        acc[r][c] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[r][jj], bf[c][jj], acc[r][c], 0, 0, 0);
        }
        if (kc + 1 < nk) {
            __syncthreads();  // every wave finished reading this chunk
            store_tiles();
            __syncthreads();
        }
    }

    // ---- epilogue: D[row][col], col = lane&31, row = (v&3) + 8*(v>>2) + 4*(lane>>5) ----
    float *outp = p.out + (size_t)z * p.sC;
#pragma unroll
    for (int c = 0; c < RN; ++c) {
        const int n = n0 + wn * (BN / 2) + c * 32 + li;
        const bool nok = n < p.N;
        const float bias = (p.bias && nok) ? p.bias[n] : 0.f;
#pragma unroll
        for (int r = 0; r < RM; ++r) {
#pragma unroll
            for (int v = 0; v < 16; ++v) {
                const int m = m0 + wm * (BM / 2) + r * 32 + (v & 3) + 8 * (v >> 2) + 4 * lh;
                if (nok && m < p.M) {
                    float val = p.alpha * acc[r][c][v] + bias;
                    if (p.rowbias) val += p.rowbias[(size_t)(m / p.rows_per_image) * p.ld_rowbias + n];
                    if (p.res) val += p.res[(size_t)m * p.ldres + n];
                    outp[(size_t)m * p.ldc + n] = val * p.out_scale;
                }
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
    if (bm == 128 && bn == 128) hipLaunchKernelGGL((igemm_f32_kernel<128, 128>), grid, block, 0, s, p);
    else if (bm == 128) hipLaunchKernelGGL((igemm_f32_kernel<128, 64>), grid, block, 0, s, p);
    else hipLaunchKernelGGL((igemm_f32_kernel<64, 64>), grid, block, 0, s, p);
    GD_HIP(hipGetLastError());
    return 0;
}

}  // namespace gd
