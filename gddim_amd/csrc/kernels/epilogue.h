// Epilogue shared by the split-precision GEMM kernels: D[row][col] of a 32x32 accumulator, col = lane & 31, row = (v & 3) + 8 (v >> 2) +
// 4 (lane >> 5).  Buffer addressed: the lane part of the offset is computed once per column tile, the row part is a scalar soffset,
// rows >= M fall outside the descriptor (dropped by the bounds check), columns >= N are pushed out of bounds once.  Adds bias, the
// per-image bias row (rows_per_image is a multiple of 8), the residual, scales, stores, and emits the fused GroupNorm statistics.
#pragma once
#ifndef GD_EPI_AUX
#define GD_EPI_AUX 0  // cache policy of the shared GEMM epilogue's output stores (A/B, round 5: 2 = nt)
#endif
#include "../common.h"
#include "qstats.h"

namespace gd {

typedef float epi_f32x16 __attribute__((ext_vector_type(16)));

__device__ __forceinline__ unsigned epi_div(unsigned n, unsigned d, unsigned magic, unsigned shift) {
    if (d == 1) return n;
    const unsigned t = __umulhi(magic, n);
    return (t + ((n - t) >> 1)) >> (shift - 1);
}

// mbase / nbase: first row / column of the wave's (RM x RN) x 32 tile block (wave-uniform)
template <int RM, int RN>
__device__ __forceinline__ void epilogue_rows(const GemmParams &p, const epi_f32x16 (&acc)[RM][RN], int mbase, int nbase, int li,
                                              int lh, bool no_store, float *out_base = nullptr, int z = 0) {
    constexpr int OOB = (int)0x80000000u;
    char *outc = reinterpret_cast<char *>(out_base ? out_base : p.out);  // out_base: the z slice of a batched product
    const __amdgpu_buffer_rsrc_t srd_o = __builtin_amdgcn_make_buffer_rsrc(outc, 0, p.M * p.ldc * 4, 0x00020000);
    const __amdgpu_buffer_rsrc_t srd_r =
        __builtin_amdgcn_make_buffer_rsrc(const_cast<char *>(reinterpret_cast<const char *>(p.res ? p.res + (size_t)z * p.sR : p.out)), 0,
                                          p.M * p.ldres * 4, 0x00020000);
    const unsigned rpi = p.rows_per_image;
    // all residual values of the wave's tile are requested at once (RM*RN*16 loads in flight) before any is consumed
    float resv[RN][RM][16];
    if (p.res) {
#pragma unroll
        for (int c = 0; c < RN; ++c) {
            const int n = nbase + c * 32 + li;
            const int vo_r = n < p.N ? (4 * lh * p.ldres + n) * 4 : OOB;
#pragma unroll
            for (int r = 0; r < RM; ++r) {
                const int mrow = mbase + r * 32;
#pragma unroll
                for (int v = 0; v < 16; ++v)
                    resv[c][r][v] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(srd_r, vo_r, (mrow + (v & 3) + 8 * (v >> 2)) * p.ldres * 4, 0));
            }
        }
        __builtin_amdgcn_sched_barrier(0);
    }
#pragma unroll
    for (int c = 0; c < RN; ++c) {
        const int n = nbase + c * 32 + li;
        const bool nok = n < p.N;
        const float bias = (p.bias && nok) ? p.bias[n] : 0.f;
        const int vo_o = nok ? (4 * lh * p.ldc + n) * 4 : OOB;
#pragma unroll
        for (int r = 0; r < RM; ++r) {
            const int mrow = mbase + r * 32;
            float addv[16];
#pragma unroll
            for (int v = 0; v < 16; ++v) addv[v] = bias;
            if (p.res) {
#pragma unroll
                for (int v = 0; v < 16; ++v) addv[v] += resv[c][r][v];
            }
            if (p.rowbias) {
                // rows_per_image is a multiple of 8: the 8 rows (v & 3) + 4*lh of a row group v >> 2 belong to one image
                // and that image index is wave-uniform -> four scalar divisions and four loads per tile
                float t[4];
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const unsigned mr = (unsigned)(mrow + 8 * g);
                    const unsigned img = mr < (unsigned)p.M ? epi_div(mr, rpi, p.mg_rpi, p.sh_rpi) : 0u;
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
                if (no_store) { asm volatile("" ::"v"(val)); continue; }
                __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, val), srd_o, vo_o,
                                                      (mrow + (v & 3) + 8 * (v >> 2)) * p.ldc * 4, GD_EPI_AUX);
            }
            if (p.qstats) {
                if (!nok || mrow + 32 > p.M) {
#pragma unroll
                    for (int v = 0; v < 16; ++v)
                        if (!nok || mrow + (v & 3) + 8 * (v >> 2) + 4 * lh >= p.M) vals[v] = 0.f;
                }
                emit_qstats(p, vals, mrow, n, nok, li, lh, p.qstats + (size_t)z * p.sQ);
            }
        }
    }
}

}  // namespace gd
