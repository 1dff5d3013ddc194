// Fused GroupNorm statistics: called by the GEMM epilogues once per 32x32 accumulator tile with the lane's 16 final
// output values (row (v&3) + 8*(v>>2) + 4*(lane>>5) of the tile, column n = lane&31 of the tile).  Writes
// qstats[img][slice][n/4] = (sum, sum of squares) over the tile's rows for each group of four consecutive columns.
// One writer per entry (deterministic, no atomics).  Requires rows_per_image % 32 == 0 (one slice per 32 rows) or
// rows_per_image == 16 (the tile holds two images, one slice each); the planner guarantees it.
#pragma once
#include "../common.h"

namespace gd {

// qbase: p.qstats advanced to the z slice of a batched product (p.sQ floats per slice); null = p.qstats
__device__ __forceinline__ void emit_qstats(const GemmParams &p, const float (&vals)[16], int mrow, int n, bool nok, int li,
                                            int lh, float *qbase = nullptr) {
    float *const qs = qbase ? qbase : p.qstats;
    float s0 = 0.f, q0 = 0.f, s1 = 0.f, q1 = 0.f;
#pragma unroll
    for (int v = 0; v < 8; ++v) { s0 += vals[v]; q0 += vals[v] * vals[v]; }
#pragma unroll
    for (int v = 8; v < 16; ++v) { s1 += vals[v]; q1 += vals[v] * vals[v]; }
    // four consecutive columns = four consecutive lanes; rows +4 live in the other half-wave
    s0 += __shfl_xor(s0, 1); q0 += __shfl_xor(q0, 1); s1 += __shfl_xor(s1, 1); q1 += __shfl_xor(q1, 1);
    s0 += __shfl_xor(s0, 2); q0 += __shfl_xor(q0, 2); s1 += __shfl_xor(s1, 2); q1 += __shfl_xor(q1, 2);
    s0 += __shfl_xor(s0, 32); q0 += __shfl_xor(q0, 32); s1 += __shfl_xor(s1, 32); q1 += __shfl_xor(q1, 32);
    if ((li & 3) == 0 && lh == 0 && nok && mrow < p.M) {
        const int Q = p.N >> 2;
        const int rpi = p.rows_per_image;
        if (rpi == 16) {
            const int img = mrow >> 4;
            float2 *o = reinterpret_cast<float2 *>(qs) + (size_t)img * Q + (n >> 2);
            o[0] = make_float2(s0, q0);
            if (mrow + 16 < p.M) o[Q] = make_float2(s1, q1);
        } else {
            const int img = mrow / rpi;
            const int sl = (mrow - img * rpi) >> 5;
            float2 *o = reinterpret_cast<float2 *>(qs) + ((size_t)img * p.qs_slices + sl) * Q + (n >> 2);
            o[0] = make_float2(s0 + s1, q0 + q1);
        }
    }
}

}  // namespace gd
