// Counter-based Gaussian noise for the stochastic CLD samplers: Philox4x32-10 (Salmon et al., SC'11) keyed by the call's seed,
// counter = (element index lo, hi, draw index, 0); words 0 and 1 feed one Box-Muller pair = the (x, v) noise of that element.
// Stateless: the value depends only on (seed, element, draw), not on launch geometry, batch split or replay order, and the
// oracle restates it bit for bit on the integer side (oracle/rng.py).  The reference draws jax.random.normal /
// multivariate_normal (threefry) at these points (cld_jax/sampling.py:399,559,648); that stream cannot be reproduced.
#pragma once
#include <hip/hip_runtime.h>

namespace gd {

struct Normal2 { float a, b; };

__device__ __forceinline__ void philox4x32_10(unsigned c0, unsigned c1, unsigned c2, unsigned c3, unsigned k0, unsigned k1,
                                               unsigned &o0, unsigned &o1) {
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        const unsigned hi0 = __umulhi(0xD2511F53u, c0), lo0 = 0xD2511F53u * c0;
        const unsigned hi1 = __umulhi(0xCD9E8D57u, c2), lo1 = 0xCD9E8D57u * c2;
        const unsigned n0 = hi1 ^ c1 ^ k0, n2 = hi0 ^ c3 ^ k1;
        c0 = n0; c1 = lo1; c2 = n2; c3 = lo0;
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
    o0 = c0; o1 = c1;
}

__device__ __forceinline__ Normal2 normal_pair(const unsigned *__restrict__ seed, unsigned long long elem, unsigned draw) {
    unsigned r0, r1;
    philox4x32_10((unsigned)elem, (unsigned)(elem >> 32), draw, 0u, seed[0], seed[1], r0, r1);
    const float u1 = ((float)(r0 >> 8) + 0.5f) * 5.9604644775390625e-08f;  // (0, 1), 2^-24 grid
    const float u2 = ((float)(r1 >> 8) + 0.5f) * 5.9604644775390625e-08f;
    const float rad = sqrtf(-2.0f * logf(u1));
    const float ang = 6.283185307179586f * u2;
    return {rad * cosf(ang), rad * sinf(ang)};
}

}  // namespace gd
