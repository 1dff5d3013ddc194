// gDDIM update rules as fused per-pixel kernels (HBM-bound, <0.1 % of a sampler call).
//
// Reference semantics:
//   multistep_ab_step   u' = Psi u + sum_o C_o eps_o, history shift        cld_jax/deis.py:141-151
//   ab_step (scalar)    x' = c_0 x + sum_o c_{o+1} eps_o                    blur_jax/multistep.py:94-97
//   sbmm / bmm          per-pixel 2x2 mat-vec                                cld_jax/utils.py:179-185
//   denoising step      u + F u dt - G G score dt, score = -R^-T eps         cld_jax/sampling.py:30-39
//   output              x = inverse_scaler(u[...,0]), v = u[...,1]           cld_jax/sampling.py:228-229
//   pack / unpack       'b ... d g -> b ... (g d)'                           cld_jax/models/utils.py:153,158
//   blur order-0 step   y' = m(t') (y - s(t) e_y)/m(t) + s(t') e_y           blur_jax/sampling.py:62-75
//   2-D ortho DCT       batch_img_dct / batch_img_idct                       blur_jax/blur.py:99-106
//
// The fused CLD sampler keeps the state channel-planar, (pixels, CP) = [x_0..x_{C-1}, v_0..v_{C-1}, 0..],
// which IS the U-Net input, so no pack/unpack runs inside the NFE loop.  The epsilon history is a
// ring of n_hist slots in the U-Net output layout (pixels, 2C); the ring head and the coefficient
// row are derived from a device-side step counter so one captured hipGraph serves every step.
#include <cstdint>

#include "../common.h"
#include "rng.h"

namespace gd {

// ---- generic operators on (n, 2) pairs ---------------------------------------------------------------
__global__ __launch_bounds__(256) void ab_step_kernel(const float2 *__restrict__ u, const float *__restrict__ coef,
                                                       const float2 *__restrict__ new_eps,
                                                       const float2 *__restrict__ hist_in, float2 *__restrict__ u_out,
                                                       float2 *__restrict__ hist_out, int n_hist, long long n) {
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const float2 uv = u[i];
    const float2 e0 = new_eps[i];
    float ox = coef[0] * uv.x + coef[1] * uv.y;
    float oy = coef[2] * uv.x + coef[3] * uv.y;
    ox += coef[4] * e0.x + coef[5] * e0.y;
    oy += coef[6] * e0.x + coef[7] * e0.y;
    float2 prev = e0;
    for (int j = 0; j < n_hist; ++j) {
        const float2 h = hist_in[(size_t)j * n + i];
        const float *c = coef + 8 + 4 * j;
        ox += c[0] * h.x + c[1] * h.y;
        oy += c[2] * h.x + c[3] * h.y;
        hist_out[(size_t)j * n + i] = prev;  // full_eps[:-1]
        prev = h;
    }
    u_out[i] = make_float2(ox, oy);
}

int launch_ab_step(const float *u, const float *coef, const float *new_eps, const float *hist_in, float *u_out,
                   float *hist_out, int n_hist, long long n_pairs, hipStream_t s) {
    hipLaunchKernelGGL(ab_step_kernel, dim3((unsigned)((n_pairs + 255) / 256)), dim3(256), 0, s, (const float2 *)u, coef,
                       (const float2 *)new_eps, (const float2 *)hist_in, (float2 *)u_out, (float2 *)hist_out, n_hist,
                       n_pairs);
    GD_HIP(hipGetLastError());
    return 0;
}

__global__ __launch_bounds__(256) void ab_step_scalar_kernel(const float *__restrict__ x, const float *__restrict__ coef,
                                                              const float *__restrict__ new_eps,
                                                              const float *__restrict__ hist_in, float *__restrict__ x_out,
                                                              float *__restrict__ hist_out, int n_hist, long long n) {
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const float e0 = new_eps[i];
    float o = coef[0] * x[i] + coef[1] * e0;
    float prev = e0;
    for (int j = 0; j < n_hist; ++j) {
        const float h = hist_in[(size_t)j * n + i];
        o += coef[2 + j] * h;
        hist_out[(size_t)j * n + i] = prev;
        prev = h;
    }
    x_out[i] = o;
}

int launch_ab_step_scalar(const float *x, const float *coef, const float *new_eps, const float *hist_in, float *x_out,
                          float *hist_out, int n_hist, long long n, hipStream_t s) {
    hipLaunchKernelGGL(ab_step_scalar_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, x, coef, new_eps,
                       hist_in, x_out, hist_out, n_hist, n);
    GD_HIP(hipGetLastError());
    return 0;
}

__global__ __launch_bounds__(256) void bmm2x2_kernel(const float *__restrict__ m, const float2 *__restrict__ in,
                                                      float2 *__restrict__ out, long long n, int per_sample,
                                                      long long pps) {
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const float *mm = per_sample ? m + 4 * (i / pps) : m;
    const float2 v = in[i];
    out[i] = make_float2(mm[0] * v.x + mm[1] * v.y, mm[2] * v.x + mm[3] * v.y);
}

int launch_bmm2x2(const float *m, const float *in, float *out, long long n_pairs, int per_sample,
                  long long pairs_per_sample, hipStream_t s) {
    hipLaunchKernelGGL(bmm2x2_kernel, dim3((unsigned)((n_pairs + 255) / 256)), dim3(256), 0, s, m, (const float2 *)in,
                       (float2 *)out, n_pairs, per_sample, pairs_per_sample);
    GD_HIP(hipGetLastError());
    return 0;
}

// ---- CLD fused sampler ---------------------------------------------------------------------------------
// (pixels, C, 2) -> planar (pixels, CP) [+ every ring slot (pixels, 2C) initialised with the same values,
// cld_jax/sampling.py:225: eps_pred = stack([u] * (deis_order + 1))]
__global__ __launch_bounds__(256) void cld_pack_kernel(const float2 *__restrict__ u, float *__restrict__ u8,
                                                        float *__restrict__ ring, int ring_slots, long long pixels, int C,
                                                        int CP) {
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= pixels * C) return;
    const long long p = i / C;
    const int c = (int)(i - p * C);
    const float2 v = u[i];
    u8[p * CP + c] = v.x;
    u8[p * CP + C + c] = v.y;
    if (c == 0)
        for (int k = 2 * C; k < CP; ++k) u8[p * CP + k] = 0.f;
    const size_t slot = (size_t)pixels * 2 * C;
    for (int sidx = 0; sidx < ring_slots; ++sidx) {
        ring[sidx * slot + p * 2 * C + c] = v.x;
        ring[sidx * slot + p * 2 * C + C + c] = v.y;
    }
}

int launch_cld_pack(const float *u, float *u8, float *ring, int ring_slots, long long pixels, int C, hipStream_t s) {
    const int CP = (2 * C + 3) / 4 * 4;
    const long long n = pixels * C;
    hipLaunchKernelGGL(cld_pack_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, (const float2 *)u, u8, ring,
                       ring_slots, pixels, C, CP);
    GD_HIP(hipGetLastError());
    return 0;
}

__global__ __launch_bounds__(256) void cld_unpack_kernel(const float *__restrict__ planar, int ld,
                                                          float2 *__restrict__ pairs, long long pixels, int C) {
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= pixels * C) return;
    const long long p = i / C;
    const int c = (int)(i - p * C);
    pairs[i] = make_float2(planar[p * ld + c], planar[p * ld + C + c]);
}

int launch_cld_unpack_pairs(const float *planar, int ld, float *pairs, long long pixels, int C, hipStream_t s) {
    const long long n = pixels * C;
    hipLaunchKernelGGL(cld_unpack_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, planar, ld, (float2 *)pairs,
                       pixels, C);
    GD_HIP(hipGetLastError());
    return 0;
}

// One AB step in place.  coef row = coef_table[*step][n_slots][4]; ring logical index j lives in physical
// slot (head + j) % ring_slots with head = (-step) mod ring_slots; the new eps overwrites the oldest slot,
// which becomes the next step's head.
__global__ __launch_bounds__(256) void cld_ab_step_kernel(float *__restrict__ u8, const float *__restrict__ eps,
                                                           float *__restrict__ ring, int ring_slots,
                                                           const float *__restrict__ coef_table, int n_slots,
                                                           const int *__restrict__ step_ptr, long long pixels, int C,
                                                           int CP, const float *__restrict__ tf,
                                                           const float *__restrict__ nz, const unsigned *__restrict__ seed,
                                                           unsigned long long elem0) {
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= pixels * C) return;
    const long long p = i / C;
    const int c = (int)(i - p * C);
    const int step = *step_ptr;
    const float *coef = coef_table + (size_t)step * n_slots * 4;
    const float x = u8[p * CP + c], v = u8[p * CP + C + c];
    const size_t eo = (size_t)p * 2 * C + c;
    float ex = eps[eo], ev = eps[eo + C];
    if (tf) {  // eps <- E_i eps_net + U_i u  (L-parametrised / mixed-score / rotated-frame samplers; tf row = [P, E, U])
        const float *E = tf + (size_t)step * 12 + 4, *U = E + 4;
        const float nx = E[0] * ex + E[1] * ev + U[0] * x + U[1] * v;
        const float nv = E[2] * ex + E[3] * ev + U[2] * x + U[3] * v;
        ex = nx;
        ev = nv;
    }
    float ox = coef[0] * x + coef[1] * v;
    float ov = coef[2] * x + coef[3] * v;
    ox += coef[4] * ex + coef[5] * ev;
    ov += coef[6] * ex + coef[7] * ev;
    if (ring_slots > 0) {
        const size_t slot = (size_t)pixels * 2 * C;
        const int head = (ring_slots - step % ring_slots) % ring_slots;
        int phys = head;
        for (int j = 0; j < ring_slots; ++j) {
            const float hx = ring[phys * slot + eo], hv = ring[phys * slot + eo + C];
            const float *cc = coef + 8 + 4 * j;
            ox += cc[0] * hx + cc[1] * hv;
            ov += cc[2] * hx + cc[3] * hv;
            if (++phys == ring_slots) phys = 0;
        }
        const int nh = (head + ring_slots - 1) % ring_slots;
        ring[nh * slot + eo] = ex;
        ring[nh * slot + eo + C] = ev;
    }
    if (nz) {  // + S_i xi: the stochastic samplers' noise of this step (draw 2 i + 1; nz row = [M_pre, S_pre, S_post])
        const float *S = nz + (size_t)step * 12 + 8;
        const Normal2 xi = normal_pair(seed, elem0 + (unsigned long long)i, 2u * (unsigned)step + 1u);
        ox += S[0] * xi.a + S[1] * xi.b;
        ov += S[2] * xi.a + S[3] * xi.b;
    }
    u8[p * CP + c] = ox;
    u8[p * CP + C + c] = ov;
}

// State update before the network evaluation of step i (first Ornstein-Uhlenbeck half step of the SSCS sampler):
// u <- M_i u + S_i xi (draw 2 i)
__global__ __launch_bounds__(256) void cld_noise_pre_kernel(float *__restrict__ u8, const float *__restrict__ nz,
                                                             const int *__restrict__ step_ptr, const unsigned *__restrict__ seed,
                                                             unsigned long long elem0, long long pixels, int C, int CP) {
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= pixels * C) return;
    const long long p = i / C;
    const int c = (int)(i - p * C);
    const int step = *step_ptr;
    const float *M = nz + (size_t)step * 12, *S = M + 4;
    const float x = u8[p * CP + c], v = u8[p * CP + C + c];
    const Normal2 xi = normal_pair(seed, elem0 + (unsigned long long)i, 2u * (unsigned)step);
    u8[p * CP + c] = M[0] * x + M[1] * v + S[0] * xi.a + S[1] * xi.b;
    u8[p * CP + C + c] = M[2] * x + M[3] * v + S[2] * xi.a + S[3] * xi.b;
}

int launch_cld_noise_pre(float *u8, const float *nz, const int *step, const unsigned *seed, unsigned long long elem0,
                         long long pixels, int C, hipStream_t s) {
    const int CP = (2 * C + 3) / 4 * 4;
    const long long n = pixels * C;
    hipLaunchKernelGGL(cld_noise_pre_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, u8, nz, step, seed, elem0, pixels,
                       C, CP);
    GD_HIP(hipGetLastError());
    return 0;
}

__global__ __launch_bounds__(256) void randn_pairs_kernel(float2 *__restrict__ out, long long n, const unsigned *__restrict__ seed,
                                                           unsigned draw, unsigned long long elem0) {
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const Normal2 xi = normal_pair(seed, elem0 + (unsigned long long)i, draw);
    out[i] = make_float2(xi.a, xi.b);
}

int launch_randn_pairs(float *out, long long n, const unsigned *seed, unsigned draw, unsigned long long elem0, hipStream_t s) {
    hipLaunchKernelGGL(randn_pairs_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, (float2 *)out, n, seed, draw, elem0);
    GD_HIP(hipGetLastError());
    return 0;
}

int launch_cld_ab_step(float *u8, const float *eps, float *ring, int ring_slots, const float *coef_table, int n_slots,
                       const int *step, long long pixels, int C, const float *tf, const float *nz, const unsigned *seed,
                       unsigned long long elem0, hipStream_t s) {
    const int CP = (2 * C + 3) / 4 * 4;
    const long long n = pixels * C;
    hipLaunchKernelGGL(cld_ab_step_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, u8, eps, ring, ring_slots,
                       coef_table, n_slots, step, pixels, C, CP, tf, nz, seed, elem0);
    GD_HIP(hipGetLastError());
    return 0;
}

// U-Net input of step i when the sampler integrates in a transformed frame: xin = P_i u (P = first matrix of the tf row)
__global__ __launch_bounds__(256) void cld_pre_kernel(const float *__restrict__ u8, float *__restrict__ xin,
                                                       const float *__restrict__ tf, const int *__restrict__ step_ptr,
                                                       long long pixels, int C, int CP) {
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= pixels * C) return;
    const long long p = i / C;
    const int c = (int)(i - p * C);
    const float *P = tf + (size_t)(*step_ptr) * 12;
    const float x = u8[p * CP + c], v = u8[p * CP + C + c];
    xin[p * CP + c] = P[0] * x + P[1] * v;
    xin[p * CP + C + c] = P[2] * x + P[3] * v;
    if (c == 0)
        for (int k = 2 * C; k < CP; ++k) xin[p * CP + k] = 0.f;
}

int launch_cld_pre(const float *u8, float *xin, const float *tf, const int *step, long long pixels, int C, hipStream_t s) {
    const int CP = (2 * C + 3) / 4 * 4;
    const long long n = pixels * C;
    hipLaunchKernelGGL(cld_pre_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, u8, xin, tf, step, pixels, C, CP);
    GD_HIP(hipGetLastError());
    return 0;
}

__global__ __launch_bounds__(256) void cld_finish_kernel(const float *__restrict__ u8, const float *__restrict__ eps,
                                                          const float *__restrict__ dn, float dt, int denoise,
                                                          int inverse_scale, float *__restrict__ x_out,
                                                          float *__restrict__ v_out, long long pixels, int C, int CP,
                                                          int flags) {
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= pixels * C) return;
    const long long p = i / C;
    const int c = (int)(i - p * C);
    float x = u8[p * CP + c], v = u8[p * CP + C + c];
    if (denoise) {
        float ex = eps[(size_t)p * 2 * C + c], ev = eps[(size_t)p * 2 * C + C + c];
        const float *F = dn, *GG = dn + 4, *Rt = dn + 8;
        if (flags & 1) {  // mixed score: eps += X u (dn + 12)
            const float *X = dn + 12;
            ex += X[0] * x + X[1] * v;
            ev += X[2] * x + X[3] * v;
        }
        // score = -R^-T eps
        const float sx = -(Rt[0] * ex + Rt[1] * ev), sv = -(Rt[2] * ex + Rt[3] * ev);
        const float fx = F[0] * x + F[1] * v, fv = F[2] * x + F[3] * v;
        const float gx = GG[0] * sx + GG[1] * sv, gv = GG[2] * sx + GG[3] * sv;
        x = x + fx * dt - gx * dt;
        v = v + fv * dt - gv * dt;
    }
    if (flags & 2) {  // leave the transformed frame: u <- POST u (dn + 16)
        const float *Q = dn + 16;
        const float nx = Q[0] * x + Q[1] * v, nv = Q[2] * x + Q[3] * v;
        x = nx;
        v = nv;
    }
    x_out[i] = inverse_scale ? (x + 1.0f) / 2.0f : x;
    v_out[i] = v;
}

int launch_cld_finish(const float *u8, const float *eps, const float *dn, float dt, int denoise, int inverse_scale,
                      float *x_out, float *v_out, long long pixels, int C, int flags, hipStream_t s) {
    const int CP = (2 * C + 3) / 4 * 4;
    const long long n = pixels * C;
    hipLaunchKernelGGL(cld_finish_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, u8, eps, dn, dt, denoise,
                       inverse_scale, x_out, v_out, pixels, C, CP, flags);
    GD_HIP(hipGetLastError());
    return 0;
}

__global__ void advance_step_kernel(int *step) { *step += 1; }

int launch_advance_step(int *step, hipStream_t s) {
    hipLaunchKernelGGL(advance_step_kernel, dim3(1), dim3(1), 0, s, step);
    GD_HIP(hipGetLastError());
    return 0;
}

// dst[0:n] = src[0:n] as a kernel (n floats; 16-byte pieces when both pointers allow it).  Used wherever a sampler call moves data
// between the caller's buffers, the engine's staging copies and the sampler state: since round 6 a sampler call consists of kernel
// launches (and graphs of kernel nodes) only -- see launch_reset_step_row for what a runtime fill node did inside a replayed graph.
__global__ void copy_floats_kernel(const float *__restrict__ src, float *__restrict__ dst, long long n, int vec) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (vec) {
        if (i * 4 + 3 < n) reinterpret_cast<float4 *>(dst)[i] = reinterpret_cast<const float4 *>(src)[i];
        else for (long long k = i * 4; k < n; ++k) dst[k] = src[k];
    } else if (i < n) {
        dst[i] = src[i];
    }
}

int launch_copy_floats(const float *src, float *dst, long long n, hipStream_t s) {
    if (n <= 0) return 0;
    const int vec = ((reinterpret_cast<uintptr_t>(src) | reinterpret_cast<uintptr_t>(dst)) & 15) == 0 ? 1 : 0;
    const long long items = vec ? (n + 3) / 4 : n;
    hipLaunchKernelGGL(copy_floats_kernel, dim3((unsigned)((items + 255) / 256)), dim3(256), 0, s, src, dst, n, vec);
    GD_HIP(hipGetLastError());
    return 0;
}

// step++ and, in the same launch, the NEXT step's row of the hoisted time-embedding table into the buffer the convolutions read their
// per-image bias from (engine.cpp: build_temb_table): row[0:width] = table[step + 1][0:width] when that knot exists.  One workgroup:
// every thread reads the counter before thread 0 rewrites it.
__global__ void __launch_bounds__(1024) advance_step_row_kernel(int *step, const float4 *__restrict__ table, float4 *__restrict__ row,
                                                                 int width4, int rows) {
    const int next = *step + 1;
    __syncthreads();
    if (next < rows) {
        const float4 *src = table + (size_t)next * width4;
        for (int i = threadIdx.x; i < width4; i += 1024) row[i] = src[i];
    }
    if (threadIdx.x == 0) *step = next;
}

// *step = 0 (and the first knot's row of the hoisted table into `row` when there is one), as a KERNEL: inside a captured whole-call graph a
// hipMemsetAsync node replayed with other bytes than it was captured with -- the counter read 0x05050505 after an unrelated framework
// launch with large kernel arguments (a 5-d broadcast torch.where) had run between two replays, and the AB step indexed its tables with it
// (round 6: rocgdb on tests/test_gpu_parity_abs.py).  Kernel nodes keep their arguments in the graph; nothing else goes into a capture now.
__global__ void __launch_bounds__(1024) reset_step_row_kernel(int *step, const float4 *__restrict__ table, float4 *__restrict__ row, int width4) {
    for (int i = threadIdx.x; i < width4; i += 1024) row[i] = table[i];
    if (threadIdx.x == 0) *step = 0;
}

int launch_reset_step_row(int *step, const float *table, float *row, int width, hipStream_t s) {
    GD_CHECK(width % 4 == 0, "reset_step_row: width %d is not a multiple of 4", width);
    hipLaunchKernelGGL(reset_step_row_kernel, dim3(1), dim3(1024), 0, s, step, (const float4 *)table, (float4 *)row, row ? width / 4 : 0);
    GD_HIP(hipGetLastError());
    return 0;
}

int launch_advance_step_row(int *step, const float *table, float *row, int width, int rows, hipStream_t s) {
    GD_CHECK(width % 4 == 0, "advance_step_row: width %d is not a multiple of 4", width);
    hipLaunchKernelGGL(advance_step_row_kernel, dim3(1), dim3(1024), 0, s, step, (const float4 *)table, (float4 *)row, width / 4, rows);
    GD_HIP(hipGetLastError());
    return 0;
}

// ---- 32x32 orthonormal DCT, LDS resident -------------------------------------------------------------
// D[k][n] = s_k cos(pi (2n+1) k / 64), s_0 = 1/sqrt(32), s_k = sqrt(2/32).
// forward  Y = D X D^T ; inverse X = D^T Y D.   One block of 256 threads per (b, c) plane.
constexpr int DN = 32, DL = 33;

__device__ __forceinline__ void dct_fill_basis(float *D) {
    for (int i = threadIdx.x; i < DN * DN; i += 256) {
        const int k = i / DN, n = i - k * DN;
        const double sc = (k == 0) ? 0.17677669529663688 /* 1/sqrt(32) */ : 0.25 /* sqrt(2/32) */;
        D[k * DL + n] = (float)(sc * cospi((double)((2 * n + 1) * k) / 64.0));
    }
}

// X (in LDS, [32][33]) -> Y (in LDS); T scratch.  inverse: uses D^T.
__device__ __forceinline__ void dct_plane(const float *D, float *X, float *T, int inverse) {
    // stage 1: T = Dm X   (Dm = D forward, D^T inverse)
    for (int i = threadIdx.x; i < DN * DN; i += 256) {
        const int r = i / DN, cidx = i - r * DN;
        float acc = 0.f;
#pragma unroll 8
        for (int n = 0; n < DN; ++n) acc += (inverse ? D[n * DL + r] : D[r * DL + n]) * X[n * DL + cidx];
        T[r * DL + cidx] = acc;
    }
    __syncthreads();
    // stage 2: X = T Dm^T
    for (int i = threadIdx.x; i < DN * DN; i += 256) {
        const int r = i / DN, cidx = i - r * DN;
        float acc = 0.f;
#pragma unroll 8
        for (int m = 0; m < DN; ++m) acc += T[r * DL + m] * (inverse ? D[m * DL + cidx] : D[cidx * DL + m]);
        X[r * DL + cidx] = acc;
    }
    __syncthreads();
}

__global__ __launch_bounds__(256) void dct2d_32_kernel(const float *__restrict__ in, float *__restrict__ out, int C,
                                                        int inverse) {
    __shared__ float D[DN * DL], X[DN * DL], T[DN * DL];
    const int b = blockIdx.x / C, c = blockIdx.x - b * C;
    dct_fill_basis(D);
    const float *src = in + (size_t)b * DN * DN * C + c;
    for (int i = threadIdx.x; i < DN * DN; i += 256) X[(i / DN) * DL + (i % DN)] = src[(size_t)i * C];
    __syncthreads();
    dct_plane(D, X, T, inverse);
    float *dst = out + (size_t)b * DN * DN * C + c;
    for (int i = threadIdx.x; i < DN * DN; i += 256) dst[(size_t)i * C] = X[(i / DN) * DL + (i % DN)];
}

int launch_dct2d_32(const float *in, float *out, int batch, int channels, int inverse, hipStream_t s) {
    hipLaunchKernelGGL(dct2d_32_kernel, dim3(batch * channels), dim3(256), 0, s, in, out, channels, inverse);
    GD_HIP(hipGetLastError());
    return 0;
}

// ---- blur fused sampler ------------------------------------------------------------------------------
// x8 (B,32,32,CP) = [IDCT(y) , 0...]: the U-Net input of blur_jax/models/utils.py:155-160 (y2x)
__global__ __launch_bounds__(256) void blur_idct_pad_kernel(const float *__restrict__ y, float *__restrict__ x8, int C,
                                                             int CP) {
    __shared__ float D[DN * DL], X[DN * DL], T[DN * DL];
    const int b = blockIdx.x / CP, c = blockIdx.x - b * CP;
    float *dst = x8 + (size_t)b * DN * DN * CP + c;
    if (c >= C) {
        for (int i = threadIdx.x; i < DN * DN; i += 256) dst[(size_t)i * CP] = 0.f;
        return;
    }
    dct_fill_basis(D);
    const float *src = y + (size_t)b * DN * DN * C + c;
    for (int i = threadIdx.x; i < DN * DN; i += 256) X[(i / DN) * DL + (i % DN)] = src[(size_t)i * C];
    __syncthreads();
    dct_plane(D, X, T, 1);
    for (int i = threadIdx.x; i < DN * DN; i += 256) dst[(size_t)i * CP] = X[(i / DN) * DL + (i % DN)];
}

int launch_blur_idct_pad(const float *y, float *x8, int batch, int C, hipStream_t s) {
    const int CP = (C + 3) / 4 * 4;
    hipLaunchKernelGGL(blur_idct_pad_kernel, dim3(batch * CP), dim3(256), 0, s, y, x8, C, CP);
    GD_HIP(hipGetLastError());
    return 0;
}

// y <- m(t')/1 * [ (y - s(t) e_y) / m(t) ] + s(t') e_y with e_y = DCT(eps_x); coefficient rows indexed by *step
__global__ __launch_bounds__(256) void blur_step_kernel(float *__restrict__ y, const float *__restrict__ eps_x,
                                                         const float *__restrict__ mean_coef,
                                                         const float *__restrict__ std_coef,
                                                         const int *__restrict__ step_ptr, int C,
                                                         const float *__restrict__ m2e) {
    __shared__ float D[DN * DL], X[DN * DL], T[DN * DL];
    const int b = blockIdx.x / C, c = blockIdx.x - b * C;
    const int step = *step_ptr;
    dct_fill_basis(D);
    const float *src = eps_x + (size_t)b * DN * DN * C + c;
    for (int i = threadIdx.x; i < DN * DN; i += 256) X[(i / DN) * DL + (i % DN)] = src[(size_t)i * C];
    __syncthreads();
    dct_plane(D, X, T, 0);
    const float *m0 = mean_coef + (size_t)step * DN * DN, *m1 = m0 + DN * DN;
    const float s0 = std_coef[step], s1 = std_coef[step + 1];
    // sde.model2eps in linear form (blur_jax/models/utils.py:147): eps_y = a DCT(net) + b (.) y with a per-frequency plane b
    const float *mrow = m2e ? m2e + (size_t)step * (1 + DN * DN) : nullptr;
    const float ma = mrow ? mrow[0] : 1.f;
    float *yy = y + (size_t)b * DN * DN * C + c;
    for (int i = threadIdx.x; i < DN * DN; i += 256) {
        const float ycur = yy[(size_t)i * C];
        float ey = X[(i / DN) * DL + (i % DN)];
        if (mrow) ey = ma * ey + mrow[1 + i] * ycur;
        const float y0 = 1.0f / m0[i] * (ycur - s0 * ey);  // blur_jax/sampling.py:69
        yy[(size_t)i * C] = m1[i] * y0 + s1 * ey;           // :71-74
    }
}

int launch_blur_step(float *y, const float *eps_x, const float *mean_coef, const float *std_coef, const int *step,
                     int batch, int C, const float *m2e, hipStream_t s) {
    hipLaunchKernelGGL(blur_step_kernel, dim3(batch * C), dim3(256), 0, s, y, eps_x, mean_coef, std_coef, step, C, m2e);
    GD_HIP(hipGetLastError());
    return 0;
}

// Multistep (DEIS) blur update with per-frequency coefficients: y' = c[step][0] y + c[step][1] e_y + sum_j c[step][2+j] hist_j, all
// (32 x 32) planes; e_y = DCT(eps_x) of this step becomes the newest history entry.  The history ring works like the CLD one
// (logical slot j = physical (head + j) % slots, head = (-step) mod slots).
__global__ __launch_bounds__(256) void blur_ab_step_kernel(float *__restrict__ y, const float *__restrict__ eps_x,
                                                            float *__restrict__ ring, int ring_slots,
                                                            const float *__restrict__ coef, int n_slots,
                                                            const int *__restrict__ step_ptr, int C, long long slot_stride,
                                                            const float *__restrict__ m2e) {
    __shared__ float D[DN * DL], X[DN * DL], T[DN * DL];
    const int b = blockIdx.x / C, c = blockIdx.x - b * C;
    const int step = *step_ptr;
    dct_fill_basis(D);
    const size_t plane = (size_t)b * DN * DN * C + c;
    const float *src = eps_x + plane;
    for (int i = threadIdx.x; i < DN * DN; i += 256) X[(i / DN) * DL + (i % DN)] = src[(size_t)i * C];
    __syncthreads();
    dct_plane(D, X, T, 0);
    const float *cf = coef + (size_t)step * n_slots * DN * DN;
    const int head = ring_slots > 0 ? (ring_slots - step % ring_slots) % ring_slots : 0;
    float *yy = y + plane;
    const float *mrow = m2e ? m2e + (size_t)step * (1 + DN * DN) : nullptr;
    const float ma = mrow ? mrow[0] : 1.f;
    for (int i = threadIdx.x; i < DN * DN; i += 256) {
        float ey = X[(i / DN) * DL + (i % DN)];
        if (mrow) ey = ma * ey + mrow[1 + i] * yy[(size_t)i * C];
        float acc = cf[i] * yy[(size_t)i * C] + cf[DN * DN + i] * ey;
        int phys = head;
        for (int j = 0; j < ring_slots; ++j) {
            acc += cf[(size_t)(2 + j) * DN * DN + i] * ring[(size_t)phys * slot_stride + plane + (size_t)i * C];
            if (++phys == ring_slots) phys = 0;
        }
        if (ring_slots > 0) ring[(size_t)((head + ring_slots - 1) % ring_slots) * slot_stride + plane + (size_t)i * C] = ey;
        yy[(size_t)i * C] = acc;
    }
}

int launch_blur_ab_step(float *y, const float *eps_x, float *ring, int ring_slots, const float *coef, int n_slots, const int *step,
                        int batch, int C, long long slot_stride, const float *m2e, hipStream_t s) {
    hipLaunchKernelGGL(blur_ab_step_kernel, dim3(batch * C), dim3(256), 0, s, y, eps_x, ring, ring_slots, coef, n_slots, step, C,
                       slot_stride, m2e);
    GD_HIP(hipGetLastError());
    return 0;
}

__global__ __launch_bounds__(256) void blur_finish_kernel(const float *__restrict__ y, float *__restrict__ x_out, int C,
                                                           int inverse_scale) {
    __shared__ float D[DN * DL], X[DN * DL], T[DN * DL];
    const int b = blockIdx.x / C, c = blockIdx.x - b * C;
    dct_fill_basis(D);
    const float *src = y + (size_t)b * DN * DN * C + c;
    for (int i = threadIdx.x; i < DN * DN; i += 256) X[(i / DN) * DL + (i % DN)] = src[(size_t)i * C];
    __syncthreads();
    dct_plane(D, X, T, 1);
    float *dst = x_out + (size_t)b * DN * DN * C + c;
    for (int i = threadIdx.x; i < DN * DN; i += 256) {
        const float v = X[(i / DN) * DL + (i % DN)];
        dst[(size_t)i * C] = inverse_scale ? (v + 1.0f) / 2.0f : v;
    }
}

int launch_blur_finish(const float *y, float *x_out, int batch, int C, int inverse_scale, hipStream_t s) {
    hipLaunchKernelGGL(blur_finish_kernel, dim3(batch * C), dim3(256), 0, s, y, x_out, C, inverse_scale);
    GD_HIP(hipGetLastError());
    return 0;
}

}  // namespace gd
