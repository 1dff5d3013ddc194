// Bandwidth-bound kernels of the NCSN++ forward: GroupNorm(+swish), FIR resampling, softmax,
// Fourier time embedding, channel padding.  All tensors NHWC fp32, accessed as float4 along C
// (every channel count in the network is a multiple of 4), one wave64 reading 1 KiB contiguous.
//
// Reference semantics:
//   GroupNorm  flax nn.GroupNorm(num_groups=min(C//4,32)), epsilon 1e-6, var = E[x^2]-E[x]^2, stats over
//              (H, W, C/G) per sample, contiguous channel groups  (call sites layerspp.py:69,196,218,
//              ncsnpp.py:236)
//   swish      x*sigmoid(x)                                   (layers.py:39-40)
//   FIR        upsample_2d / downsample_2d with k=[1,3,3,1]   (up_or_down_sampling.py:333-411);
//              closed forms in SURVEY.md App. E
//   softmax    jax.nn.softmax over the key axis               (layerspp.py:76)
//   Fourier    [sin, cos](2 pi W log(labels))                 (layerspp.py:33-43, ncsnpp.py:74-77)
#include <cstdlib>

#include "../common.h"
#include "act.h"

namespace gd {

typedef float f32x4 __attribute__((ext_vector_type(4)));


int gn_slices(int batch, int HW) {
    // enough blocks to fill 256 CUs several times over, but at least 64 pixels per slice
    int s = 1;
    while (s < GN_MAX_SLICES && (long long)batch * s < 2048 && HW / (s * 2) >= 64) s *= 2;
    return s;
}

// ---- GroupNorm ----------------------------------------------------------------------------------------------------
// Statistics arrive as "quad partial sums" qs[b][slice][c/4] = (sum, sumsq) over a slice of the image's pixels and four
// consecutive channels.  Normally the GEMM that produced the tensor emits them in its epilogue (one slice per 32 output
// rows); gn_stats_kernel is the stand-alone producer for tensors that have none.  gn_apply_kernel reduces slices and quads
// to (mean, rstd) per (image, group) in its prologue and normalises (+ swish).  Every step is deterministic (no atomics).

// grid (slices, B), 256 threads; T = (256 / C4) * C4 threads active so that each thread keeps one fixed float4 column
__global__ __launch_bounds__(256) void gn_stats_kernel(const float *__restrict__ x, int C, float *__restrict__ qs, int HW,
                                                        int slices) {
    __shared__ float s_sum[256], s_sq[256];
    const int C4 = C >> 2;
    const int rows_pp = 256 / C4;
    const int T = rows_pp * C4;
    const int tid = threadIdx.x;
    const int b = blockIdx.y, sl = blockIdx.x;
    const int per = (HW + slices - 1) / slices;
    const int p_beg = sl * per, p_end = min(HW, p_beg + per);
    float s = 0.f, q = 0.f;
    if (tid < T) {
        const int f = tid % C4, rp = tid / C4;
        const float *src = x + (size_t)b * HW * C + f * 4;
        for (int p = p_beg + rp; p < p_end; p += rows_pp) {
            const f32x4 v = *reinterpret_cast<const f32x4 *>(src + (size_t)p * C);
            s += (v[0] + v[1]) + (v[2] + v[3]);
            q += (v[0] * v[0] + v[1] * v[1]) + (v[2] * v[2] + v[3] * v[3]);
        }
    }
    s_sum[tid] = s;
    s_sq[tid] = q;
    __syncthreads();
    if (tid < C4) {
        float ts = 0.f, tq = 0.f;
        for (int rp = 0; rp < rows_pp; ++rp) { ts += s_sum[rp * C4 + tid]; tq += s_sq[rp * C4 + tid]; }
        float *o = qs + (((size_t)b * slices + sl) * C4 + tid) * 2;
        o[0] = ts;
        o[1] = tq;
    }
}

#ifndef GN_UNROLL
#define GN_UNROLL 4
#endif
#ifndef GN_NT
#define GN_NT 1  // 1: nontemporal loads (the raw tensor is dead after this pass: 164.3 -> 159.5 ms per 50 evaluations in-box), 2: nontemporal stores (slower: the consumer reads the output next)
#endif
// grid (blocks per image, B), 256 threads.  Prologue: the block reduces the quad partial sums of its image to (mean, rstd)
// per group in LDS (thread t -> group t % G, slice phase t / G), across two channel-concatenated sources when the GroupNorm
// runs on a skip concat; then every thread normalises its fixed float4 column over the block's pixels.
__global__ __launch_bounds__(256) void gn_apply_kernel(const float *__restrict__ x0, int C0, const float *__restrict__ x1,
                                                        int C1, const float *__restrict__ qs0, int S0,
                                                        const float *__restrict__ qs1, int S1,
                                                        const float *__restrict__ gamma, const float *__restrict__ beta,
                                                        float *__restrict__ out, int HW, int groups, int act, float inv_n,
                                                        float eps, int split_out, float *__restrict__ table) {
    __shared__ float s_sum[256], s_sq[256];
    const int C = C0 + C1, C4 = C >> 2;
    const int Q0 = C0 >> 2, Q1 = C1 >> 2;
    const int tid = threadIdx.x;
    const int b = blockIdx.y;
    {
        const int phases = 256 / groups;
        const int g = tid % groups, ph = tid / groups;
        const int qpg = C4 / groups;  // quads per group
        float s = 0.f, q = 0.f;
        if (ph < phases) {
            for (int k = 0; k < qpg; ++k) {
                const int qg = g * qpg + k;
                const bool first = qg < Q0;
                const float *base = first ? qs0 : qs1;
                const int Q = first ? Q0 : Q1, S = first ? S0 : S1, ql = first ? qg : qg - Q0;
                for (int sl = ph; sl < S; sl += phases) {
                    const float *pp = base + (((size_t)b * S + sl) * Q + ql) * 2;
                    s += pp[0];
                    q += pp[1];
                }
            }
        }
        s_sum[tid] = s;
        s_sq[tid] = q;
        __syncthreads();
        if (tid < groups) {
            float ts = 0.f, tq = 0.f;
            for (int k = 0; k < phases; ++k) { ts += s_sum[k * groups + tid]; tq += s_sq[k * groups + tid]; }
            const float mean = ts * inv_n;
            const float var = tq * inv_n - mean * mean;  // flax: E[x^2] - E[x]^2
            s_sum[tid] = mean;
            s_sq[tid] = rsqrtf(var + eps);
        }
        __syncthreads();
    }
    const int rows_pp = 256 / C4;
    const int T = rows_pp * C4;
    if (tid >= T) return;
    const int nblk = gridDim.x;
    const int per = (HW + nblk - 1) / nblk;
    const int p_beg = blockIdx.x * per, p_end = min(HW, p_beg + per);
    const int f = tid % C4, rp = tid / C4;
    const int c = f * 4;
    const int g = f / (C4 / groups);
    const float mean = s_sum[g];
    const float rstd = s_sq[g];
    const f32x4 ga = *reinterpret_cast<const f32x4 *>(gamma + c);
    const f32x4 be = *reinterpret_cast<const f32x4 *>(beta + c);
    f32x4 sc, sh;
#pragma unroll
    for (int i = 0; i < 4; ++i) { sc[i] = rstd * ga[i]; sh[i] = be[i] - mean * rstd * ga[i]; }
    if (table) {  // table mode: only the per-channel affine map of this image, table[b] = [C scale | C shift]; the consumer
                  // GEMM applies it (and the activation) while it stages its input (conv3x3_halo_bf16x3.hip)
        if (blockIdx.x == 0 && rp == 0) {
            *reinterpret_cast<f32x4 *>(table + (size_t)b * 2 * C + c) = sc;
            *reinterpret_cast<f32x4 *>(table + (size_t)b * 2 * C + C + c) = sh;
        }
        return;
    }
    const float *src;
    int ld, cc;
    if (c < C0) { src = x0; ld = C0; cc = c; } else { src = x1; ld = C1; cc = c - C0; }
    src += (size_t)b * HW * ld + cc;
    float *dst = out + (size_t)b * HW * C + c;
    auto finish = [&](f32x4 v, int p) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            float y = v[i] * sc[i] + sh[i];
            v[i] = act ? swish_e(y) : y;
        }
        if (split_out) {  // row = [C bf16 hi | C bf16 lo], hi = rne(x), lo = rne(x - hi)
            typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
            bf16x4 h, l;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                h[i] = (__bf16)v[i];
                l[i] = (__bf16)(v[i] - (float)h[i]);
            }
            __bf16 *row = reinterpret_cast<__bf16 *>(out + ((size_t)b * HW + p) * C);
            *reinterpret_cast<bf16x4 *>(row + c) = h;
            *reinterpret_cast<bf16x4 *>(row + C + c) = l;
        } else {
            if (GN_NT & 2) __builtin_nontemporal_store(v, reinterpret_cast<f32x4 *>(dst + (size_t)p * C));
            else *reinterpret_cast<f32x4 *>(dst + (size_t)p * C) = v;
        }
    };
    // GN_UNROLL independent 16-byte loads in flight per thread (one per iteration leaves the kernel latency bound)
    int p = p_beg + rp;
    for (; p + (GN_UNROLL - 1) * rows_pp < p_end; p += GN_UNROLL * rows_pp) {
        f32x4 v[GN_UNROLL];
#pragma unroll
        for (int u = 0; u < GN_UNROLL; ++u) {
            const f32x4 *q = reinterpret_cast<const f32x4 *>(src + (size_t)(p + u * rows_pp) * ld);
            v[u] = (GN_NT & 1) ? __builtin_nontemporal_load(q) : *q;
        }
#pragma unroll
        for (int u = 0; u < GN_UNROLL; ++u) finish(v[u], p + u * rows_pp);
    }
    for (; p < p_end; p += rows_pp) finish(*reinterpret_cast<const f32x4 *>(src + (size_t)p * ld), p);
}

// ---- generic GroupNorm: any group size, up to two channel-concatenated sources (rare configurations) ----
// grid (G, B): (mean, rstd) of group g of image b
__global__ __launch_bounds__(256) void gn_group_stats_kernel(const float *__restrict__ x0, int C0, const float *__restrict__ x1,
                                                              int C1, float *__restrict__ mr, int HW, int groups, float eps) {
    __shared__ float s_sum[256], s_sq[256];
    const int C = C0 + C1, cpg = C / groups;
    const int g = blockIdx.x, b = blockIdx.y, tid = threadIdx.x;
    float s = 0.f, q = 0.f;
    for (long long i = tid; i < (long long)HW * cpg; i += 256) {
        const int p = (int)(i / cpg), c = g * cpg + (int)(i - (long long)p * cpg);
        const float v = c < C0 ? x0[((size_t)b * HW + p) * C0 + c] : x1[((size_t)b * HW + p) * C1 + (c - C0)];
        s += v;
        q += v * v;
    }
    s_sum[tid] = s;
    s_sq[tid] = q;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if (tid < o) { s_sum[tid] += s_sum[tid + o]; s_sq[tid] += s_sq[tid + o]; }
        __syncthreads();
    }
    if (tid == 0) {
        const float n = (float)HW * (float)cpg;
        const float mean = s_sum[0] / n;
        const float var = s_sq[0] / n - mean * mean;
        mr[((size_t)b * groups + g) * 2] = mean;
        mr[((size_t)b * groups + g) * 2 + 1] = rsqrtf(var + eps);
    }
}

__global__ __launch_bounds__(256) void gn_apply_generic_kernel(const float *__restrict__ x0, int C0, const float *__restrict__ x1,
                                                                int C1, const float *__restrict__ mr,
                                                                const float *__restrict__ gamma, const float *__restrict__ beta,
                                                                float *__restrict__ out, int HW, int groups, int act,
                                                                long long total) {
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= total) return;
    const int C = C0 + C1, cpg = C / groups;
    const int c = (int)(i % C);
    const long long bp = i / C;
    const int b = (int)(bp / HW);
    const float v = c < C0 ? x0[(size_t)bp * C0 + c] : x1[(size_t)bp * C1 + (c - C0)];
    const float *m = mr + ((size_t)b * groups + c / cpg) * 2;
    const float y = (v - m[0]) * m[1] * gamma[c] + beta[c];
    out[i] = act ? swish_e(y) : y;
}

int launch_gn_generic(const float *x0, int C0, const float *x1, int C1, float *mr, const float *gamma, const float *beta,
                      float *out, int batch, int HW, int groups, int act, float eps, hipStream_t s) {
    const int C = C0 + C1;
    GD_CHECK(groups >= 1 && C % groups == 0, "group_norm: %d channels do not split into %d groups", C, groups);
    hipLaunchKernelGGL(gn_group_stats_kernel, dim3(groups, batch), dim3(256), 0, s, x0, C0, x1, C1, mr, HW, groups, eps);
    const long long total = (long long)batch * HW * C;
    hipLaunchKernelGGL(gn_apply_generic_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, x0, C0, x1, C1, mr, gamma,
                       beta, out, HW, groups, act, total);
    GD_HIP(hipGetLastError());
    return 0;
}

int launch_gn_stats(const float *x, int C, float *qs, int batch, int HW, int slices, hipStream_t s) {
    GD_CHECK(C % 4 == 0 && C / 4 <= 256, "group_norm: unsupported C=%d", C);
    hipLaunchKernelGGL(gn_stats_kernel, dim3(slices, batch), dim3(256), 0, s, x, C, qs, HW, slices);
    GD_HIP(hipGetLastError());
    return 0;
}

int launch_gn_apply(const float *x0, int C0, const float *x1, int C1, const float *qs0, int S0, const float *qs1, int S1,
                    const float *gamma, const float *beta, float *out, int batch, int HW, int groups, int act, float eps,
                    int split_out, float *table, hipStream_t s) {
    const int C = C0 + C1;
    GD_CHECK(C % 4 == 0 && C0 % 4 == 0 && C / 4 <= 256 && (C / 4) % groups == 0 && groups <= 256 && groups >= 1,
             "group_norm: unsupported C=%d (C0=%d) groups=%d", C, C0, groups);
    int nblk = 1;
    while (!table && nblk < 64 && (long long)batch * nblk < 4096 && HW / (nblk * 2) >= 16) nblk *= 2;
    const float inv_n = 1.0f / ((float)HW * (float)(C / groups));
    hipLaunchKernelGGL(gn_apply_kernel, dim3(nblk, batch), dim3(256), 0, s, x0, C0, x1, C1, qs0, S0, qs1, S1, gamma, beta, out,
                       HW, groups, act, inv_n, eps, split_out, table);
    GD_HIP(hipGetLastError());
    return 0;
}

// ---- naive 2x resampling of the fir=False configurations (up_or_down_sampling.py:76-86) ----
// up: nearest neighbour (jnp.tile);  down: mean of the 2x2 block
__global__ __launch_bounds__(256) void naive_resample2_kernel(const float *__restrict__ x, float *__restrict__ out, int H, int W,
                                                              int C4, long long total, int up) {
    const long long idx = (long long)blockIdx.x * 256 + threadIdx.x;
    if (idx >= total) return;
    const int f = (int)(idx % C4);
    long long r = idx / C4;
    const int OW = up ? 2 * W : W / 2, OH = up ? 2 * H : H / 2;
    const int ox = (int)(r % OW);
    r /= OW;
    const int oy = (int)(r % OH);
    const int b = (int)(r / OH);
    const f32x4 *xb = reinterpret_cast<const f32x4 *>(x) + (size_t)b * H * W * C4 + f;
    f32x4 v;
    if (up) {
        v = xb[((size_t)(oy >> 1) * W + (ox >> 1)) * C4];
    } else {
        const size_t o = ((size_t)(2 * oy) * W + 2 * ox) * C4;
        // jnp.mean over the two block axes: the sum of the four values times 1/4
        v = ((xb[o] + xb[o + C4]) + (xb[o + (size_t)W * C4] + xb[o + (size_t)W * C4 + C4])) * 0.25f;
    }
    reinterpret_cast<f32x4 *>(out)[idx] = v;
}

int launch_naive_resample2(const float *x, float *out, int batch, int H, int W, int C, int up, hipStream_t s) {
    GD_CHECK(C % 4 == 0 && (up || (H % 2 == 0 && W % 2 == 0)), "naive resample: unsupported shape %dx%dx%d", H, W, C);
    const int C4 = C / 4;
    const long long total = (long long)batch * (up ? 4 * H * W : (H / 2) * (W / 2)) * C4;
    hipLaunchKernelGGL(naive_resample2_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, x, out, H, W, C4, total, up);
    GD_HIP(hipGetLastError());
    return 0;
}

// ---- FIR [1,3,3,1] 2x resampling, zero boundary -------------------------------
// up:   y[2m] = (x[m-1] + 3 x[m]) / 4,  y[2m+1] = (3 x[m] + x[m+1]) / 4   per axis
// down: y[i]  = (x[2i-1] + 3 x[2i] + 3 x[2i+1] + x[2i+2]) / 8             per axis
__global__ __launch_bounds__(256) void fir_up2_kernel(const float *__restrict__ x, float *__restrict__ out, int H, int W,
                                                       int C4, long long total) {
    const long long idx = (long long)blockIdx.x * 256 + threadIdx.x;
    if (idx >= total) return;
    const int f = (int)(idx % C4);
    long long r = idx / C4;
    const int OW = 2 * W, OH = 2 * H;
    const int ox = (int)(r % OW);
    r /= OW;
    const int oy = (int)(r % OH);
    const int b = (int)(r / OH);
    // per axis: even output 2m uses (m-1: 1/4, m: 3/4); odd output 2m+1 uses (m: 3/4, m+1: 1/4)
    const int my = oy >> 1, mx = ox >> 1;
    const int y0 = (oy & 1) ? my : my - 1, y1 = y0 + 1;
    const int x0 = (ox & 1) ? mx : mx - 1, x1 = x0 + 1;
    const float wy0 = (oy & 1) ? 0.75f : 0.25f, wy1 = 1.0f - wy0;
    const float wx0 = (ox & 1) ? 0.75f : 0.25f, wx1 = 1.0f - wx0;
    const f32x4 *xb = reinterpret_cast<const f32x4 *>(x) + (size_t)b * H * W * C4 + f;
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    auto tap = [&](int yy, int xx, float w) {
        if ((unsigned)yy < (unsigned)H && (unsigned)xx < (unsigned)W) {
            const f32x4 v = xb[((size_t)yy * W + xx) * C4];
            acc += v * w;
        }
    };
    tap(y0, x0, wy0 * wx0);
    tap(y0, x1, wy0 * wx1);
    tap(y1, x0, wy1 * wx0);
    tap(y1, x1, wy1 * wx1);
    reinterpret_cast<f32x4 *>(out)[idx] = acc;
}

__global__ __launch_bounds__(256) void fir_down2_kernel(const float *__restrict__ x, float *__restrict__ out, int H,
                                                         int W, int C4, long long total) {
    const long long idx = (long long)blockIdx.x * 256 + threadIdx.x;
    if (idx >= total) return;
    const int f = (int)(idx % C4);
    long long r = idx / C4;
    const int OW = W / 2, OH = H / 2;
    const int ox = (int)(r % OW);
    r /= OW;
    const int oy = (int)(r % OH);
    const int b = (int)(r / OH);
    const f32x4 *xb = reinterpret_cast<const f32x4 *>(x) + (size_t)b * H * W * C4 + f;
    const float wt[4] = {0.125f, 0.375f, 0.375f, 0.125f};
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int a = 0; a < 4; ++a) {
        const int yy = 2 * oy - 1 + a;
        if ((unsigned)yy >= (unsigned)H) continue;
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            const int xx = 2 * ox - 1 + c;
            if ((unsigned)xx >= (unsigned)W) continue;
            acc += xb[((size_t)yy * W + xx) * C4] * (wt[a] * wt[c]);
        }
    }
    reinterpret_cast<f32x4 *>(out)[idx] = acc;
}

// ---- the FIR of conv_downsample_2d at full resolution (up_or_down_sampling.py:168-209: upfirdn with k = outer([1,3,3,1]) / 64 and
// pads (2, 2)): out (H + 1, W + 1), out[y][x] = sum_{i,j < 4} k[i] k[j] x[y + i - 2][x + j - 2], zero outside the image.  The 3x3
// stride-2 VALID convolution that follows reads it as an ordinary GEMM source.
__global__ __launch_bounds__(256) void fir_pad_kernel(const float *__restrict__ x, float *__restrict__ out, int H, int W, int C4,
                                                       long long total) {
    const long long idx = (long long)blockIdx.x * 256 + threadIdx.x;
    if (idx >= total) return;
    const int f = (int)(idx % C4);
    long long r = idx / C4;
    const int OW = W + 1, OH = H + 1;
    const int ox = (int)(r % OW);
    r /= OW;
    const int oy = (int)(r % OH);
    const int b = (int)(r / OH);
    const float wt[4] = {0.125f, 0.375f, 0.375f, 0.125f};
    const f32x4 *xb = reinterpret_cast<const f32x4 *>(x) + (size_t)b * H * W * C4 + f;
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int a = 0; a < 4; ++a) {
        const int yy = oy + a - 2;
        if ((unsigned)yy >= (unsigned)H) continue;
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            const int xx = ox + c - 2;
            if ((unsigned)xx >= (unsigned)W) continue;
            acc += xb[((size_t)yy * W + xx) * C4] * (wt[a] * wt[c]);
        }
    }
    reinterpret_cast<f32x4 *>(out)[idx] = acc;
}

int launch_fir_pad(const float *x, float *out, int batch, int H, int W, int C, hipStream_t s) {
    GD_CHECK(C % 4 == 0, "fir_pad: C %% 4 != 0");
    const long long total = (long long)batch * (H + 1) * (W + 1) * (C / 4);
    hipLaunchKernelGGL(fir_pad_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, x, out, H, W, C / 4, total);
    GD_HIP(hipGetLastError());
    return 0;
}

int launch_fir_up2(const float *x, float *out, int batch, int H, int W, int C, hipStream_t s) {
    GD_CHECK(C % 4 == 0, "fir_up2: C %% 4 != 0");
    const long long total = (long long)batch * 4 * H * W * (C / 4);
    hipLaunchKernelGGL(fir_up2_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, x, out, H, W, C / 4, total);
    GD_HIP(hipGetLastError());
    return 0;
}

int launch_fir_down2(const float *x, float *out, int batch, int H, int W, int C, hipStream_t s) {
    GD_CHECK(C % 4 == 0 && H % 2 == 0 && W % 2 == 0, "fir_down2: bad shape");
    const long long total = (long long)batch * (H / 2) * (W / 2) * (C / 4);
    hipLaunchKernelGGL(fir_down2_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, x, out, H, W, C / 4, total);
    GD_HIP(hipGetLastError());
    return 0;
}

// ---- row softmax, one wave64 per row ----------------------------------------------------------------
__global__ __launch_bounds__(256) void softmax_rows_kernel(float *__restrict__ x, long long rows, int cols) {
    const long long row = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const int lane = threadIdx.x & 63;
    float *r = x + (size_t)row * cols;
    float mx = -INFINITY;
    for (int c = lane; c < cols; c += 64) mx = fmaxf(mx, r[c]);
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o));
    float sum = 0.f;
    for (int c = lane; c < cols; c += 64) {
        const float e = expf(r[c] - mx);
        r[c] = e;
        sum += e;
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) sum += __shfl_xor(sum, o);
    const float inv = 1.0f / sum;
    for (int c = lane; c < cols; c += 64) r[c] *= inv;
}

int launch_softmax_rows(float *x, long long rows, int cols, hipStream_t s) {
    hipLaunchKernelGGL(softmax_rows_kernel, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, s, x, rows, cols);
    GD_HIP(hipGetLastError());
    return 0;
}

// ---- small elementwise helpers ------------------------------------------------------------------------
__global__ void swish_kernel(const float *__restrict__ x, float *__restrict__ out, long long n) {
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i < n) out[i] = swish_e(x[i]);
}

int launch_swish(const float *x, float *out, long long n, hipStream_t s) {
    hipLaunchKernelGGL(swish_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, x, out, n);
    GD_HIP(hipGetLastError());
    return 0;
}

// positional = 0: GaussianFourierProjection, sin/cos(2 pi W_j log(label)) (layerspp.py:33-43)
// positional = 1: get_timestep_embedding, sin/cos(label * W_j) with W_j = 10000^(-j / (E - 1)) (layers.py:450-464)
__global__ void fourier_kernel(const float *__restrict__ labels, const float *__restrict__ ts_table,
                               const int *__restrict__ step, const float *__restrict__ W, float *__restrict__ out,
                               int batch, int E, int positional) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= batch * E) return;
    const int b = i / E, j = i - b * E;
    const float label = labels ? labels[b] : 999.0f * ts_table[*step];
    const float xp = positional ? label * W[j] : logf(label) * W[j] * 2.0f * 3.14159265358979323846f;
    float sn, cs;
    sincosf(xp, &sn, &cs);
    out[(size_t)b * 2 * E + j] = sn;
    out[(size_t)b * 2 * E + E + j] = cs;
}

int launch_fourier(const float *labels, const float *ts_table, const int *step, const float *W, float *out, int batch,
                   int E, int positional, hipStream_t s) {
    hipLaunchKernelGGL(fourier_kernel, dim3((batch * E + 255) / 256), dim3(256), 0, s, labels, ts_table, step, W, out,
                       batch, E, positional);
    GD_HIP(hipGetLastError());
    return 0;
}

// out[p][0:Cpad] = [x[p*ld_in + 0:Cin] * mul + add, 0...]
__global__ void pad_channels_kernel(const float *__restrict__ x, int ld_in, float *__restrict__ out, long long pixels,
                                    int Cin, int Cpad, float mul, float add) {
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= pixels * Cpad) return;
    const long long p = i / Cpad;
    const int c = (int)(i - p * Cpad);
    out[i] = c < Cin ? x[p * ld_in + c] * mul + add : 0.f;
}

int launch_pad_channels(const float *x, int ld_in, float *out, long long pixels, int Cin, int Cpad, float mul, float add,
                        hipStream_t s) {
    const long long n = pixels * Cpad;
    hipLaunchKernelGGL(pad_channels_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, x, ld_in, out, pixels, Cin,
                       Cpad, mul, add);
    GD_HIP(hipGetLastError());
    return 0;
}

// x[b][...] /= labels[b]   (config.model.scale_by_sigma, ncsnpp.py:239-241)
__global__ void scale_rows_kernel(float *__restrict__ x, const float *__restrict__ labels, long long per_image, long long n) {
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i < n) x[i] = x[i] / labels[i / per_image];
}

int launch_scale_rows(float *x, const float *labels, int batch, long long per_image, hipStream_t s) {
    const long long n = (long long)batch * per_image;
    hipLaunchKernelGGL(scale_rows_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, x, labels, per_image, n);
    GD_HIP(hipGetLastError());
    return 0;
}

}  // namespace gd
