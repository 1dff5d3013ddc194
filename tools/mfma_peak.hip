// Sustained MFMA ceilings: every wave issues independent v_mfma_f32_32x32x2_f32 (modes 0/1) or v_mfma_f32_32x32x16_bf16
// (modes 2/3: constant / changing operands) back to back.
// Build: hipcc --offload-arch=gfx950 -O3 tools/mfma_peak.hip -o tools/mfma_peak
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef float f32x16 __attribute__((ext_vector_type(16)));
// mode 1: operands change every MFMA (per-lane pseudo-random table held in registers) -> realistic toggling
__global__ __launch_bounds__(256) void mfma_loop_rand(float *out, int iters) {
    f32x16 a0 = {}, a1 = {}, a2 = {}, a3 = {};
    float t[8];
    unsigned h = threadIdx.x * 2654435761u + blockIdx.x * 40503u + 12345u;
    for (int i = 0; i < 8; ++i) { h = h * 1664525u + 1013904223u; t[i] = ((int)(h >> 8) - (1 << 23)) * (1.0f / (1 << 23)); }
    for (int i = 0; i < iters; i += 2) {
        a0 = __builtin_amdgcn_mfma_f32_32x32x2f32(t[0], t[1], a0, 0, 0, 0);
        a1 = __builtin_amdgcn_mfma_f32_32x32x2f32(t[2], t[3], a1, 0, 0, 0);
        a2 = __builtin_amdgcn_mfma_f32_32x32x2f32(t[4], t[5], a2, 0, 0, 0);
        a3 = __builtin_amdgcn_mfma_f32_32x32x2f32(t[6], t[7], a3, 0, 0, 0);
        a0 = __builtin_amdgcn_mfma_f32_32x32x2f32(t[1], t[2], a0, 0, 0, 0);
        a1 = __builtin_amdgcn_mfma_f32_32x32x2f32(t[3], t[4], a1, 0, 0, 0);
        a2 = __builtin_amdgcn_mfma_f32_32x32x2f32(t[5], t[6], a2, 0, 0, 0);
        a3 = __builtin_amdgcn_mfma_f32_32x32x2f32(t[7], t[0], a3, 0, 0, 0);
        // decay the accumulators a little so they stay finite and keep toggling
        if ((i & 1023) == 0) { a0 *= 0.5f; a1 *= 0.5f; a2 *= 0.5f; a3 *= 0.5f; }
    }
    float s = 0;
    for (int v = 0; v < 16; ++v) s += a0[v] + a1[v] + a2[v] + a3[v];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}
__global__ __launch_bounds__(256) void mfma_loop(float *out, int iters) {
    f32x16 a0 = {}, a1 = {}, a2 = {}, a3 = {};
    float x = threadIdx.x * 1e-3f, y = 1.0f + threadIdx.x * 1e-4f;
    for (int i = 0; i < iters; ++i) {
        a0 = __builtin_amdgcn_mfma_f32_32x32x2f32(x, y, a0, 0, 0, 0);
        a1 = __builtin_amdgcn_mfma_f32_32x32x2f32(x, y, a1, 0, 0, 0);
        a2 = __builtin_amdgcn_mfma_f32_32x32x2f32(x, y, a2, 0, 0, 0);
        a3 = __builtin_amdgcn_mfma_f32_32x32x2f32(x, y, a3, 0, 0, 0);
    }
    float s = 0;
    for (int v = 0; v < 16; ++v) s += a0[v] + a1[v] + a2[v] + a3[v];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
template <bool RAND>
__global__ __launch_bounds__(256) void mfma_loop_bf16(float *out, int iters) {
    f32x16 a0 = {}, a1 = {}, a2 = {}, a3 = {};
    bf16x8 t[8];
    unsigned h = threadIdx.x * 2654435761u + blockIdx.x * 40503u + 12345u;
    for (int i = 0; i < 8; ++i)
        for (int j = 0; j < 8; ++j) {
            h = h * 1664525u + 1013904223u;
            t[i][j] = (__bf16)(RAND ? ((int)(h >> 8) - (1 << 23)) * (1.0f / (1 << 23)) : 0.5f);
        }
    for (int i = 0; i < iters; i += 2) {
        a0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(t[0], t[1], a0, 0, 0, 0);
        a1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(t[2], t[3], a1, 0, 0, 0);
        a2 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(t[4], t[5], a2, 0, 0, 0);
        a3 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(t[6], t[7], a3, 0, 0, 0);
        a0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(t[1], t[2], a0, 0, 0, 0);
        a1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(t[3], t[4], a1, 0, 0, 0);
        a2 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(t[5], t[6], a2, 0, 0, 0);
        a3 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(t[7], t[0], a3, 0, 0, 0);
        if ((i & 1023) == 0) { a0 *= 0.5f; a1 *= 0.5f; a2 *= 0.5f; a3 *= 0.5f; }
    }
    float s = 0;
    for (int v = 0; v < 16; ++v) s += a0[v] + a1[v] + a2[v] + a3[v];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}
int main(int argc, char **argv) {
    const int mode = argc > 1 ? atoi(argv[1]) : 0;
    float *out;
    hipMalloc(&out, 2048 * 256 * sizeof(float));
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    for (int rep = 0; rep < 4; ++rep) {
        const int iters = rep == 0 ? 2000 : (rep == 3 ? 600000 : 60000), blocks = 256 * 2;
        hipEventRecord(e0);
        if (mode == 2) hipLaunchKernelGGL(mfma_loop_bf16<false>, dim3(blocks), dim3(256), 0, 0, out, iters);
        else if (mode == 3) hipLaunchKernelGGL(mfma_loop_bf16<true>, dim3(blocks), dim3(256), 0, 0, out, iters);
        else if (mode) hipLaunchKernelGGL(mfma_loop_rand, dim3(blocks), dim3(256), 0, 0, out, iters);
        else hipLaunchKernelGGL(mfma_loop, dim3(blocks), dim3(256), 0, 0, out, iters);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        const double kdepth = mode >= 2 ? 16 : 2, per_clk = mode >= 2 ? 1024.0 : 64.0;
        double flops = 2.0 * 32 * 32 * kdepth * 4.0 * iters * 4 /*waves*/ * blocks;
        printf("mode %d rep %d: %.1f ms  %.1f TFLOP/s  (implied clock %.2f GHz at %.0f FLOP/clk/SIMD x 1024 SIMDs)\n", mode, rep, ms,
               flops / ms / 1e9, flops / ms / 1e6 / (per_clk * 1024), per_clk);
    }
    return 0;
}
