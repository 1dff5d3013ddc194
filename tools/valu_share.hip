// What a VALU instruction costs next to fp32 MFMAs on gfx950: the hybrid Winograd kernel's K loop as a register-only skeleton.
//
// 512-thread workgroups, one per CU (two waves per SIMD, as in conv3x3_wino42_f32_kernel); every wave runs "chunks" of 48
// v_mfma_f32_32x32x2_f32 on six accumulators in 24 gaps of two (no MFMA waits for its predecessor's accumulator) and, in every gap,
// NV / 24 VALU instructions of one kind.  Thread 0 of every workgroup stamps the shader clock (s_memtime) and the 100 MHz wall clock
// around the loop; the host prints shader cycles per chunk per SIMD (the MFMAs alone are 96 x 64 = 6 144), the sustained clock and the
// cycles ONE extra VALU instruction costs the SIMD's pipe.  profiles/r05_notes.md holds the table.
// Build: hipcc --offload-arch=gfx950 -O3 tools/valu_share.hip -o tools/valu_share ; run: tools/valu_share
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <vector>

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

enum Kind { NONE = 0, FMA = 1, ADD = 2, DPP = 3, PKFMA = 4, PKADD = 5, MOV = 6, LDSR = 7, LDSW = 8, EXP = 9, RCP = 10, PKMUL = 11 };

// (four independent destination registers in rotation: throughput, not the 4-cycle dependent latency or the DPP read-after-write hazard)
template <int KIND>
__device__ __forceinline__ void valu(float &a, float &b, f32x2 &p, f32x2 &q, float *lds, int lane) {
    if (KIND == FMA) asm volatile("v_fma_f32 %0, %0, 2.0, %1" : "+v"(a) : "v"(b));
    if (KIND == ADD) asm volatile("v_add_f32 %0, %0, %1" : "+v"(a) : "v"(b));
    if (KIND == DPP) asm volatile("v_fmac_f32_dpp %0, %0, %1 quad_perm:[2,2,1,1] row_mask:0xf bank_mask:0xf" : "+v"(a) : "v"(b));
    if (KIND == PKFMA) asm volatile("v_pk_fma_f32 %0, %0, 2.0, %1 op_sel_hi:[1,0,1]" : "+v"(p) : "v"(q));
    if (KIND == PKADD) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(p) : "v"(q));
    if (KIND == MOV) asm volatile("v_mov_b32 %0, %1" : "+v"(a) : "v"(b));
    if (KIND == EXP) asm volatile("v_exp_f32 %0, %1" : "+v"(a) : "v"(b));
    if (KIND == RCP) asm volatile("v_rcp_f32 %0, %1" : "+v"(a) : "v"(b));
    if (KIND == PKMUL) asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(p) : "v"(q));
    if (KIND == LDSR) {
        typedef float f32x4 __attribute__((ext_vector_type(4)));
        f32x4 v = *reinterpret_cast<volatile f32x4 *>(lds + lane * 4);
        asm volatile("" ::"v"(v));
    }
    if (KIND == LDSW) *reinterpret_cast<volatile f32x2 *>(lds + lane * 2) = p;
}

template <int KIND, int PER_GAP>
__global__ __launch_bounds__(512, 2) void skeleton(float *out, unsigned long long *stamps, int chunks) {
    __shared__ float lds[16384];
    f32x16 acc[6] = {};
    const int lane = threadIdx.x & 63;
    unsigned h = threadIdx.x * 2654435761u + blockIdx.x * 40503u + 12345u;
    float t[8];
    for (int i = 0; i < 8; ++i) { h = h * 1664525u + 1013904223u; t[i] = ((int)(h >> 8) - (1 << 23)) * (1.0f / (1 << 23)); }
    float av[4] = {t[0], t[1], t[6], t[7]}, b = t[1] * 1e-3f;
    f32x2 pv[4] = {{t[2], t[3]}, {t[3], t[4]}, {t[4], t[5]}, {t[5], t[6]}}, q = {t[4] * 1e-3f, t[5] * 1e-3f};
    lds[threadIdx.x] = av[0];
    __syncthreads();
    unsigned long long c0 = 0, w0 = 0;
    if (threadIdx.x == 0) { c0 = __builtin_readcyclecounter(); w0 = wall_clock64(); }
    for (int c = 0; c < chunks; ++c) {
#pragma unroll
        for (int jp = 0; jp < 3; ++jp)
#pragma unroll
            for (int g = 0; g < 8; ++g) {
                acc[2 * jp] = __builtin_amdgcn_mfma_f32_32x32x2f32(t[g], t[(g + 1) & 7], acc[2 * jp], 0, 0, 0);
                acc[2 * jp + 1] = __builtin_amdgcn_mfma_f32_32x32x2f32(t[(g + 2) & 7], t[(g + 3) & 7], acc[2 * jp + 1], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int k = 0; k < PER_GAP; ++k) {
                    const int r = (g * PER_GAP + k) & 3;
                    valu<KIND>(av[r], b, pv[r], q, lds + (threadIdx.x >> 6) * 1024, lane);
                }
                __builtin_amdgcn_sched_barrier(0);
            }
        if ((c & 255) == 255) {
#pragma unroll
            for (int k = 0; k < 6; ++k) acc[k] *= 0.5f;
#pragma unroll
            for (int k = 0; k < 4; ++k) { av[k] *= 0.5f; pv[k] *= 0.5f; }
        }
    }
    __syncthreads();  // (the older wave of a SIMD wins the arbitration and would finish early: the stamp is taken when all eight are done)
    if (threadIdx.x == 0) {
        stamps[blockIdx.x * 2] = __builtin_readcyclecounter() - c0;
        stamps[blockIdx.x * 2 + 1] = wall_clock64() - w0;
    }
    float s = 0.f;
    for (int k = 0; k < 4; ++k) s += av[k] + pv[k].x + pv[k].y;
    for (int k = 0; k < 6; ++k)
        for (int v = 0; v < 16; ++v) s += acc[k][v];
    out[blockIdx.x * 512 + threadIdx.x] = s;
}

// The same skeleton on the bf16 matrix core (v_mfma_f32_32x32x16_bf16: 32 cycles per SIMD each, 96 x 32 = 3 072 per chunk): does a VALU
// instruction cost matrix time THERE?  (What a Winograd-domain kernel on split bf16 operands would have to hide: profiles/r05_notes.md.)
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
template <int KIND, int PER_GAP>
__global__ __launch_bounds__(512, 2) void skeleton_bf16(float *out, unsigned long long *stamps, int chunks) {
    __shared__ float lds[16384];
    f32x16 acc[6] = {};
    const int lane = threadIdx.x & 63;
    unsigned h = threadIdx.x * 2654435761u + blockIdx.x * 40503u + 12345u;
    bf16x8 t[8];
    float tf[8];
    for (int i = 0; i < 8; ++i) {
        for (int j = 0; j < 8; ++j) { h = h * 1664525u + 1013904223u; t[i][j] = (__bf16)(((int)(h >> 8) - (1 << 23)) * (1.0f / (1 << 23))); }
        tf[i] = (float)t[i][0];
    }
    float av[4] = {tf[0], tf[1], tf[6], tf[7]}, b = tf[1] * 1e-3f;
    f32x2 pv[4] = {{tf[2], tf[3]}, {tf[3], tf[4]}, {tf[4], tf[5]}, {tf[5], tf[6]}}, q = {tf[4] * 1e-3f, tf[5] * 1e-3f};
    lds[threadIdx.x] = av[0];
    __syncthreads();
    unsigned long long c0 = 0, w0 = 0;
    if (threadIdx.x == 0) { c0 = __builtin_readcyclecounter(); w0 = wall_clock64(); }
    for (int c = 0; c < chunks; ++c) {
#pragma unroll
        for (int jp = 0; jp < 3; ++jp)
#pragma unroll
            for (int g = 0; g < 8; ++g) {
                acc[2 * jp] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(t[g], t[(g + 1) & 7], acc[2 * jp], 0, 0, 0);
                acc[2 * jp + 1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(t[(g + 2) & 7], t[(g + 3) & 7], acc[2 * jp + 1], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int k = 0; k < PER_GAP; ++k) {
                    const int r = (g * PER_GAP + k) & 3;
                    valu<KIND>(av[r], b, pv[r], q, lds + (threadIdx.x >> 6) * 1024, lane);
                }
                __builtin_amdgcn_sched_barrier(0);
            }
        if ((c & 255) == 255) {
#pragma unroll
            for (int k = 0; k < 6; ++k) acc[k] *= 0.5f;
#pragma unroll
            for (int k = 0; k < 4; ++k) { av[k] *= 0.5f; pv[k] *= 0.5f; }
        }
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        stamps[blockIdx.x * 2] = __builtin_readcyclecounter() - c0;
        stamps[blockIdx.x * 2 + 1] = wall_clock64() - w0;
    }
    float s = 0.f;
    for (int k = 0; k < 4; ++k) s += av[k] + pv[k].x + pv[k].y;
    for (int k = 0; k < 6; ++k)
        for (int v = 0; v < 16; ++v) s += acc[k][v];
    out[blockIdx.x * 512 + threadIdx.x] = s;
}

template <int KIND, int PER_GAP>
static double run_bf16(const char *name, float *out, unsigned long long *stamps, double base_cycles) {
    const int blocks = 256, chunks = 40000;
    hipLaunchKernelGGL((skeleton_bf16<KIND, PER_GAP>), dim3(blocks), dim3(512), 0, 0, out, stamps, 2000);
    hipLaunchKernelGGL((skeleton_bf16<KIND, PER_GAP>), dim3(blocks), dim3(512), 0, 0, out, stamps, chunks);
    hipDeviceSynchronize();
    std::vector<unsigned long long> h(blocks * 2);
    hipMemcpy(h.data(), stamps, blocks * 2 * sizeof(unsigned long long), hipMemcpyDeviceToHost);
    double cyc = 0, wall = 0;
    for (int i = 0; i < blocks; ++i) { cyc += (double)h[2 * i]; wall += (double)h[2 * i + 1]; }
    cyc /= blocks; wall /= blocks;
    const double per_chunk = cyc / chunks;
    const int nv = 24 * PER_GAP * 2;
    printf("bf16 MFMA + %-22s %3d per gap: %8.0f cycles per chunk per SIMD (96 x 32 = 3072), %.3f GHz", name, PER_GAP, per_chunk, cyc / (wall * 10.0));
    if (base_cycles > 0 && nv > 0) printf("   -> %.2f cycles per instruction", (per_chunk - base_cycles) / nv);
    printf("\n");
    return per_chunk;
}

template <int KIND, int PER_GAP>
static void run(const char *name, float *out, unsigned long long *stamps, double base_cycles) {
    const int blocks = 256, chunks = 20000;
    hipLaunchKernelGGL((skeleton<KIND, PER_GAP>), dim3(blocks), dim3(512), 0, 0, out, stamps, 2000);  // warm-up
    hipLaunchKernelGGL((skeleton<KIND, PER_GAP>), dim3(blocks), dim3(512), 0, 0, out, stamps, chunks);
    hipDeviceSynchronize();
    std::vector<unsigned long long> h(blocks * 2);
    hipMemcpy(h.data(), stamps, blocks * 2 * sizeof(unsigned long long), hipMemcpyDeviceToHost);
    double cyc = 0, wall = 0;
    for (int i = 0; i < blocks; ++i) { cyc += (double)h[2 * i]; wall += (double)h[2 * i + 1]; }
    cyc /= blocks; wall /= blocks;
    const double per_chunk = cyc / chunks, ghz = cyc / (wall * 10.0) /* wall ticks of 10 ns */, us = wall * 0.01 / chunks;
    const int nv = 24 * PER_GAP * 2;  // VALU instructions per chunk per SIMD (two waves)
    printf("%-28s %3d per gap: %8.0f cycles per chunk per SIMD, %.3f GHz, %.3f us per chunk", name, PER_GAP, per_chunk, ghz, us);
    if (base_cycles > 0 && nv > 0) printf("   -> %.2f cycles per instruction", (per_chunk - base_cycles) / nv);
    printf("\n");
}

static double g_base = 0;

int main() {
    float *out;
    unsigned long long *stamps;
    hipMalloc(&out, 256 * 512 * sizeof(float));
    hipMalloc(&stamps, 256 * 2 * sizeof(unsigned long long));
    {
        // the MFMAs alone: the base line
        const int blocks = 256, chunks = 20000;
        hipLaunchKernelGGL((skeleton<NONE, 0>), dim3(blocks), dim3(512), 0, 0, out, stamps, 2000);
        hipLaunchKernelGGL((skeleton<NONE, 0>), dim3(blocks), dim3(512), 0, 0, out, stamps, chunks);
        hipDeviceSynchronize();
        std::vector<unsigned long long> h(blocks * 2);
        hipMemcpy(h.data(), stamps, blocks * 2 * sizeof(unsigned long long), hipMemcpyDeviceToHost);
        double cyc = 0, wall = 0;
        for (int i = 0; i < blocks; ++i) { cyc += (double)h[2 * i]; wall += (double)h[2 * i + 1]; }
        g_base = cyc / blocks / chunks;
        printf("%-28s            : %8.0f cycles per chunk per SIMD (96 MFMAs x 64 = 6144), %.3f GHz\n", "MFMAs only", g_base, cyc / (wall * 10.0));
    }
    run<FMA, 1>("v_fma_f32", out, stamps, g_base);
    run<FMA, 2>("v_fma_f32", out, stamps, g_base);
    run<FMA, 4>("v_fma_f32", out, stamps, g_base);
    run<ADD, 2>("v_add_f32", out, stamps, g_base);
    run<MOV, 2>("v_mov_b32", out, stamps, g_base);
    run<DPP, 1>("v_fmac_f32_dpp quad_perm", out, stamps, g_base);
    run<DPP, 2>("v_fmac_f32_dpp quad_perm", out, stamps, g_base);
    run<PKFMA, 1>("v_pk_fma_f32", out, stamps, g_base);
    run<PKFMA, 2>("v_pk_fma_f32", out, stamps, g_base);
    run<PKADD, 1>("v_pk_add_f32", out, stamps, g_base);
    run<PKADD, 2>("v_pk_add_f32", out, stamps, g_base);
    run<EXP, 1>("v_exp_f32", out, stamps, g_base);
    run<EXP, 2>("v_exp_f32", out, stamps, g_base);
    run<RCP, 2>("v_rcp_f32", out, stamps, g_base);
    run<PKMUL, 2>("v_pk_mul_f32", out, stamps, g_base);
    run<LDSR, 1>("ds_read_b128", out, stamps, g_base);
    {
        const double b16 = run_bf16<NONE, 0>("nothing", out, stamps, 0);
        run_bf16<FMA, 2>("v_fma_f32", out, stamps, b16);
        run_bf16<FMA, 4>("v_fma_f32", out, stamps, b16);
        run_bf16<FMA, 8>("v_fma_f32", out, stamps, b16);
        run_bf16<PKFMA, 4>("v_pk_fma_f32", out, stamps, b16);
        run_bf16<EXP, 2>("v_exp_f32", out, stamps, b16);
    }
    run<LDSW, 1>("ds_write_b64", out, stamps, g_base);
    return 0;
}
