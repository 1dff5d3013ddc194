// Shared declarations of the gddim_amd native library (host side).
#pragma once
#include <hip/hip_runtime.h>

#include <cstdarg>
#include <cstdio>
#include <string>

namespace gd {

void set_error(const char *fmt, ...);
int device_cus();  // compute units of the current device (hipDeviceAttributeMultiprocessorCount, cached per device; 256 if no device answers)

#define GD_HIP(expr)                                                                             \
    do {                                                                                         \
        hipError_t _e = (expr);                                                                  \
        if (_e != hipSuccess) {                                                                  \
            gd::set_error("%s failed: %s (%s:%d)", #expr, hipGetErrorString(_e), __FILE__, __LINE__); \
            return 1;                                                                            \
        }                                                                                        \
    } while (0)

#define GD_CHECK(cond, ...)            \
    do {                               \
        if (!(cond)) {                 \
            gd::set_error(__VA_ARGS__); \
            return 1;                  \
        }                              \
    } while (0)

#define GD_TRY(expr)          \
    do {                      \
        int _r = (expr);      \
        if (_r) return _r;    \
    } while (0)

// ---- implicit-GEMM parameters (kernels/igemm.hip) ---------------------------------------------
// C[m][n] = (alpha * sum_k A[m][k] * B[n][k] + bias[n] + rowbias[img(m)][n] + res[m][n]) * out_scale
//   A: implicit im2col of up to two NHWC sources (channel concat), k = (dy*KW + dx)*Cin + c
//   B: [N][K] row-major (weights, or a K-matrix), or k-major [K][N] (b_kmajor)
struct GemmParams {
    const float *a0;
    const float *a1;
    int C0, C1;      // channels taken from a0 / a1 (Cin = C0 + C1), both multiples of 4
    int lda0, lda1;  // pixel stride (floats) of each source
    int H, W;        // input spatial size
    int OH, OW;      // output spatial size
    int KH, KW, stride, pad;
    int M;           // rows (per z-slice): images*OH*OW
    int N, K;        // K = KH*KW*(C0+C1)
    long long sA;    // z-stride of a0 (floats)
    const float *b;
    int ldb;
    int b_kmajor;
    long long sB;
    float *out;
    int ldc;
    long long sC;
    const float *bias;     // [N] or null
    const float *rowbias;  // [(m / rows_per_image) * ld_rowbias + n] or null
    int ld_rowbias;
    int rows_per_image;
    const float *res;  // [m * ldres + n] or null
    int ldres;
    long long sR;      // z-stride of res (floats)
    long long sQ;      // z-stride of qstats (floats)
    float alpha;
    float out_scale;
    int in_act;  // 1: swish applied to A elements on load
    int m_tiles, n_tiles;
    int flat_tiles;  // 1: block id -> (mtile, ntile) row-major; 0: XCD-aware grouping (set by the launchers)
    int z_count;     // persistent kernel: number of z slices (the other kernels read gridDim.z / blockIdx.z)
    const float *zeros;  // >= 16 B of device zeros (filled in by launch_igemm when null)
    // magic numbers for division by OH*OW, OW and rows_per_image (filled in by launch_igemm)
    unsigned mg_ohw, sh_ohw, mg_ow, sh_ow, mg_rpi, sh_rpi;
    int diag;  // diagnostic ablations of the fast kernel (never set by the product path)
    // bf16x3 path: the weights split as bf16 hi / lo arrays, same [N][ldb] indexing as b (null on the fp32 path)
    const unsigned short *b_hi;
    const unsigned short *b_lo;
    // bf16x3 path: 1 = a0 already holds split activations, row m = [lda0 bf16 hi | lda0 bf16 lo] (same bytes per row as fp32),
    // written by the producing GroupNorm-apply / FIR kernel
    int a_presplit;
    // conv3x3_halo_bf16x3 with a fused GroupNorm: a0 is the RAW fp32 input of the normalisation, gn_table[img] = [C scale | C shift]
    // (gn_apply_kernel in table mode), gn_act = 1 applies swish; the kernel normalises, activates and splits while staging
    const float *gn_table;
    int gn_act;
    // fused GroupNorm statistics of the OUTPUT: qstats[img][slice][n/4] = (sum, sum of squares) over the slice's rows of
    // columns 4*(n/4)..+3; slice = 32 output rows (qs_slices = rows_per_image/32 per image), or one image when
    // rows_per_image == 16 (qs_slices = 1).  Unique writer per entry: deterministic.  null = not requested.
    float *qstats;
    int qs_slices;
    // GroupNorm of the OUTPUT in this GEMM's epilogue (fp32 kernels, rows_per_image 16 or 64: one wave's 32 x 32 accumulator tile
    // holds whole images and whole groups of egn_cpg channels): egn_out[m * ldc + n] = act((y - mean) rstd gamma[n] + beta[n]) with
    // (mean, rstd) of (image, group) reduced in registers (flax: var = E[y^2] - E[y]^2, eps 1e-6); egn_skip_raw: the raw output has
    // no other reader and is not stored.  null = off.  (plan.h: Op::egn_peer)
    float *egn_out;
    const float *egn_gamma, *egn_beta;
    int egn_cpg, egn_act, egn_skip_raw;
    // fp32 Winograd kernels (conv3x3_wino2d_f32.hip, conv3x3_wino42_f32.hip): the 3x3 weights pre-transformed by launch_wino2d_weights /
    // launch_wino42_weights into the layout the kernel reads; null = direct kernel
    const float *b_wino;
    int ldb_wino;  // F(2x2,3x3): row stride of U (16 Cin floats; set by its launcher)
    // two-dimensional Winograd kernel with the channel range split over wk_split workgroups per tile (4x4 images, whose few tiles would
    // leave most CUs idle): split s multiplies channels [s, s + 1) * Cin / wk_split and stores its partial outputs -- after the output
    // transform, before bias / residual / scale -- to wk_partial[s][m][n] (row stride N); wino2d_splitk_finish_kernel adds the
    // partials in index order and runs the epilogue.  0 / 1 = off.
    int wk_split;
    float *wk_partial;
    // persistent short-K kernel, 1x1 convolutions on the RAW input of a GroupNorm (the shortcut convolution of a residual block reads
    // what the block's first GroupNorm normalises): the loader also applies that GroupNorm to every A element it stages -- gn_table[img]
    // = [C scale | C shift] (gn_apply_kernel in table mode), gn_act = 1 applies swish -- and stores the result to
    // gn_side_out[m * (C0 + C1) + c]: the separate GroupNorm-apply pass over the same tensor disappears.  null = off.
    float *gn_side_out;
};

int launch_igemm(const GemmParams &p, int batch_z, hipStream_t s);
// true when launch_igemm would run this 1x1 problem on the persistent short-K kernel, whose loader implements gn_side_out
bool igemm_gn_side_ok(const GemmParams &p, int batch_z);
// true when launch_igemm would run this problem on igemm_f32_fast_kernel (any tile, any split), whose epilogue implements egn_out
bool igemm_egn_ok(const GemmParams &p, int batch_z);
const char *igemm_variant_name(const GemmParams &p, int batch_z);
// split-precision (bf16x3) variant for weight GEMMs (kernels/igemm_bf16x3.hip)
bool igemm_bf16x3_ok(const GemmParams &p, int batch_z);
const char *igemm_bf16x3_name(const GemmParams &p);
int launch_igemm_bf16x3(const GemmParams &p, hipStream_t s);
// batched products of two fp32 activation matrices (attention) on the same kernel, both operands split while staging
bool igemm_bf16x3_act_ok(const GemmParams &p, int batch_z);
int launch_igemm_bf16x3_act(const GemmParams &p, int batch_z, hipStream_t s);
// 3x3 stride-1 convolutions on pre-split activations with the input halo staged once per channel slab
// (kernels/conv3x3_halo_bf16x3.hip); GDDIM_X3_NOHALO=1 routes them back to igemm_bf16x3_kernel
bool conv3x3_halo_ok(const GemmParams &p);
bool conv3x3_halo_fused_ok(const GemmParams &p);  // shape test of the fused-GroupNorm variant (p.a0 raw fp32, p.gn_table set)
const char *conv3x3_halo_name();
const char *conv3x3_halo_fused_name();
int launch_conv3x3_halo(const GemmParams &p, hipStream_t s);
// fp32 3x3 stride-1 convolutions through the two-dimensional Winograd transform F(2x2, 3x3) (x 2.25 fewer matrix-core flops than
// the direct sum, kernels/conv3x3_wino2d_f32.hip); p.b_wino = U[n][Cin/8][16][8] from launch_wino2d_weights (16 N Cin floats);
// GDDIM_NOWINO2=1 routes them back to the direct implicit-GEMM kernel
bool conv3x3_wino2d_ok(const GemmParams &p);
bool conv3x3_wino2d_shape_ok(const GemmParams &p);
const char *conv3x3_wino2d_name();
int launch_conv3x3_wino2d(const GemmParams &p, hipStream_t s);
int launch_wino2d_weights(const float *w, float *u, int N, int Cin, hipStream_t s);
// ... on 4x4 images (rows_per_image == 16), split over the channels: the number of splits the launcher would use (0 = not eligible),
// the floats of scratch it needs (wk_split * M * N), and the launch (main kernel + wino2d_splitk_finish_kernel; p.wk_partial set)
int conv3x3_wino2d_splitk_factor(const GemmParams &p);
const char *conv3x3_wino2d_splitk_name();  // (the engine's profile entry: the split launch and its finishing pass)
int launch_conv3x3_wino2d_splitk(const GemmParams &p, hipStream_t s);
// fp32 3x3 stride-1 convolutions through the hybrid Winograd transform F(2,3)_y x F(4,3)_x (24 products per 2 x 4 output tile: x 3 fewer
// matrix-core flops than the direct sum, x 1.33 fewer than F(2x2,3x3); kernels/conv3x3_wino42_f32.hip); p.b_wino = the weights in MFMA
// fragment order from launch_wino42_weights (wino42_weight_floats(N, Cin) floats); GDDIM_NOWINO42=1 routes them back to F(2x2,3x3)
bool conv3x3_wino42_ok(const GemmParams &p);
bool conv3x3_wino42_shape_ok(const GemmParams &p);
const char *conv3x3_wino42_name();
const char *conv3x3_wino42_name_for(const GemmParams &p);
int launch_conv3x3_wino42(const GemmParams &p, hipStream_t s);
int launch_wino42_weights(const float *w, float *u, int N, int Cin, hipStream_t s);
long long wino42_weight_floats(int N, int Cin);
// per-device zero buffer used by the staging loads; allocates on first use (call once outside graph capture)
const float *igemm_zero_buffer();

// ---- normalisation / activation / resampling (kernels/norm_act.hip) -----------------------------
constexpr int GN_MAX_SLICES = 16;
int gn_slices(int batch, int HW);
// quad partial sums of one NHWC tensor: qs[b][slice][c/4][2] = (sum, sumsq) over the slice's pixels (fallback for tensors
// whose producer could not emit them in its epilogue)
int launch_gn_stats(const float *x, int C, float *qs, int batch, int HW, int slices, hipStream_t s);
// normalise (+ swish); (mean, rstd) per (b, group) are reduced in the kernel's prologue from the quad partial sums of up to
// two channel-concatenated sources (S0 / S1 slices per image)
// split_out = 1: out rows are written as [C bf16 hi | C bf16 lo] (the A operand layout of the pre-split bf16x3 GEMM)
// table != null: nothing is normalised; table[b] = [C scale | C shift] is written for a consumer that applies the map itself
int launch_gn_apply(const float *x0, int C0, const float *x1, int C1, const float *qs0, int S0, const float *qs1, int S1,
                    const float *gamma, const float *beta, float *out, int batch, int HW, int groups, int act, float eps,
                    int split_out, float *table, hipStream_t s);
// GroupNorm for group sizes that are not a multiple of 4 channels: direct statistics + element-wise apply
int launch_gn_generic(const float *x0, int C0, const float *x1, int C1, float *mr, const float *gamma, const float *beta,
                      float *out, int batch, int HW, int groups, int act, float eps, hipStream_t s);
int launch_fir_up2(const float *x, float *out, int batch, int H, int W, int C, hipStream_t s);
int launch_naive_resample2(const float *x, float *out, int batch, int H, int W, int C, int up, hipStream_t s);
int launch_fir_down2(const float *x, float *out, int batch, int H, int W, int C, hipStream_t s);
int launch_fir_pad(const float *x, float *out, int batch, int H, int W, int C, hipStream_t s);  // (H, W) -> (H + 1, W + 1), see the kernel
int launch_softmax_rows(float *x, long long rows, int cols, hipStream_t s);
// fused attention core out = (softmax(q k^T C^-1/2) v + bias [+ res]) * out_scale for N = 256 tokens, C = 256 channels
// (kernels/attn_fused_f32.hip): q / k rows of stride ld_q / ld_k, v transposed per image (vT[c][token]), res rows of stride ld_res (null:
// none), qstats (null: none) = the fused GroupNorm statistics of out, 8 slices per image (kernels/qstats.h);
// GDDIM_NO_FUSED_ATTN=1 keeps the three-launch path
bool attn_fused_ok(int N, int C);
const char *attn_fused_name(int batch);
int launch_attn_fused(const float *q, int ld_q, const float *k, int ld_k, const float *vT, const float *bias, const float *res, int ld_res,
                      float out_scale, float *qstats, float *out, int ld_o, int batch, int N, int C, hipStream_t s);
int launch_swish(const float *x, float *out, long long n, hipStream_t s);
// temb0[b][0:E] = sin(2 pi W log(label_b)), [E:2E] = cos(...); label_b = labels[b] or 999*ts[*step]
int launch_fourier(const float *labels, const float *ts_table, const int *step, const float *W, float *out, int batch,
                   int E, int positional, hipStream_t s);
int launch_pad_channels(const float *x, int ld_in, float *out, long long pixels, int Cin, int Cpad, float mul,
                        float add, hipStream_t s);
int launch_scale_rows(float *x, const float *labels, int batch, long long per_image, hipStream_t s);

// ---- sampler update rules (kernels/update.hip) ---------------------------------------------------
int launch_ab_step(const float *u, const float *coef, const float *new_eps, const float *hist_in, float *u_out,
                   float *hist_out, int n_hist, long long n_pairs, hipStream_t s);
int launch_ab_step_scalar(const float *x, const float *coef, const float *new_eps, const float *hist_in, float *x_out,
                          float *hist_out, int n_hist, long long n, hipStream_t s);
int launch_bmm2x2(const float *m, const float *in, float *out, long long n_pairs, int per_sample,
                  long long pairs_per_sample, hipStream_t s);
int launch_dct2d_32(const float *in, float *out, int batch, int channels, int inverse, hipStream_t s);
// CLD fused-sampler kernels on the channel-planar state (B*HW, 8) = [x_0..x_{C-1}, v_0..v_{C-1}, 0..]
int launch_cld_pack(const float *u, float *u8, float *ring, int ring_slots, long long pixels, int C, hipStream_t s);
int launch_cld_unpack_pairs(const float *planar, int ld, float *pairs, long long pixels, int C, hipStream_t s);
// tf (nullable): per-step rows of 12 floats [P_i, E_i, U_i] (2x2 each): U-Net input P_i u, eps <- E_i eps_net + U_i u
// nz (nullable): per-step rows of 12 floats [M_pre, S_pre, S_post]; the step adds S_post xi with xi the counter-based Gaussian
// pair of (seed, elem0 + element, draw 2 step + 1) (kernels/rng.h)
int launch_cld_ab_step(float *u8, const float *eps, float *ring, int ring_slots, const float *coef_table, int n_slots,
                       const int *step, long long pixels, int C, const float *tf, const float *nz, const unsigned *seed,
                       unsigned long long elem0, hipStream_t s);
// u <- M_pre u + S_pre xi (draw 2 step), before the network evaluation of the step
int launch_cld_noise_pre(float *u8, const float *nz, const int *step, const unsigned *seed, unsigned long long elem0,
                         long long pixels, int C, hipStream_t s);
int launch_randn_pairs(float *out, long long n, const unsigned *seed, unsigned draw, unsigned long long elem0, hipStream_t s);
int launch_cld_pre(const float *u8, float *xin, const float *tf, const int *step, long long pixels, int C, hipStream_t s);
// dn: 20 floats F, GG, RinvT, X (mixed score, flags & 1), POST (final frame change, flags & 2)
int launch_cld_finish(const float *u8, const float *eps, const float *dn, float dt, int denoise, int inverse_scale,
                      float *x_out, float *v_out, long long pixels, int C, int flags, hipStream_t s);
// F(2x2,3x3) with the component GEMMs in the exact three-piece bf16 split (kernels/conv3x3_wino2d_split6.hip; precision mode 2 and the stand-alone operator)
bool conv3x3_wino2d_split6_shape_ok(int batch, int H, int W, int Cin, int Cout);
long long conv3x3_wino2d_split6_weight_bytes(int Cin, int Cout);
int launch_wino2d_split6_weights(const float *w, unsigned short *u, int Cout, int Cin, hipStream_t s);
bool conv3x3_wino2d_split6_gemm_ok(const GemmParams &g);  // a GEMM op the kernel takes (bias, per-image bias row, residual, scale, statistics)
int launch_conv3x3_wino2d_split6_gemm(const GemmParams &g, const unsigned short *u, hipStream_t s);
int launch_conv3x3_wino2d_split6(const float *x, const unsigned short *u, const float *bias, const float *res, float *out, int batch, int H,
                                 int W, int Cin, int Cout, float out_scale, hipStream_t s);
int launch_advance_step(int *step, hipStream_t s);
int launch_reset_step_row(int *step, const float *table, float *row, int width, hipStream_t s);  // *step = 0; row = table[0] (row may be null)
int launch_copy_floats(const float *src, float *dst, long long n, hipStream_t s);  // dst[0:n] = src[0:n], a kernel (not the runtime's copy engine)
// step++ and row[0:width] = table[step + 1][0:width] (if step + 1 < rows): the hoisted time-embedding table (engine.cpp)
int launch_advance_step_row(int *step, const float *table, float *row, int width, int rows, hipStream_t s);
// blur fused-sampler kernels
int launch_blur_idct_pad(const float *y, float *x8, int batch, int C, hipStream_t s);
// m2e: optional per-step rows [a, b[32*32]] of sde.model2eps in linear form, eps_y = a DCT(net) + b (.) y (null: identity)
int launch_blur_step(float *y, const float *eps_x, const float *mean_coef, const float *std_coef, const int *step,
                     int batch, int C, const float *m2e, hipStream_t s);
// y' = c0 y + c1 DCT(eps_x) + sum_j c_{2+j} hist_j with per-frequency coefficient planes coef[step][n_slots][32*32]
int launch_blur_ab_step(float *y, const float *eps_x, float *ring, int ring_slots, const float *coef, int n_slots, const int *step,
                        int batch, int C, long long slot_stride, const float *m2e, hipStream_t s);
int launch_blur_finish(const float *y, float *x_out, int batch, int C, int inverse_scale, hipStream_t s);

}  // namespace gd
