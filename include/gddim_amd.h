/*
 * gddim_amd -- C ABI of the MI355X-native gDDIM sampling engine.
 *
 * This is the drop-in boundary for the reference's sampler hot path.  The
 * reference (qsh-zh/gDDIM) is pure Python/JAX and has no FFI of its own; every entry point
 * below cites the reference Python interface it stands in for, and the gddim_amd Python package binds it
 * with ctypes under the reference's own function names (INTEGRATION.md shows the stub).
 *
 * Conventions
 *   - every function returns 0 on success, non-zero on error; gd_last_error() gives the text
 *     (thread-local).  Nothing falls back to a CPU path: a call that needs the GPU fails.
 *   - "dev" pointers are device (HBM) pointers, "host" pointers are host memory.  No torch
 *     or other framework types cross this boundary.
 *   - tensors are dense fp32, NHWC; a CLD state u is (B,H,W,C,2) as in the reference
 *     (last axis = (x, v)); the U-Net sees it channel-planar as (B,H,W,2C) =
 *     [x_0..x_{C-1}, v_0..v_{C-1}]  (cld_jax/models/utils.py:153,158).
 *   - `stream` is a hipStream_t passed as void*; NULL = the null stream.
 *   - an engine belongs to one GPU and is not thread-safe; gd_sample_* performs no device
 *     allocation (everything is sized at gd_engine_create for max_batch) so it can be captured
 *     into / replayed from a hipGraph.
 */
#ifndef GDDIM_AMD_H
#define GDDIM_AMD_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct gd_engine gd_engine;

#define GD_MAX_LEVELS 8

/* NCSN++ architecture switches = the fields NCSNpp.__call__ reads from `config`
 * (cld_jax/models/ncsnpp.py:41-69; values in cld_jax/configs/accr_dcifar10_config.py:24-47,
 * blur_jax/configs/ddpm_deep_cifar10_config.py:24-47).  Only the live branches are built:
 * resblock_type='biggan', fir=True with fir_kernel=[1,3,3,1], embedding_type='fourier',
 * progressive='none', nonlinearity='swish'; gd_engine_create rejects anything else. */
typedef struct gd_model_desc {
    int image_size;                       /* config.data.image_size (H = W)                      */
    int in_channels;                      /* U-Net in/out channels: 2*num_channels (CLD) or num_channels (blur) */
    int nf;                               /* config.model.nf                                      */
    int num_res_blocks;                   /* config.model.num_res_blocks                          */
    int num_levels;                       /* len(config.model.ch_mult)                            */
    int ch_mult[GD_MAX_LEVELS];           /* config.model.ch_mult                                 */
    int num_attn_resolutions;             /* len(config.model.attn_resolutions)                   */
    int attn_resolutions[GD_MAX_LEVELS];  /* config.model.attn_resolutions                        */
    int conditional;                      /* config.model.conditional                             */
    int skip_rescale;                     /* config.model.skip_rescale                            */
    int progressive_input_residual;       /* config.model.progressive_input == 'residual' (else 'none') */
    int scale_by_sigma;                   /* config.model.scale_by_sigma                          */
    int centered;                         /* config.data.centered (False: x <- 2x-1, ncsnpp.py:136-138) */
    int max_batch;                        /* largest per-GPU batch this engine will be asked for  */
    int positional_embedding;             /* config.model.embedding_type == 'positional' (else 'fourier'), ncsnpp.py:71-86 */
    int no_fir;                           /* config.model.fir == False: nearest-neighbour / 2x2-mean resampling in the BigGAN
                                             blocks (up_or_down_sampling.py:76-86) instead of the [1,3,3,1] FIR          */
} gd_model_desc;

/* ---- diagnostics ---------------------------------------------------------------------- */
const char *gd_last_error(void);
int gd_version(void);
/* Number of visible HIP devices (0 when there is no GPU / driver). */
int gd_device_count(void);

/* The library's run-time switches (GDDIM_* / GD_* environment variables: A/B measurements and tests, none needed in normal use) live in
 * one table (csrc/options.h), parsed when first needed, at every gd_engine_create, and here.  gd_engine_plan_json prints the table an
 * engine was built under ("options").  No reference counterpart. */
int gd_options_reload(void);

/* ---- host-side SDE math (no GPU) -------------------------------------------------------- */
/* R(t) table of the CLD SDE: fixed-step integration of dR/dt = F R + 1/2 G G^T R^-T from
 * R_0 = diag(sqrt(numerical_eps), sqrt(vv_gamma/m_inv + numerical_eps)), emitting the state
 * before each of n steps (out: n x 4 doubles, row-major 2x2).
 * Replaces: cld_jax/sde_lib.py:93-118 (_get_s_R_fn: lax.scan over scan_body_fn) and
 * cld_jax/deis.py:5-17 (runge_kutta).  Two instantiations of one body that performs the reference's
 * operations in the order its expressions are written, without fused multiply-add:
 *   gd_cld_r_table      float64 -- config.model.x64 = True  (jax_enable_x64, cld_jax/sde_lib.py:112)
 *   gd_cld_r_table_f32  float32 -- config.model.x64 = False, the reference's default
 *                       (cld_jax/configs/default_cifar10_config.py); out: n x 4 floats.
 * The mid-point Euler branch (is_rk = 0, sde_lib.py:103-107) inverts with the closed 2x2 form where the
 * reference calls jnp.linalg.inv (an LU solve). */
int gd_cld_r_table(double m_inv, double beta_0, double beta_1, double vv_gamma, double numerical_eps,
                   int is_rk, double R_dt, size_t n, double *out_host);
int gd_cld_r_table_f32(double m_inv, double beta_0, double beta_1, double vv_gamma, double numerical_eps,
                       int is_rk, double R_dt, size_t n, float *out_host);

/* psi2(t) table of the rotated-frame ("ML") CLD sampler: RK4 scan of d psi2/dt = psi1(t)^-1 F2(t) psi1(t) psi2 from
 * psi2(0) = I with step dt, emitting the state before each of n steps (out: n x 4 doubles).  psi1(t) = expm(int_0^t F1),
 * F = F1 + F2 is the split of cld_jax/sde_lib.py:121-180.
 * Replaces: get_ml_psi2_fn (cld_jax/sampling.py:272-285; lax.scan of runge_kutta, N = 100000). */
int gd_cld_psi2_table(double m_inv, double beta_0, double beta_1, double dt, size_t n, double *out_host);

/* Tables of the lambda-family reverse SDE (cld_jax/sde_lib.py:334-399, class LambdaSDE); r_xp / r_fp are the knots of the R(t)
 * interpolant (r_n times, r_n x 4 values), exactly what get_interp_fn holds.
 *   hat_psi_table: RK4 scan of d psi/dt = \hat F(t) psi from I with step dt on linspace(0, 1 + dt, n, endpoint=False), emitting
 *                  the state before each step (replaces _get_hat_psi_02t, :357-376);
 *   cond_rev_cov:  the 10 000-step RK4 integration of cond_rev_cov(s, t) as coded (:382-399), out4 = 2x2 row-major. */
int gd_cld_lambda_hat_psi_table(double m_inv, double beta_0, double beta_1, double lambda, const double *r_xp,
                                const double *r_fp, size_t r_n, double dt, size_t n, double *out_host);
int gd_cld_lambda_cond_rev_cov(double m_inv, double beta_0, double beta_1, double lambda, const double *r_xp,
                               const double *r_fp, size_t r_n, double s_time, double t_time, int n_step, double *out4);

/* ---- engine life cycle ------------------------------------------------------------------ */
/* device >= 0: build the plan, allocate weights + activation arena on that GPU.
 * device <  0: plan-only (host) mode -- no GPU touched; used to inspect the plan / packed
 *              weights on machines without a GPU.  Compute calls fail in this mode.
 * Replaces: mutils.init_model / model.init (cld_jax/models/utils.py:109-125) as far as
 * shapes go, and the jit/pmap compilation of `sampler` (cld_jax/sampling.py:212-237). */
int gd_engine_create(const gd_model_desc *desc, int device, gd_engine **out);
int gd_engine_destroy(gd_engine *e);

/* Plan introspection: a JSON document listing every op, tensor and weight slot. */
int gd_engine_plan_json(gd_engine *e, char *buf, size_t cap, size_t *needed);
/* Algorithmic FLOPs (2*MAC of every conv/dense/attention contraction) of one U-Net
 * evaluation of one image; conv3x3_only != 0 counts only 3x3 convolutions. */
double gd_engine_flops_per_image(gd_engine *e, int conv3x3_only);

/* ---- weights ------------------------------------------------------------------------------
 * One call per leaf of the flax parameter tree `state.params_ema`
 * (cld_jax/sampling.py:216, cld_jax/models/utils.py:32-40), named by its path joined with '/',
 * e.g. "ResnetBlockBigGANpp_3/Conv_0/kernel"; data is host fp32 in flax layout (conv HWIO,
 * Dense (in,out), NIN W (in,out), vectors).  The engine re-packs into its GEMM layout.
 * Replaces: flax_utils.replicate(state) host->device copy (cld_jax/run_lib.py:711). */
int gd_engine_num_params(gd_engine *e);
int gd_engine_param_info(gd_engine *e, int index, char *name_buf, size_t name_cap, int64_t *shape4, int *ndim);
int gd_engine_load_param(gd_engine *e, const char *name, const float *host_data, const int64_t *shape, int ndim);
/* Fails (listing what is missing) unless every parameter has been loaded. */
int gd_engine_params_ready(gd_engine *e);
/* Export a packed weight slot to host memory (plan-only and GPU mode). */
int gd_engine_packed_weight(gd_engine *e, const char *slot, float *host_out, size_t cap_floats, size_t *n_floats);
/* Bytes of device memory the engine's weights occupy: out4 = {packed GEMM layout, F(2x2,3x3) copies, hybrid F(2,3)y x F(4,3)x copies
 * (MFMA fragment order), bf16 hi + lo copies of the split-precision mode (0 until that mode is used)}.  No reference counterpart. */
int gd_engine_weight_bytes(gd_engine *e, long long out4[4]);

/* Arithmetic of the weight contractions (convolutions, NIN, dense):
 *   0  fp32 operands and accumulation on the fp32 matrix core (v_mfma_f32_32x32x2_f32), the default and the mode every
 *      parity claim refers to; 3x3 stride-1 convolutions large enough to fill the chip go through the Winograd F(2,3)
 *      transform along x (additions in fp32, weights transformed in double; GDDIM_NOWINO=1 keeps the direct sum);
 *   1  "bf16x3": fp32 operands split into bf16 hi + lo, a*b ~= a_hi*b_hi + a_hi*b_lo + a_lo*b_hi on the bf16 matrix
 *      core with fp32 accumulation (per-product error ~2^-16, measured as close to float64 as mode 0 on this network).
 *   2  "split6" (round 6): mode 0 with the 3x3 stride-1 convolutions of the levels whose images are multiples of 16 x 16 pixels through
 *      the two-dimensional Winograd transform F(2x2,3x3) whose component GEMMs run in the EXACT three-piece bf16 split (six bf16 MFMAs per
 *      fp32 product, every dropped term <= 2^-24 of the product, fp32 accumulation): rms error 0.33 of mode 0's hybrid Winograd kernel
 *      against float64; per launch 1.07-1.23x its speed stand-alone, end to end 0.98-1.01 of mode 0's (the modes that keep the bf16
 *      matrix core busy run at a 12-14 % lower shader clock: profiles/r06_notes.md).  Bias, per-image bias row, residual, scale, the fused GroupNorm
 *      statistics and -- on 16x16 images -- the GroupNorm (+ swish) itself in its epilogue.
 * Also selectable with GDDIM_PRECISION=bf16x3 / split6 at engine creation.  The reference has one arithmetic only (XLA fp32);
 * this switch has no counterpart there. */
int gd_engine_set_precision(gd_engine *e, int mode);

/* ---- score network ------------------------------------------------------------------------
 * eps = NCSNpp(x, labels): x_dev (B,H,W,in_channels), labels_dev (B,) = 999*t, out_dev like x.
 * Replaces: model.apply(variables, x, labels, train=False, mutable=False)
 * (cld_jax/models/utils.py:156; NCSNpp.__call__, cld_jax/models/ncsnpp.py:41-243). */
int gd_unet_forward(gd_engine *e, const float *x_dev, const float *labels_dev, float *out_dev, int batch, void *stream);
/* CLD wrapper: u_dev / eps_dev are (B,H,W,C,2); labels = 999*t_dev (t per sample).
 * Replaces: get_eps_fn(...)(u, t) with mixed_score=False (cld_jax/models/utils.py:168-182). */
int gd_cld_eps(gd_engine *e, const float *u_dev, const float *t_dev, float *eps_dev, int batch, void *stream);

/* ---- CLD gDDIM sampler ---------------------------------------------------------------------
 * Install the schedule of one sampler instance:
 *   ts_host[num_step]                 rev_ts[i], the time fed to the network at step i
 *   coef_host[num_step][n_slots][4]   deis_coef[i] = [Psi_i, C_i0 .. C_i(n_slots-2)], 2x2 row-major,
 *                                     n_slots = deis_order + 3 (order0 sampler: n_slots = 2)
 *   denoise != 0: after the loop apply u + F u dt - GG score dt at t_denoise with dt = -t_denoise
 *   dn_F/dn_GG/dn_RinvT: F(t), G(t)@G(t), inv(R(t))^T at t = t_denoise (host, 4 floats each)
 * Replaces: the closure state of _impl_deis_sampler / get_order0_sampler
 * (cld_jax/sampling.py:204-211,156-177): deis_coef = sde.get_deis_coef(order, rev_ts),
 * step_fn = get_denoising_step(...) (cld_jax/sampling.py:30-39). */
int gd_engine_set_schedule_cld(gd_engine *e, int num_step, int n_slots, const float *ts_host, const float *coef_host,
                               int denoise, float t_denoise, const float *dn_F, const float *dn_GG,
                               const float *dn_RinvT);
/* Optional per-step 2x2 maps of the installed CLD schedule (call after gd_engine_set_schedule_cld, which resets them; all
 * arrays host, row-major, NULL = identity / zero):
 *   pre[num_step][4]      the network is evaluated on P_i u instead of u              (mldeis: y2x, cld_jax/sampling.py:349-351)
 *   eps_lin[num_step][4]  eps <- E_i eps_net + U_i u before the AB step               (ldeis: epsR2epsL, sde_lib.py:494-500)
 *   u_lin[num_step][4]    U_i above                                                   (mixed score: models/utils.py:174-176)
 *   dn_u_lin[4]           eps <- eps_net + X u in the denoising step                  (mixed score at t_denoise)
 *   post[4]               u <- POST u after the denoising step, before inverse scaling (mldeis: y2x at eps/2, sampling.py:363)
 * Replaces: the extra closure state of _impl_Ldeis_sampler / get_mldeis_sampler / get_eps_fn(mixed_score=True). */
int gd_engine_set_transforms_cld(gd_engine *e, const float *pre, const float *eps_lin, const float *u_lin,
                                 const float *dn_u_lin, const float *post);
/* Per-step Gaussian noise of the stochastic CLD samplers (call after gd_engine_set_schedule_cld, which resets it; host arrays of
 * num_step x 4 floats, row-major 2x2, NULL = none):
 *   pre_lin, pre_noise   u <- M_i u + S_i xi before the network evaluation of step i  (SSCS: first OU half step,
 *                        cld_jax/sampling.py:542-561,595)
 *   post_noise           + S_i xi' added by the update of step i                       (em: lambda G sqrt|dt| noise, :648-650;
 *                        SSCS: second OU half step, :597)
 * xi, xi' are standard normal pairs from a counter-based generator keyed by (seed of the call, element, step): Philox4x32-10 +
 * Box-Muller (gddim_amd/csrc/kernels/rng.h).  The reference draws from jax's threefry stream, which cannot be reproduced;
 * only the distribution is matched.  gd_engine_set_seed selects the stream of the following gd_sample_cld calls. */
int gd_engine_set_noise_cld(gd_engine *e, const float *pre_lin, const float *pre_noise, const float *post_noise);
int gd_engine_set_seed(gd_engine *e, unsigned long long seed);
/* n_pairs standard normal pairs of that generator: out_dev[2 i], out_dev[2 i + 1] = pair of element elem0 + i, draw `draw`.
 * Replaces: jax.random.normal at the call sites above (stand-alone, for tests and for prior_sampling-style draws). */
int gd_randn_pairs(float *out_dev, long long n_pairs, unsigned long long seed, unsigned draw, unsigned long long elem0,
                   void *stream);
/* One sampler call: u_in_dev (B,H,W,C,2) initial noise -> x_out_dev (B,H,W,C) = (u_x+1)/2 when
 * inverse_scale != 0 else u_x, v_out_dev (B,H,W,C).  use_graph: 0 eager launches, 1 replay a
 * captured hipGraph per step (captured on first use for this batch size), 2 ONE hipGraph for the
 * whole call (pack, every step, the denoising evaluation, finish; input / outputs pass through
 * engine-owned staging buffers because a graph bakes its addresses).
 * Replaces: sampler(rng, state, batch_size, u) (cld_jax/sampling.py:212-230 / :178-193):
 * fori_loop(body_fn) = eps_fn + multistep_ab_step, then step_fn, inverse_scaler.
 * The call only QUEUES work (on the engine's stream, ordered after / before `stream` by events) and leaves the calling thread's
 * current HIP device as it found it: one host thread can drive one engine per GPU back to back -- the reference's
 * jax.pmap(sampler) over jax.local_device_count() devices (cld_jax/sampling.py:232-237; gddim_amd/devices.py).
 * Inside the loop every image carries the step's label 999 ts[step] (cld_jax/sampling.py:217,221), and ts is fixed by
 * gd_engine_set_schedule_*: the time embedding (ncsnpp.py:71-91) and the per-block Dense_0(act(temb)) rows (layerspp.py:215-216)
 * are computed once per knot when the schedule, the parameters or the precision mode change, and the loop reads row `step` of
 * that (knots, sum Cout) table (GDDIM_NO_TEMB_HOIST=1 at engine creation: computed inside every step; same bits). */
int gd_sample_cld(gd_engine *e, int batch, const float *u_in_dev, float *x_out_dev, float *v_out_dev,
                  int inverse_scale, int use_graph, void *stream);

/* ---- blur gDDIM sampler --------------------------------------------------------------------
 *   ts_host[nfe]                     rev_ts[i]
 *   mean_coef_host[nfe+1][32*32]     y_mean_coef(rev_ts[i]) = sqrt(alpha) * D_t per frequency (h,w)
 *   std_coef_host[nfe+1]             y_std_coef(rev_ts[i]) = sqrt(1 - alpha)
 * Replaces: get_order0_sampler closure (blur_jax/sampling.py:53-61) + sde.y_mean_coef /
 * y_std_coef (blur_jax/sde_lib.py:90-97). */
int gd_engine_set_schedule_blur(gd_engine *e, int nfe, const float *ts_host, const float *mean_coef_host,
                                const float *std_coef_host);
/* Multistep (DEIS) schedule of the blur sampler with per-frequency coefficients -- the completion of blur_jax/multistep.py for
 * the blur SDE's per-frequency transition (the reference never runs it, there is no behaviour to match):
 *   coef_host[nfe][n_slots][32*32]: slot 0 = psi_i = m(t_{i+1}) / m(t_i), slot 1 = coefficient of the new eps_y, slots 2.. of the
 *   history (newest first); n_slots = order + 2.  gd_sample_blur then runs y' = c0 y + c1 e_y + sum_j c_{2+j} hist_j per step. */
int gd_engine_set_schedule_blur_deis(gd_engine *e, int nfe, int n_slots, const float *ts_host, const float *coef_host);
/* sde.model2eps(x, t, model_output) (blur_jax/sde_lib.py:150-152, applied in get_eps_fn, blur_jax/models/utils.py:147) in linear
 * form, stated in DCT space where the sampler lives:
 *   rows_host[nfe][1 + 32*32] = [a_i, b_i(h,w)]:  eps_y = a_i * DCT(model_output) + b_i (.) y  at step i
 *   (a scalar b_i, i.e. a constant plane, is eps_x = a_i * model_output + b_i * x; identity = a 1, b 0: the reference's own SDE
 *   class).  Pass NULL to remove it.  Call after gd_engine_set_schedule_blur[_deis]. */
int gd_engine_set_model2eps_blur(gd_engine *e, const float *rows_host);
/* y_in_dev (B,32,32,C) -> x_out_dev (B,32,32,C) = inverse_scaler(IDCT(y_nfe)).  use_graph as in gd_sample_cld:
 * 0 eager, 1 one captured graph per step, 2 ONE hipGraph for the whole call (state load, nfe steps, finish).
 * Replaces: sampler(rng, state, batch_size, u) (blur_jax/sampling.py:55-80). */
int gd_sample_blur(gd_engine *e, int batch, const float *y_in_dev, float *x_out_dev, int inverse_scale,
                   int use_graph, void *stream);

/* ---- profiling -----------------------------------------------------------------------------
 * When enabled, eager launches are bracketed by hipEvents on the launch stream and the
 * accumulated per-kernel-class durations / algorithmic FLOPs / bytes can be read back. */
typedef struct gd_prof_entry {
    char name[128];
    double ms;        /* accumulated device time                   */
    double flops;     /* accumulated algorithmic FLOPs (2*MAC)     */
    double bytes;     /* accumulated algorithmic HBM bytes         */
    long long launches;
} gd_prof_entry;
int gd_engine_profile(gd_engine *e, int enable);
int gd_engine_profile_read(gd_engine *e, gd_prof_entry *out, int cap, int *n);

/* ---- stand-alone operators (no engine; tests + the Python mirror of the reference helpers) ----
 * All pointers are device pointers. */
/* u' = Psi u + sum_o C_o eps_o with full_eps = [new_eps, eps_pred[0..n_hist-1]];
 * coef_dev: (n_hist + 2) x 4 floats [Psi, C_0..C_{n_hist}]; tensors (n, 2) pairs, n = B*H*W*C.
 * hist_out_dev (n_hist, n, 2) receives full_eps[:-1] (may not alias hist_in).
 * Replaces: multistep_ab_step (cld_jax/deis.py:141-151). */
int gd_ab_step(const float *u_dev, const float *coef_dev, const float *new_eps_dev, const float *hist_in_dev,
               float *u_out_dev, float *hist_out_dev, int n_hist, long long n_pairs, void *stream);
/* x' = c_0 x + sum_o c_{o+1} eps_o, scalar coefficients; same history convention.
 * Replaces: ab_step (blur_jax/multistep.py:94-97). */
int gd_ab_step_scalar(const float *x_dev, const float *coef_dev, const float *new_eps_dev, const float *hist_in_dev,
                      float *x_out_dev, float *hist_out_dev, int n_hist, long long n, void *stream);
/* out[..., i] = sum_j m[i][j] in[..., j] on (n,2) pairs; per_sample != 0: m_dev is (B,2,2) and
 * pairs_per_sample gives the grouping.  Replaces: sbmm / bmm (cld_jax/utils.py:179-185). */
int gd_bmm2x2(const float *m_dev, const float *in_dev, float *out_dev, long long n_pairs, int per_sample,
              long long pairs_per_sample, void *stream);
/* Orthonormal 2-D DCT-II (inverse == 0) or its inverse (DCT-III) over the 32x32 plane of every
 * (b, c) of an NHWC tensor (B,32,32,C).
 * Replaces: batch_img_dct / batch_img_idct (blur_jax/blur.py:99-106). */
int gd_dct2d_32(const float *in_dev, float *out_dev, int batch, int channels, int inverse, void *stream);
/* Generic implicit-GEMM convolution (the MFMA kernel), NHWC x [Cout][KH*KW*Cin] -> NHWC:
 * out = (alpha * conv(x) + bias + res) * out_scale.  w_dev is the packed [Cout][KH][KW][Cin] form.
 * Replaces: nn.Conv / lax.conv_general_dilated call sites (cld_jax/models/layers.py:66-107). */
int gd_conv2d(const float *x_dev, const float *w_dev, const float *bias_dev, const float *res_dev, float *out_dev,
              int batch, int H, int W, int Cin, int Cout, int KH, int KW, int stride, int pad, float alpha,
              float out_scale, void *stream);
/* The same convolution through the two-dimensional transform F(2x2, 3x3) (x 2.25 fewer matrix-core products than the direct sum):
 * u_scratch_dev receives the transformed weights, 16 * Cout * Cin floats.  Shapes: W in {8, 16, 32, 64}, H even, H*W/4 % 16 == 0,
 * Cin % 32 == 0, any Cout.  What the engine runs for the 3x3 stride-1 convolutions of the 8x8 .. 64x64 levels at production
 * batch sizes (kernels/conv3x3_wino2d_f32.hip).
 * Replaces: the same nn.Conv call sites (cld_jax/models/layers.py:66-107). */
int gd_conv3x3_winograd2d(const float *x_dev, const float *w_dev, float *u_scratch_dev, const float *bias_dev,
                          const float *res_dev, float *out_dev, int batch, int H, int W, int Cin, int Cout, float alpha,
                          float out_scale, void *stream);
/* The same convolution through the hybrid transform F(2,3) along y x F(4,3) along x (24 products per 2 x 4 output tile: x 3 fewer
 * matrix-core products than the direct sum, x 1.33 fewer than F(2x2,3x3)): u_scratch_dev receives the transformed weights in MFMA
 * fragment order, 24 * (32 * ceil(Cout / 32)) * Cin floats.  Shapes: W in {8, 16, 32, 64}, H even, H*W/8 % 8 == 0, Cin % 32 == 0, any
 * Cout.  What the engine runs for the 3x3 stride-1 convolutions of the 8x8 .. 64x64 levels at production batch sizes in round 3
 * (kernels/conv3x3_wino42_f32.hip; measured fp32 error in profiles/r03_winograd_error_study.md).
 * Replaces: the same nn.Conv call sites (cld_jax/models/layers.py:66-107). */
int gd_conv3x3_winograd42(const float *x_dev, const float *w_dev, float *u_scratch_dev, const float *bias_dev,
                          const float *res_dev, float *out_dev, int batch, int H, int W, int Cin, int Cout, float alpha,
                          float out_scale, void *stream);
/* gd_conv3x3_winograd42 in two steps, as the engine runs it: the Winograd-domain weight copy once (u_dev: the scratch size of
 * gd_conv3x3_winograd42), then the convolution alone (what tools/bench_split6.py times against the split6 operator below). */
int gd_conv3x3_winograd42_weights(const float *w_dev, float *u_dev, int Cin, int Cout, void *stream);
int gd_conv3x3_winograd42_pretransformed(const float *x_dev, const float *u_dev, const float *bias_dev, const float *res_dev, float *out_dev,
                                         int batch, int H, int W, int Cin, int Cout, float alpha, float out_scale, void *stream);
/* The operator behind precision mode 2 (round 6): the same convolution -- nn.Conv(features, (3, 3), padding 1),
 * cld_jax/models/layers.py:95-107 -- through F(2x2,3x3) with every component GEMM in the exact three-piece bf16 split on
 * v_mfma_f32_32x32x16_bf16 (six products per fp32 product, fp32 accumulation; kernels/conv3x3_wino2d_split6.hip).
 *   gd_conv3x3_winograd2d_split6_weights: w_dev [Cout][9 Cin] (gd_conv2d's packing) -> u_dev, ..._weight_bytes(Cin, Cout) bytes of
 *     pre-split fragments (done once per parameter load);
 *   gd_conv3x3_winograd2d_split6: out = (conv(x, w) + bias + res) * out_scale.  H % 16 == 0, W % 16 == 0, Cin % 16 == 0, Cout % 64 == 0. */
long long gd_conv3x3_winograd2d_split6_weight_bytes(int Cin, int Cout);
int gd_conv3x3_winograd2d_split6_weights(const float *w_dev, void *u_dev, int Cin, int Cout, void *stream);
int gd_conv3x3_winograd2d_split6(const float *x_dev, const void *u_dev, const float *bias_dev, const float *res_dev, float *out_dev, int batch,
                                 int H, int W, int Cin, int Cout, float out_scale, void *stream);
/* ... and with the GroupNorm (+ swish) that follows the convolution in its epilogue, as gd_conv3x3_winograd42_gn below (split6 mode's
 * Op::egn_peer path): 16x16 images only (a workgroup tile is a whole image x 64 outputs), Cout / groups = 4, 8 or 16; out_dev null =
 * the convolution output itself is not stored.  Replaces: nn.GroupNorm + swish after conv3x3 (cld_jax/models/layerspp.py:196-220). */
int gd_conv3x3_winograd2d_split6_gn(const float *x_dev, const void *u_dev, const float *bias_dev, const float *res_dev, float *out_dev,
                                    const float *gamma_dev, const float *beta_dev, float *gn_out_dev, int groups, int act, int batch, int H, int W,
                                    int Cin, int Cout, float out_scale, void *stream);
/* gd_conv3x3_winograd42 with the GroupNorm (+ swish) that follows the convolution computed in its epilogue (the engine's Op::egn_peer path): gn_out_dev
 * (B,H,W,Cout) = act(GroupNorm_groups(conv output)) with flax's epsilon 1e-6; out_dev receives the convolution output itself, or is
 * null when nothing else reads it.  Images of 8x8, 16x16 (one workgroup tile) and 32x32 pixels (four tiles walked by one persistent
 * workgroup, Cin >= 64).  Replaces: nn.GroupNorm + swish after conv3x3 (cld_jax/models/layerspp.py:196-220). */
int gd_conv3x3_winograd42_gn(const float *x_dev, const float *w_dev, float *u_scratch_dev, const float *bias_dev, const float *res_dev,
                             float *out_dev, const float *gamma_dev, const float *beta_dev, float *gn_out_dev, int groups, int act,
                             int batch, int H, int W, int Cin, int Cout, float alpha, float out_scale, void *stream);
/* flax GroupNorm(num_groups, eps=1e-6) (+ swish when act != 0) on NHWC (B,H,W,C).
 * Replaces: nn.GroupNorm + nn.swish call sites (cld_jax/models/layerspp.py:69,196,218). */
int gd_group_norm(const float *x_dev, const float *scale_dev, const float *bias_dev, float *out_dev, int batch,
                  int HW, int C, int groups, int act, void *stream);
/* FIR [1,3,3,1] 2x resampling of NHWC tensors.  Replaces: upsample_2d / downsample_2d
 * (cld_jax/models/up_or_down_sampling.py:333-411). */
/* up: 0 FIR down, 1 FIR up, 2 naive (2x2 mean) down, 3 naive (nearest) up */
int gd_fir_resample(const float *x_dev, float *out_dev, int batch, int H, int W, int C, int up, void *stream);
/* Single-head self-attention core on projected q,k,v packed as qkv_dev (B,N,3C):
 * out (B,N,C) = softmax(q k^T C^-1/2) v; scratch_dev must hold B*N*N floats.
 * Replaces: the two einsums + softmax of AttnBlockpp (cld_jax/models/layerspp.py:74-78). */
int gd_attention(const float *qkv_dev, float *scratch_dev, float *out_dev, int batch, int N, int C, void *stream);
/* The same core as ONE launch without the B*N*N scratch, in the operand layout the engine's attention block produces: q and k packed
 * as qk_dev (B,N,2C), v transposed per image vT_dev (B,C,N), optional value bias (C): out (B,N,C) = softmax(q k^T C^-1/2) v + b_v.
 * Built for N = 256, C = 256 (the 16x16 attention blocks of the CIFAR-10 configurations); other sizes are refused.
 * Replaces: the same call site (cld_jax/models/layerspp.py:74-78). */
int gd_attention_fused(const float *qk_dev, const float *vT_dev, const float *bias_v_dev, float *out_dev, int batch, int N, int C,
                       void *stream);
/* The whole tail of AttnBlockpp in that launch, as the engine runs it after folding the projections at parameter-load time (W_q W_k^T
 * into the query projection, W_v W_3 into the value projection: the key is the GroupNorm output itself and NIN_3 disappears):
 *   out (B,N,C) = (res + softmax(q k^T C^-1/2) v + bias) * out_scale,   q_dev, k_dev, res_dev, out_dev (B,N,C) row-major, vT_dev (B,C,N);
 * bias_dev (C), res_dev, qstats_dev optional (null).  qstats_dev (B, N/32, C/4, 2) receives the fused GroupNorm statistics of out: sum
 * and sum of squares over each 32-token slice of every four consecutive channels.  N = C = 256 only.
 * Replaces: cld_jax/models/layerspp.py:74-83 (einsums, softmax, NIN_3, skip_rescale residual). */
int gd_attention_block_fused(const float *q_dev, const float *k_dev, const float *vT_dev, const float *bias_dev, const float *res_dev,
                             float out_scale, float *qstats_dev, float *out_dev, int batch, int N, int C, void *stream);

#ifdef __cplusplus
}
#endif
#endif /* GDDIM_AMD_H */
