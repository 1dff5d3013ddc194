// gddim_amd engine: owns the plan, the packed weights, the activation arena and the sampler state of one GPU,
// and implements the C ABI of include/gddim_amd.h.  No CPU compute fallback exists: every compute entry point
// launches HIP kernels or fails.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdarg>
#include <cstdlib>
#include <cstring>
#include <map>
#include <mutex>
#include <string>
#include <vector>

#include "common.h"
#include "options.h"
#include "plan.h"

namespace gd {

static thread_local std::string g_err;
// Everything that is not pure queueing -- engine creation and destruction, parameter upload, weight transforms, schedules, precision
// switches (hipMalloc / synchronous copies / hipDeviceSynchronize) and every stream capture -- runs under ONE process-wide lock: with
// one issuing thread per device (gddim_amd/devices.py) a capture on one thread was invalidated by such a call on another ("operation
// failed due to a previous error during capture": tests/test_gpu_devices.py, round 6).  A sampler call takes the lock on entry and
// gives it back before it replays its graphs (hipGraphLaunch of a 21 350-node graph is 185 ms of host time: the part that must overlap
// across devices); a warm call holds it for microseconds.
static std::recursive_mutex g_cold;
#define GD_COLD std::lock_guard<std::recursive_mutex> gd_cold_guard(g_cold)

void set_error(const char *fmt, ...) {
    char buf[2048];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    g_err = buf;
}

}  // namespace gd

using namespace gd;

struct ProfAcc {
    double ms = 0, flops = 0, bytes = 0;
    long long launches = 0;
};

struct PendingEvent {
    hipEvent_t a, b;
    std::string name;
    double flops, bytes;
};

struct gd_engine {
    Plan plan;
    const Options *opts = nullptr;  // the option snapshot this engine was created under; in force during every call on it (options.h)
    int device = -1;
    bool plan_only = true;
    bool params_uploaded = false;
    // device memory
    float *arena = nullptr;
    float *wbuf = nullptr;
    // precision mode: 0 = exact fp32 MFMA (default), 1 = bf16x3 split precision for the weight GEMMs, 2 = split6: the 3x3 convolutions of
    // the levels whose images are multiples of 16 x 16 through F(2x2,3x3) in the exact three-piece bf16 split (everything else as mode 0)
    int precision = 0;
    unsigned short *wsplit6 = nullptr;      // pre-split Winograd-domain fragments of those slots (kernels/conv3x3_wino2d_split6.hip)
    std::vector<long long> split6_off;      // byte offsets into wsplit6, -1 = the slot has none
    long long wsplit6_bytes = 0;
    bool wsplit6_valid = false;
    unsigned short *wsplit_hi = nullptr, *wsplit_lo = nullptr;  // bf16 hi / lo of wbuf, same element offsets
    bool wsplit_valid = false;
    // ... and their two-dimensional F(2x2, 3x3) copies (kernels/conv3x3_wino2d_f32.hip), 16 N Cin floats per eligible slot
    float *wwino2 = nullptr;
    float *splitk_ws = nullptr;  // partial outputs of the channel-split 2-D Winograd launches (4x4 level): 8 * max_batch * 16 * max N floats
    long long splitk_ws_floats = 0, splitk_per_image = 0;  // (a lane's sub-batch uses the slice that starts at its first image)
    std::vector<long long> wino2_off;
    // ... and their hybrid F(2,3)_y x F(4,3)_x copies in MFMA fragment order (kernels/conv3x3_wino42_f32.hip), wino42_weight_floats() per slot
    float *wwino42 = nullptr;
    std::vector<long long> wino42_off;
    std::map<int, std::vector<char>> fuse_ok;      // batch size -> per GEMM op: the fused-GroupNorm halo kernel takes it
    std::map<int, std::vector<char>> egn_ok;       // batch size -> per GEMM op: normalises its output in the epilogue (Op::egn_peer)
    std::map<int, std::vector<char>> side_ok;      // batch size -> per GEMM op: writes a GroupNorm of its input from its loader (Op::side_peer)
    std::map<int, std::vector<char>> presplit_ok;  // batch size -> per tensor: may be stored pre-split (see presplit_allowed)
    std::vector<long long> slot_off;  // floats into wbuf
    long long wbuf_floats = 0;
    long long wwino2_floats = 0, wwino42_floats = 0;
    float *xin = nullptr, *epsb = nullptr, *labels = nullptr, *eps_planar = nullptr;
    // sampler state
    float *u8 = nullptr, *ring = nullptr, *ystate = nullptr;
    long long ring_cap = 0;
    float *ts_dev = nullptr, *coef_dev = nullptr, *dn_dev = nullptr, *mean_dev = nullptr, *std_dev = nullptr;
    float *m2e_dev = nullptr;     // blur: per-step [a, b[1024]] of sde.model2eps in linear form (gd_engine_set_model2eps_blur), null = identity
    float *tf_dev = nullptr;      // per-step [P, E, U] rows of the CLD sampler (gd_engine_set_transforms_cld), null = none
    bool tf_pre = false;          // the rows carry a non-identity P: the U-Net reads xin = P u instead of u
    int fin_flags = 0;            // cld_finish: 1 mixed-score term, 2 final frame change
    float *nz_dev = nullptr;      // per-step [M_pre, S_pre, S_post] rows of the stochastic samplers (gd_engine_set_noise_cld)
    bool nz_pre = false;          // the rows carry a state update before the network evaluation
    unsigned *seed_dev = nullptr; // 2 words, rewritten by every sampler call (graph replays read it from memory)
    unsigned long long seed = 0;
    unsigned seed_words[2] = {0, 0};  // host staging of seed_dev (must outlive the async copy)
    int *step_dev = nullptr;  // GD_MAX_LANES counters (one per concurrent sub-batch)
    // Hoisted time embedding (cld_jax/models/ncsnpp.py:71-91, layerspp.py:215-216): inside the fused samplers the label of a
    // step is 999 ts[step] for every image, and ts is fixed when the schedule is installed -- so Fourier -> Dense_0 -> Dense_1 -> the
    // concatenated per-block projection run ONCE per knot (ensure_temb_table, the same kernels on the same inputs: bit-identical rows) and the
    // loop reads row `step` of the (knots, sum Cout) table: the step-advance launch copies the next row into the lane's temb_cur.
    float *temb_tab = nullptr, *temb_cur = nullptr;  // [temb_rows][temb_w]; [MAX_LANES][temb_w]
    int temb_rows = 0, temb_w = 0, temb_t = -1;      // temb_t: the plan tensor every per-image bias row is read from (-1: no hoist)
    long long temb_cap = 0;
    bool temb_valid = false;  // the table matches the installed schedule, weights and precision mode
    int sched_kind = 0;  // 1 cld, 2 blur order0, 3 blur multistep (per-frequency DEIS)
    int num_step = 0, n_slots = 0, denoise = 0;
    float t_denoise = 0;
    // the sampler loop runs on the engine's own stream (graph capture is illegal on the null stream); it is
    // ordered after / before the caller's stream with two events
    // A sampler call is split into `lanes` independent sub-batches, each on its own stream: while one sub-batch is in a
    // phase that leaves the matrix pipe idle (GroupNorm and other HBM-bound kernels, the prologue/epilogue round of a GEMM
    // launch) the other one computes.  GD_LANES=1 disables the split.
    static constexpr int MAX_LANES = 4;
    hipStream_t stream[MAX_LANES] = {nullptr, nullptr, nullptr, nullptr};
    hipEvent_t ev_in = nullptr, ev_out[MAX_LANES] = {nullptr, nullptr, nullptr, nullptr};
    int lanes = 1;  // measured on MI355X (B=256): 2 lanes overlap in the trace but change throughput by < 1 %
    // graphs keyed by (sub-batch size, first image, lane)
    struct Graphs { hipGraphExec_t step = nullptr, fin = nullptr, loop = nullptr; };
    // whole-loop graphs (use_graph = 2) bake every address, so the call's input / outputs pass through engine-owned staging buffers
    float *uin_stage = nullptr, *xout_stage = nullptr, *vout_stage = nullptr;
    std::map<long long, Graphs> graphs;
    // profiling
    bool prof = false;
    std::map<std::string, ProfAcc> prof_acc;
    std::vector<PendingEvent> pending;
};

static const float *slot_ptr(gd_engine *e, int slot) { return slot < 0 ? nullptr : e->wbuf + e->slot_off[slot]; }

static void drop_graphs(gd_engine *e) {
    for (auto &kv : e->graphs) {
        if (kv.second.step) (void)hipGraphExecDestroy(kv.second.step);
        if (kv.second.fin) (void)hipGraphExecDestroy(kv.second.fin);
        if (kv.second.loop) (void)hipGraphExecDestroy(kv.second.loop);
    }
    e->graphs.clear();
}

struct Bind {
    const float *x_in;
    const float *labels;  // nullptr: labels = 999 * ts_dev[*step]
    float *eps_out;
    float *arena;         // activation arena of this sub-batch
    const int *step;      // device step counter of this sub-batch
    const float *temb_row = nullptr;  // the current row of the hoisted time-embedding table (null: the embedding ops run in the loop)
};

static float *tptr(gd_engine *e, const Bind &b, const TRef &r, int batch) {
    if (r.t < 0) return nullptr;
    const Tensor &t = e->plan.tensors[r.t];
    float *base;
    switch (t.ext) {
        case EXT_X_IN: base = const_cast<float *>(b.x_in); break;
        case EXT_LABELS: base = const_cast<float *>(b.labels); break;
        case EXT_EPS_OUT: base = b.eps_out; break;
        default: base = b.arena + (size_t)t.offset * batch; break;
    }
    return base ? base + r.col : nullptr;
}

struct ProfScope {
    gd_engine *e;
    hipStream_t s;
    PendingEvent ev;
    bool on;
    ProfScope(gd_engine *e_, hipStream_t s_, const std::string &name, double flops, double bytes) : e(e_), s(s_), on(e_->prof) {
        if (!on) return;
        ev.name = name; ev.flops = flops; ev.bytes = bytes;
        (void)hipEventCreate(&ev.a);
        (void)hipEventCreate(&ev.b);
        (void)hipEventRecord(ev.a, s);
    }
    ~ProfScope() {
        if (!on) return;
        (void)hipEventRecord(ev.b, s);
        e->pending.push_back(ev);
    }
};

// GD_PROF_BY_LABEL=1: profile entries are keyed by the op's plan label with digits removed (e.g. "AttnBlockpp_.qk")
static std::string label_class(const Op &op) {
    std::string r;
    for (char c : op.label)
        if (c < '0' || c > '9') r += c;
    char buf[64];
    snprintf(buf, sizeof buf, "@%dx%d:%d>%d", op.H, op.W, op.C0 + op.C1, op.N);
    return r + buf;
}

// The key of a profile entry: the kernel's name (default), the op's label class (GD_PROF_BY_LABEL=1), or both, "class|kernel"
// (GD_PROF_BY_LABEL=2: which kernel every level of the network ran on, tools/batch_sweep.py).  Empty when the profiler is off.
static std::string pkey(const gd_engine *e, const Op &op, const std::string &kernel, const char *class_prefix = "") {
    if (!e->prof) return std::string();
    switch (opts().prof_key) {
        case 1: return class_prefix + label_class(op);
        case 2: return class_prefix + label_class(op) + "|" + kernel;
        default: return kernel;
    }
}

// pointer of a tensor reference advanced to image i0 of the batch (tensors are stored image after image, per_image floats each)
static float *tptr_at(gd_engine *e, const Bind &b, const TRef &r, int batch, int i0) {
    float *p = tptr(e, b, r, batch);
    return p && i0 ? p + (size_t)i0 * e->plan.tensors[r.t].per_image : p;
}

// The buffer-addressed kernels (Winograd, fast / persistent GEMM) reach 2 GiB per operand.  A weight GEMM over a larger tensor -- the
// 256-channel tensors of the 64x64 level at batch 512 are exactly 2 GiB -- runs as several launches over image sub-ranges instead of
// falling back to the 64-bit-addressed generic kernel (images are independent rows of the GEMM).  Returns the number of launches.
static int gemm_parts(const gd_engine *e, const Op &op, int batch) {
    if (e->precision == 1 || op.kind != OP_GEMM || op.z_per_image || op.w_slot < 0 || op.a_slot >= 0 || batch < 2) return 1;
    long long per = (long long)op.H * op.W * std::max(op.lda0, op.C1 > 0 ? op.lda1 : 0) * 4;
    per = std::max(per, (long long)op.m_per_image * op.ldc * 4);
    if (op.res.t >= 0) per = std::max(per, (long long)op.m_per_image * op.ldres * 4);
    const long long limit = (1LL << 31) - (8LL << 20);  // (the kernels' own tests add halo rows and a few cache lines)
    int parts = 1;
    while (parts < batch && (long long)((batch + parts - 1) / parts) * per >= limit) ++parts;
    return parts;
}

// GemmParams of a GEMM op for images [i0, i0 + nb) of this batch (nb < 0: the whole batch); returns the z-batch count
static int fill_gemm(gd_engine *e, const Op &op, const Bind &b, int batch, GemmParams &g, int i0 = 0, int nb = -1) {
    if (nb < 0) nb = batch;
    g.a0 = op.a_slot >= 0 ? slot_ptr(e, op.a_slot) + op.a_off : tptr_at(e, b, op.a0, batch, i0);
    g.a1 = tptr_at(e, b, op.a1, batch, i0);
    g.C0 = op.C0; g.C1 = op.C1; g.lda0 = op.lda0; g.lda1 = op.lda1 ? op.lda1 : 4;
    g.H = op.H; g.W = op.W; g.OH = op.OH; g.OW = op.OW; g.KH = op.KH; g.KW = op.KW; g.stride = op.stride; g.pad = op.pad;
    g.N = op.N; g.K = op.K;
    int z = 1;
    if (op.z_per_image) { g.M = op.m_per_image; z = nb; g.sA = op.sA; g.sB = op.sB; g.sC = op.sC; }
    else g.M = op.m_per_image * nb;
    if (op.w_slot >= 0) g.b = slot_ptr(e, op.w_slot);
    else g.b = tptr_at(e, b, op.bten, batch, i0);
    g.ldb = op.ldb; g.b_kmajor = op.b_kmajor;
    g.out = tptr_at(e, b, op.out, batch, i0); g.ldc = op.ldc;
    g.bias = op.bias_slot >= 0 ? slot_ptr(e, op.bias_slot) + op.bias_off : nullptr;
    // the per-image bias rows are the time-embedding projections: inside the fused samplers every image of the batch carries the same
    // label (b.labels == nullptr: 999 t of the step), one row is computed (run_op) and every image reads it (row stride 0)
    const bool one_label = b.labels == nullptr && op.rowbias.t >= 0;
    g.rowbias = one_label ? (b.temb_row ? const_cast<float *>(b.temb_row) + op.rowbias.col : tptr(e, b, op.rowbias, batch))
                          : tptr_at(e, b, op.rowbias, batch, i0);
    g.ld_rowbias = one_label ? 0 : op.ld_rowbias; g.rows_per_image = op.m_per_image;
    g.res = tptr_at(e, b, op.res, batch, i0); g.ldres = op.ldres; g.sR = op.sR; g.sQ = op.sQ;
    g.alpha = op.alpha; g.out_scale = op.out_scale; g.in_act = op.in_act;
    g.qstats = tptr_at(e, b, op.qstats, batch, i0); g.qs_slices = op.qs_slices;
    return z;
}

// ... of the launch that covers the first image: what the per-batch-size kernel choices below are made on
static int fill_gemm_first(gd_engine *e, const Op &op, const Bind &b, int batch, GemmParams &g) {
    const int parts = gemm_parts(e, op, batch);
    return fill_gemm(e, op, b, batch, g, 0, (batch + parts - 1) / parts);
}

// ALGORITHMIC HBM bytes of a GEMM launch (bench.py: roofline.algorithmic_bytes): every operand once -- the input pixels (not the
// im2col expansion), the B matrix / weights, the residual, the output (raw and / or normalised copy)
static double gemm_alg_bytes(const GemmParams &g, int z) {
    const double in_rows = g.KH * g.KW > 1 ? (double)(g.M / (g.OH * g.OW > 0 ? g.OH * g.OW : 1)) * g.H * g.W : (double)g.M;
    double fl = in_rows * (g.C0 + g.C1) + (double)g.N * g.K;
    fl += (double)g.M * g.N * ((g.egn_skip_raw ? 0 : 1) + (g.res ? 1 : 0) + (g.egn_out ? 1 : 0));
    if (g.gn_side_out) fl += (double)g.M * (g.C0 + g.C1);
    return 4.0 * z * fl;
}

// A tensor is stored pre-split (bf16x3 mode) only when every GEMM that reads it is eligible for the split-precision kernels at
// THIS batch size (the 2 GiB buffer-addressing limit depends on it); decided once per batch size, per tensor.
static bool presplit_allowed(gd_engine *e, const Bind &b, int batch, int tensor) {
    auto it = e->presplit_ok.find(batch);
    if (it == e->presplit_ok.end()) {
        std::vector<char> ok(e->plan.tensors.size(), 1);
        for (const Op &op : e->plan.ops) {
            if (op.kind != OP_GEMM || !op.a_split) continue;
            GemmParams g{};
            const int z = fill_gemm(e, op, b, batch, g);
            g.b_hi = e->wsplit_hi + e->slot_off[op.w_slot];
            g.b_lo = e->wsplit_lo + e->slot_off[op.w_slot];
            g.a_presplit = 1;
            if (!igemm_bf16x3_ok(g, z)) ok[op.a0.t] = 0;
        }
        it = e->presplit_ok.emplace(batch, std::move(ok)).first;
    }
    return tensor >= 0 && it->second[tensor] != 0;
}

// Fused GroupNorm (plan.h: Op::fuse_peer): taken when, at this batch size, the halo kernel accepts the convolution and the
// tensor between the two ops would otherwise be stored pre-split.  gemm_index: the convolution's op index.  Decided once per batch.
static bool gn_fused(gd_engine *e, const Bind &b, int batch, int gemm_index) {
    if (e->precision != 1 || !e->wsplit_valid) return false;
    auto it = e->fuse_ok.find(batch);
    if (it == e->fuse_ok.end()) {
        std::vector<char> ok(e->plan.ops.size(), 0);
        for (size_t i = 0; i < e->plan.ops.size(); ++i) {
            const Op &op = e->plan.ops[i];
            if (op.kind != OP_GEMM || op.fuse_peer < 0 || !op.a_split) continue;
            if (!presplit_allowed(e, b, batch, op.a0.t)) continue;
            GemmParams g{};
            const int z = fill_gemm(e, op, b, batch, g);
            g.b_hi = e->wsplit_hi + e->slot_off[op.w_slot];
            g.b_lo = e->wsplit_lo + e->slot_off[op.w_slot];
            g.a0 = tptr(e, b, op.gn_src, batch); g.a1 = tptr(e, b, op.gn_src1, batch);
            g.C0 = op.gn_C0; g.C1 = op.gn_C1; g.lda0 = op.gn_C0; g.lda1 = op.gn_C1 ? op.gn_C1 : 4;
            ok[i] = (z == 1 && conv3x3_halo_fused_ok(g)) ? 1 : 0;
        }
        it = e->fuse_ok.emplace(batch, std::move(ok)).first;
    }
    return it->second[gemm_index] != 0;
}

// Epilogue GroupNorm (plan.h: Op::egn_peer), fp32 mode: taken when, at this batch size, the kernel the GEMM runs on implements it
// (the Winograd kernels on 16x16 / 8x8 images, the finishing pass of the channel-split launch and igemm_f32_fast_kernel on 4x4 images).
// Fills the egn_* fields of g.  Decided once per batch size; GDDIM_NO_EGN=1 switches it off (A/B).
static void fill_egn(gd_engine *e, const Op &op, const Bind &b, int batch, GemmParams &g, int i0 = 0) {
    const Op &a = e->plan.ops[op.egn_peer];
    g.egn_out = tptr_at(e, b, a.out, batch, i0);
    g.egn_gamma = slot_ptr(e, a.gamma_slot);
    g.egn_beta = slot_ptr(e, a.beta_slot);
    g.egn_cpg = a.C0 / a.groups;
    g.egn_act = a.act;
    g.egn_skip_raw = op.egn_sole;
}

// split6 mode: the convolution goes to the split6 kernel (g without the egn_* fields: egn_fused decides those for the kernel that runs)
static bool split6_takes(const gd_engine *e, const Op &op, const GemmParams &g, int z) {
    return e->precision == 2 && e->wsplit6_valid && z == 1 && op.w_slot >= 0 && op.w_slot < (int)e->split6_off.size() &&
           e->split6_off[op.w_slot] >= 0 && conv3x3_wino2d_split6_gemm_ok(g);
}

static bool egn_fused(gd_engine *e, const Bind &b, int batch, int gemm_index) {
    if (opts().no_egn || e->precision == 1 || gemm_index < 0) return false;
    auto it = e->egn_ok.find(batch);
    if (it == e->egn_ok.end()) {
        std::vector<char> ok(e->plan.ops.size(), 0);
        for (size_t i = 0; i < e->plan.ops.size(); ++i) {
            const Op &op = e->plan.ops[i];
            if (op.kind != OP_GEMM || op.egn_peer < 0) continue;
            GemmParams g{};
            const int z = fill_gemm_first(e, op, b, batch, g);
            if (split6_takes(e, op, g, z)) {  // split6 mode: its kernel normalises whole 16 x 16 images; elsewhere the stand-alone pass runs
                fill_egn(e, op, b, batch, g);
                ok[i] = conv3x3_wino2d_split6_gemm_ok(g) ? 1 : 0;
                continue;
            }
            fill_egn(e, op, b, batch, g);
            const bool wino2 = z == 1 && op.w_slot >= 0 && e->wwino2 && e->wino2_off[op.w_slot] >= 0;
            const bool wino42 = z == 1 && op.w_slot >= 0 && e->wwino42 && e->wino42_off[op.w_slot] >= 0;
            if (wino42 && conv3x3_wino42_ok(g)) { ok[i] = 1; continue; }  // (its shape test covers the egn_* fields)
            if (wino2 && conv3x3_wino2d_ok(g)) ok[i] = 1;  // (its shape test covers the egn_* fields)
            else if (wino2 && e->splitk_ws && conv3x3_wino2d_splitk_factor(g) > 1) ok[i] = 1;  // (the finishing pass normalises)
            else ok[i] = igemm_egn_ok(g, z) ? 1 : 0;
        }
        it = e->egn_ok.emplace(batch, std::move(ok)).first;
    }
    return it->second[gemm_index] != 0;
}

// GroupNorm as a side output of the shortcut convolution (plan.h: Op::side_peer), fp32 mode: taken when, at this batch size, the 1x1
// convolution runs on the persistent short-K kernel and the GroupNorm is not already written by its producer's epilogue.  Decided once
// per batch size; GDDIM_NO_SIDE_GN=1 switches it off (A/B).
static void fill_side(gd_engine *e, const Op &op, const Bind &b, int batch, GemmParams &g, int i0 = 0) {
    const Op &a = e->plan.ops[op.side_peer];
    g.gn_table = tptr_at(e, b, a.gn_table, batch, i0);
    g.gn_act = a.act;
    g.gn_side_out = tptr_at(e, b, a.out, batch, i0);
}

static bool side_fused(gd_engine *e, const Bind &b, int batch, int gemm_index) {
    if (opts().no_side_gn || e->precision == 1 || gemm_index < 0) return false;
    auto it = e->side_ok.find(batch);
    if (it == e->side_ok.end()) {
        std::vector<char> ok(e->plan.ops.size(), 0);
        for (size_t i = 0; i < e->plan.ops.size(); ++i) {
            const Op &op = e->plan.ops[i];
            if (op.kind != OP_GEMM || op.side_peer < 0) continue;
            const Op &a = e->plan.ops[op.side_peer];
            if (a.kind != OP_GN_APPLY || a.a0.t != op.a0.t || a.a1.t != op.a1.t || a.C0 != op.C0 || a.C1 != op.C1 || a.a0.col != 0 || op.a0.col != 0 ||
                a.gn_table.t < 0 || e->plan.tensors[a.out.t].ext != EXT_NONE)
                continue;
            // the affine table is built at the convolution's op index: the GroupNorm's statistics must already exist there, i.e. no
            // stand-alone statistics op may sit between the two (the apply op directly follows the convolution, both sources with
            // statistics from their producers' epilogues)
            if (op.side_peer != (int)i + 1 || a.stats.t < 0 || (a.C1 > 0 && a.stats1.t < 0)) continue;
            if (a.egn_peer >= 0 && egn_fused(e, b, batch, a.egn_peer)) continue;  // its producer writes it already
            GemmParams g{};
            const int z = fill_gemm_first(e, op, b, batch, g);
            fill_side(e, op, b, batch, g);
            ok[i] = igemm_gn_side_ok(g, z) ? 1 : 0;
        }
        it = e->side_ok.emplace(batch, std::move(ok)).first;
    }
    return it->second[gemm_index] != 0;
}

static int launch_apply_op(gd_engine *e, const Op &op, const Bind &b, int batch, bool table_only, hipStream_t s);

// The attention core q k^T -> softmax -> p v as ONE launch (kernels/attn_fused_f32.hip), fp32 mode, 256 tokens x 256 channels: op i is
// the q k^T product of such a core when it is a per-image product of two row-major activation matrices (two column blocks of one tensor,
// or -- the planner's folded form -- the folded query projection and the GroupNorm output), followed by the in-place softmax of its
// output and by the product of that output with a k-contiguous V^T (whose epilogue may carry the block's residual, scale and statistics).
static bool attn_core_at(const gd_engine *e, int i) {
    const std::vector<Op> &ops = e->plan.ops;
    if (e->precision == 1 || i < 0 || i + 2 >= (int)ops.size()) return false;
    const Op &qk = ops[i], &sm = ops[i + 1], &av = ops[i + 2];
    if (qk.kind != OP_GEMM || sm.kind != OP_SOFTMAX || av.kind != OP_GEMM) return false;
    if (!qk.z_per_image || qk.w_slot >= 0 || qk.a_slot >= 0 || qk.bten.t < 0 || qk.b_kmajor ||
        qk.C1 != 0 || qk.K != qk.C0 || qk.m_per_image != qk.N || qk.res.t >= 0 || qk.rowbias.t >= 0 || qk.bias_slot >= 0 ||
        qk.qstats.t >= 0)
        return false;
    if (sm.out.t != qk.out.t || sm.cols != qk.N || sm.rows_per_image != qk.N) return false;
    if (!av.z_per_image || av.w_slot >= 0 || av.a_slot >= 0 || av.a0.t != qk.out.t || av.C0 != qk.N || av.K != qk.N || av.N != qk.C0 || av.b_kmajor ||
        av.ldb != qk.N || av.bten.col != 0 || av.rowbias.t >= 0 || av.alpha != 1.f || av.out.col != 0)
        return false;
    // what launch_attn_fused hard-codes: no activation on load, a single source, q / k rows of one image contiguous at their strides
    // (image stride N * ld), V^T as [C][N] per image (image stride C * N), scores of one image N x N, residual / output rows of one image
    // contiguous, eight statistics slices per image, and no other reader of the scores
    if (qk.in_act || av.in_act || av.C1 != 0 || av.lda0 != qk.N || qk.sA != (long long)qk.N * qk.lda0 || qk.sB != (long long)qk.N * qk.ldb ||
        qk.sC != (long long)qk.N * qk.N || av.sA != (long long)qk.N * qk.N || av.sB != (long long)qk.C0 * qk.N || qk.ldc != qk.N || qk.out.col != 0 ||
        av.sC != (long long)qk.N * av.ldc)
        return false;
    if (av.res.t >= 0 && av.sR != (long long)qk.N * av.ldres) return false;
    if (av.qstats.t >= 0 && (av.qs_slices != qk.N / 32 || av.sQ != (long long)av.qs_slices * (av.N / 4) * 2)) return false;
    for (int j = 0; j < (int)ops.size(); ++j) {
        if (j == i + 1 || j == i + 2) continue;
        const Op &o = ops[j];
        if (j != i && (o.a0.t == qk.out.t || o.a1.t == qk.out.t || o.bten.t == qk.out.t || o.res.t == qk.out.t)) return false;
    }
    const float want = 1.0f / std::sqrt((float)qk.C0);
    if (std::fabs(qk.alpha - want) > 1e-6f * want || qk.out_scale != 1.f) return false;
    return attn_fused_ok(qk.N, qk.C0);
}

static int launch_apply_op(gd_engine *e, const Op &op, const Bind &b, int batch, bool table_only, hipStream_t s) {
    const int HW = op.H * op.W;
    const int rt = gn_slices(batch, HW);
    ProfScope ps(e, s, pkey(e, op, std::string(table_only ? "gn_table" : "gn_apply"), table_only ? "gn_table:" : "gn_apply:"), 0,
                 table_only ? 0.0 : 8.0 * batch * HW * (op.C0 + op.C1));
    return launch_gn_apply(tptr(e, b, op.a0, batch), op.C0, tptr(e, b, op.a1, batch), op.C1, tptr(e, b, op.stats, batch),
                           op.S0 > 0 ? op.S0 : rt, tptr(e, b, op.stats1, batch), op.S1 > 0 ? op.S1 : rt,
                           slot_ptr(e, op.gamma_slot), slot_ptr(e, op.beta_slot), tptr(e, b, op.out, batch), batch, HW,
                           op.groups, op.act, 1e-6f,
                           (e->precision == 1 && e->wsplit_valid && op.split_out && presplit_allowed(e, b, batch, op.out.t)) ? 1 : 0,
                           table_only ? tptr(e, b, op.gn_table, batch) : nullptr, s);
}

// One launch of a GEMM op over images [i0, i0 + nb) of the batch (run_op: the whole batch, or one of gemm_parts sub-ranges).  The
// kernel choices that other ops depend on (egn_fused, side_fused, gn_fused) are keyed by the FULL batch size.
static int run_gemm(gd_engine *e, int op_index, const Bind &b, int batch, int i0, int nb, bool first, hipStream_t s) {
    const Op &op = e->plan.ops[op_index];
    const Plan &P = e->plan;
    GemmParams g{};
    const int z = fill_gemm(e, op, b, batch, g, i0, nb);
    if (e->precision == 1 && op.w_slot >= 0 && e->wsplit_valid) {
        g.b_hi = e->wsplit_hi + e->slot_off[op.w_slot];
        g.b_lo = e->wsplit_lo + e->slot_off[op.w_slot];
        g.a_presplit = op.a_split && presplit_allowed(e, b, batch, op.a0.t);
        GD_CHECK(!g.a_presplit || igemm_bf16x3_ok(g, z), "%s: pre-split input but the GEMM is not eligible for bf16x3", op.label.c_str());
        if (op.fuse_peer >= 0 && gn_fused(e, b, batch, op_index)) {
            g.a0 = tptr(e, b, op.gn_src, batch);  // the raw input(s) of the GroupNorm
            g.a1 = tptr(e, b, op.gn_src1, batch);
            g.C0 = op.gn_C0; g.C1 = op.gn_C1; g.lda0 = op.gn_C0; g.lda1 = op.gn_C1 ? op.gn_C1 : 4;
            g.a_presplit = 0;
            g.gn_table = tptr(e, b, op.gn_table, batch);
            g.gn_act = e->plan.ops[op.fuse_peer].act;
            ProfScope ps(e, s, pkey(e, op, std::string(conv3x3_halo_fused_name())), op.flops_per_image * nb, gemm_alg_bytes(g, z));
            return launch_conv3x3_halo(g, s);
        }
        if (z == 1 && conv3x3_halo_ok(g)) {
            ProfScope ps(e, s, pkey(e, op, std::string(conv3x3_halo_name())), op.flops_per_image * nb, gemm_alg_bytes(g, z));
            return launch_conv3x3_halo(g, s);
        }
        if (igemm_bf16x3_ok(g, z)) {
            ProfScope ps(e, s, pkey(e, op, std::string(igemm_bf16x3_name(g))), op.flops_per_image * nb, gemm_alg_bytes(g, z));
            return launch_igemm_bf16x3(g, s);
        }
        g.b_hi = g.b_lo = nullptr;
    }
    GD_CHECK(!g.a_presplit, "%s: pre-split input without a split-precision kernel", op.label.c_str());
    if (op.egn_peer >= 0 && egn_fused(e, b, batch, op_index)) fill_egn(e, op, b, batch, g, i0);
    if (e->precision == 1 && op.w_slot < 0 && igemm_bf16x3_act_ok(g, z)) {  // attention products: activation x activation
        ProfScope ps(e, s, pkey(e, op, std::string("igemm_bf16x3_kernel<256,128,4,2,act>")), op.flops_per_image * nb, gemm_alg_bytes(g, z));
        return launch_igemm_bf16x3_act(g, z, s);
    }
    if (split6_takes(e, op, g, z)) {
        ProfScope ps(e, s, pkey(e, op, std::string(g.egn_out ? "conv3x3_wino2d_split6_kernel+gn" : "conv3x3_wino2d_split6_kernel")), op.flops_per_image * nb, gemm_alg_bytes(g, z));
        return launch_conv3x3_wino2d_split6_gemm(g, reinterpret_cast<const unsigned short *>(reinterpret_cast<const char *>(e->wsplit6) + e->split6_off[op.w_slot]), s);
    }
    if (e->precision != 1 && z == 1 && op.w_slot >= 0 && e->wwino42 && e->wino42_off[op.w_slot] >= 0 && conv3x3_wino42_ok(g)) {
        g.b_wino = e->wwino42 + e->wino42_off[op.w_slot];
        ProfScope ps(e, s, pkey(e, op, std::string(conv3x3_wino42_name_for(g))), op.flops_per_image * nb, gemm_alg_bytes(g, z));
        return launch_conv3x3_wino42(g, s);
    }
    if (e->precision != 1 && z == 1 && op.w_slot >= 0 && e->wwino2 && e->wino2_off[op.w_slot] >= 0 && conv3x3_wino2d_ok(g)) {
        g.b_wino = e->wwino2 + e->wino2_off[op.w_slot];
        ProfScope ps(e, s, pkey(e, op, std::string(conv3x3_wino2d_name())), op.flops_per_image * nb, gemm_alg_bytes(g, z));
        return launch_conv3x3_wino2d(g, s);
    }
    if (e->precision != 1 && z == 1 && op.w_slot >= 0 && e->wwino2 && e->wino2_off[op.w_slot] >= 0 && e->splitk_ws &&
        conv3x3_wino2d_splitk_factor(g) > 1) {  // 4x4 images: the channel range split over workgroups + a finishing pass
        g.b_wino = e->wwino2 + e->wino2_off[op.w_slot];
        const long long img0 = (long long)(b.arena - e->arena) / e->plan.arena_per_image + i0;  // first image of this sub-batch (lanes)
        g.wk_partial = e->splitk_ws + img0 * e->splitk_per_image;
        GD_CHECK(img0 * e->splitk_per_image + (long long)conv3x3_wino2d_splitk_factor(g) * g.M * g.N <= e->splitk_ws_floats,
                 "%s: split-K scratch too small", op.label.c_str());
        ProfScope ps(e, s, pkey(e, op, std::string(conv3x3_wino2d_splitk_name())), op.flops_per_image * nb, gemm_alg_bytes(g, z));
        return launch_conv3x3_wino2d_splitk(g, s);
    }
    if (op.side_peer >= 0 && side_fused(e, b, batch, op_index)) {  // the per-image affine table first, then the GEMM whose loader applies it
        if (first) GD_TRY(launch_apply_op(e, P.ops[op.side_peer], b, batch, true, s));  // (the table of the whole batch)
        fill_side(e, op, b, batch, g, i0);
    }
    ProfScope ps(e, s, pkey(e, op, std::string(igemm_variant_name(g, z)) + (g.gn_side_out ? "+gn" : "")),
                 op.flops_per_image * nb, gemm_alg_bytes(g, z));
    return launch_igemm(g, z, s);
}

static int run_op(gd_engine *e, int op_index, const Bind &b, int batch, hipStream_t s) {
    const Op &op = e->plan.ops[op_index];
    const Plan &P = e->plan;
    if (op.temb && b.temb_row) return 0;  // hoisted: the row of this step is already in place (ensure_temb_table)
    switch (op.kind) {
        case OP_FOURIER: {
            ProfScope ps(e, s, "fourier", 0, 0);
            return launch_fourier(b.labels, e->ts_dev, b.step, slot_ptr(e, op.w_slot), tptr(e, b, op.out, batch), b.labels ? batch : 1,
                                  op.N, op.act, s);
        }
        case OP_GEMM: {
            if (op.temb && b.labels == nullptr) return run_gemm(e, op_index, b, batch, 0, 1, true, s);  // one label: one row (see fill_gemm)
            if (attn_core_at(e, op_index)) {  // q k^T, softmax and p v in one launch; the next two ops return at once
                const Op &av = P.ops[op_index + 2];
                ProfScope ps(e, s, pkey(e, op, std::string(attn_fused_name(batch))),
                             (op.flops_per_image + av.flops_per_image) * batch, 4.0 * batch * (av.res.t >= 0 ? 5.0 : 4.0) * op.N * op.C0);  // q, k, v^T, out (, residual) once each
                return launch_attn_fused(tptr(e, b, op.a0, batch), op.lda0, tptr(e, b, op.bten, batch), op.ldb, tptr(e, b, av.bten, batch),
                                         av.bias_slot >= 0 ? slot_ptr(e, av.bias_slot) + av.bias_off : nullptr, tptr(e, b, av.res, batch), av.ldres,
                                         av.out_scale, tptr(e, b, av.qstats, batch), tptr(e, b, av.out, batch), av.ldc, batch, op.N, op.C0, s);
            }
            if (attn_core_at(e, op_index - 2)) return 0;  // the p v product of a fused core
            const int parts = gemm_parts(e, op, batch);
            if (parts == 1) return run_gemm(e, op_index, b, batch, 0, batch, true, s);
            for (int pi = 0, i0 = 0; pi < parts; ++pi) {  // image sub-ranges (operands beyond the 2 GiB buffer range)
                const int nb = batch / parts + (pi < batch % parts ? 1 : 0);
                GD_TRY(run_gemm(e, op_index, b, batch, i0, nb, pi == 0, s));
                i0 += nb;
            }
            return 0;
        }
        case OP_GN_STATS: {
            const int HW = op.H * op.W;
            const int sl = gn_slices(batch, HW);
            ProfScope ps(e, s, "gn_stats", 0, 4.0 * batch * HW * op.C0);
            return launch_gn_stats(tptr(e, b, op.a0, batch), op.C0, tptr(e, b, op.stats, batch), batch, HW, sl, s);
        }
        case OP_GN_APPLY: {
            if (op.egn_peer >= 0 && egn_fused(e, b, batch, op.egn_peer)) return 0;  // the producer's epilogue wrote op.out
            if (op.side_peer >= 0 && side_fused(e, b, batch, op.side_peer)) return 0;  // the shortcut convolution's loader wrote op.out
            const bool table_only = op.fuse_peer >= 0 && gn_fused(e, b, batch, op.fuse_peer);
            return launch_apply_op(e, op, b, batch, table_only, s);
        }
        case OP_GN_GENERIC: {
            ProfScope ps(e, s, "gn_generic", 0, 12.0 * batch * op.H * op.W * (op.C0 + op.C1));
            return launch_gn_generic(tptr(e, b, op.a0, batch), op.C0, tptr(e, b, op.a1, batch), op.C1, tptr(e, b, op.stats, batch),
                                     slot_ptr(e, op.gamma_slot), slot_ptr(e, op.beta_slot), tptr(e, b, op.out, batch), batch,
                                     op.H * op.W, op.groups, op.act, 1e-6f, s);
        }
        case OP_FIR_UP: {
            ProfScope ps(e, s, "fir_up2", 0, 4.0 * batch * op.H * op.W * op.C0 * 5);
            if (op.act) return launch_naive_resample2(tptr(e, b, op.a0, batch), tptr(e, b, op.out, batch), batch, op.H, op.W, op.C0, 1, s);
            return launch_fir_up2(tptr(e, b, op.a0, batch), tptr(e, b, op.out, batch), batch, op.H, op.W, op.C0, s);
        }
        case OP_FIR_DOWN: {
            ProfScope ps(e, s, "fir_down2", 0, 4.0 * batch * op.H * op.W * op.C0 * 1.25);
            if (op.act) return launch_naive_resample2(tptr(e, b, op.a0, batch), tptr(e, b, op.out, batch), batch, op.H, op.W, op.C0, 0, s);
            return launch_fir_down2(tptr(e, b, op.a0, batch), tptr(e, b, op.out, batch), batch, op.H, op.W, op.C0, s);
        }
        case OP_FIR_PAD: {
            ProfScope ps(e, s, "fir_pad", 0, 4.0 * batch * op.C0 * ((double)op.H * op.W + (double)(op.H + 1) * (op.W + 1)));
            return launch_fir_pad(tptr(e, b, op.a0, batch), tptr(e, b, op.out, batch), batch, op.H, op.W, op.C0, s);
        }
        case OP_SOFTMAX: {
            if (attn_core_at(e, op_index - 1)) return 0;  // inside a fused attention core
            ProfScope ps(e, s, "softmax", 0, 8.0 * batch * op.rows_per_image * op.cols);
            return launch_softmax_rows(tptr(e, b, op.out, batch), (long long)batch * op.rows_per_image, op.cols, s);
        }
        case OP_AFFINE_PAD: {
            ProfScope ps(e, s, "affine_pad", 0, 0);
            return launch_pad_channels(tptr(e, b, op.a0, batch), op.lda0, tptr(e, b, op.out, batch),
                                       (long long)batch * op.H * op.W, op.cin, op.cpad, op.mul, op.add, s);
        }
        case OP_SCALE_ROWS: {
            GD_CHECK(b.labels != nullptr, "scale_by_sigma needs explicit labels (not supported inside the fused sampler)");
            return launch_scale_rows(tptr(e, b, op.out, batch), b.labels, batch, op.m_per_image, s);
        }
    }
    GD_CHECK(false, "unknown op kind %d", op.kind);
    (void)P;
}

static int run_unet(gd_engine *e, const Bind &b, int batch, hipStream_t s) {
    for (int i = 0; i < (int)e->plan.ops.size(); ++i) GD_TRY(run_op(e, i, b, batch, s));
    return 0;
}

// Every entry point that touches an engine's GPU makes it the calling thread's current HIP device and puts the previous one back on
// return: one host thread drives several engines on several GPUs (gddim_amd.devices: the reference's pmap), and the caller's own
// runtime (PyTorch) must find the device it left.
struct DeviceScope {
    int prev = -1;
    bool changed = false;
    explicit DeviceScope(const gd_engine *e) {
        if (!e || e->plan_only || e->device < 0) return;
        if (hipGetDevice(&prev) != hipSuccess) prev = -1;
        if (prev != e->device) changed = hipSetDevice(e->device) == hipSuccess;
    }
    ~DeviceScope() {
        if (changed && prev >= 0) (void)hipSetDevice(prev);
    }
    DeviceScope(const DeviceScope &) = delete;
    DeviceScope &operator=(const DeviceScope &) = delete;
};

// The hoisted time-embedding table of the installed schedule: one row per knot -- ts[0 .. num_step) and, for CLD, the denoising time --
// computed by the plan's own time-embedding ops (Op::temb) on one row, exactly what the loop would compute at that step.
static int run_op(gd_engine *e, int op_index, const Bind &b, int batch, hipStream_t s);
static int ensure_temb_table(gd_engine *e) {
    if (e->temb_valid) return 0;
    e->temb_rows = 0;
    const int rows = e->sched_kind == 1 ? e->num_step + 1 : e->num_step;
    const int width = e->temb_t >= 0 ? e->plan.tensors[e->temb_t].C : 0;
    if (opts().no_temb_hoist || e->temb_t < 0 || e->sched_kind == 0 || !e->ts_dev || rows < 1 || width % 4 != 0) {
        e->temb_valid = true;  // (nothing to hoist: the loop computes the embedding itself)
        return 0;
    }
    if ((long long)rows * width > e->temb_cap) {
        if (e->temb_tab) (void)hipFree(e->temb_tab);
        e->temb_tab = nullptr; e->temb_cap = 0;
        GD_HIP(hipMalloc((void **)&e->temb_tab, (size_t)rows * width * sizeof(float)));
        e->temb_cap = (long long)rows * width;
    }
    if (!e->temb_cur || e->temb_w != width) {
        if (e->temb_cur) (void)hipFree(e->temb_cur);
        e->temb_cur = nullptr;
        GD_HIP(hipMalloc((void **)&e->temb_cur, (size_t)gd_engine::MAX_LANES * width * sizeof(float)));
    }
    e->temb_w = width;
    hipStream_t s = e->stream[0];
    const bool prof = e->prof;
    e->prof = false;  // (set-up work, not part of a sampler call's profile)
    int *counter = e->step_dev + gd_engine::MAX_LANES - 1;  // a lane counter no sampler call is using now (calls on an engine are serial)
    Bind b{e->xin, nullptr, e->epsb, e->arena, counter};
    const TRef row{e->temb_t, 0};
    int r = 0;
    for (int k = 0; k < rows && !r; ++k) {
        if (hipMemcpyAsync(counter, &k, sizeof(int), hipMemcpyHostToDevice, s) != hipSuccess || hipStreamSynchronize(s) != hipSuccess) { r = 1; break; }
        for (int i = 0; i < (int)e->plan.ops.size() && !r; ++i)
            if (e->plan.ops[i].temb) r = run_op(e, i, b, 1, s);
        if (!r && hipMemcpyAsync(e->temb_tab + (size_t)k * width, tptr(e, b, row, 1), (size_t)width * sizeof(float), hipMemcpyDeviceToDevice, s) != hipSuccess) r = 1;
    }
    if (!r && hipStreamSynchronize(s) != hipSuccess) r = 1;
    e->prof = prof;
    GD_CHECK(!r, "building the time-embedding table failed: %s", g_err.c_str());
    e->temb_rows = rows;
    e->temb_valid = true;
    return 0;
}

static int check_compute(gd_engine *e, int batch) {
    GD_CHECK(e != nullptr, "null engine");
    GD_CHECK(!e->plan_only, "engine was created in plan-only mode (device < 0): no GPU compute available");
    GD_CHECK(e->params_uploaded, "parameters not loaded: call gd_engine_params_ready first");
    GD_CHECK(batch >= 1 && batch <= e->plan.desc.max_batch, "batch %d outside [1, max_batch=%d]", batch, e->plan.desc.max_batch);
    GD_HIP(hipSetDevice(e->device));
    return 0;
}

extern "C" {

const char *gd_last_error(void) { return g_err.c_str(); }
int gd_version(void) { return 100; }

int gd_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

// ---- engine ----
int gd_engine_create(const gd_model_desc *desc, int device, gd_engine **out) {
    GD_COLD;
    GD_CHECK(desc && out, "null argument");
    gd_engine *e = new gd_engine();
    e->opts = options_reload();  // an engine is built under the environment of its creation and keeps THAT table for life (options.h)
    OptScope scope(e->opts);
    if (e->plan.build(*desc)) { delete e; return 1; }
    e->device = device;
    e->plan_only = device < 0;
    e->slot_off.resize(e->plan.slots.size());
    long long off = 0;
    for (size_t i = 0; i < e->plan.slots.size(); ++i) {
        e->slot_off[i] = off;
        off += (e->plan.slots[i].rows * e->plan.slots[i].cols + 63) / 64 * 64;
    }
    e->wbuf_floats = off;
    for (const Op &op : e->plan.ops)  // the tensor the per-image bias rows come from (one: the concatenated projection); else no hoist
        if (op.rowbias.t >= 0) e->temb_t = (e->temb_t == -1 || e->temb_t == op.rowbias.t) ? op.rowbias.t : -2;
    if (e->temb_t < 0) e->temb_t = -1;
    if (!e->plan_only) {
        DeviceScope dscope(e);
        auto fail = [&](const char *what, hipError_t err) {
            set_error("%s: %s", what, hipGetErrorString(err));
            gd_engine_destroy(e);
            return 1;
        };
        hipError_t err;
        if ((err = hipSetDevice(device)) != hipSuccess) return fail("hipSetDevice", err);
        const size_t B = desc->max_batch;
        const size_t pix = B * desc->image_size * desc->image_size;
        const int CP = e->plan.CP;
#define GD_ALLOC(ptr, n)                                                                \
    if ((err = hipMalloc((void **)&(ptr), (size_t)(n) * sizeof(float))) != hipSuccess) \
        return fail("hipMalloc " #ptr, err);
        GD_ALLOC(e->arena, (size_t)e->plan.arena_per_image * B);
        GD_ALLOC(e->wbuf, (size_t)e->wbuf_floats);
        GD_ALLOC(e->xin, pix * CP);
        GD_ALLOC(e->epsb, pix * desc->in_channels);
        GD_ALLOC(e->eps_planar, pix * desc->in_channels);
        GD_ALLOC(e->labels, B);
        GD_ALLOC(e->u8, pix * CP);
        GD_ALLOC(e->ystate, pix * desc->in_channels);
        GD_ALLOC(e->uin_stage, pix * desc->in_channels);
        GD_ALLOC(e->xout_stage, pix * desc->in_channels);
        GD_ALLOC(e->vout_stage, pix * desc->in_channels);
        GD_ALLOC(e->dn_dev, 20);
        GD_HIP(hipMalloc((void **)&e->seed_dev, 2 * sizeof(unsigned)));
        GD_ALLOC(e->step_dev, gd_engine::MAX_LANES);
#undef GD_ALLOC
        (void)hipMemset(e->step_dev, 0, sizeof(int) * gd_engine::MAX_LANES);
        e->precision = opts().precision;
        e->lanes = std::max(1, std::min((int)gd_engine::MAX_LANES, opts().lanes));
        if (!igemm_zero_buffer()) return fail("zero buffer", hipErrorOutOfMemory);  // never allocate inside a graph capture
        if ((err = hipEventCreateWithFlags(&e->ev_in, hipEventDisableTiming)) != hipSuccess) return fail("hipEventCreate", err);
        for (int i = 0; i < gd_engine::MAX_LANES; ++i) {
            if ((err = hipStreamCreateWithFlags(&e->stream[i], hipStreamNonBlocking)) != hipSuccess) return fail("hipStreamCreate", err);
            if ((err = hipEventCreateWithFlags(&e->ev_out[i], hipEventDisableTiming)) != hipSuccess) return fail("hipEventCreate", err);
        }
    }
    *out = e;
    return 0;
}

int gd_engine_destroy(gd_engine *e) {
    GD_COLD;
    OptScope scope(e ? e->opts : nullptr);
    if (!e) return 0;
    if (!e->plan_only) {
        DeviceScope dscope(e);
        (void)hipSetDevice(e->device);
        (void)hipDeviceSynchronize();
        drop_graphs(e);
        for (auto &p : e->pending) { (void)hipEventDestroy(p.a); (void)hipEventDestroy(p.b); }
        if (e->wwino2) (void)hipFree(e->wwino2);
        if (e->wwino42) (void)hipFree(e->wwino42);
        if (e->splitk_ws) (void)hipFree(e->splitk_ws);
        if (e->wsplit_hi) (void)hipFree(e->wsplit_hi);
        if (e->wsplit_lo) (void)hipFree(e->wsplit_lo);
        if (e->wsplit6) (void)hipFree(e->wsplit6);
        float *bufs[] = {e->arena, e->wbuf, e->xin, e->epsb, e->eps_planar, e->labels, e->u8, e->ring, e->ystate, e->uin_stage,
                         e->xout_stage, e->vout_stage,
                         e->ts_dev, e->coef_dev, e->dn_dev, e->mean_dev, e->std_dev, e->tf_dev, e->nz_dev, e->m2e_dev, e->temb_tab, e->temb_cur, (float *)e->seed_dev,
                         (float *)e->step_dev};
        for (float *b : bufs)
            if (b) (void)hipFree(b);
        if (e->ev_in) (void)hipEventDestroy(e->ev_in);
        for (int i = 0; i < gd_engine::MAX_LANES; ++i) {
            if (e->ev_out[i]) (void)hipEventDestroy(e->ev_out[i]);
            if (e->stream[i]) (void)hipStreamDestroy(e->stream[i]);
        }
    }
    delete e;
    return 0;
}

int gd_engine_plan_json(gd_engine *e, char *buf, size_t cap, size_t *needed) {
    OptScope scope(e ? e->opts : nullptr);
    GD_CHECK(e, "null engine");
    const std::string j = e->plan.json();
    if (needed) *needed = j.size() + 1;
    if (buf && cap >= j.size() + 1) memcpy(buf, j.c_str(), j.size() + 1);
    return 0;
}

double gd_engine_flops_per_image(gd_engine *e, int conv3x3_only) {
    if (!e) return 0;
    return conv3x3_only ? e->plan.conv3x3_flops_per_image : e->plan.flops_per_image;
}

// ---- weights ----
int gd_engine_num_params(gd_engine *e) { return e ? (int)e->plan.params.size() : 0; }

int gd_engine_param_info(gd_engine *e, int index, char *name_buf, size_t name_cap, int64_t *shape4, int *ndim) {
    GD_CHECK(e && index >= 0 && index < (int)e->plan.params.size(), "bad param index");
    const Param &p = e->plan.params[index];
    if (name_buf && name_cap) { strncpy(name_buf, p.name.c_str(), name_cap - 1); name_buf[name_cap - 1] = 0; }
    if (ndim) *ndim = (int)p.shape.size();
    if (shape4)
        for (size_t i = 0; i < 4; ++i) shape4[i] = i < p.shape.size() ? p.shape[i] : 1;
    return 0;
}

int gd_engine_load_param(gd_engine *e, const char *name, const float *host_data, const int64_t *shape, int ndim) {
    OptScope scope(e ? e->opts : nullptr);
    GD_CHECK(e && name && host_data && shape, "null argument");
    e->params_uploaded = false;
    e->wsplit_valid = false;
    e->wsplit6_valid = false;
    return e->plan.load_param(name, host_data, shape, ndim);
}

static inline unsigned short f32_to_bf16_rne(float x) {
    unsigned u;
    memcpy(&u, &x, 4);
    if ((u & 0x7fffffffu) > 0x7f800000u) return (unsigned short)((u >> 16) | 0x40);  // NaN
    u += 0x7fffu + ((u >> 16) & 1u);
    return (unsigned short)(u >> 16);
}

// hi = bf16_rne(w), lo = bf16_rne(w - hi) for every packed weight; uploaded next to the fp32 copy
static int upload_split_weights(gd_engine *e) {
    GD_HIP(hipSetDevice(e->device));
    if (!e->wsplit_hi) {
        GD_HIP(hipMalloc((void **)&e->wsplit_hi, (size_t)e->wbuf_floats * 2));
        GD_HIP(hipMalloc((void **)&e->wsplit_lo, (size_t)e->wbuf_floats * 2));
    }
    std::vector<unsigned short> hi, lo;
    for (size_t i = 0; i < e->plan.slots.size(); ++i) {
        const WSlot &sl = e->plan.slots[i];
        hi.resize(sl.host.size());
        lo.resize(sl.host.size());
        for (size_t k = 0; k < sl.host.size(); ++k) {
            const float w = sl.host[k];
            const unsigned short h = f32_to_bf16_rne(w);
            const unsigned hb = (unsigned)h << 16;
            float hf;
            memcpy(&hf, &hb, 4);
            hi[k] = h;
            lo[k] = f32_to_bf16_rne(w - hf);
        }
        GD_HIP(hipMemcpy(e->wsplit_hi + e->slot_off[i], hi.data(), hi.size() * 2, hipMemcpyHostToDevice));
        GD_HIP(hipMemcpy(e->wsplit_lo + e->slot_off[i], lo.data(), lo.size() * 2, hipMemcpyHostToDevice));
    }
    e->wsplit_valid = true;
    return 0;
}

// The Winograd-domain copies of the 3x3 / stride 1 / pad 1 convolution weights, built on the device from wbuf when the parameters are
// uploaded (allocated once: the plan fixes their sizes, and captured graphs keep pointing into them when parameters are loaded again).
// Which launches use a copy is decided per batch size by the kernels' *_ok tests.  A slot the hybrid kernel can take gets the hybrid copy
// only -- the F(2x2,3x3) copy is for what the hybrid kernel cannot tile (the 4x4 level) and, under GDDIM_NOWINO42=1, for the A/B of the
// older kernel: 1.4 GB of weights for the benchmarked network instead of round 3's 2.8 GB (gd_engine_weight_bytes).
static int build_wino_weights(gd_engine *e) {
    const Plan &P = e->plan;
    std::vector<std::pair<int, int>> dims(P.slots.size(), {0, 0});  // (N, Cin) per slot
    auto hybrid_takes = [&](const Op &op) {
        return !opts().no_wino42 && (op.W == 8 || op.W == 16 || op.W == 32 || op.W == 64) && op.H % 2 == 0 && (op.H * op.W / 8) % 8 == 0;
    };
    // the two-dimensional form F(2x2,3x3)
    e->wino2_off.assign(P.slots.size(), -1);
    long long total2 = 0, need_ws = 0;
    for (const Op &op : P.ops) {
        if (op.kind != OP_GEMM || op.w_slot < 0 || op.a_slot >= 0 || op.z_per_image || e->wino2_off[op.w_slot] >= 0) continue;
        const int Cin = op.C0 + op.C1;
        if (op.KH != 3 || op.KW != 3 || op.stride != 1 || op.pad != 1 || Cin % 32 != 0 || op.K != 9 * Cin || op.ldb != op.K) continue;
        if (P.slots[op.w_slot].rows != op.N || P.slots[op.w_slot].cols != op.K) continue;
        if ((op.N % 4 != 0 || op.N < 32) && (op.qstats.t >= 0 || op.egn_peer >= 0)) continue;
        const bool img4 = op.H == 4 && op.W == 4;  // the channel-split launch (conv3x3_wino2d_splitk_factor)
        if (!img4 && ((op.W != 8 && op.W != 16 && op.W != 32 && op.W != 64) || op.H % 2 != 0 || (op.H * op.W / 4) % 16 != 0)) continue;
        // (Cin = 32 slots keep their F(2x2,3x3) copy: conv3x3_wino42_ok may still refuse such a launch at run time -- too few wide tiles --
        // and it must then fall to this kernel, not to the direct one: ADVICE r4)
        if (!img4 && hybrid_takes(op) && Cin != 32) continue;
        if (img4) need_ws = std::max(need_ws, 8LL * e->plan.desc.max_batch * 16 * op.N);
        if (img4) e->splitk_per_image = std::max(e->splitk_per_image, 8LL * 16 * op.N);
        e->wino2_off[op.w_slot] = total2;
        dims[op.w_slot] = {op.N, Cin};
        total2 += (long long)op.N * 16 * Cin;
    }
    if (need_ws > e->splitk_ws_floats) {  // (once: the plan and max_batch fix it)
        if (e->splitk_ws) (void)hipFree(e->splitk_ws);
        GD_HIP(hipMalloc((void **)&e->splitk_ws, (size_t)need_ws * sizeof(float)));
        e->splitk_ws_floats = need_ws;
    }
    GD_CHECK(!e->wwino2 || total2 == e->wwino2_floats, "internal error: the F(2x2,3x3) weight layout of an engine changed between two parameter loads");
    if (total2 > 0) {
        if (!e->wwino2) GD_HIP(hipMalloc((void **)&e->wwino2, (size_t)total2 * sizeof(float)));
        e->wwino2_floats = total2;
        for (size_t i = 0; i < P.slots.size(); ++i)
            if (e->wino2_off[i] >= 0)
                GD_TRY(launch_wino2d_weights(e->wbuf + e->slot_off[i], e->wwino2 + e->wino2_off[i], dims[i].first, dims[i].second, nullptr));
    }
    // the hybrid F(2,3)_y x F(4,3)_x form for the slots its kernel can take (8x8 .. 64x64 images; which launches use it is decided per
    // batch size by conv3x3_wino42_ok)
    e->wino42_off.assign(P.slots.size(), -1);
    long long total42 = 0;
    for (const Op &op : P.ops) {
        if (op.kind != OP_GEMM || op.w_slot < 0 || op.a_slot >= 0 || op.z_per_image || e->wino42_off[op.w_slot] >= 0) continue;
        const int Cin = op.C0 + op.C1;
        if (op.KH != 3 || op.KW != 3 || op.stride != 1 || op.pad != 1 || Cin % 32 != 0 || op.K != 9 * Cin || op.ldb != op.K) continue;
        if (P.slots[op.w_slot].rows != op.N || P.slots[op.w_slot].cols != op.K) continue;
        if ((op.N % 4 != 0 || op.N < 32) && (op.qstats.t >= 0 || op.egn_peer >= 0)) continue;
        if (!hybrid_takes(op)) continue;
        e->wino42_off[op.w_slot] = total42;
        dims[op.w_slot] = {op.N, Cin};
        total42 += wino42_weight_floats(op.N, Cin);
    }
    GD_CHECK(!e->wwino42 || total42 == e->wwino42_floats, "internal error: the hybrid Winograd weight layout of an engine changed between two parameter loads");
    if (total42 > 0) {
        if (!e->wwino42) GD_HIP(hipMalloc((void **)&e->wwino42, (size_t)total42 * sizeof(float)));
        e->wwino42_floats = total42;
        for (size_t i = 0; i < P.slots.size(); ++i)
            if (e->wino42_off[i] >= 0)
                GD_TRY(launch_wino42_weights(e->wbuf + e->slot_off[i], e->wwino42 + e->wino42_off[i], dims[i].first, dims[i].second, nullptr));
    }
    GD_HIP(hipDeviceSynchronize());
    return 0;
}

// split6 mode: pre-split Winograd-domain fragments of every 3x3 / stride 1 slot whose images are multiples of 16 x 16 (allocated once; the
// plan fixes the layout), built on the device from wbuf
static int build_split6_weights(gd_engine *e) {
    const Plan &P = e->plan;
    std::vector<std::pair<int, int>> dims(P.slots.size(), {0, 0});
    e->split6_off.assign(P.slots.size(), -1);
    long long total = 0;
    for (const Op &op : P.ops) {
        if (op.kind != OP_GEMM || op.w_slot < 0 || op.a_slot >= 0 || op.z_per_image || e->split6_off[op.w_slot] >= 0) continue;
        if (op.KH != 3 || op.KW != 3 || op.stride != 1 || op.pad != 1 || op.C1 != 0 || op.K != 9 * op.C0 || op.ldb != op.K) continue;
        if (P.slots[op.w_slot].rows != op.N || P.slots[op.w_slot].cols != op.K) continue;
        if (op.H % 16 != 0 || op.W % 16 != 0 || op.C0 % 16 != 0 || op.N % 64 != 0) continue;
        e->split6_off[op.w_slot] = total;
        dims[op.w_slot] = {op.N, op.C0};
        total += conv3x3_wino2d_split6_weight_bytes(op.C0, op.N);
    }
    GD_CHECK(!e->wsplit6 || total == e->wsplit6_bytes, "internal error: the split6 weight layout of an engine changed between two parameter loads");
    if (total > 0) {
        if (!e->wsplit6) GD_HIP(hipMalloc((void **)&e->wsplit6, (size_t)total));
        e->wsplit6_bytes = total;
        for (size_t i = 0; i < P.slots.size(); ++i)
            if (e->split6_off[i] >= 0)
                GD_TRY(launch_wino2d_split6_weights(e->wbuf + e->slot_off[i], reinterpret_cast<unsigned short *>(reinterpret_cast<char *>(e->wsplit6) + e->split6_off[i]),
                                                    dims[i].first, dims[i].second, nullptr));
    }
    GD_HIP(hipDeviceSynchronize());
    e->wsplit6_valid = true;
    return 0;
}

int gd_engine_set_precision(gd_engine *e, int mode) {
    GD_COLD;
    OptScope scope(e ? e->opts : nullptr);
    GD_CHECK(e, "null engine");
    GD_CHECK(mode == 0 || mode == 1 || mode == 2, "unknown precision mode %d (0 = fp32, 1 = bf16x3, 2 = split6)", mode);
    DeviceScope dscope(e);
    e->temb_valid = false;  // (the embedding GEMMs may run on other kernels)
    if (!e->plan_only) {
        GD_HIP(hipSetDevice(e->device));
        GD_HIP(hipDeviceSynchronize());
        drop_graphs(e);  // captured graphs bake the kernel choice
        e->egn_ok.clear();
        e->side_ok.clear();
        if (mode == 1 && e->params_uploaded && !e->wsplit_valid) GD_TRY(upload_split_weights(e));
        if (mode == 2 && e->params_uploaded && !e->wsplit6_valid) GD_TRY(build_split6_weights(e));
    }
    e->precision = mode;
    return 0;
}

int gd_engine_params_ready(gd_engine *e) {
    GD_COLD;
    OptScope scope(e ? e->opts : nullptr);
    GD_CHECK(e, "null engine");
    std::string missing;
    int nmiss = 0;
    for (const Param &p : e->plan.params)
        if (!p.loaded) {
            if (nmiss < 8) missing += (nmiss ? ", " : "") + p.name;
            ++nmiss;
        }
    GD_CHECK(nmiss == 0, "%d parameters not loaded: %s%s", nmiss, missing.c_str(), nmiss > 8 ? ", ..." : "");
    e->plan.fold_params();  // derived weight slots (the folded attention projections)
    DeviceScope dscope(e);
    e->temb_valid = false;
    if (!e->plan_only) {
        GD_HIP(hipSetDevice(e->device));
        // the upload and the Winograd weight transforms below run on the null stream while the engine's lanes are non-blocking streams:
        // a sampler call still in flight (another parameter tree re-uploaded into this engine) must not see its weights change under it
        GD_HIP(hipDeviceSynchronize());
        for (size_t i = 0; i < e->plan.slots.size(); ++i) {
            const WSlot &s = e->plan.slots[i];
            GD_HIP(hipMemcpy(e->wbuf + e->slot_off[i], s.host.data(), s.host.size() * sizeof(float), hipMemcpyHostToDevice));
        }
        e->params_uploaded = true;
        e->wsplit_valid = false;
        e->wsplit6_valid = false;
        GD_TRY(build_wino_weights(e));
        if (e->precision == 1) GD_TRY(upload_split_weights(e));
        if (e->precision == 2) GD_TRY(build_split6_weights(e));
    }
    return 0;
}

int gd_engine_weight_bytes(gd_engine *e, long long out4[4]) {
    GD_CHECK(e && out4, "null argument");
    out4[0] = e->wbuf_floats * 4;
    out4[1] = e->wwino2_floats * 4;
    out4[2] = e->wwino42_floats * 4;
    out4[3] = (e->wsplit_hi ? e->wbuf_floats * 4 : 0) + (e->wsplit6 ? e->wsplit6_bytes : 0);  // bf16 hi + lo (mode 1), split6 fragments (mode 2)
    return 0;
}

int gd_engine_packed_weight(gd_engine *e, const char *slot, float *host_out, size_t cap_floats, size_t *n_floats) {
    GD_CHECK(e && slot, "null argument");
    for (const WSlot &s : e->plan.slots)
        if (s.name == slot) {
            if (n_floats) *n_floats = s.host.size();
            if (host_out) {
                GD_CHECK(cap_floats >= s.host.size(), "buffer too small for slot '%s'", slot);
                memcpy(host_out, s.host.data(), s.host.size() * sizeof(float));
            }
            return 0;
        }
    GD_CHECK(false, "unknown weight slot '%s'", slot);
}

// ---- score network ----
int gd_unet_forward(gd_engine *e, const float *x_dev, const float *labels_dev, float *out_dev, int batch, void *stream) {
    OptScope scope(e ? e->opts : nullptr);
    DeviceScope dscope(e);
    GD_TRY(check_compute(e, batch));
    GD_CHECK(x_dev && labels_dev && out_dev, "null device pointer");
    hipStream_t s = (hipStream_t)stream;
    const gd_model_desc &d = e->plan.desc;
    const long long pix = (long long)batch * d.image_size * d.image_size;
    GD_TRY(launch_pad_channels(x_dev, d.in_channels, e->xin, pix, d.in_channels, e->plan.CP, 1.f, 0.f, s));
    Bind b{e->xin, labels_dev, out_dev, e->arena, e->step_dev};
    return run_unet(e, b, batch, s);
}

int gd_cld_eps(gd_engine *e, const float *u_dev, const float *t_dev, float *eps_dev, int batch, void *stream) {
    OptScope scope(e ? e->opts : nullptr);
    DeviceScope dscope(e);
    GD_TRY(check_compute(e, batch));
    GD_CHECK(u_dev && t_dev && eps_dev, "null device pointer");
    const gd_model_desc &d = e->plan.desc;
    GD_CHECK(d.in_channels % 2 == 0, "CLD needs an even number of U-Net channels");
    hipStream_t s = (hipStream_t)stream;
    const int C = d.in_channels / 2;
    const long long pix = (long long)batch * d.image_size * d.image_size;
    GD_TRY(launch_pad_channels(t_dev, 1, e->labels, batch, 1, 1, 999.f, 0.f, s));  // labels = t * 999 (models/utils.py:172)
    GD_TRY(launch_cld_pack(u_dev, e->xin, nullptr, 0, pix, C, s));
    Bind b{e->xin, e->labels, e->eps_planar, e->arena, e->step_dev};
    GD_TRY(run_unet(e, b, batch, s));
    return launch_cld_unpack_pairs(e->eps_planar, d.in_channels, eps_dev, pix, C, s);
}

// ---- schedules ----
static int upload(float **dst, const float *src, size_t n) {
    if (*dst) { (void)hipFree(*dst); *dst = nullptr; }
    GD_HIP(hipMalloc((void **)dst, n * sizeof(float)));
    GD_HIP(hipMemcpy(*dst, src, n * sizeof(float), hipMemcpyHostToDevice));
    return 0;
}

int gd_engine_set_schedule_cld(gd_engine *e, int num_step, int n_slots, const float *ts_host, const float *coef_host,
                               int denoise, float t_denoise, const float *dn_F, const float *dn_GG, const float *dn_RinvT) {
    GD_COLD;
    OptScope scope(e ? e->opts : nullptr);
    GD_CHECK(e && !e->plan_only, "engine has no GPU");
    GD_CHECK(num_step >= 0 && n_slots >= 2 && ts_host && coef_host, "bad schedule");
    GD_CHECK(e->plan.desc.in_channels % 2 == 0, "CLD needs an even number of U-Net channels");
    DeviceScope dscope(e);
    e->temb_valid = false;
    GD_HIP(hipSetDevice(e->device));
    GD_HIP(hipDeviceSynchronize());
    drop_graphs(e);
    std::vector<float> ts(ts_host, ts_host + num_step);
    ts.push_back(t_denoise);  // ts[num_step]: the time of the denoising evaluation
    GD_TRY(upload(&e->ts_dev, ts.data(), ts.size()));
    if (num_step > 0) GD_TRY(upload(&e->coef_dev, coef_host, (size_t)num_step * n_slots * 4));
    else {  // nfe = 1 with noise removal: no AB step runs, the caller's coef buffer is empty -- upload one zero row
        std::vector<float> zero((size_t)n_slots * 4, 0.f);
        GD_TRY(upload(&e->coef_dev, zero.data(), zero.size()));
    }
    if (denoise) {
        GD_CHECK(dn_F && dn_GG && dn_RinvT, "denoise matrices missing");
        float dn[12];
        memcpy(dn, dn_F, 16); memcpy(dn + 4, dn_GG, 16); memcpy(dn + 8, dn_RinvT, 16);
        GD_HIP(hipMemcpy(e->dn_dev, dn, sizeof dn, hipMemcpyHostToDevice));
    }
    const int ring_slots = n_slots - 2;
    const gd_model_desc &d = e->plan.desc;
    const long long need = (long long)ring_slots * d.max_batch * d.image_size * d.image_size * d.in_channels;
    if (need > e->ring_cap) {
        if (e->ring) (void)hipFree(e->ring);
        e->ring = nullptr;
        GD_HIP(hipMalloc((void **)&e->ring, (size_t)need * sizeof(float)));
        e->ring_cap = need;
    }
    e->sched_kind = 1; e->num_step = num_step; e->n_slots = n_slots; e->denoise = denoise; e->t_denoise = t_denoise;
    if (e->tf_dev) { (void)hipFree(e->tf_dev); e->tf_dev = nullptr; }
    e->tf_pre = false; e->fin_flags = 0;
    if (e->nz_dev) { (void)hipFree(e->nz_dev); e->nz_dev = nullptr; }
    e->nz_pre = false;
    return 0;
}

int gd_engine_set_noise_cld(gd_engine *e, const float *pre_lin, const float *pre_noise, const float *post_noise) {
    GD_COLD;
    OptScope scope(e ? e->opts : nullptr);
    GD_CHECK(e && !e->plan_only, "engine has no GPU");
    GD_CHECK(e->sched_kind == 1, "install the CLD schedule first (gd_engine_set_schedule_cld)");
    GD_CHECK((pre_lin == nullptr) == (pre_noise == nullptr), "pre_lin and pre_noise come together");
    DeviceScope dscope(e);
    GD_HIP(hipSetDevice(e->device));
    GD_HIP(hipDeviceSynchronize());
    drop_graphs(e);
    if (e->nz_dev) { (void)hipFree(e->nz_dev); e->nz_dev = nullptr; }
    e->nz_pre = false;
    if (!pre_lin && !post_noise) return 0;
    const int n = std::max(1, e->num_step);
    std::vector<float> rows((size_t)n * 12, 0.f);
    for (int i = 0; i < n; ++i) {
        float *r = rows.data() + (size_t)i * 12;
        r[0] = r[3] = 1.f;  // M_pre = I, S_pre = S_post = 0
        if (i >= e->num_step) continue;
        if (pre_lin) { memcpy(r, pre_lin + (size_t)i * 4, 16); memcpy(r + 4, pre_noise + (size_t)i * 4, 16); }
        if (post_noise) memcpy(r + 8, post_noise + (size_t)i * 4, 16);
    }
    GD_TRY(upload(&e->nz_dev, rows.data(), rows.size()));
    e->nz_pre = pre_lin != nullptr;
    return 0;
}

int gd_engine_set_seed(gd_engine *e, unsigned long long seed) {
    GD_CHECK(e != nullptr, "null engine");
    e->seed = seed;
    return 0;
}

int gd_randn_pairs(float *out_dev, long long n_pairs, unsigned long long seed, unsigned draw, unsigned long long elem0,
                   void *stream) {
    GD_CHECK(out_dev && n_pairs >= 0, "bad argument");
    if (n_pairs == 0) return 0;
    hipStream_t s = (hipStream_t)stream;
    unsigned *sd = nullptr;
    const unsigned words[2] = {(unsigned)seed, (unsigned)(seed >> 32)};
    GD_HIP(hipMalloc((void **)&sd, sizeof words));
    GD_HIP(hipMemcpyAsync(sd, words, sizeof words, hipMemcpyHostToDevice, s));
    int r = launch_randn_pairs(out_dev, n_pairs, sd, draw, elem0, s);
    (void)hipStreamSynchronize(s);
    (void)hipFree(sd);
    return r;
}

int gd_engine_set_transforms_cld(gd_engine *e, const float *pre, const float *eps_lin, const float *u_lin, const float *dn_u_lin,
                                 const float *post) {
    GD_COLD;
    OptScope scope(e ? e->opts : nullptr);
    GD_CHECK(e && !e->plan_only, "engine has no GPU");
    GD_CHECK(e->sched_kind == 1, "install the CLD schedule first (gd_engine_set_schedule_cld)");
    DeviceScope dscope(e);
    GD_HIP(hipSetDevice(e->device));
    GD_HIP(hipDeviceSynchronize());
    drop_graphs(e);
    if (e->tf_dev) { (void)hipFree(e->tf_dev); e->tf_dev = nullptr; }
    e->tf_pre = false; e->fin_flags = 0;
    if (pre || eps_lin || u_lin) {
        const int n = std::max(1, e->num_step);
        std::vector<float> rows((size_t)n * 12, 0.f);
        for (int i = 0; i < n; ++i) {
            float *r = rows.data() + (size_t)i * 12;
            r[0] = r[3] = r[4] = r[7] = 1.f;  // P = E = I, U = 0
            if (pre && i < e->num_step) memcpy(r, pre + (size_t)i * 4, 16);
            if (eps_lin && i < e->num_step) memcpy(r + 4, eps_lin + (size_t)i * 4, 16);
            if (u_lin && i < e->num_step) memcpy(r + 8, u_lin + (size_t)i * 4, 16);
        }
        GD_TRY(upload(&e->tf_dev, rows.data(), rows.size()));
        e->tf_pre = pre != nullptr;
    }
    float tail[8] = {0, 0, 0, 0, 1, 0, 0, 1};
    if (dn_u_lin) { memcpy(tail, dn_u_lin, 16); e->fin_flags |= 1; }
    if (post) { memcpy(tail + 4, post, 16); e->fin_flags |= 2; }
    GD_HIP(hipMemcpy(e->dn_dev + 12, tail, sizeof tail, hipMemcpyHostToDevice));
    return 0;
}

// step <- 0 and, with the hoisted time embedding, the first knot's row into the lane's current-row buffer.  One kernel: no memset / memcpy
// node ever goes into a captured graph (launch_reset_step_row: a replayed hipMemsetAsync node wrote other bytes than it was captured with).
static int reset_step(gd_engine *e, int *step, float *temb_row, hipStream_t s) {
    return launch_reset_step_row(step, e->temb_tab, temb_row, temb_row ? e->temb_w : 0, s);
}

// step++ (and the next knot's row)
static int advance_step(gd_engine *e, int *step, float *temb_row, hipStream_t s) {
    if (temb_row) return launch_advance_step_row(step, e->temb_tab, temb_row, e->temb_w, e->temb_rows, s);
    return launch_advance_step(step, s);
}

// GD_LANES > 1 splits a call over concurrent streams in the eager and per-step-graph forms; the whole-call graph is one lane.  Say so
// once instead of ignoring the switch silently.
static void warn_lanes_ignored(const gd_engine *e, bool whole) {
    static bool said = false;
    if (!whole || e->lanes <= 1 || said) return;
    said = true;
    fprintf(stderr, "gddim_amd: GD_LANES=%d applies to use_graph = 0 / 1; the whole-call hipGraph (use_graph = 2, the samplers' default) "
                    "runs one lane\n", e->lanes);
}

static int capture_begin(hipStream_t s) {
    GD_HIP(hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal));
    return 0;
}

static int capture_end(hipStream_t s, hipGraphExec_t *exec) {
    hipGraph_t g = nullptr;
    GD_HIP(hipStreamEndCapture(s, &g));
    hipError_t err = hipGraphInstantiate(exec, g, nullptr, nullptr, 0);
    (void)hipGraphDestroy(g);
    GD_CHECK(err == hipSuccess, "hipGraphInstantiate: %s", hipGetErrorString(err));
    return 0;
}

// split `batch` images over the engine's lanes: sub-batch i = images [start[i], start[i] + size[i])
static int split_lanes(gd_engine *e, int batch, bool allow, int *start, int *size) {
    int n = allow ? e->lanes : 1;
    if (batch < 2 * n) n = 1;
    for (int i = 0, at = 0; i < n; ++i) {
        size[i] = batch / n + (i < batch % n ? 1 : 0);
        start[i] = at;
        at += size[i];
    }
    return n;
}

int gd_sample_cld(gd_engine *e, int batch, const float *u_in_dev, float *x_out_dev, float *v_out_dev, int inverse_scale,
                  int use_graph, void *stream) {
    OptScope scope(e ? e->opts : nullptr);
    DeviceScope dscope(e);
    std::unique_lock<std::recursive_mutex> cold(g_cold);  // table rebuilds and captures under the lock; released before the replay
    GD_TRY(check_compute(e, batch));
    GD_CHECK(e->sched_kind == 1, "no CLD schedule installed (gd_engine_set_schedule_cld)");
    GD_CHECK(u_in_dev && x_out_dev && v_out_dev, "null device pointer");
    GD_CHECK(!e->plan.desc.scale_by_sigma, "scale_by_sigma is not supported inside the fused sampler");
    GD_TRY(ensure_temb_table(e));
    hipStream_t caller = (hipStream_t)stream;
    const gd_model_desc &d = e->plan.desc;
    const int C = d.in_channels / 2;
    const size_t HW = (size_t)d.image_size * d.image_size;
    const int ring_slots = e->n_slots - 2;
    const bool graph = use_graph && !e->prof;
    const bool whole = graph && use_graph == 2;  // one graph for the whole call (single lane)
    warn_lanes_ignored(e, whole);
    int start[gd_engine::MAX_LANES], size[gd_engine::MAX_LANES];
    const int nl = split_lanes(e, batch, !e->prof && !whole, start, size);
    if (e->nz_dev) {  // the seed of this call, in device memory so that captured graphs read the current one
        e->seed_words[0] = (unsigned)e->seed;
        e->seed_words[1] = (unsigned)(e->seed >> 32);
        GD_HIP(hipMemcpyAsync(e->seed_dev, e->seed_words, 2 * sizeof(unsigned), hipMemcpyHostToDevice, caller));
    }
    GD_HIP(hipEventRecord(e->ev_in, caller));

    struct Lane {
        hipStream_t s;
        int b;
        long long pix;
        unsigned long long elem0;
        float *u8, *eps, *ring, *xin, *temb;
        int *step;
        Bind bind, bind_u;  // bind: U-Net reads the (possibly transformed) step input; bind_u: reads the state itself
        gd_engine::Graphs *g;
    } L[gd_engine::MAX_LANES];
    for (int i = 0; i < nl; ++i) {
        Lane &l = L[i];
        l.s = e->stream[i];
        l.b = size[i];
        l.pix = (long long)size[i] * HW;
        const size_t p0 = (size_t)start[i] * HW;
        l.u8 = e->u8 + p0 * e->plan.CP;
        l.xin = e->xin + p0 * e->plan.CP;
        l.elem0 = (unsigned long long)p0 * C;
        l.eps = e->epsb + p0 * d.in_channels;
        l.ring = e->ring ? e->ring + p0 * d.in_channels * ring_slots : nullptr;
        l.step = e->step_dev + i;
        l.temb = e->temb_rows ? e->temb_cur + (size_t)i * e->temb_w : nullptr;
        l.bind_u = Bind{l.u8, nullptr, l.eps, e->arena + (size_t)e->plan.arena_per_image * start[i], l.step, l.temb};
        l.bind = l.bind_u;
        if (e->tf_pre) l.bind.x_in = l.xin;
        l.g = graph ? &e->graphs[((long long)size[i] << 40) | ((long long)start[i] << 8) | i | (whole ? (1LL << 62) : 0) |
                                 (whole && inverse_scale ? (1LL << 61) : 0)] : nullptr;
        GD_HIP(hipStreamWaitEvent(l.s, e->ev_in, 0));
        if (whole) continue;  // pack and the counter reset are nodes of the whole-loop graph
        GD_TRY(launch_cld_pack(u_in_dev + p0 * C * 2, l.u8, l.ring, ring_slots, l.pix, C, l.s));
        GD_TRY(reset_step(e, l.step, l.temb, l.s));
    }
    auto step_body = [&](Lane &l) -> int {
        if (e->nz_pre) GD_TRY(launch_cld_noise_pre(l.u8, e->nz_dev, l.step, e->seed_dev, l.elem0, l.pix, C, l.s));
        if (e->tf_pre) GD_TRY(launch_cld_pre(l.u8, l.xin, e->tf_dev, l.step, l.pix, C, l.s));
        GD_TRY(run_unet(e, l.bind, l.b, l.s));
        GD_TRY(launch_cld_ab_step(l.u8, l.eps, l.ring, ring_slots, e->coef_dev, e->n_slots, l.step, l.pix, C, e->tf_dev, e->nz_dev,
                                  e->seed_dev, l.elem0, l.s));
        return advance_step(e, l.step, l.temb, l.s);
    };
    if (whole) {
        // ONE graph for the whole call: pack -> num_step x [U-Net, AB step, step++] -> denoising U-Net -> finish.  Every address is
        // baked, so the caller's buffers are copied to / from engine-owned staging buffers around the single hipGraphLaunch.
        Lane &l = L[0];
        const size_t n_in = (size_t)l.pix * C * 2, n_out = (size_t)l.pix * C;
        GD_TRY(launch_copy_floats(u_in_dev, e->uin_stage, (long long)n_in, l.s));  // (kernels only around and inside the graph: launch_copy_floats)
        if (!l.g->loop) {
            GD_TRY(capture_begin(l.s));
            int r = launch_cld_pack(e->uin_stage, l.u8, l.ring, ring_slots, l.pix, C, l.s);
            if (!r) r = reset_step(e, l.step, l.temb, l.s);
            for (int k = 0; k < e->num_step && !r; ++k) r = step_body(l);
            if (!r && e->denoise) r = run_unet(e, l.bind_u, l.b, l.s);
            if (!r) r = launch_cld_finish(l.u8, l.eps, e->dn_dev, -e->t_denoise, e->denoise, inverse_scale, e->xout_stage, e->vout_stage,
                                          l.pix, C, e->fin_flags, l.s);
            int r2 = capture_end(l.s, &l.g->loop);
            if (r || r2) return 1;
        }
        cold.unlock();
        GD_HIP(hipGraphLaunch(l.g->loop, l.s));
        GD_TRY(launch_copy_floats(e->xout_stage, x_out_dev, (long long)n_out, l.s));
        GD_TRY(launch_copy_floats(e->vout_stage, v_out_dev, (long long)n_out, l.s));
        GD_HIP(hipEventRecord(e->ev_out[0], l.s));
        GD_HIP(hipStreamWaitEvent(caller, e->ev_out[0], 0));
        return 0;
    }
    if (graph) {
        for (int i = 0; i < nl; ++i) {
            Lane &l = L[i];
            if (!l.g->step) {
                GD_TRY(capture_begin(l.s));
                int r = step_body(l);
                int r2 = capture_end(l.s, &l.g->step);
                if (r || r2) return 1;
            }
            if (e->denoise && !l.g->fin) {
                GD_TRY(capture_begin(l.s));
                int r = run_unet(e, l.bind_u, l.b, l.s);  // the denoising evaluation reads the state as it is
                int r2 = capture_end(l.s, &l.g->fin);
                if (r || r2) return 1;
            }
        }
        cold.unlock();
        for (int k = 0; k < e->num_step; ++k)
            for (int i = 0; i < nl; ++i) GD_HIP(hipGraphLaunch(L[i].g->step, L[i].s));
        if (e->denoise)
            for (int i = 0; i < nl; ++i) GD_HIP(hipGraphLaunch(L[i].g->fin, L[i].s));
    } else {
        cold.unlock();
        for (int k = 0; k < e->num_step; ++k)
            for (int i = 0; i < nl; ++i) GD_TRY(step_body(L[i]));
        if (e->denoise)
            for (int i = 0; i < nl; ++i) GD_TRY(run_unet(e, L[i].bind_u, L[i].b, L[i].s));  // step == num_step -> t = t_denoise
    }
    for (int i = 0; i < nl; ++i) {
        Lane &l = L[i];
        const size_t o0 = (size_t)start[i] * HW * C;
        GD_TRY(launch_cld_finish(l.u8, l.eps, e->dn_dev, -e->t_denoise, e->denoise, inverse_scale, x_out_dev + o0, v_out_dev + o0,
                                 l.pix, C, e->fin_flags, l.s));
        GD_HIP(hipEventRecord(e->ev_out[i], l.s));
        GD_HIP(hipStreamWaitEvent(caller, e->ev_out[i], 0));
    }
    return 0;
}

int gd_engine_set_schedule_blur(gd_engine *e, int nfe, const float *ts_host, const float *mean_coef_host,
                                const float *std_coef_host) {
    GD_COLD;
    OptScope scope(e ? e->opts : nullptr);
    GD_CHECK(e && !e->plan_only, "engine has no GPU");
    GD_CHECK(nfe >= 1 && ts_host && mean_coef_host && std_coef_host, "bad schedule");
    GD_CHECK(e->plan.desc.image_size == 32, "the blur SDE is hard-wired to 32x32 (blur_jax/sde_lib.py:24)");
    DeviceScope dscope(e);
    e->temb_valid = false;
    GD_HIP(hipSetDevice(e->device));
    GD_HIP(hipDeviceSynchronize());
    drop_graphs(e);
    GD_TRY(upload(&e->ts_dev, ts_host, nfe));
    GD_TRY(upload(&e->mean_dev, mean_coef_host, (size_t)(nfe + 1) * 1024));
    GD_TRY(upload(&e->std_dev, std_coef_host, (size_t)(nfe + 1)));
    e->sched_kind = 2; e->num_step = nfe;
    if (e->m2e_dev) { (void)hipFree(e->m2e_dev); e->m2e_dev = nullptr; }
    return 0;
}

int gd_engine_set_schedule_blur_deis(gd_engine *e, int nfe, int n_slots, const float *ts_host, const float *coef_host) {
    GD_COLD;
    OptScope scope(e ? e->opts : nullptr);
    GD_CHECK(e && !e->plan_only, "engine has no GPU");
    GD_CHECK(nfe >= 1 && n_slots >= 2 && ts_host && coef_host, "bad schedule");
    GD_CHECK(e->plan.desc.image_size == 32, "the blur SDE is hard-wired to 32x32 (blur_jax/sde_lib.py:24)");
    DeviceScope dscope(e);
    e->temb_valid = false;
    GD_HIP(hipSetDevice(e->device));
    GD_HIP(hipDeviceSynchronize());
    drop_graphs(e);
    GD_TRY(upload(&e->ts_dev, ts_host, nfe));
    GD_TRY(upload(&e->coef_dev, coef_host, (size_t)nfe * n_slots * 1024));
    const int ring_slots = n_slots - 2;
    const gd_model_desc &d = e->plan.desc;
    const long long need = (long long)ring_slots * d.max_batch * 1024 * d.in_channels;
    if (need > e->ring_cap) {
        if (e->ring) (void)hipFree(e->ring);
        e->ring = nullptr;
        GD_HIP(hipMalloc((void **)&e->ring, (size_t)need * sizeof(float)));
        e->ring_cap = need;
    }
    e->sched_kind = 3; e->num_step = nfe; e->n_slots = n_slots;
    if (e->m2e_dev) { (void)hipFree(e->m2e_dev); e->m2e_dev = nullptr; }
    return 0;
}

int gd_engine_set_model2eps_blur(gd_engine *e, const float *rows_host) {
    GD_COLD;
    OptScope scope(e ? e->opts : nullptr);
    GD_CHECK(e && !e->plan_only, "engine has no GPU");
    GD_CHECK(e->sched_kind == 2 || e->sched_kind == 3, "install a blur schedule first (gd_engine_set_schedule_blur[_deis])");
    DeviceScope dscope(e);
    GD_HIP(hipSetDevice(e->device));
    GD_HIP(hipDeviceSynchronize());
    drop_graphs(e);  // captured launches bake the table pointer
    if (e->m2e_dev) { (void)hipFree(e->m2e_dev); e->m2e_dev = nullptr; }
    if (rows_host) GD_TRY(upload(&e->m2e_dev, rows_host, (size_t)e->num_step * 1025));
    return 0;
}

int gd_sample_blur(gd_engine *e, int batch, const float *y_in_dev, float *x_out_dev, int inverse_scale, int use_graph,
                   void *stream) {
    OptScope scope(e ? e->opts : nullptr);
    DeviceScope dscope(e);
    std::unique_lock<std::recursive_mutex> cold(g_cold);  // (as in gd_sample_cld)
    GD_TRY(check_compute(e, batch));
    GD_CHECK(e->sched_kind == 2 || e->sched_kind == 3, "no blur schedule installed (gd_engine_set_schedule_blur[_deis])");
    GD_TRY(ensure_temb_table(e));
    const bool multistep = e->sched_kind == 3;
    const int ring_slots = multistep ? e->n_slots - 2 : 0;
    const long long slot_stride = (long long)e->plan.desc.max_batch * 1024 * e->plan.desc.in_channels;
    // y <- the initial state; the reference's loops start their epsilon history from the state itself
    // (copy kernels, not hipMemcpyAsync: this runs inside the captured graphs too -- see reset_step)
    auto load_state = [&](float *y, float *ring, const float *src, size_t n, hipStream_t s) -> int {
        GD_TRY(launch_copy_floats(src, y, (long long)n, s));
        for (int j = 0; j < ring_slots; ++j) GD_TRY(launch_copy_floats(src, ring + (size_t)j * slot_stride, (long long)n, s));
        return 0;
    };
    GD_CHECK(y_in_dev && x_out_dev, "null device pointer");
    GD_CHECK(!e->plan.desc.scale_by_sigma, "scale_by_sigma is not supported inside the fused sampler");
    hipStream_t caller = (hipStream_t)stream;
    const gd_model_desc &d = e->plan.desc;
    const int C = d.in_channels;
    const bool graph = use_graph && !e->prof;
    const bool whole = graph && use_graph == 2;  // one graph for the whole call (single lane), as in gd_sample_cld
    warn_lanes_ignored(e, whole);
    int start[gd_engine::MAX_LANES], size[gd_engine::MAX_LANES];
    const int nl = split_lanes(e, batch, !e->prof && !whole, start, size);
    GD_HIP(hipEventRecord(e->ev_in, caller));
    struct Lane {
        hipStream_t s;
        int b;
        float *y, *xin, *eps, *ring, *temb;
        int *step;
        Bind bind;
        gd_engine::Graphs *g;
    } L[gd_engine::MAX_LANES];
    for (int i = 0; i < nl; ++i) {
        Lane &l = L[i];
        l.s = e->stream[i];
        l.b = size[i];
        const size_t p0 = (size_t)start[i] * 1024;
        l.y = e->ystate + p0 * C;
        l.xin = e->xin + p0 * e->plan.CP;
        l.eps = e->epsb + p0 * C;
        l.step = e->step_dev + i;
        l.temb = e->temb_rows ? e->temb_cur + (size_t)i * e->temb_w : nullptr;
        l.bind = Bind{l.xin, nullptr, l.eps, e->arena + (size_t)e->plan.arena_per_image * start[i], l.step, l.temb};
        l.g = graph ? &e->graphs[((long long)size[i] << 40) | ((long long)start[i] << 8) | i | (whole ? (1LL << 62) : 0) |
                                 (whole && inverse_scale ? (1LL << 61) : 0)] : nullptr;
        l.ring = (multistep && e->ring) ? e->ring + p0 * C : nullptr;
        GD_HIP(hipStreamWaitEvent(l.s, e->ev_in, 0));
        if (whole) continue;  // the state / history copies and the counter reset are nodes of the whole-call graph
        GD_TRY(load_state(l.y, l.ring, y_in_dev + p0 * C, (size_t)l.b * 1024 * C, l.s));
        GD_TRY(reset_step(e, l.step, l.temb, l.s));
    }
    auto step_body = [&](Lane &l) -> int {
        GD_TRY(launch_blur_idct_pad(l.y, l.xin, l.b, C, l.s));
        GD_TRY(run_unet(e, l.bind, l.b, l.s));
        if (multistep)
            GD_TRY(launch_blur_ab_step(l.y, l.eps, l.ring, ring_slots, e->coef_dev, e->n_slots, l.step, l.b, C, slot_stride, e->m2e_dev, l.s));
        else
            GD_TRY(launch_blur_step(l.y, l.eps, e->mean_dev, e->std_dev, l.step, l.b, C, e->m2e_dev, l.s));
        return advance_step(e, l.step, l.temb, l.s);
    };
    if (whole) {
        // ONE graph for the whole call (blur_jax/sampling.py:53-89: the fori_loop and the final y2x): state load -> nfe x [IDCT, U-Net,
        // DCT + update, step++] -> finish.  Addresses are baked: the caller's buffers pass through the engine's staging copies.
        Lane &l = L[0];
        const size_t n = (size_t)l.b * 1024 * C;
        GD_TRY(launch_copy_floats(y_in_dev, e->uin_stage, (long long)n, l.s));
        if (!l.g->loop) {
            GD_TRY(capture_begin(l.s));
            int r = load_state(l.y, l.ring, e->uin_stage, n, l.s);
            if (!r) r = reset_step(e, l.step, l.temb, l.s);
            for (int k = 0; k < e->num_step && !r; ++k) r = step_body(l);
            if (!r) r = launch_blur_finish(l.y, e->xout_stage, l.b, C, inverse_scale, l.s);
            int r2 = capture_end(l.s, &l.g->loop);
            if (r || r2) return 1;
        }
        cold.unlock();
        GD_HIP(hipGraphLaunch(l.g->loop, l.s));
        GD_TRY(launch_copy_floats(e->xout_stage, x_out_dev, (long long)n, l.s));
        GD_HIP(hipEventRecord(e->ev_out[0], l.s));
        GD_HIP(hipStreamWaitEvent(caller, e->ev_out[0], 0));
        return 0;
    }
    if (graph) {
        for (int i = 0; i < nl; ++i) {
            Lane &l = L[i];
            if (!l.g->step) {
                GD_TRY(capture_begin(l.s));
                int r = step_body(l);
                int r2 = capture_end(l.s, &l.g->step);
                if (r || r2) return 1;
            }
        }
        cold.unlock();
        for (int k = 0; k < e->num_step; ++k)
            for (int i = 0; i < nl; ++i) GD_HIP(hipGraphLaunch(L[i].g->step, L[i].s));
    } else {
        cold.unlock();
        for (int k = 0; k < e->num_step; ++k)
            for (int i = 0; i < nl; ++i) GD_TRY(step_body(L[i]));
    }
    for (int i = 0; i < nl; ++i) {
        Lane &l = L[i];
        GD_TRY(launch_blur_finish(l.y, x_out_dev + (size_t)start[i] * 1024 * C, l.b, C, inverse_scale, l.s));
        GD_HIP(hipEventRecord(e->ev_out[i], l.s));
        GD_HIP(hipStreamWaitEvent(caller, e->ev_out[i], 0));
    }
    return 0;
}

// ---- profiling ----
int gd_engine_profile(gd_engine *e, int enable) {
    OptScope scope(e ? e->opts : nullptr);
    GD_CHECK(e, "null engine");
    e->prof = enable != 0;
    if (enable) {
        e->prof_acc.clear();
        for (auto &p : e->pending) { (void)hipEventDestroy(p.a); (void)hipEventDestroy(p.b); }
        e->pending.clear();
    }
    return 0;
}

int gd_engine_profile_read(gd_engine *e, gd_prof_entry *out, int cap, int *n) {
    OptScope scope(e ? e->opts : nullptr);
    GD_CHECK(e && !e->plan_only, "engine has no GPU");
    DeviceScope dscope(e);
    GD_HIP(hipSetDevice(e->device));
    GD_HIP(hipDeviceSynchronize());
    for (auto &p : e->pending) {
        float ms = 0;
        if (hipEventElapsedTime(&ms, p.a, p.b) == hipSuccess) {
            ProfAcc &a = e->prof_acc[p.name];
            a.ms += ms; a.flops += p.flops; a.bytes += p.bytes; a.launches++;
        }
        (void)hipEventDestroy(p.a);
        (void)hipEventDestroy(p.b);
    }
    e->pending.clear();
    int i = 0;
    for (auto &kv : e->prof_acc) {
        if (out && i < cap) {
            memset(&out[i], 0, sizeof out[i]);
            strncpy(out[i].name, kv.first.c_str(), sizeof(out[i].name) - 1);
            out[i].ms = kv.second.ms; out[i].flops = kv.second.flops; out[i].bytes = kv.second.bytes;
            out[i].launches = kv.second.launches;
        }
        ++i;
    }
    if (n) *n = i;
    return 0;
}

// ---- stand-alone operators ----
int gd_ab_step(const float *u_dev, const float *coef_dev, const float *new_eps_dev, const float *hist_in_dev, float *u_out_dev,
               float *hist_out_dev, int n_hist, long long n_pairs, void *stream) {
    GD_CHECK(u_dev && coef_dev && new_eps_dev && u_out_dev && (n_hist == 0 || (hist_in_dev && hist_out_dev)), "null pointer");
    return launch_ab_step(u_dev, coef_dev, new_eps_dev, hist_in_dev, u_out_dev, hist_out_dev, n_hist, n_pairs, (hipStream_t)stream);
}

int gd_ab_step_scalar(const float *x_dev, const float *coef_dev, const float *new_eps_dev, const float *hist_in_dev,
                      float *x_out_dev, float *hist_out_dev, int n_hist, long long n, void *stream) {
    GD_CHECK(x_dev && coef_dev && new_eps_dev && x_out_dev && (n_hist == 0 || (hist_in_dev && hist_out_dev)), "null pointer");
    return launch_ab_step_scalar(x_dev, coef_dev, new_eps_dev, hist_in_dev, x_out_dev, hist_out_dev, n_hist, n, (hipStream_t)stream);
}

int gd_bmm2x2(const float *m_dev, const float *in_dev, float *out_dev, long long n_pairs, int per_sample,
              long long pairs_per_sample, void *stream) {
    GD_CHECK(m_dev && in_dev && out_dev, "null pointer");
    return launch_bmm2x2(m_dev, in_dev, out_dev, n_pairs, per_sample, pairs_per_sample, (hipStream_t)stream);
}

int gd_dct2d_32(const float *in_dev, float *out_dev, int batch, int channels, int inverse, void *stream) {
    GD_CHECK(in_dev && out_dev && batch >= 1 && channels >= 1, "bad argument");
    return launch_dct2d_32(in_dev, out_dev, batch, channels, inverse, (hipStream_t)stream);
}

int gd_conv2d(const float *x_dev, const float *w_dev, const float *bias_dev, const float *res_dev, float *out_dev, int batch,
              int H, int W, int Cin, int Cout, int KH, int KW, int stride, int pad, float alpha, float out_scale, void *stream) {
    GD_CHECK(x_dev && w_dev && out_dev, "null pointer");
    GemmParams g{};
    g.a0 = x_dev; g.C0 = Cin; g.lda0 = Cin; g.lda1 = 4;
    g.H = H; g.W = W; g.KH = KH; g.KW = KW; g.stride = stride; g.pad = pad;
    g.OH = (H + 2 * pad - KH) / stride + 1;
    g.OW = (W + 2 * pad - KW) / stride + 1;
    g.M = batch * g.OH * g.OW; g.N = Cout; g.K = KH * KW * Cin;
    g.b = w_dev; g.ldb = g.K;
    g.out = out_dev; g.ldc = Cout; g.bias = bias_dev; g.res = res_dev; g.ldres = Cout;
    g.rows_per_image = g.OH * g.OW;
    g.alpha = alpha; g.out_scale = out_scale;
    return launch_igemm(g, 1, (hipStream_t)stream);
}

int gd_conv3x3_winograd2d(const float *x_dev, const float *w_dev, float *u_scratch_dev, const float *bias_dev, const float *res_dev,
                          float *out_dev, int batch, int H, int W, int Cin, int Cout, float alpha, float out_scale, void *stream) {
    GD_CHECK(x_dev && w_dev && u_scratch_dev && out_dev, "null pointer");
    GemmParams g{};
    g.a0 = x_dev; g.C0 = Cin; g.lda0 = Cin; g.lda1 = 4;
    g.H = H; g.W = W; g.KH = 3; g.KW = 3; g.stride = 1; g.pad = 1; g.OH = H; g.OW = W;
    g.M = batch * H * W; g.N = Cout; g.K = 9 * Cin;
    g.b = w_dev; g.ldb = g.K;
    g.out = out_dev; g.ldc = Cout; g.bias = bias_dev; g.res = res_dev; g.ldres = Cout;
    g.rows_per_image = H * W;
    g.alpha = alpha; g.out_scale = out_scale;
    if (H == 4 && W == 4) {  // 4x4 images: the channel-split launch (scratch of its own here; the engine keeps one)
        g.qs_slices = 1;
        const int ks = conv3x3_wino2d_splitk_factor(g);
        GD_CHECK(ks > 1, "conv3x3_winograd2d: shape not eligible (B=%d H=%d W=%d Cin=%d Cout=%d)", batch, H, W, Cin, Cout);
        float *ws = nullptr;
        GD_HIP(hipMalloc((void **)&ws, (size_t)ks * g.M * g.N * sizeof(float)));
        GD_TRY(launch_wino2d_weights(w_dev, u_scratch_dev, Cout, Cin, (hipStream_t)stream));
        g.b_wino = u_scratch_dev;
        g.wk_partial = ws;
        const int r = launch_conv3x3_wino2d_splitk(g, (hipStream_t)stream);
        (void)hipStreamSynchronize((hipStream_t)stream);
        (void)hipFree(ws);
        return r;
    }
    GD_CHECK(conv3x3_wino2d_shape_ok(g), "conv3x3_winograd2d: shape not eligible (B=%d H=%d W=%d Cin=%d Cout=%d)", batch, H, W, Cin, Cout);
    GD_TRY(launch_wino2d_weights(w_dev, u_scratch_dev, Cout, Cin, (hipStream_t)stream));
    g.b_wino = u_scratch_dev;
    return launch_conv3x3_wino2d(g, (hipStream_t)stream);
}

int gd_conv3x3_winograd42(const float *x_dev, const float *w_dev, float *u_scratch_dev, const float *bias_dev, const float *res_dev,
                          float *out_dev, int batch, int H, int W, int Cin, int Cout, float alpha, float out_scale, void *stream) {
    GD_CHECK(x_dev && w_dev && u_scratch_dev && out_dev, "null pointer");
    GemmParams g{};
    g.a0 = x_dev; g.C0 = Cin; g.lda0 = Cin; g.lda1 = 4;
    g.H = H; g.W = W; g.KH = 3; g.KW = 3; g.stride = 1; g.pad = 1; g.OH = H; g.OW = W;
    g.M = batch * H * W; g.N = Cout; g.K = 9 * Cin;
    g.b = w_dev; g.ldb = g.K;
    g.out = out_dev; g.ldc = Cout; g.bias = bias_dev; g.res = res_dev; g.ldres = Cout;
    g.rows_per_image = H * W;
    g.alpha = alpha; g.out_scale = out_scale;
    GD_CHECK(conv3x3_wino42_shape_ok(g), "conv3x3_winograd42: shape not eligible (B=%d H=%d W=%d Cin=%d Cout=%d)", batch, H, W, Cin, Cout);
    GD_TRY(launch_wino42_weights(w_dev, u_scratch_dev, Cout, Cin, (hipStream_t)stream));
    g.b_wino = u_scratch_dev;
    return launch_conv3x3_wino42(g, (hipStream_t)stream);
}

long long gd_conv3x3_winograd2d_split6_weight_bytes(int Cin, int Cout) { return conv3x3_wino2d_split6_weight_bytes(Cin, Cout); }

int gd_conv3x3_winograd2d_split6_weights(const float *w_dev, void *u_dev, int Cin, int Cout, void *stream) {
    GD_CHECK(w_dev && u_dev, "null pointer");
    GD_CHECK(Cin % 16 == 0 && Cout % 64 == 0, "conv3x3_winograd2d_split6_weights: Cin %% 16 and Cout %% 64 must be 0 (Cin=%d Cout=%d)", Cin, Cout);
    return launch_wino2d_split6_weights(w_dev, (unsigned short *)u_dev, Cout, Cin, (hipStream_t)stream);
}

int gd_conv3x3_winograd2d_split6(const float *x_dev, const void *u_dev, const float *bias_dev, const float *res_dev, float *out_dev, int batch,
                                 int H, int W, int Cin, int Cout, float out_scale, void *stream) {
    GD_CHECK(x_dev && u_dev && out_dev, "null pointer");
    return launch_conv3x3_wino2d_split6(x_dev, (const unsigned short *)u_dev, bias_dev, res_dev, out_dev, batch, H, W, Cin, Cout, out_scale,
                                        (hipStream_t)stream);
}

int gd_conv3x3_winograd2d_split6_gn(const float *x_dev, const void *u_dev, const float *bias_dev, const float *res_dev, float *out_dev,
                                    const float *gamma_dev, const float *beta_dev, float *gn_out_dev, int groups, int act, int batch, int H, int W,
                                    int Cin, int Cout, float out_scale, void *stream) {
    GD_CHECK(x_dev && u_dev && gamma_dev && beta_dev && gn_out_dev, "null pointer");
    GD_CHECK(groups >= 1 && Cout % groups == 0, "conv3x3_winograd2d_split6_gn: %d channels do not split into %d groups", Cout, groups);
    GemmParams g{};
    g.a0 = x_dev; g.C0 = Cin; g.lda0 = Cin; g.lda1 = 4;
    g.H = H; g.W = W; g.KH = 3; g.KW = 3; g.stride = 1; g.pad = 1; g.OH = H; g.OW = W;
    g.M = batch * H * W; g.N = Cout; g.K = 9 * Cin;
    g.ldb = g.K;
    g.out = out_dev ? out_dev : gn_out_dev; g.ldc = Cout; g.bias = bias_dev; g.res = res_dev; g.ldres = Cout;
    g.rows_per_image = H * W;
    g.alpha = 1.f; g.out_scale = out_scale;
    g.egn_out = gn_out_dev; g.egn_gamma = gamma_dev; g.egn_beta = beta_dev; g.egn_cpg = Cout / groups; g.egn_act = act;
    g.egn_skip_raw = out_dev ? 0 : 1;
    GD_CHECK(conv3x3_wino2d_split6_gemm_ok(g), "conv3x3_winograd2d_split6_gn: shape not eligible (B=%d H=%d W=%d Cin=%d Cout=%d groups=%d; 16x16 images, 4 / 8 / 16 channels per group)",
             batch, H, W, Cin, Cout, groups);
    return launch_conv3x3_wino2d_split6_gemm(g, (const unsigned short *)u_dev, (hipStream_t)stream);
}

int gd_conv3x3_winograd42_weights(const float *w_dev, float *u_dev, int Cin, int Cout, void *stream) {
    GD_CHECK(w_dev && u_dev, "null pointer");
    return launch_wino42_weights(w_dev, u_dev, Cout, Cin, (hipStream_t)stream);
}

int gd_conv3x3_winograd42_pretransformed(const float *x_dev, const float *u_dev, const float *bias_dev, const float *res_dev, float *out_dev,
                                         int batch, int H, int W, int Cin, int Cout, float alpha, float out_scale, void *stream) {
    GD_CHECK(x_dev && u_dev && out_dev, "null pointer");
    GemmParams g{};
    g.a0 = x_dev; g.C0 = Cin; g.lda0 = Cin; g.lda1 = 4;
    g.H = H; g.W = W; g.KH = 3; g.KW = 3; g.stride = 1; g.pad = 1; g.OH = H; g.OW = W;
    g.M = batch * H * W; g.N = Cout; g.K = 9 * Cin;
    g.b = u_dev; g.ldb = g.K;  // (only the Winograd-domain copy is read)
    g.out = out_dev; g.ldc = Cout; g.bias = bias_dev; g.res = res_dev; g.ldres = Cout;
    g.rows_per_image = H * W;
    g.alpha = alpha; g.out_scale = out_scale;
    GD_CHECK(conv3x3_wino42_shape_ok(g), "conv3x3_winograd42_pretransformed: shape not eligible (B=%d H=%d W=%d Cin=%d Cout=%d)", batch, H, W, Cin, Cout);
    g.b_wino = u_dev;
    return launch_conv3x3_wino42(g, (hipStream_t)stream);
}

int gd_conv3x3_winograd42_gn(const float *x_dev, const float *w_dev, float *u_scratch_dev, const float *bias_dev, const float *res_dev,
                             float *out_dev, const float *gamma_dev, const float *beta_dev, float *gn_out_dev, int groups, int act,
                             int batch, int H, int W, int Cin, int Cout, float alpha, float out_scale, void *stream) {
    GD_CHECK(x_dev && w_dev && u_scratch_dev && gamma_dev && beta_dev && gn_out_dev, "null pointer");
    GD_CHECK(groups >= 1 && Cout % groups == 0, "conv3x3_winograd42_gn: %d channels do not split into %d groups", Cout, groups);
    GemmParams g{};
    g.a0 = x_dev; g.C0 = Cin; g.lda0 = Cin; g.lda1 = 4;
    g.H = H; g.W = W; g.KH = 3; g.KW = 3; g.stride = 1; g.pad = 1; g.OH = H; g.OW = W;
    g.M = batch * H * W; g.N = Cout; g.K = 9 * Cin;
    g.b = w_dev; g.ldb = g.K;
    g.out = out_dev ? out_dev : gn_out_dev; g.ldc = Cout; g.bias = bias_dev; g.res = res_dev; g.ldres = Cout;
    g.rows_per_image = H * W;
    g.alpha = alpha; g.out_scale = out_scale;
    g.egn_out = gn_out_dev; g.egn_gamma = gamma_dev; g.egn_beta = beta_dev; g.egn_cpg = Cout / groups; g.egn_act = act;
    g.egn_skip_raw = out_dev ? 0 : 1;
    GD_CHECK(conv3x3_wino42_shape_ok(g), "conv3x3_winograd42_gn: shape not eligible (B=%d H=%d W=%d Cin=%d Cout=%d groups=%d)", batch, H, W, Cin,
             Cout, groups);
    GD_TRY(launch_wino42_weights(w_dev, u_scratch_dev, Cout, Cin, (hipStream_t)stream));
    g.b_wino = u_scratch_dev;
    return launch_conv3x3_wino42(g, (hipStream_t)stream);
}

int gd_group_norm(const float *x_dev, const float *scale_dev, const float *bias_dev, float *out_dev, int batch, int HW, int C,
                  int groups, int act, void *stream) {
    GD_CHECK(x_dev && scale_dev && bias_dev && out_dev, "null pointer");
    hipStream_t s = (hipStream_t)stream;
    GD_CHECK(groups >= 1 && C % groups == 0, "group_norm: %d channels do not split into %d groups", C, groups);
    if (C % 4 != 0 || (C / groups) % 4 != 0) {  // groups that are not whole float4 columns: the generic path
        float *mr = nullptr;
        GD_HIP(hipMalloc((void **)&mr, (size_t)batch * groups * 2 * sizeof(float)));
        int r = launch_gn_generic(x_dev, C, nullptr, 0, mr, scale_dev, bias_dev, out_dev, batch, HW, groups, act, 1e-6f, s);
        (void)hipStreamSynchronize(s);
        (void)hipFree(mr);
        return r;
    }
    const int sl = gn_slices(batch, HW);
    float *qs = nullptr;
    GD_HIP(hipMalloc((void **)&qs, (size_t)batch * sl * (C / 4) * 2 * sizeof(float)));
    int r = launch_gn_stats(x_dev, C, qs, batch, HW, sl, s);
    if (!r) r = launch_gn_apply(x_dev, C, nullptr, 0, qs, sl, nullptr, 0, scale_dev, bias_dev, out_dev, batch, HW, groups, act, 1e-6f, 0, nullptr, s);
    (void)hipStreamSynchronize(s);
    (void)hipFree(qs);
    return r;
}

int gd_fir_resample(const float *x_dev, float *out_dev, int batch, int H, int W, int C, int up, void *stream) {
    GD_CHECK(x_dev && out_dev, "null pointer");
    if (up >= 2) return launch_naive_resample2(x_dev, out_dev, batch, H, W, C, up == 3, (hipStream_t)stream);
    return up ? launch_fir_up2(x_dev, out_dev, batch, H, W, C, (hipStream_t)stream)
              : launch_fir_down2(x_dev, out_dev, batch, H, W, C, (hipStream_t)stream);
}

int gd_attention(const float *qkv_dev, float *scratch_dev, float *out_dev, int batch, int N, int C, void *stream) {
    GD_CHECK(qkv_dev && scratch_dev && out_dev, "null pointer");
    hipStream_t s = (hipStream_t)stream;
    GemmParams g{};
    g.a0 = qkv_dev; g.C0 = C; g.lda0 = 3 * C; g.lda1 = 4; g.H = 1; g.W = N; g.OH = 1; g.OW = N; g.KH = g.KW = 1; g.stride = 1;
    g.M = N; g.N = N; g.K = C; g.sA = (long long)N * 3 * C;
    g.b = qkv_dev + C; g.ldb = 3 * C; g.sB = g.sA;
    g.out = scratch_dev; g.ldc = N; g.sC = (long long)N * N;
    g.alpha = (float)std::pow((double)C, -0.5); g.out_scale = 1.f; g.rows_per_image = N;
    GD_TRY(launch_igemm(g, batch, s));
    GD_TRY(launch_softmax_rows(scratch_dev, (long long)batch * N, N, s));
    GemmParams h{};
    h.a0 = scratch_dev; h.C0 = N; h.lda0 = N; h.lda1 = 4; h.H = 1; h.W = N; h.OH = 1; h.OW = N; h.KH = h.KW = 1; h.stride = 1;
    h.M = N; h.N = C; h.K = N; h.sA = (long long)N * N;
    h.b = qkv_dev + 2 * C; h.ldb = 3 * C; h.b_kmajor = 1; h.sB = (long long)N * 3 * C;
    h.out = out_dev; h.ldc = C; h.sC = (long long)N * C;
    h.alpha = 1.f; h.out_scale = 1.f; h.rows_per_image = N;
    return launch_igemm(h, batch, s);
}

int gd_attention_fused(const float *qk_dev, const float *vT_dev, const float *bias_v_dev, float *out_dev, int batch, int N, int C,
                       void *stream) {
    GD_CHECK(qk_dev && vT_dev && out_dev, "null pointer");
    GD_CHECK(attn_fused_ok(N, C), "attention_fused: built for N = 256 tokens and C = 256 channels (N=%d C=%d)", N, C);
    return launch_attn_fused(qk_dev, 2 * C, qk_dev + C, 2 * C, vT_dev, bias_v_dev, nullptr, 0, 1.f, nullptr, out_dev, C, batch, N, C, (hipStream_t)stream);
}

int gd_attention_block_fused(const float *q_dev, const float *k_dev, const float *vT_dev, const float *bias_dev, const float *res_dev,
                             float out_scale, float *qstats_dev, float *out_dev, int batch, int N, int C, void *stream) {
    GD_CHECK(q_dev && k_dev && vT_dev && out_dev, "null pointer");
    GD_CHECK(attn_fused_ok(N, C), "attention_block_fused: built for N = 256 tokens and C = 256 channels (N=%d C=%d)", N, C);
    return launch_attn_fused(q_dev, C, k_dev, C, vT_dev, bias_dev, res_dev, C, out_scale, qstats_dev, out_dev, C, batch, N, C, (hipStream_t)stream);
}

}  // extern "C"
