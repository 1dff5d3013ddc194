// U-Net execution plan: the NCSN++ forward (cld_jax/models/ncsnpp.py:41-243) lowered to a flat list of
// kernel-level ops over an activation arena, plus the weight slots the flax parameter tree is packed into.
// Pure host code (no HIP): the plan can be built and inspected on a machine without a GPU.
#pragma once
#include <cstdint>
#include <map>
#include <string>
#include <vector>

#include "../../include/gddim_amd.h"

namespace gd {

enum OpKind {
    OP_FOURIER = 0,
    OP_GEMM,
    OP_GN_STATS,
    OP_GN_APPLY,
    OP_FIR_UP,
    OP_FIR_DOWN,
    OP_SOFTMAX,
    OP_AFFINE_PAD,
    OP_SCALE_ROWS,
    OP_GN_GENERIC,  // GroupNorm whose groups are not whole float4 columns (e.g. 192 channels in 32 groups of 6): direct two-kernel path
    OP_FIR_PAD,     // a0 (H, W, C0) -> out (H + 1, W + 1, C0): the [1,3,3,1]^2 / 64 FIR at full resolution with pads (2, 2) -- the first half
                    // of conv_downsample_2d (up_or_down_sampling.py:168-209); its 3x3 stride-2 VALID convolution follows as a GEMM
};

// external tensors are bound to caller pointers at run time
enum ExtKind { EXT_NONE = 0, EXT_X_IN, EXT_LABELS, EXT_EPS_OUT };

struct Tensor {
    std::string name;
    int H = 1, W = 1, C = 1;
    long long per_image = 0;  // floats per image
    long long offset = -1;    // arena offset (floats per image); address = arena + offset * batch
    int ext = EXT_NONE;
    int refs = 0;
    // companion tensor holding the fused GroupNorm quad statistics of this tensor (emitted by the producing GEMM), and
    // its slices per image; -1 when the producer does not emit them
    int stats_t = -1;
    int qs_slices = 0;
    int table_t = -1;  // companion holding the affine map of the GroupNorm that produced this tensor (see Op::gn_table)
    int fresh = 0;     // allocated above everything allocated before it (never in a recycled block): see Op::egn_peer
};

struct TRef {
    int t = -1;
    int col = 0;  // channel offset inside the tensor
};

struct Op {
    int kind = OP_GEMM;
    std::string label;
    // ---- GEMM / conv ----
    TRef a0, a1;
    int C0 = 0, C1 = 0, lda0 = 0, lda1 = 0;
    int H = 1, W = 1, OH = 1, OW = 1, KH = 1, KW = 1, stride = 1, pad = 0;
    int m_per_image = 1;  // rows contributed by one image
    int z_per_image = 0;  // 1: blockIdx.z = image, M = m_per_image (attention matmuls)
    int N = 0, K = 0;
    int a_slot = -1;      // A rows from a weight slot instead of a tensor (V^T = W_v h^T in attention)
    long long a_off = 0;  // float offset into that slot
    int w_slot = -1;      // B from a weight slot ([N][K] row-major) ...
    TRef bten;            // ... or from a tensor
    int ldb = 0, b_kmajor = 0;
    long long sA = 0, sB = 0, sC = 0;  // per-image strides when z_per_image
    long long sR = 0, sQ = 0;          // ... of the residual and of the fused statistics (floats), when z_per_image
    int bias_slot = -1;
    int bias_off = 0;
    TRef rowbias;
    int ld_rowbias = 0;
    TRef res;
    int ldres = 0;
    TRef out;
    int ldc = 0;
    float alpha = 1.f, out_scale = 1.f;
    int in_act = 0;
    int temb = 0;  // part of the time-embedding path: depends on the label only (the engine computes one row when the batch shares one label)
    // split precision (bf16x3 mode only): GN_APPLY writes its output as bf16 hi|lo rows (split_out) when every consumer is
    // a weight GEMM that can read them (a_split); decided by Plan::mark_split at build time
    int split_out = 0, a_split = 0;
    // fused GroupNorm (bf16x3 mode, run-time decision): a GN_APPLY op and the 3x3 convolution that is its only reader point at each
    // other (fuse_peer = op index).  When the halo kernel takes the convolution at the current batch size the apply op only
    // writes gn_table (per image [C scale | C shift]) and the convolution reads gn_src, the apply op's raw input.
    int fuse_peer = -1;
    TRef gn_table, gn_src, gn_src1;
    int gn_C0 = 0, gn_C1 = 0;
    // GroupNorm in the producer's epilogue (fp32 mode, run-time decision): a single-source GN_APPLY op on a 16x16, 8x8 or 4x4 tensor
    // and the GEMM that produces that tensor point at each other (egn_peer = op index).  At 8x8 / 4x4 one wave's 32 x 32 accumulator
    // tile holds whole images and whole groups, so the epilogue can reduce (mean, rstd) in registers and write the normalised
    // (+ swish) tensor itself; at 16x16 an image is one workgroup tile of the two-dimensional Winograd kernel, which adds the
    // column sums of its waves through LDS; the apply op is then skipped.  egn_sole: the apply op is the only reader of the raw output, which
    // is then not stored at all.  The apply op's output tensor is `fresh` so that writing it early cannot hit live data.
    int egn_peer = -1, egn_sole = 0;
    // GroupNorm as a side output of the 1x1 shortcut convolution that reads the same raw tensor(s) (fp32 mode, run-time decision): the
    // GN_APPLY op of a residual block's GroupNorm_0 and the block's Conv_2 point at each other (side_peer = op index); Conv_2 is planned
    // in FRONT of the GroupNorm.  When the persistent short-K kernel takes the convolution at the current batch size its loader writes
    // the normalised tensor (GemmParams::gn_side_out) and the apply op only produces the per-image affine table, ahead of it.
    int side_peer = -1;
    TRef qstats;          // GEMM: quad statistics of the output (fused GroupNorm statistics), t = -1: none
    int qs_slices = 0;
    // ---- GroupNorm ----
    // GN_STATS: a0 -> stats (quad partial sums, slices chosen at run time)
    // GN_APPLY: a0 [+ a1] normalised; statistics from the quad sums stats (S0 slices per image) [+ stats1 (S1 slices)];
    //           S = 0 means "the run-time slice count of GN_STATS"
    int groups = 0, act = 0, gamma_slot = -1, beta_slot = -1;
    int S0 = 0, S1 = 0;
    TRef stats, stats1;
    // ---- softmax ----
    int rows_per_image = 0, cols = 0;
    // ---- affine pad ----
    int cin = 0, cpad = 0;
    float mul = 1.f, add = 0.f;
    // ---- accounting ----
    double flops_per_image = 0;  // algorithmic 2*MAC
    int is_conv3x3 = 0;
};

struct WSlot {
    std::string name;
    long long rows = 0, cols = 0;
    std::vector<float> host;  // packed [rows][cols]
};

enum ParamKind {
    P_VEC = 0,     // (n,) -> slot[row_off ... row_off+n)   (slot viewed as flat)
    P_DENSE,       // (in, out) -> rows [row_off, row_off+out), cols = in   (transposed)
    P_CONV,        // (kh, kw, I, O) -> [O][kh*kw*Ipad], input channels zero-padded to ipad
    P_CONV_FIR6,   // (3, 3, I, O) composed with the [1,3,3,1]^2/64 FIR -> [O][36*Ipad] (6x6, stride 2, pad 2)
};

struct Param {
    std::string name;  // flax path joined by '/'
    std::vector<int64_t> shape;
    int kind = P_VEC;
    int slot = -1;
    long long row_off = 0;
    int ipad = 0;
    bool loaded = false;
};

// An attention block whose four NIN projections are folded into two at parameter-load time (Plan::attn): the checkpoint's
// NIN_0..3 land in the raw slots, Plan::fold_params writes the derived ones the GEMMs read.
struct AttnFold {
    int C = 0;
    int raw_w = -1, raw_b = -1;    // [3C][C] rows q | k | v (row o = W[:, o]) and [3C] biases
    int raw_wo = -1, raw_bo = -1;  // NIN_3: [C][C], [C]
    int fold_w = -1, fold_b = -1;  // [2C][C]: rows 0..C-1 of W_q W_k^T (row c' = column c'), rows C..2C-1 of (W_v W_3)^T ; [2C] biases
};

// A bias slot that is the sum of two loaded ones (Plan::resblock: the shortcut convolution of an up-sampling block runs BEFORE the FIR, its
// bias is added after it -- folded into Conv_1's).
struct BiasFold { int dst = -1, a = -1, b = -1; };

struct Plan {
    gd_model_desc desc{};
    int CP = 0;  // U-Net input channels padded to a multiple of 4
    std::vector<Tensor> tensors;
    std::vector<Op> ops;
    std::vector<WSlot> slots;
    std::vector<Param> params;
    std::map<std::string, int> param_index;
    long long arena_per_image = 0;  // floats
    int t_x_in = -1, t_labels = -1, t_eps_out = -1;
    double flops_per_image = 0, conv3x3_flops_per_image = 0;
    // algorithmic flops of the reference the engine does not issue: the k and output projections of the folded attention blocks
    // (Plan::attn).  flops_per_image (what bench.py prices) includes them; the ops' own flops_per_image are what they really multiply.
    double folded_flops_per_image = 0;
    std::vector<AttnFold> attn_folds;
    std::vector<BiasFold> bias_folds;
    bool no_attn_fold = false, no_upswap = false;  // GDDIM_NO_ATTN_FOLD / GDDIM_NO_UPSWAP at build(): the written forms (A/B, tests)
    std::string options_at_build;                  // options_json() when build() ran: printed by json()
    void fold_params();  // derived weight slots from the loaded ones (float64 on the host); call once all parameters are loaded

    int build(const gd_model_desc &d);  // 0 ok
    void mark_split();
    void link_fused_gn(int gn_out_tensor);
    void link_epilogue_gn();
    int load_param(const std::string &name, const float *data, const int64_t *shape, int ndim);
    std::string json() const;

   private:
    // arena
    struct Block { long long off, size; long long since; };  // since: the op count from which the block has been free without interruption
    std::vector<Block> free_list_;
    int fresh_producer_ = -1;  // op index of the GEMM that may write the next `fresh` tensor early (set by Plan::gn)
    long long arena_alloc(long long size);
    long long arena_alloc_fresh(long long size, int producer);
    void arena_free(long long off, long long size);
    // builders
    int new_tensor(const std::string &name, int H, int W, int C, int ext = EXT_NONE, long long per_image = -1, bool fresh = false);
    void retain(int t) { tensors[t].refs++; }
    void release(int t);
    int new_slot(const std::string &name, long long rows, long long cols);
    void add_param(const std::string &name, std::vector<int64_t> shape, int kind, int slot, long long row_off, int ipad = 0);
    int gn(const std::string &prefix, const std::string &pname, int xa, int Ca, int xb, int Cb, int H, int W, int act);
    int conv(const std::string &label, const std::string &pname, int xa, int Ca, int xb, int Cb, int H, int W, int ksize,
             int O, int res_t, float out_scale, int rowbias_t, int rowbias_col, int out_t = -1, bool want_stats = false);
    void attach_stats(int t, int rows_per_image);
    int resblock(int idx, int xa, int Ca, int xb, int Cb, int H, int W, int O, bool up, bool down, int &outH);
    int attn(int idx, int x, int H, int W, int C);
    int attn_unfolded(int idx, int x, int H, int W, int C);
    int temb_t_ = -1, tproj_t_ = -1, tproj_slot_w_ = -1, tproj_slot_b_ = -1;
    std::vector<int> rb_tproj_off_;
    int sum_cout_ = 0;
};

}  // namespace gd
