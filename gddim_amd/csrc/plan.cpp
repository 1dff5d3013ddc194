// Builds the NCSN++ execution plan.  Every block below follows the reference module it lowers:
//   NCSNpp.__call__           cld_jax/models/ncsnpp.py:41-243
//   ResnetBlockBigGANpp       cld_jax/models/layerspp.py:180-227
//   AttnBlockpp               cld_jax/models/layerspp.py:61-83
//   Downsample(fir,with_conv) cld_jax/models/layerspp.py:134-140 -> up_or_down_sampling.py:40-73,168-209
// Parameter names are flax's auto-names (module class + per-parent call counter), SURVEY.md App. A.2.
#include "plan.h"

#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <sstream>

#include "common.h"
#include "options.h"

namespace gd {

// ---------------------------------------------------------------- arena (first fit, per-image floats)
long long Plan::arena_alloc(long long size) {
    size = (size + 63) / 64 * 64;
    for (size_t i = 0; i < free_list_.size(); ++i) {
        if (free_list_[i].size >= size) {
            long long off = free_list_[i].off;
            free_list_[i].off += size;
            free_list_[i].size -= size;
            if (free_list_[i].size == 0) free_list_.erase(free_list_.begin() + i);
            return off;
        }
    }
    long long off = arena_per_image;
    // extend: if the last free block touches the end, grow from it
    if (!free_list_.empty() && free_list_.back().off + free_list_.back().size == arena_per_image) {
        off = free_list_.back().off;
        free_list_.pop_back();
    }
    arena_per_image = off + size;
    return off;
}

// A block for a tensor that its producer's epilogue may write EARLY, at op index `producer` (Op::egn_peer: the GroupNorm output written
// by the GEMM in front of the GroupNorm op): the block must have been free, without interruption, since before that op -- a block freed
// or handed out in between could belong to a tensor that lives across the early write.  Free blocks carry the op count since which they
// have been free; when none qualifies the arena grows (round 2 always grew: 3.0 of 4.5 M floats per image were such one-shot blocks).
long long Plan::arena_alloc_fresh(long long size, int producer) {
    size = (size + 63) / 64 * 64;
    if (producer >= 0) {
        for (size_t i = 0; i < free_list_.size(); ++i) {
            if (free_list_[i].size >= size && free_list_[i].since <= producer) {
                const long long off = free_list_[i].off;
                free_list_[i].off += size;
                free_list_[i].size -= size;
                if (free_list_[i].size == 0) free_list_.erase(free_list_.begin() + i);
                return off;
            }
        }
    }
    const long long off = arena_per_image;
    arena_per_image = off + size;
    return off;
}

void Plan::arena_free(long long off, long long size) {
    size = (size + 63) / 64 * 64;
    Block b{off, size, (long long)ops.size()};
    auto it = std::lower_bound(free_list_.begin(), free_list_.end(), b,
                               [](const Block &x, const Block &y) { return x.off < y.off; });
    it = free_list_.insert(it, b);
    // coalesce with next / previous
    size_t i = it - free_list_.begin();
    if (i + 1 < free_list_.size() && free_list_[i].off + free_list_[i].size == free_list_[i + 1].off) {
        free_list_[i].size += free_list_[i + 1].size;
        free_list_[i].since = std::max(free_list_[i].since, free_list_[i + 1].since);  // (the merged block is as young as its youngest part)
        free_list_.erase(free_list_.begin() + i + 1);
    }
    if (i > 0 && free_list_[i - 1].off + free_list_[i - 1].size == free_list_[i].off) {
        free_list_[i - 1].size += free_list_[i].size;
        free_list_[i - 1].since = std::max(free_list_[i - 1].since, free_list_[i].since);
        free_list_.erase(free_list_.begin() + i);
    }
}

int Plan::new_tensor(const std::string &name, int H, int W, int C, int ext, long long per_image, bool fresh) {
    Tensor t;
    t.name = name;
    t.H = H; t.W = W; t.C = C;
    t.per_image = per_image >= 0 ? per_image : (long long)H * W * C;
    t.ext = ext;
    t.refs = 1;
    if (ext == EXT_NONE) t.offset = fresh ? arena_alloc_fresh(t.per_image, fresh_producer_) : arena_alloc(t.per_image);
    t.fresh = fresh ? 1 : 0;
    tensors.push_back(t);
    return (int)tensors.size() - 1;
}

void Plan::release(int t) {
    if (t < 0) return;
    Tensor &T = tensors[t];
    if (--T.refs == 0 && T.ext == EXT_NONE) {
        arena_free(T.offset, T.per_image);
        if (T.stats_t >= 0) release(T.stats_t);
        if (T.table_t >= 0) release(T.table_t);
    }
}

// Give tensor t a companion for the quad statistics its producing GEMM will emit (kernels/qstats.h): one slice per 32
// output rows, or per image when an image has 16 rows.  Other image sizes keep the stand-alone statistics kernel.
void Plan::attach_stats(int t, int rows_per_image) {
    int S = 0;
    if (rows_per_image % 32 == 0) S = rows_per_image / 32;
    else if (rows_per_image == 16) S = 1;
    if (S == 0 || tensors[t].C % 4 != 0 || tensors[t].ext != EXT_NONE) return;
    const int st = new_tensor(tensors[t].name + ".qstats", 1, 1, 1, EXT_NONE, (long long)S * (tensors[t].C / 4) * 2);
    tensors[t].stats_t = st;
    tensors[t].qs_slices = S;
}

int Plan::new_slot(const std::string &name, long long rows, long long cols) {
    WSlot s;
    s.name = name;
    s.rows = rows; s.cols = cols;
    s.host.assign((size_t)(rows * cols), 0.f);
    slots.push_back(std::move(s));
    return (int)slots.size() - 1;
}

void Plan::add_param(const std::string &name, std::vector<int64_t> shape, int kind, int slot, long long row_off, int ipad) {
    Param p;
    p.name = name; p.shape = std::move(shape); p.kind = kind; p.slot = slot; p.row_off = row_off; p.ipad = ipad;
    param_index[name] = (int)params.size();
    params.push_back(std::move(p));
}

// ---------------------------------------------------------------- GroupNorm (+swish): [stats] + apply
// returns the normalised tensor (H, W, Ca+Cb); inputs are not released
int Plan::gn(const std::string &label, const std::string &pname, int xa, int Ca, int xb, int Cb, int H, int W, int act) {
    const int C = Ca + Cb;
    const int G = std::min(C / 4, 32);  // nn.GroupNorm(num_groups=min(C // 4, 32))
    int gs = new_slot(pname + "/scale", 1, C), bs = new_slot(pname + "/bias", 1, C);
    add_param(pname + "/scale", {C}, P_VEC, gs, 0);
    add_param(pname + "/bias", {C}, P_VEC, bs, 0);
    if ((C / G) % 4 != 0 || Ca % 4 != 0) {
        // groups that cut through a float4 column (nf = 64: 192 channels -> 32 groups of 6): the quad statistics do not apply;
        // a direct (mean, rstd) pass over the sources and an element-wise apply (rare configurations, not tuned)
        int mr = new_tensor(label + ".mean_rstd", 1, 1, 2 * G);
        int out = new_tensor(label + ".out", H, W, C);
        Op a;
        a.kind = OP_GN_GENERIC; a.label = label + ".generic";
        a.a0 = {xa, 0}; a.C0 = Ca; a.a1 = {xb, 0}; a.C1 = Cb; a.H = H; a.W = W; a.groups = G;
        a.stats = {mr, 0}; a.act = act; a.gamma_slot = gs; a.beta_slot = bs; a.out = {out, 0};
        ops.push_back(a);
        release(mr);
        return out;
    }
    // quad statistics of each source: emitted by its producer, or computed here by the stand-alone kernel
    int qs[2] = {-1, -1}, S[2] = {0, 0}, tmp[2] = {-1, -1};
    const int src[2] = {xa, xb}, Cs[2] = {Ca, Cb};
    for (int i = 0; i < 2; ++i) {
        if (src[i] < 0 || Cs[i] == 0) continue;
        if (tensors[src[i]].stats_t >= 0) {
            qs[i] = tensors[src[i]].stats_t;
            S[i] = tensors[src[i]].qs_slices;
        } else {
            tmp[i] = new_tensor(label + ".qstats" + std::to_string(i), 1, 1, 1, EXT_NONE, (long long)GN_MAX_SLICES * (Cs[i] / 4) * 2);
            Op st;
            st.kind = OP_GN_STATS; st.label = label + ".stats" + std::to_string(i);
            st.a0 = {src[i], 0}; st.C0 = Cs[i]; st.H = H; st.W = W; st.stats = {tmp[i], 0};
            ops.push_back(st);
            qs[i] = tmp[i];
            S[i] = 0;  // run-time slice count
        }
    }
    // candidate for normalisation in the producer's epilogue (Op::egn_peer): one source that carries fused statistics, an image of
    // 16, 64 or 256 pixels (256: one workgroup tile of the two-dimensional Winograd kernels), groups of 4..32 channels that divide a
    // 32-column tile
    const int cpg = C / G;
    const bool egn_candidate = Cb == 0 && (H * W == 16 || H * W == 64 || (H == 16 && W == 16)) && tensors[xa].stats_t >= 0 && tensors[xa].C == C &&
                               (cpg == 4 || cpg == 8 || cpg == 16 || cpg == 32) && C % 32 == 0;
    fresh_producer_ = -1;
    if (egn_candidate)
        for (int j = (int)ops.size() - 1; j >= 0; --j)
            if (ops[j].out.t == xa) { fresh_producer_ = ops[j].kind == OP_GEMM ? j : -1; break; }
    int out = new_tensor(label + ".out", H, W, C, EXT_NONE, -1, egn_candidate);
    tensors[out].table_t = new_tensor(label + ".affine", 1, 1, 2 * C);  // lives as long as the output
    Op a;
    a.kind = OP_GN_APPLY; a.label = label + ".apply";
    a.a0 = {xa, 0}; a.C0 = Ca; a.a1 = {xb, 0}; a.C1 = Cb; a.H = H; a.W = W; a.groups = G;
    a.stats = {qs[0], 0}; a.S0 = S[0]; a.stats1 = {qs[1], 0}; a.S1 = S[1];
    a.act = act; a.gamma_slot = gs; a.beta_slot = bs; a.out = {out, 0};
    a.gn_table = {tensors[out].table_t, 0};
    ops.push_back(a);
    release(tmp[0]);
    release(tmp[1]);
    return out;
}

// ---------------------------------------------------------------- nn.Conv (SAME) as implicit GEMM
// pname: flax module path holding {kernel, bias}.  out_t >= 0 writes into an existing (external) tensor.
int Plan::conv(const std::string &label, const std::string &pname, int xa, int Ca, int xb, int Cb, int H, int W, int ksize,
               int O, int res_t, float out_scale, int rowbias_t, int rowbias_col, int out_t, bool want_stats) {
    const int Cin = Ca + Cb;
    int ws = new_slot(pname + "/kernel", O, (long long)ksize * ksize * Cin);
    int bs = new_slot(pname + "/bias", 1, O);
    add_param(pname + "/kernel", {ksize, ksize, Cin, O}, P_CONV, ws, 0, Cin);
    add_param(pname + "/bias", {O}, P_VEC, bs, 0);
    int out = out_t >= 0 ? out_t : new_tensor(label + ".out", H, W, O);
    if (want_stats && out_t < 0) attach_stats(out, H * W);
    Op g;
    g.kind = OP_GEMM; g.label = label;
    g.a0 = {xa, 0}; g.C0 = Ca; g.lda0 = tensors[xa].C;
    if (Cb) { g.a1 = {xb, 0}; g.C1 = Cb; g.lda1 = tensors[xb].C; }
    g.H = H; g.W = W; g.OH = H; g.OW = W; g.KH = g.KW = ksize; g.stride = 1; g.pad = ksize / 2;
    g.m_per_image = H * W; g.N = O; g.K = ksize * ksize * Cin;
    g.w_slot = ws; g.ldb = g.K; g.bias_slot = bs;
    if (rowbias_t >= 0) { g.rowbias = {rowbias_t, rowbias_col}; g.ld_rowbias = tensors[rowbias_t].C; }
    if (res_t >= 0) { g.res = {res_t, 0}; g.ldres = tensors[res_t].C; }
    g.out = {out, 0}; g.ldc = tensors[out].C; g.out_scale = out_scale;
    if (tensors[out].stats_t >= 0) { g.qstats = {tensors[out].stats_t, 0}; g.qs_slices = tensors[out].qs_slices; }
    g.flops_per_image = 2.0 * H * W * O * g.K;
    g.is_conv3x3 = ksize == 3;
    ops.push_back(g);
    return out;
}

// ---------------------------------------------------------------- ResnetBlockBigGANpp (layerspp.py:180-227)
int Plan::resblock(int idx, int xa, int Ca, int xb, int Cb, int H, int W, int O, bool up, bool down, int &outH) {
    const std::string n = "ResnetBlockBigGANpp_" + std::to_string(idx);
    const int C = Ca + Cb;
    const float scale = desc.skip_rescale ? (float)(1.0 / std::sqrt(2.0)) : 1.f;
    // shortcut: conv1x1 iff C != out_ch or up or down (layerspp.py:221-222).  Without resampling it reads the block input itself --
    // what GroupNorm_0 normalises -- and is planned FIRST, so that its loader can write act(GroupNorm_0(x)) on the way (Op::side_peer)
    int xs = -1;
    const bool early_shortcut = C != O && !up && !down;
    if (early_shortcut) xs = conv(n + ".conv2", n + "/Conv_2", xa, Ca, xb, Cb, H, W, 1, O, -1, 1.f, -1, 0);
    const int shortcut_op = early_shortcut ? (int)ops.size() - 1 : -1;
    // h = act(GroupNorm_0(x))
    int h0 = gn(n + ".gn0", n + "/GroupNorm_0", xa, Ca, xb, Cb, H, W, 1);
    if (shortcut_op >= 0 && Ca % 4 == 0 && Cb % 4 == 0) {
        const int apply_op = (int)ops.size() - 1;  // (Plan::gn pushes the apply op last)
        if (ops[apply_op].kind == OP_GN_APPLY && ops[apply_op].out.t == h0) {
            ops[apply_op].side_peer = shortcut_op;
            ops[shortcut_op].side_peer = apply_op;
        }
    }
    int xs_src = xa;  // shortcut source (single tensor unless concat)
    int up_b2 = -1;   // Conv_2's bias slot when the up-sampling shortcut runs before the FIR
    int H2 = H, W2 = W;
    int xr = -1;  // resampled x
    if (up || down) {
        // both h and x are FIR-resampled (layerspp.py:198-211); never combined with a concat input
        H2 = up ? 2 * H : H / 2;
        W2 = up ? 2 * W : W / 2;
        int h0r = new_tensor(n + ".h_resampled", H2, W2, C);
        Op f;
        f.kind = up ? OP_FIR_UP : OP_FIR_DOWN; f.label = n + (up ? ".fir_up_h" : ".fir_down_h");
        f.act = desc.no_fir ? 1 : 0;  // 1: naive_upsample_2d / naive_downsample_2d instead of the FIR (layerspp.py:203-211)
        f.a0 = {h0, 0}; f.C0 = C; f.H = H; f.W = W; f.out = {h0r, 0};
        ops.push_back(f);
        release(h0);
        h0 = h0r;
        // the shortcut of a resampling block is Conv_2(FIR(x)) (layerspp.py:198-222).  The FIR acts per channel and Conv_2 per pixel: they
        // commute up to Conv_2's bias (the FIR's zero boundary does not pass a constant through unchanged).  Down: FIR first, the 1x1
        // convolution at the low resolution, as written.  Up: the 1x1 convolution WITHOUT its bias at the low resolution -- a quarter of
        // the rows --, then the FIR on its output; the bias joins Conv_1's, which adds the shortcut as its residual (Plan::bias_folds).
        if (up && !no_upswap) {
            const int xlo = conv(n + ".conv2", n + "/Conv_2", xa, C, -1, 0, H, W, 1, O, -1, 1.f, -1, 0);
            up_b2 = ops.back().bias_slot;
            ops.back().bias_slot = -1;
            folded_flops_per_image += 2.0 * ((double)H2 * W2 - (double)H * W) * O * C;  // (the reference multiplies at the high resolution)
            xs = new_tensor(n + ".shortcut_resampled", H2, W2, O);
            f.label = n + ".fir_up_shortcut";
            f.a0 = {xlo, 0}; f.C0 = O; f.out = {xs, 0};
            ops.push_back(f);
            release(xlo);
        } else {
            xr = new_tensor(n + ".x_resampled", H2, W2, C);
            f.label = n + (up ? ".fir_up_x" : ".fir_down_x");
            f.a0 = {xa, 0}; f.out = {xr, 0};
            ops.push_back(f);
            xs_src = xr;
        }
    }
    // h = Conv_0(h) + Dense_0(act(temb))[:, None, None, :]
    const int tcol = tproj_t_ >= 0 ? rb_tproj_off_[idx] : 0;
    int h1 = conv(n + ".conv0", n + "/Conv_0", h0, C, -1, 0, H2, W2, 3, O, -1, 1.f, tproj_t_, tcol, -1, true);
    if (!up && !down) link_fused_gn(h0);  // the source of GroupNorm_0 is the block input, alive until the end of the block
    release(h0);
    if (tproj_t_ >= 0) {
        add_param(n + "/Dense_0/kernel", {desc.nf * 4, O}, P_DENSE, tproj_slot_w_, tcol);
        add_param(n + "/Dense_0/bias", {O}, P_VEC, tproj_slot_b_, tcol);
    }
    // h = act(GroupNorm_1(h)); Dropout is the identity at inference
    int h2 = gn(n + ".gn1", n + "/GroupNorm_1", h1, O, -1, 0, H2, W2, 1);  // h1 stays alive until Conv_1 is out (fused GroupNorm)
    // the shortcut of a resampling block reads the resampled input
    int res_t;
    if (C != O || up || down) {
        if ((up || down) && xs < 0) xs = conv(n + ".conv2", n + "/Conv_2", xs_src, C, -1, 0, H2, W2, 1, O, -1, 1.f, -1, 0);
        res_t = xs;
    } else {
        res_t = xa;
    }
    // (x + Conv_1(h)) / sqrt(2)
    int out = conv(n + ".conv1", n + "/Conv_1", h2, O, -1, 0, H2, W2, 3, O, res_t, scale, -1, 0, -1, true);
    if (up_b2 >= 0) {
        BiasFold bf;
        bf.a = ops.back().bias_slot; bf.b = up_b2;
        bf.dst = new_slot(n + "/Conv_1/bias+Conv_2/bias", 1, O);
        ops.back().bias_slot = bf.dst;
        bias_folds.push_back(bf);
    }
    link_fused_gn(h2);
    release(h1);
    release(h2);
    release(xs);
    release(xr);
    outH = H2;
    return out;
}

// ---------------------------------------------------------------- AttnBlockpp (layerspp.py:61-83)
// Folded form (default; GDDIM_NO_ATTN_FOLD=1 at engine creation keeps the four projections): two of the four C x C projections are
// algebra, not work.  With q = h W_q + b_q, k = h W_k + b_k the score q_i . k_j = (h_i W_q + b_q) W_k^T h_j^T + (a term that depends on i
// only and cancels in the softmax over j), so q' = h (W_q W_k^T) + W_k b_q meets k' = h, the GroupNorm output itself; and because
// softmax rows sum to one, NIN_3(P (h W_v + b_v)) = P (h W_v W_3) + (b_v W_3 + b_3), so v'' = h (W_v W_3) and the output projection
// disappears: residual, 1/sqrt(2) and the fused statistics move to the P v'' epilogue.  The products W_q W_k^T and W_v W_3 are formed
// once per parameter load in float64 (Plan::fold_params) and rounded once.
int Plan::attn(int idx, int x, int H, int W, int C) {
    if (no_attn_fold) return attn_unfolded(idx, x, H, W, C);
    const std::string n = "AttnBlockpp_" + std::to_string(idx);
    const int HW = H * W;
    const float scale = desc.skip_rescale ? (float)(1.0 / std::sqrt(2.0)) : 1.f;
    int hn = gn(n + ".gn", n + "/GroupNorm_0", x, C, -1, 0, H, W, 0);
    AttnFold f;
    f.C = C;
    f.raw_w = new_slot(n + "/qkv_W", 3 * C, C); f.raw_b = new_slot(n + "/qkv_b", 1, 3 * C);
    for (int i = 0; i < 3; ++i) {
        add_param(n + "/NIN_" + std::to_string(i) + "/W", {C, C}, P_DENSE, f.raw_w, (long long)i * C);
        add_param(n + "/NIN_" + std::to_string(i) + "/b", {C}, P_VEC, f.raw_b, (long long)i * C);
    }
    f.raw_wo = new_slot(n + "/NIN_3/W", C, C); f.raw_bo = new_slot(n + "/NIN_3/b", 1, C);
    add_param(n + "/NIN_3/W", {C, C}, P_DENSE, f.raw_wo, 0);
    add_param(n + "/NIN_3/b", {C}, P_VEC, f.raw_bo, 0);
    f.fold_w = new_slot(n + "/fold_W", 2 * C, C); f.fold_b = new_slot(n + "/fold_b", 1, 2 * C);
    attn_folds.push_back(f);
    // q' = h (W_q W_k^T) + W_k b_q
    int qp = new_tensor(n + ".qf", H, W, C);
    {
        Op g;
        g.kind = OP_GEMM; g.label = n + ".q_proj";
        g.a0 = {hn, 0}; g.C0 = C; g.lda0 = C; g.H = H; g.W = W; g.OH = H; g.OW = W;
        g.m_per_image = HW; g.N = C; g.K = C; g.w_slot = f.fold_w; g.ldb = C; g.bias_slot = f.fold_b;
        g.out = {qp, 0}; g.ldc = C;
        g.flops_per_image = 2.0 * HW * C * C;
        ops.push_back(g);
    }
    // v''^T[c][token] = sum_k (W_v W_3)[k][c] h[token][k], per image, k-contiguous for the P v'' product
    int vT = new_tensor(n + ".vT", 1, 1, 1, EXT_NONE, (long long)C * HW);
    {
        Op g;
        g.kind = OP_GEMM; g.label = n + ".v_proj_T";
        g.a_slot = f.fold_w; g.a_off = (long long)C * C; g.C0 = C; g.lda0 = C; g.H = 1; g.W = C; g.OH = 1; g.OW = C;
        g.m_per_image = C; g.z_per_image = 1; g.N = HW; g.K = C;
        g.bten = {hn, 0}; g.ldb = C; g.b_kmajor = 0;
        g.sA = 0; g.sB = (long long)HW * C; g.sC = (long long)C * HW;
        g.out = {vT, 0}; g.ldc = HW;
        g.flops_per_image = 2.0 * HW * C * C;
        ops.push_back(g);
    }
    // w = softmax(q' h^T * C^-1/2)
    int S = new_tensor(n + ".scores", 1, 1, 1, EXT_NONE, (long long)HW * HW);
    {
        Op g;
        g.kind = OP_GEMM; g.label = n + ".qk";
        g.a0 = {qp, 0}; g.C0 = C; g.lda0 = C; g.H = 1; g.W = HW; g.OH = 1; g.OW = HW;
        g.m_per_image = HW; g.z_per_image = 1; g.N = HW; g.K = C;
        g.bten = {hn, 0}; g.ldb = C; g.b_kmajor = 0;
        g.sA = (long long)HW * C; g.sB = g.sA; g.sC = (long long)HW * HW;
        g.out = {S, 0}; g.ldc = HW; g.alpha = (float)std::pow((double)C, -0.5);
        g.flops_per_image = 2.0 * HW * HW * C;
        ops.push_back(g);
        Op sm;
        sm.kind = OP_SOFTMAX; sm.label = n + ".softmax"; sm.out = {S, 0}; sm.rows_per_image = HW; sm.cols = HW;
        ops.push_back(sm);
    }
    // (x + w v'' + (b_v W_3 + b_3)) / sqrt(2), with the statistics of the block output.  The fused core (attn256_f32_kernel) runs qk,
    // softmax and av as ONE launch whose workgroups store rows of `out` while others still read q' and h: `out` is allocated while
    // both are live, so first-fit cannot lay it over them at a shifted offset (64x64 configurations did: out 2 rows below q').
    int out = new_tensor(n + ".out", H, W, C);
    attach_stats(out, HW);
    release(qp);
    release(hn);
    {
        Op g;
        g.kind = OP_GEMM; g.label = n + ".av";
        g.a0 = {S, 0}; g.C0 = HW; g.lda0 = HW; g.H = 1; g.W = HW; g.OH = 1; g.OW = HW;
        g.m_per_image = HW; g.z_per_image = 1; g.N = C; g.K = HW;
        g.bten = {vT, 0}; g.ldb = HW; g.b_kmajor = 0;
        g.sA = (long long)HW * HW; g.sB = (long long)C * HW; g.sC = (long long)HW * C;
        g.bias_slot = f.fold_b; g.bias_off = C;
        g.res = {x, 0}; g.ldres = C; g.sR = (long long)HW * C;
        g.out = {out, 0}; g.ldc = C; g.out_scale = scale;
        if (tensors[out].stats_t >= 0) {
            g.qstats = {tensors[out].stats_t, 0}; g.qs_slices = tensors[out].qs_slices;
            g.sQ = (long long)g.qs_slices * (C / 4) * 2;
        }
        g.flops_per_image = 2.0 * HW * HW * C;
        ops.push_back(g);
    }
    release(S);
    release(vT);
    folded_flops_per_image += 2 * (2.0 * HW * C * C);  // NIN_1 (k) and NIN_3 (output) of the reference
    return out;
}

void Plan::fold_params() {
    for (const BiasFold &f : bias_folds)
        for (size_t i = 0; i < slots[f.dst].host.size(); ++i) slots[f.dst].host[i] = slots[f.a].host[i] + slots[f.b].host[i];
    for (const AttnFold &f : attn_folds) {
        const int C = f.C;
        const std::vector<float> &w = slots[f.raw_w].host, &b = slots[f.raw_b].host, &wo = slots[f.raw_wo].host, &bo = slots[f.raw_bo].host;
        std::vector<float> &fw = slots[f.fold_w].host, &fb = slots[f.fold_b].host;
        const float *Q = w.data(), *K = w.data() + (size_t)C * C, *V = w.data() + 2 * (size_t)C * C;  // row m = W[:, m]
        std::vector<double> acc((size_t)C * C);
        // rows 0..C-1: fold[c'][i] = sum_m W_q[i][m] W_k[c'][m] = sum_m Q[m][i] K[m][c']
        std::fill(acc.begin(), acc.end(), 0.0);
        for (int m = 0; m < C; ++m)
            for (int c = 0; c < C; ++c) {
                const double kc = K[(size_t)m * C + c];
                double *a = acc.data() + (size_t)c * C;
                const float *q = Q + (size_t)m * C;
                for (int i = 0; i < C; ++i) a[i] += kc * (double)q[i];
            }
        for (size_t i = 0; i < (size_t)C * C; ++i) fw[i] = (float)acc[i];
        for (int c = 0; c < C; ++c) {  // b'[c'] = sum_m b_q[m] W_k[c'][m]
            double s = 0;
            for (int m = 0; m < C; ++m) s += (double)b[m] * (double)K[(size_t)m * C + c];
            fb[c] = (float)s;
        }
        // rows C..2C-1: fold[C + c][k] = sum_m W_v[k][m] W_3[m][c] = sum_m V[m][k] wo[c][m]
        std::fill(acc.begin(), acc.end(), 0.0);
        for (int c = 0; c < C; ++c)
            for (int m = 0; m < C; ++m) {
                const double w3 = wo[(size_t)c * C + m];
                double *a = acc.data() + (size_t)c * C;
                const float *v = V + (size_t)m * C;
                for (int k = 0; k < C; ++k) a[k] += w3 * (double)v[k];
            }
        for (size_t i = 0; i < (size_t)C * C; ++i) fw[(size_t)C * C + i] = (float)acc[i];
        for (int c = 0; c < C; ++c) {  // b''[c] = sum_m b_v[m] W_3[m][c] + b_3[c]
            double s = bo[c];
            for (int m = 0; m < C; ++m) s += (double)b[2 * C + m] * (double)wo[(size_t)c * C + m];
            fb[C + c] = (float)s;
        }
    }
}

int Plan::attn_unfolded(int idx, int x, int H, int W, int C) {
    const std::string n = "AttnBlockpp_" + std::to_string(idx);
    const int HW = H * W;
    const float scale = desc.skip_rescale ? (float)(1.0 / std::sqrt(2.0)) : 1.f;
    int hn = gn(n + ".gn", n + "/GroupNorm_0", x, C, -1, 0, H, W, 0);
    // q, k = NIN_0..1(h) as one GEMM with N = 2C.  v is produced directly TRANSPOSED per image, V^T[c][token] =
    // sum_k W_v[c][k] h[token][k], so that the PV product below reads both operands k-contiguous; its bias moves to the
    // PV epilogue (softmax rows sum to one: P (V + 1 b^T) = P V + b^T).
    int wq = new_slot(n + "/qkv_W", 3 * C, C), bq = new_slot(n + "/qkv_b", 1, 3 * C);
    for (int i = 0; i < 3; ++i) {
        add_param(n + "/NIN_" + std::to_string(i) + "/W", {C, C}, P_DENSE, wq, (long long)i * C);
        add_param(n + "/NIN_" + std::to_string(i) + "/b", {C}, P_VEC, bq, (long long)i * C);
    }
    int qk = new_tensor(n + ".qk", H, W, 2 * C);
    {
        Op g;
        g.kind = OP_GEMM; g.label = n + ".qk_proj";
        g.a0 = {hn, 0}; g.C0 = C; g.lda0 = C; g.H = H; g.W = W; g.OH = H; g.OW = W;
        g.m_per_image = HW; g.N = 2 * C; g.K = C; g.w_slot = wq; g.ldb = C; g.bias_slot = bq;
        g.out = {qk, 0}; g.ldc = 2 * C;
        g.flops_per_image = 2.0 * HW * 2 * C * C;
        ops.push_back(g);
    }
    int vT = new_tensor(n + ".vT", 1, 1, 1, EXT_NONE, (long long)C * HW);
    {
        Op g;
        g.kind = OP_GEMM; g.label = n + ".v_proj_T";
        g.a_slot = wq; g.a_off = 2LL * C * C; g.C0 = C; g.lda0 = C; g.H = 1; g.W = C; g.OH = 1; g.OW = C;
        g.m_per_image = C; g.z_per_image = 1; g.N = HW; g.K = C;
        g.bten = {hn, 0}; g.ldb = C; g.b_kmajor = 0;
        g.sA = 0; g.sB = (long long)HW * C; g.sC = (long long)C * HW;
        g.out = {vT, 0}; g.ldc = HW;
        g.flops_per_image = 2.0 * HW * C * C;
        ops.push_back(g);
    }
    release(hn);
    // w = softmax(q k^T * C^-1/2)
    int S = new_tensor(n + ".scores", 1, 1, 1, EXT_NONE, (long long)HW * HW);
    {
        Op g;
        g.kind = OP_GEMM; g.label = n + ".qk";
        g.a0 = {qk, 0}; g.C0 = C; g.lda0 = 2 * C; g.H = 1; g.W = HW; g.OH = 1; g.OW = HW;
        g.m_per_image = HW; g.z_per_image = 1; g.N = HW; g.K = C;
        g.bten = {qk, C}; g.ldb = 2 * C; g.b_kmajor = 0;
        g.sA = (long long)HW * 2 * C; g.sB = g.sA; g.sC = (long long)HW * HW;
        g.out = {S, 0}; g.ldc = HW; g.alpha = (float)std::pow((double)C, -0.5);
        g.flops_per_image = 2.0 * HW * HW * C;
        ops.push_back(g);
        Op sm;
        sm.kind = OP_SOFTMAX; sm.label = n + ".softmax"; sm.out = {S, 0}; sm.rows_per_image = HW; sm.cols = HW;
        ops.push_back(sm);
    }
    // h = w v (+ b_v); allocated while q, k are live: one fused launch reads them and writes this (see Plan::attn)
    int ho = new_tensor(n + ".av", H, W, C);
    release(qk);
    {
        Op g;
        g.kind = OP_GEMM; g.label = n + ".av";
        g.a0 = {S, 0}; g.C0 = HW; g.lda0 = HW; g.H = 1; g.W = HW; g.OH = 1; g.OW = HW;
        g.m_per_image = HW; g.z_per_image = 1; g.N = C; g.K = HW;
        g.bten = {vT, 0}; g.ldb = HW; g.b_kmajor = 0;
        g.sA = (long long)HW * HW; g.sB = (long long)C * HW; g.sC = (long long)HW * C;
        g.bias_slot = bq; g.bias_off = 2 * C;
        g.out = {ho, 0}; g.ldc = C;
        g.flops_per_image = 2.0 * HW * HW * C;
        ops.push_back(g);
    }
    release(S);
    release(vT);
    // (x + NIN_3(h)) / sqrt(2)
    int wo = new_slot(n + "/NIN_3/W", C, C), bo = new_slot(n + "/NIN_3/b", 1, C);
    add_param(n + "/NIN_3/W", {C, C}, P_DENSE, wo, 0);
    add_param(n + "/NIN_3/b", {C}, P_VEC, bo, 0);
    int out = new_tensor(n + ".out", H, W, C);
    attach_stats(out, HW);
    {
        Op g;
        g.kind = OP_GEMM; g.label = n + ".proj";
        g.a0 = {ho, 0}; g.C0 = C; g.lda0 = C; g.H = H; g.W = W; g.OH = H; g.OW = W;
        g.m_per_image = HW; g.N = C; g.K = C; g.w_slot = wo; g.ldb = C; g.bias_slot = bo;
        g.res = {x, 0}; g.ldres = C; g.out = {out, 0}; g.ldc = C; g.out_scale = scale;
        if (tensors[out].stats_t >= 0) { g.qstats = {tensors[out].stats_t, 0}; g.qs_slices = tensors[out].qs_slices; }
        g.flops_per_image = 2.0 * HW * C * C;
        ops.push_back(g);
    }
    release(ho);
    return out;
}

// ---------------------------------------------------------------- NCSNpp.__call__
int Plan::build(const gd_model_desc &d) {
    desc = d;
    // A/B switches of the lowering, read per plan (a second engine in the process, or a test, may build the other form)
    no_attn_fold = opts().no_attn_fold != 0;
    no_upswap = opts().no_upswap != 0;
    options_at_build = options_json();
    GD_CHECK(d.num_levels >= 1 && d.num_levels <= GD_MAX_LEVELS, "num_levels out of range");
    GD_CHECK(d.nf >= 16 && d.nf % 16 == 0, "nf must be a multiple of 16 (GroupNorm groups of >= 4 channels), got %d", d.nf);
    GD_CHECK(d.in_channels >= 1 && d.image_size >= (1 << (d.num_levels - 1)), "bad in_channels / image_size");
    GD_CHECK(d.image_size % (1 << (d.num_levels - 1)) == 0, "image_size must be divisible by 2^(levels-1)");
    GD_CHECK(!(d.no_fir && d.progressive_input_residual), "fir=False with progressive_input='residual' is not built (no reference "
             "config uses it)");
    GD_CHECK(d.max_batch >= 1, "max_batch must be >= 1");
    const int nf = d.nf, nres = d.num_levels, nrb = d.num_res_blocks;
    CP = (d.in_channels + 3) / 4 * 4;
    auto is_attn = [&](int res) {
        for (int i = 0; i < d.num_attn_resolutions; ++i)
            if (d.attn_resolutions[i] == res) return true;
        return false;
    };

    // pre-walk: out_ch of every ResnetBlock in call order -> column offsets of the fused temb projection
    {
        std::vector<int> outs;
        int c = nf;
        std::vector<int> hs{nf};
        for (int l = 0; l < nres; ++l) {
            for (int b = 0; b < nrb; ++b) { c = nf * d.ch_mult[l]; outs.push_back(c); hs.push_back(c); }
            if (l != nres - 1) { outs.push_back(c); hs.push_back(c); }
        }
        outs.push_back(c); outs.push_back(c);
        for (int l = nres - 1; l >= 0; --l) {
            for (int b = 0; b < nrb + 1; ++b) { hs.pop_back(); c = nf * d.ch_mult[l]; outs.push_back(c); }
            if (l != 0) outs.push_back(c);
        }
        sum_cout_ = 0;
        for (int o : outs) { rb_tproj_off_.push_back(sum_cout_); sum_cout_ += o; }
    }

    t_x_in = new_tensor("x_in", d.image_size, d.image_size, CP, EXT_X_IN);
    t_labels = new_tensor("labels", 1, 1, 1, EXT_LABELS);
    t_eps_out = new_tensor("eps_out", d.image_size, d.image_size, d.in_channels, EXT_EPS_OUT);

    // ---- time embedding (ncsnpp.py:71-91) ----
    if (d.conditional) {
        // Gaussian Fourier features of log(label) (2 nf wide, learned-free W is a checkpoint parameter), or the sinusoidal
        // positional embedding of the label itself (nf wide, fixed frequencies 10000^(-j/(nf/2-1)), layers.py:450-464)
        const int E = d.positional_embedding ? nf / 2 : nf;
        int wf;
        if (d.positional_embedding) {
            GD_CHECK(nf % 2 == 0 && nf >= 4, "positional embedding needs an even nf >= 4");
            wf = new_slot("positional_frequencies", 1, E);
            const double c = std::log(10000.0) / (double)(E - 1);
            for (int j = 0; j < E; ++j) slots[wf].host[j] = (float)std::exp(-(double)(float)((float)j * (float)c));
        } else {
            wf = new_slot("GaussianFourierProjection_0/W", 1, nf);
            add_param("GaussianFourierProjection_0/W", {nf}, P_VEC, wf, 0);
        }
        int t0 = new_tensor("temb.fourier", 1, 1, 2 * E);
        Op f;
        f.kind = OP_FOURIER; f.label = "temb.fourier"; f.a0 = {t_labels, 0}; f.w_slot = wf; f.N = E; f.out = {t0, 0};
        f.act = d.positional_embedding ? 1 : 0;
        f.temb = 1;
        ops.push_back(f);
        auto dense = [&](const std::string &label, const std::string &pname, int in_t, int in_c, int out_c, int act_in) {
            int ws = new_slot(pname + "/kernel", out_c, in_c), bs = new_slot(pname + "/bias", 1, out_c);
            add_param(pname + "/kernel", {in_c, out_c}, P_DENSE, ws, 0);
            add_param(pname + "/bias", {out_c}, P_VEC, bs, 0);
            int o = new_tensor(label, 1, 1, out_c);
            Op g;
            g.kind = OP_GEMM; g.label = label; g.a0 = {in_t, 0}; g.C0 = in_c; g.lda0 = in_c; g.m_per_image = 1;
            g.N = out_c; g.K = in_c; g.w_slot = ws; g.ldb = in_c; g.bias_slot = bs; g.out = {o, 0}; g.ldc = out_c;
            g.in_act = act_in; g.flops_per_image = 2.0 * out_c * in_c;
            g.temb = 1;
            ops.push_back(g);
            return o;
        };
        int t1 = dense("temb.dense0", "Dense_0", t0, 2 * E, 4 * nf, 0);
        release(t0);
        temb_t_ = dense("temb.dense1", "Dense_1", t1, 4 * nf, 4 * nf, 1);
        release(t1);
        // all per-block Dense_0(act(temb)) as one GEMM: (B, 4nf) x (4nf, sum Cout)
        tproj_slot_w_ = new_slot("temb_proj/kernel", sum_cout_, 4 * nf);
        tproj_slot_b_ = new_slot("temb_proj/bias", 1, sum_cout_);
        tproj_t_ = new_tensor("temb.proj", 1, 1, sum_cout_);
        Op g;
        g.kind = OP_GEMM; g.label = "temb.proj"; g.a0 = {temb_t_, 0}; g.C0 = 4 * nf; g.lda0 = 4 * nf; g.m_per_image = 1;
        g.N = sum_cout_; g.K = 4 * nf; g.w_slot = tproj_slot_w_; g.ldb = 4 * nf; g.bias_slot = tproj_slot_b_;
        g.out = {tproj_t_, 0}; g.ldc = sum_cout_; g.in_act = 1; g.flops_per_image = 2.0 * sum_cout_ * 4 * nf;
        g.temb = 1;
        ops.push_back(g);
        release(temb_t_);
    }

    // ---- input (ncsnpp.py:136-146) ----
    int x = t_x_in;
    if (!d.centered) {
        int xc = new_tensor("x_centered", d.image_size, d.image_size, CP);
        Op a;
        a.kind = OP_AFFINE_PAD; a.label = "x_2x_minus_1"; a.a0 = {x, 0}; a.lda0 = CP; a.cin = d.in_channels; a.cpad = CP;
        a.mul = 2.f; a.add = -1.f; a.out = {xc, 0}; a.H = a.W = d.image_size;
        ops.push_back(a);
        x = xc;
    } else {
        retain(x);
    }
    int pyr = -1, pyrC = CP;
    if (d.progressive_input_residual) { pyr = x; retain(pyr); }

    int rb = 0, at = 0, dn = 0;
    int res = d.image_size;
    struct Skip { int t, C; };
    std::vector<Skip> hs;
    // stem conv: K uses the padded channel count; algorithmic FLOPs count the real ones
    int h = conv("stem", "Conv_0", x, CP, -1, 0, res, res, 3, nf, -1, 1.f, -1, 0, -1, true);
    params[param_index["Conv_0/kernel"]].shape = {3, 3, d.in_channels, nf};
    ops.back().flops_per_image = 2.0 * res * res * nf * 9 * d.in_channels;
    release(x);
    int C = nf;
    hs.push_back({h, C});  // hs owns the reference
    // ---- down path (ncsnpp.py:147-173) ----
    for (int l = 0; l < nres; ++l) {
        for (int b = 0; b < nrb; ++b) {
            int oh;
            int o = resblock(rb++, hs.back().t, hs.back().C, -1, 0, res, res, nf * d.ch_mult[l], false, false, oh);
            C = nf * d.ch_mult[l];
            if (is_attn(res)) {
                int o2 = attn(at++, o, res, res, C);
                release(o);
                o = o2;
            }
            hs.push_back({o, C});
        }
        if (l != nres - 1) {
            int oh;
            int o = resblock(rb++, hs.back().t, hs.back().C, -1, 0, res, res, C, false, true, oh);
            res = oh;
            if (d.progressive_input_residual) {
                // Downsample_k/Conv2d_0 = conv_downsample_2d: FIR at full resolution with pads (2, 2), then a 3x3 stride-2 VALID
                // convolution.  With whole 32-channel slabs the two run as written (an HBM pass over the small pyramid tensor + a
                // quarter of the products); the 8-channel first level composes them into one 6x6 stride-2 pad-2 convolution.
                const std::string pn = "Downsample_" + std::to_string(dn++) + "/Conv2d_0";
                const bool two_step = pyrC % 32 == 0 && !d.no_fir;
                int ws = new_slot(pn + "/weight", C, (two_step ? 9LL : 36LL) * pyrC), bs = new_slot(pn + "/bias", 1, C);
                const int real_in = (pyr == t_x_in || tensors[pyr].name == "x_centered") ? d.in_channels : pyrC;
                add_param(pn + "/weight", {3, 3, real_in, C}, two_step ? P_CONV : P_CONV_FIR6, ws, 0, pyrC);
                add_param(pn + "/bias", {C}, P_VEC, bs, 0);
                int src = pyr;
                if (two_step) {
                    src = new_tensor(pn + ".fir", 2 * res + 1, 2 * res + 1, pyrC);
                    Op f;
                    f.kind = OP_FIR_PAD; f.label = pn + ".fir_pad";
                    f.a0 = {pyr, 0}; f.C0 = pyrC; f.H = 2 * res; f.W = 2 * res; f.out = {src, 0};
                    ops.push_back(f);
                }
                int np = new_tensor(pn + ".out", res, res, C);
                attach_stats(np, res * res);
                Op g;
                g.kind = OP_GEMM; g.label = "pyramid_down_" + std::to_string(dn - 1);
                g.a0 = {src, 0}; g.C0 = pyrC; g.lda0 = pyrC; g.OH = res; g.OW = res; g.stride = 2; g.m_per_image = res * res; g.N = C;
                if (two_step) { g.H = 2 * res + 1; g.W = 2 * res + 1; g.KH = g.KW = 3; g.pad = 0; g.K = 9 * pyrC; }
                else { g.H = 2 * res; g.W = 2 * res; g.KH = g.KW = 6; g.pad = 2; g.K = 36 * pyrC; }
                g.w_slot = ws; g.ldb = g.K; g.bias_slot = bs; g.res = {o, 0}; g.ldres = C; g.out = {np, 0}; g.ldc = C;
                g.out_scale = d.skip_rescale ? (float)(1.0 / std::sqrt(2.0)) : 1.f;
                if (tensors[np].stats_t >= 0) { g.qstats = {tensors[np].stats_t, 0}; g.qs_slices = tensors[np].qs_slices; }
                g.flops_per_image = 2.0 * res * res * C * 9 * real_in;
                g.is_conv3x3 = 1;
                ops.push_back(g);
                if (two_step) release(src);
                release(pyr);
                release(o);
                pyr = np; pyrC = C;
                retain(pyr);  // one ref for `pyr`, one for hs below
                o = np;
            }
            hs.push_back({o, C});
        }
    }
    release(pyr);
    // ---- middle (ncsnpp.py:175-178) ----
    h = hs.back().t;
    retain(h);
    {
        int oh;
        int o = resblock(rb++, h, C, -1, 0, res, res, C, false, false, oh);
        release(h);
        int o2 = attn(at++, o, res, res, C);
        release(o);
        h = resblock(rb++, o2, C, -1, 0, res, res, C, false, false, oh);
        release(o2);
    }
    // ---- up path (ncsnpp.py:183-229) ----
    for (int l = nres - 1; l >= 0; --l) {
        for (int b = 0; b < nrb + 1; ++b) {
            Skip s = hs.back();
            hs.pop_back();
            int oh;
            int o = resblock(rb++, h, C, s.t, s.C, res, res, nf * d.ch_mult[l], false, false, oh);
            release(h);
            release(s.t);
            h = o;
            C = nf * d.ch_mult[l];
        }
        if (is_attn(res)) {
            int o = attn(at++, h, res, res, C);
            release(h);
            h = o;
        }
        if (l != 0) {
            int oh;
            int o = resblock(rb++, h, C, -1, 0, res, res, C, true, false, oh);
            release(h);
            h = o;
            res = oh;
        }
    }
    GD_CHECK(hs.empty(), "internal: skip stack not empty");
    // ---- head (ncsnpp.py:236-241) ----
    {
        int hh = gn("head.gn", "GroupNorm_0", h, C, -1, 0, res, res, 1);
        release(h);
        conv("head.conv", "Conv_1", hh, C, -1, 0, res, res, 3, d.in_channels, -1, 1.f, -1, 0, t_eps_out);
        release(hh);
        if (d.scale_by_sigma) {
            Op sgm;
            sgm.kind = OP_SCALE_ROWS; sgm.label = "scale_by_sigma"; sgm.out = {t_eps_out, 0}; sgm.a0 = {t_labels, 0};
            sgm.m_per_image = res * res * d.in_channels;
            ops.push_back(sgm);
        }
    }
    if (tproj_t_ >= 0) release(tproj_t_);
    GD_CHECK((int)rb_tproj_off_.size() == rb, "internal: resblock count mismatch (%d vs %d)", (int)rb_tproj_off_.size(), rb);

    mark_split();
    link_epilogue_gn();
    // The engine computes the time-embedding ops (Op::temb) on ONE row when the whole batch carries one label and lets every image read
    // that row with stride 0 (engine.cpp: fill_gemm).  That is only right while the tensors those ops produce are read as per-image bias
    // rows or by other time-embedding ops, and while every per-image bias row comes from them: rows 1.. of such a tensor are stale.
    {
        std::vector<char> from_temb(tensors.size(), 0);
        for (const Op &o : ops)
            if (o.temb && o.out.t >= 0) from_temb[o.out.t] = 1;
        for (const Op &o : ops) {
            GD_CHECK(o.rowbias.t < 0 || from_temb[o.rowbias.t], "internal: %s reads a per-image bias row that no time-embedding op produces",
                     o.label.c_str());
            if (o.temb) continue;
            for (const TRef *r : {&o.a0, &o.a1, &o.bten, &o.res})
                GD_CHECK(r->t < 0 || !from_temb[r->t], "internal: %s reads a time-embedding tensor as an operand (only its first row is "
                         "computed when the batch shares one label)", o.label.c_str());
        }
    }
    flops_per_image = conv3x3_flops_per_image = 0;
    for (const Op &o : ops) {
        flops_per_image += o.flops_per_image;
        if (o.is_conv3x3) conv3x3_flops_per_image += o.flops_per_image;
    }
    flops_per_image += folded_flops_per_image;  // (the reference multiplies them; the price of an evaluation is the reference's)
    return 0;
}

// Called right after the convolution that reads GroupNorm output `t` has been emitted (it is the last op): if the apply op that
// produced `t` (one source, or a channel concat of two) and the convolution have the halo kernel's static shape (3x3, stride 1, 16x16 or 32x32 images,
// 32-channel slabs, more than 64 outputs), the two ops are linked for the run-time fusion.  The caller keeps the apply op's source
// tensor alive until the convolution has been emitted.
void Plan::link_fused_gn(int t) {
    if (t < 0 || tensors[t].table_t < 0 || ops.empty()) return;
    const int gi = (int)ops.size() - 1;
    Op &g = ops[gi];
    if (g.kind != OP_GEMM || g.a0.t != t || g.a1.t >= 0 || g.KH != 3 || g.KW != 3 || g.stride != 1 || g.pad != 1) return;
    if (g.H != g.W || (g.W != 32 && g.W != 16) || g.C0 % 32 != 0 || g.N <= 64 || g.N % 4 != 0 || g.w_slot < 0) return;
    for (int i = gi - 1; i >= 0; --i) {
        Op &a = ops[i];
        if (a.kind == OP_GN_APPLY && a.out.t == t) {
            if (a.gn_table.t < 0 || a.fuse_peer >= 0 || a.C0 % 32 != 0 || a.C1 % 32 != 0) return;
            for (int j = i + 1; j < gi; ++j)  // the convolution must be the only reader
                if (ops[j].a0.t == t || ops[j].a1.t == t || ops[j].bten.t == t || ops[j].res.t == t) return;
            a.fuse_peer = gi;
            g.fuse_peer = i;
            g.gn_src = a.a0;
            g.gn_src1 = a.a1;
            g.gn_C0 = a.C0;
            g.gn_C1 = a.C1;
            g.gn_table = a.gn_table;
            return;
        }
    }
}

// Pairs every epilogue-normalisation candidate (Plan::gn: `fresh` output) with the GEMM that produces its source (Op::egn_peer).
void Plan::link_epilogue_gn() {
    for (int i = 0; i < (int)ops.size(); ++i) {
        Op &a = ops[i];
        if (a.kind != OP_GN_APPLY || a.C1 != 0 || a.a0.col != 0 || !tensors[a.out.t].fresh) continue;
        const int t = a.a0.t;
        int prod = -1;
        for (int j = i - 1; j >= 0; --j)
            if ((ops[j].kind == OP_GEMM || ops[j].kind == OP_GN_APPLY || ops[j].kind == OP_FIR_UP || ops[j].kind == OP_FIR_DOWN) &&
                ops[j].out.t == t) { prod = j; break; }
        if (prod < 0) continue;
        Op &g = ops[prod];
        if (g.kind != OP_GEMM || g.z_per_image || g.out.col != 0 || g.qstats.t < 0 || g.N != a.C0 || g.ldc != a.C0 ||
            g.m_per_image != a.H * a.W || g.w_slot < 0 || g.egn_peer >= 0)
            continue;
        bool sole = true;
        for (int k = 0; k < (int)ops.size() && sole; ++k) {
            if (k == i) continue;
            const Op &o = ops[k];
            if (o.a0.t == t || o.a1.t == t || o.bten.t == t || o.res.t == t || o.rowbias.t == t || (k != prod && o.out.t == t)) sole = false;
        }
        a.egn_peer = prod;
        g.egn_peer = i;
        g.egn_sole = sole ? 1 : 0;
    }
}

// A GroupNorm-apply output may be stored split (bf16 hi|lo rows) in bf16x3 mode when all of its readers are weight GEMMs the
// split-precision kernel accepts with a pre-split A operand: whole tensor as the single source, 32-channel slabs.
void Plan::mark_split() {
    for (size_t i = 0; i < ops.size(); ++i) {
        Op &ap = ops[i];
        if (ap.kind != OP_GN_APPLY || tensors[ap.out.t].ext != EXT_NONE) continue;
        const int t = ap.out.t;
        const int C = ap.C0 + ap.C1;
        std::vector<size_t> readers;
        bool ok = C % 32 == 0;
        for (size_t j = 0; j < ops.size() && ok; ++j) {
            if (j == i) continue;
            const Op &o = ops[j];
            const bool reads = o.a0.t == t || o.a1.t == t || o.bten.t == t || o.res.t == t || o.rowbias.t == t ||
                               (o.kind != OP_GEMM && o.out.t == t);
            if (!reads) continue;
            if (o.kind == OP_GEMM && o.a0.t == t && o.a0.col == 0 && o.a1.t < 0 && o.C1 == 0 && o.C0 == C && o.lda0 == C &&
                o.bten.t != t && o.res.t != t && o.rowbias.t != t && o.a_slot < 0 && o.w_slot >= 0 && !o.z_per_image &&
                !o.b_kmajor && !o.in_act && o.KH * o.KW <= 32 && o.K % 32 == 0 && o.ldb % 8 == 0 &&
                !(o.rowbias.t >= 0 && o.m_per_image % 8 != 0))
                readers.push_back(j);
            else
                ok = false;
        }
        if (!ok || readers.empty()) continue;
        ap.split_out = 1;
        for (size_t j : readers) ops[j].a_split = 1;
    }
}

// ---------------------------------------------------------------- parameter packing
int Plan::load_param(const std::string &name, const float *data, const int64_t *shape, int ndim) {
    auto it = param_index.find(name);
    GD_CHECK(it != param_index.end(), "unknown parameter '%s'", name.c_str());
    Param &p = params[it->second];
    GD_CHECK((int)p.shape.size() == ndim, "parameter '%s': expected %d dims, got %d", name.c_str(), (int)p.shape.size(), ndim);
    for (int i = 0; i < ndim; ++i)
        GD_CHECK(p.shape[i] == shape[i], "parameter '%s': dim %d is %lld, expected %lld", name.c_str(), i,
                 (long long)shape[i], (long long)p.shape[i]);
    WSlot &s = slots[p.slot];
    switch (p.kind) {
        case P_VEC: {
            for (int64_t i = 0; i < shape[0]; ++i) s.host[(size_t)(p.row_off + i)] = data[i];
            break;
        }
        case P_DENSE: {  // (in, out) -> rows row_off.., cols = in
            const int64_t in = shape[0], out = shape[1];
            GD_CHECK(in == s.cols, "parameter '%s': inner dim %lld != slot cols %lld", name.c_str(), (long long)in, s.cols);
            for (int64_t o = 0; o < out; ++o)
                for (int64_t i = 0; i < in; ++i) s.host[(size_t)((p.row_off + o) * s.cols + i)] = data[i * out + o];
            break;
        }
        case P_CONV: {  // HWIO -> [O][(dy*KW+dx)*Ipad + i]
            const int64_t kh = shape[0], kw = shape[1], I = shape[2], O = shape[3];
            const int64_t ip = p.ipad;
            for (int64_t o = 0; o < O; ++o)
                for (int64_t t = 0; t < kh * kw; ++t)
                    for (int64_t i = 0; i < I; ++i)
                        s.host[(size_t)(o * s.cols + t * ip + i)] = data[(t * I + i) * O + o];
            break;
        }
        case P_CONV_FIR6: {
            // conv_downsample_2d (up_or_down_sampling.py:168-209): upfirdn with k = outer([1,3,3,1])/64, pads (2,2),
            // then 3x3 stride-2 VALID conv.  Both are correlations (the FIR is symmetric), so the composition is a
            // 6x6 stride-2 pad-2 correlation with W6[u][v] = sum_{a+s=u, b+t=v} w[a][b] k[s][t].
            const int64_t I = shape[2], O = shape[3];
            const int64_t ip = p.ipad;
            const double k1[4] = {1.0 / 8, 3.0 / 8, 3.0 / 8, 1.0 / 8};
            std::vector<double> w6((size_t)36);
            for (int64_t o = 0; o < O; ++o)
                for (int64_t i = 0; i < I; ++i) {
                    std::fill(w6.begin(), w6.end(), 0.0);
                    for (int a = 0; a < 3; ++a)
                        for (int b = 0; b < 3; ++b) {
                            const double w = data[((a * 3 + b) * I + i) * O + o];
                            for (int ss = 0; ss < 4; ++ss)
                                for (int tt = 0; tt < 4; ++tt) w6[(a + ss) * 6 + (b + tt)] += w * k1[ss] * k1[tt];
                        }
                    for (int t = 0; t < 36; ++t) s.host[(size_t)(o * s.cols + t * ip + i)] = (float)w6[t];
                }
            break;
        }
    }
    p.loaded = true;
    return 0;
}

// ---------------------------------------------------------------- JSON dump
static void jref(std::ostringstream &o, const char *k, const TRef &r) { o << "\"" << k << "\":[" << r.t << "," << r.col << "],"; }

std::string Plan::json() const {
    std::ostringstream o;
    o.precision(9);
    o << "{\"options\":" << (options_at_build.empty() ? std::string("{}") : options_at_build) << ",\"CP\":" << CP << ",\"arena_per_image\":" << arena_per_image << ",\"flops_per_image\":" << flops_per_image
      << ",\"conv3x3_flops_per_image\":" << conv3x3_flops_per_image << ",\"folded_flops_per_image\":" << folded_flops_per_image << ",\"gn_max_slices\":" << GN_MAX_SLICES
      << ",\"tensors\":[";
    for (size_t i = 0; i < tensors.size(); ++i) {
        const Tensor &t = tensors[i];
        o << (i ? "," : "") << "{\"name\":\"" << t.name << "\",\"H\":" << t.H << ",\"W\":" << t.W << ",\"C\":" << t.C
          << ",\"per_image\":" << t.per_image << ",\"offset\":" << t.offset << ",\"ext\":" << t.ext << "}";
    }
    o << "],\"slots\":[";
    for (size_t i = 0; i < slots.size(); ++i)
        o << (i ? "," : "") << "{\"name\":\"" << slots[i].name << "\",\"rows\":" << slots[i].rows << ",\"cols\":" << slots[i].cols << "}";
    o << "],\"ops\":[";
    for (size_t i = 0; i < ops.size(); ++i) {
        const Op &p = ops[i];
        o << (i ? "," : "") << "{\"kind\":" << p.kind << ",\"label\":\"" << p.label << "\",";
        jref(o, "a0", p.a0); jref(o, "a1", p.a1); jref(o, "bten", p.bten); jref(o, "rowbias", p.rowbias);
        jref(o, "res", p.res); jref(o, "out", p.out); jref(o, "stats", p.stats); jref(o, "stats1", p.stats1); jref(o, "qstats", p.qstats);
        o << "\"split_out\":" << p.split_out << ",\"a_split\":" << p.a_split << ",\"fuse_peer\":" << p.fuse_peer << ",\"egn_peer\":" << p.egn_peer
          << ",\"egn_sole\":" << p.egn_sole << ",\"side_peer\":" << p.side_peer << ",";
        o << "\"C0\":" << p.C0 << ",\"C1\":" << p.C1 << ",\"lda0\":" << p.lda0 << ",\"lda1\":" << p.lda1 << ",\"H\":" << p.H
          << ",\"W\":" << p.W << ",\"OH\":" << p.OH << ",\"OW\":" << p.OW << ",\"KH\":" << p.KH << ",\"KW\":" << p.KW
          << ",\"stride\":" << p.stride << ",\"pad\":" << p.pad << ",\"m_per_image\":" << p.m_per_image
          << ",\"z_per_image\":" << p.z_per_image << ",\"N\":" << p.N << ",\"K\":" << p.K << ",\"a_slot\":" << p.a_slot << ",\"a_off\":" << p.a_off << ",\"w_slot\":" << p.w_slot
          << ",\"ldb\":" << p.ldb << ",\"b_kmajor\":" << p.b_kmajor << ",\"sA\":" << p.sA << ",\"sB\":" << p.sB
          << ",\"sC\":" << p.sC << ",\"sR\":" << p.sR << ",\"sQ\":" << p.sQ << ",\"bias_slot\":" << p.bias_slot << ",\"bias_off\":" << p.bias_off
          << ",\"ld_rowbias\":" << p.ld_rowbias << ",\"ldres\":" << p.ldres << ",\"ldc\":" << p.ldc << ",\"alpha\":" << p.alpha
          << ",\"out_scale\":" << p.out_scale << ",\"in_act\":" << p.in_act << ",\"temb\":" << p.temb << ",\"qs_slices\":" << p.qs_slices << ",\"S0\":" << p.S0 << ",\"S1\":" << p.S1 << ",\"groups\":" << p.groups
          << ",\"act\":" << p.act << ",\"gamma_slot\":" << p.gamma_slot << ",\"beta_slot\":" << p.beta_slot
          << ",\"rows_per_image\":" << p.rows_per_image << ",\"cols\":" << p.cols << ",\"cin\":" << p.cin
          << ",\"cpad\":" << p.cpad << ",\"mul\":" << p.mul << ",\"add\":" << p.add << ",\"flops_per_image\":" << p.flops_per_image
          << ",\"is_conv3x3\":" << p.is_conv3x3 << "}";
    }
    o << "]}";
    return o.str();
}

}  // namespace gd
