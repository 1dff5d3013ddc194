// Run-time switches of the native library: ONE table, parsed from the environment in one place.
//
// None is needed in normal use; they exist for in-box A/B measurements and for tests that force a kernel a small problem would not
// select.  The environment is parsed into an IMMUTABLE snapshot when the library first needs it, again at every gd_engine_create and on
// gd_options_reload(); a new snapshot is published through an atomic pointer, old ones stay valid for whoever still reads them.  An
// engine keeps the snapshot it was created under for its whole life (gd_engine::opts) and installs it on the calling thread for the
// duration of each of its calls (OptScope), so every selector and launcher underneath -- they all read opts() -- sees the table the
// engine's plan, weight copies, cached fusion decisions and captured graphs were made with: engines built under different settings
// coexist in one process, and a reload never changes a live engine.  Stand-alone operator entry points (gd_conv2d, ...)
// read the process-wide snapshot.  gd_engine_plan_json prints the engine's own table ("options").
#pragma once
#include <string>

namespace gd {

struct Options {
    // ---- engine ----
    int precision = 0;          // GDDIM_PRECISION=bf16x3 -> 1: engines start in split-precision mode; =split6 -> 2 (gd_engine_set_precision)
    int lanes = 1;              // GD_LANES: concurrent streams a sampler call is split over
    int prof_key = 0;           // GD_PROF_BY_LABEL: profile entries keyed by 0 kernel name, 1 plan-label class, 2 "label class|kernel"
    int no_egn = 0;             // GDDIM_NO_EGN: no GroupNorm in the producer's epilogue
    int no_side_gn = 0;         // GDDIM_NO_SIDE_GN: no GroupNorm as a side output of the shortcut convolution's loader
    int no_fused_attn = 0;      // GDDIM_NO_FUSED_ATTN: attention core as three launches
    int no_temb_hoist = 0;      // GDDIM_NO_TEMB_HOIST: the time-embedding ops run inside every step instead of once per schedule knot
    int attn_q = 0;             // GDDIM_ATTN_Q: 0 the launcher's choice, 32 / 64 force the fused attention kernel's query block (A/B)
    // ---- lowering (Plan::build) ----
    int no_attn_fold = 0;       // GDDIM_NO_ATTN_FOLD: attention blocks keep their four projections
    int no_upswap = 0;          // GDDIM_NO_UPSWAP: up-sampling blocks multiply their 1x1 shortcut after the FIR
    // ---- fp32 3x3 convolutions ----
    int no_wino = 0;            // GDDIM_NOWINO: every 3x3 convolution on the direct implicit-GEMM kernel
    int no_wino2 = 0;           // GDDIM_NOWINO2: no F(2x2,3x3) kernel (and no hybrid kernel): the same as GDDIM_NOWINO since round 4
    int no_wino42 = 0;          // GDDIM_NOWINO42: no hybrid F(2,3)y x F(4,3)x kernel
    int no_wino2_splitk = 0;    // GDDIM_NOWINO2_SPLITK: 4x4 level back on the direct split-K tile
    int wino2_ks = 0;           // GDDIM_WINO2_KS: forced channel-split factor of the 4x4 level's launch (2, 4, 8)
    int wino2_min_blocks = 256; // GDDIM_WINO2_MINBLOCKS: workgroup tiles a launch must have for the F(2x2,3x3) kernel
    int wino42_min_blocks = 256;  // GDDIM_WINO42_MINBLOCKS: wide tiles a Cin = 32 launch must have for the hybrid kernel's two-chunk form (other shapes: rounds x cost)
    int wino42_grid = 0;        // GDDIM_WINO42_GRID: persistent workgroups per launch (tests: several tiles per workgroup)
    int wino42_nw = 0;          // GDDIM_WINO42_NW: 0 the launcher's choice, 1 / 2 force the 32- / 64-output workgroup tile
    // ---- fp32 implicit GEMM ----
    int igemm_variant = 0;      // GDDIM_IGEMM_VARIANT: 0 128x128 / 8 waves, 1 128x128 / 4 waves, 2 256x128
    long long igemm_bigmin = 384, igemm_midmin = 384;  // GDDIM_IGEMM_BIGMIN / _MIDMIN: workgroups needed for 128x128 / 128x64 tiles
    int igemm_ks = 0;           // GDDIM_IGEMM_KS: forced intra-workgroup split-K factor of the 64x64 tile
    int igemm_nopersist = 0;    // GDDIM_IGEMM_NOPERSIST: persistent short-K kernel off
    int igemm_pers_maxk = 1024; // GDDIM_IGEMM_PERS_MAXK
    int igemm_small4 = 0;       // GDDIM_IGEMM_SMALL4: 4-wave 128x64 tiles
    int igemm_nofast = 0;       // GDDIM_IGEMM_NOFAST: every GEMM on the generic kernel
    int igemm_diag = 0;         // GDDIM_IGEMM_DIAG: timing ablations (results wrong by construction)
    // ---- bf16x3 mode ----
    int x3_nohalo = 0, x3_nofuse = 0, x3_fuse_maxn = 1 << 30, x3_nopersist = 0, x3_noattn = 0, x3_variant = 0;  // GDDIM_X3_*
};

const Options &opts();               // the table in force on this thread: the running engine call's snapshot, else the process-wide one
const Options *options_snapshot();   // the process-wide snapshot (immutable, never freed)
const Options *options_reload();     // parse the environment again, publish a new snapshot and return THAT snapshot
std::string options_json();          // {"GDDIM_...": value, ...}: every entry of opts(), in the order above

// Installs an engine's snapshot as this thread's opts() for the lifetime of the object (nests; null = leave as is).
struct OptScope {
    explicit OptScope(const Options *o);
    ~OptScope();
    OptScope(const OptScope &) = delete;
    OptScope &operator=(const OptScope &) = delete;

private:
    const Options *prev_;
    bool set_;
};

}  // namespace gd
