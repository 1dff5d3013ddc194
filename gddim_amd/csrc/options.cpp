// The one place the native library reads its environment (options.h).
#include "options.h"

#include <atomic>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <deque>
#include <mutex>
#include <sstream>
#include <strings.h>

#include "../../include/gddim_amd.h"
#include "common.h"

namespace gd {
namespace {

struct Entry {
    const char *env;
    int Options::*i;
    long long Options::*ll;
    bool numeric;  // a number that must parse, as opposed to a switch (anything set = on)
};

// (name, member, numeric): the ONE table gd_engine_plan_json prints and parse() fills
const Entry kEntries[] = {
    {"GD_LANES", &Options::lanes, nullptr, true},
    {"GD_PROF_BY_LABEL", &Options::prof_key, nullptr, true},
    {"GDDIM_NO_EGN", &Options::no_egn, nullptr, false},
    {"GDDIM_NO_SIDE_GN", &Options::no_side_gn, nullptr, false},
    {"GDDIM_NO_FUSED_ATTN", &Options::no_fused_attn, nullptr, false},
    {"GDDIM_NO_TEMB_HOIST", &Options::no_temb_hoist, nullptr, false},
    {"GDDIM_ATTN_Q", &Options::attn_q, nullptr, true},
    {"GDDIM_NO_ATTN_FOLD", &Options::no_attn_fold, nullptr, false},
    {"GDDIM_NO_UPSWAP", &Options::no_upswap, nullptr, false},
    {"GDDIM_NOWINO", &Options::no_wino, nullptr, false},
    {"GDDIM_NOWINO2", &Options::no_wino2, nullptr, false},
    {"GDDIM_NOWINO42", &Options::no_wino42, nullptr, false},
    {"GDDIM_NOWINO2_SPLITK", &Options::no_wino2_splitk, nullptr, false},
    {"GDDIM_WINO2_KS", &Options::wino2_ks, nullptr, true},
    {"GDDIM_WINO2_MINBLOCKS", &Options::wino2_min_blocks, nullptr, true},
    {"GDDIM_WINO42_MINBLOCKS", &Options::wino42_min_blocks, nullptr, true},
    {"GDDIM_WINO42_GRID", &Options::wino42_grid, nullptr, true},
    {"GDDIM_WINO42_NW", &Options::wino42_nw, nullptr, true},
    {"GDDIM_IGEMM_VARIANT", &Options::igemm_variant, nullptr, true},
    {"GDDIM_IGEMM_BIGMIN", nullptr, &Options::igemm_bigmin, true},
    {"GDDIM_IGEMM_MIDMIN", nullptr, &Options::igemm_midmin, true},
    {"GDDIM_IGEMM_KS", &Options::igemm_ks, nullptr, true},
    {"GDDIM_IGEMM_NOPERSIST", &Options::igemm_nopersist, nullptr, false},
    {"GDDIM_IGEMM_PERS_MAXK", &Options::igemm_pers_maxk, nullptr, true},
    {"GDDIM_IGEMM_SMALL4", &Options::igemm_small4, nullptr, false},
    {"GDDIM_IGEMM_NOFAST", &Options::igemm_nofast, nullptr, false},
    {"GDDIM_IGEMM_DIAG", &Options::igemm_diag, nullptr, true},
    {"GDDIM_X3_NOHALO", &Options::x3_nohalo, nullptr, false},
    {"GDDIM_X3_NOFUSE", &Options::x3_nofuse, nullptr, false},
    {"GDDIM_X3_FUSE_MAXN", &Options::x3_fuse_maxn, nullptr, true},
    {"GDDIM_X3_NOPERSIST", &Options::x3_nopersist, nullptr, false},
    {"GDDIM_X3_NOATTN", &Options::x3_noattn, nullptr, false},
    {"GDDIM_X3_VARIANT", &Options::x3_variant, nullptr, true},
};

// every snapshot ever published stays alive (a few hundred bytes each): a reader that picked one up keeps a valid table whatever
// other threads reload meanwhile
std::deque<Options> g_history;
std::mutex g_mutex;
std::atomic<const Options *> g_current{nullptr};
thread_local const Options *t_scope = nullptr;

const Options *parse() {
    Options o;
    if (const char *pm = getenv("GDDIM_PRECISION")) o.precision = strcmp(pm, "bf16x3") == 0 ? 1 : strcmp(pm, "split6") == 0 ? 2 : 0;
    for (const Entry &e : kEntries) {
        const char *v = getenv(e.env);
        if (!v) continue;
        char *end = nullptr;
        long long x = strtoll(v, &end, 10);
        if (end == v) {
            if (!e.numeric) {
                // a switch set to nothing or to a word is on (`GDDIM_NO_EGN= cmd` = `GDDIM_NO_EGN=1 cmd`) -- except the words that say off
                x = (strcasecmp(v, "off") == 0 || strcasecmp(v, "false") == 0 || strcasecmp(v, "no") == 0) ? 0 : 1;
            } else {
                fprintf(stderr, "gddim_amd: %s='%s' is not a number: ignored\n", e.env, v);
                continue;
            }
        }
        if (e.i) o.*(e.i) = (int)x; else o.*(e.ll) = x;
    }
    if (o.lanes < 1) o.lanes = 1;
    // the switches nest, and the outer one wins: no Winograd at all (GDDIM_NOWINO) forces no two-dimensional forms (GDDIM_NOWINO2), which
    // forces no hybrid form and no channel-split 4x4 launch -- whatever the nested variables say
    if (o.no_wino) o.no_wino2 = 1;
    if (o.no_wino2) o.no_wino42 = o.no_wino2_splitk = 1;
    std::lock_guard<std::mutex> lock(g_mutex);
    g_history.push_back(o);
    const Options *p = &g_history.back();  // deque: growing never moves its elements
    g_current.store(p, std::memory_order_release);
    return p;
}

}  // namespace

const Options *options_snapshot() {
    const Options *p = g_current.load(std::memory_order_acquire);
    return p ? p : parse();
}

const Options &opts() { return t_scope ? *t_scope : *options_snapshot(); }

const Options *options_reload() { return parse(); }

OptScope::OptScope(const Options *o) : prev_(t_scope), set_(o != nullptr) {
    if (set_) t_scope = o;
}

OptScope::~OptScope() {
    if (set_) t_scope = prev_;
}

std::string options_json() {
    const Options &o = opts();
    std::ostringstream s;
    s << "{\"GDDIM_PRECISION\":\"" << (o.precision == 1 ? "bf16x3" : o.precision == 2 ? "split6" : "fp32") << "\"";
    for (const Entry &e : kEntries) s << ",\"" << e.env << "\":" << (e.i ? (long long)(o.*(e.i)) : o.*(e.ll));
    s << "}";
    return s.str();
}

}  // namespace gd

namespace gd {
int device_cus() {
    static std::atomic<int> cached[16] = {};
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 16) return 256;
    int c = cached[dev].load(std::memory_order_relaxed);
    if (c > 0) return c;
    if (hipDeviceGetAttribute(&c, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || c <= 0) c = 256;
    cached[dev].store(c, std::memory_order_relaxed);
    return c;
}
}  // namespace gd

extern "C" int gd_options_reload(void) {
    gd::options_reload();
    return 0;
}
