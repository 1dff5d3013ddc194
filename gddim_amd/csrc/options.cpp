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
};

// (name, member): the table gd_engine_plan_json prints and parse() fills
const Entry kEntries[] = {
    {"GD_LANES", &Options::lanes, nullptr},
    {"GD_PROF_BY_LABEL", &Options::prof_key, nullptr},
    {"GDDIM_NO_EGN", &Options::no_egn, nullptr},
    {"GDDIM_NO_SIDE_GN", &Options::no_side_gn, nullptr},
    {"GDDIM_NO_FUSED_ATTN", &Options::no_fused_attn, nullptr},
    {"GDDIM_ATTN_Q", &Options::attn_q, nullptr},
    {"GDDIM_NO_ATTN_FOLD", &Options::no_attn_fold, nullptr},
    {"GDDIM_NO_UPSWAP", &Options::no_upswap, nullptr},
    {"GDDIM_NOWINO", &Options::no_wino, nullptr},
    {"GDDIM_NOWINO2", &Options::no_wino2, nullptr},
    {"GDDIM_NOWINO42", &Options::no_wino42, nullptr},
    {"GDDIM_NOWINO2_SPLITK", &Options::no_wino2_splitk, nullptr},
    {"GDDIM_WINO2_KS", &Options::wino2_ks, nullptr},
    {"GDDIM_WINO2_MINBLOCKS", &Options::wino2_min_blocks, nullptr},
    {"GDDIM_WINO42_MINBLOCKS", &Options::wino42_min_blocks, nullptr},
    {"GDDIM_WINO42_GRID", &Options::wino42_grid, nullptr},
    {"GDDIM_WINO42_NW", &Options::wino42_nw, nullptr},
    {"GDDIM_IGEMM_VARIANT", &Options::igemm_variant, nullptr},
    {"GDDIM_IGEMM_BIGMIN", nullptr, &Options::igemm_bigmin},
    {"GDDIM_IGEMM_MIDMIN", nullptr, &Options::igemm_midmin},
    {"GDDIM_IGEMM_KS", &Options::igemm_ks, nullptr},
    {"GDDIM_IGEMM_NOPERSIST", &Options::igemm_nopersist, nullptr},
    {"GDDIM_IGEMM_PERS_MAXK", &Options::igemm_pers_maxk, nullptr},
    {"GDDIM_IGEMM_SMALL4", &Options::igemm_small4, nullptr},
    {"GDDIM_IGEMM_NOFAST", &Options::igemm_nofast, nullptr},
    {"GDDIM_IGEMM_DIAG", &Options::igemm_diag, nullptr},
    {"GDDIM_X3_NOHALO", &Options::x3_nohalo, nullptr},
    {"GDDIM_X3_NOFUSE", &Options::x3_nofuse, nullptr},
    {"GDDIM_X3_FUSE_MAXN", &Options::x3_fuse_maxn, nullptr},
    {"GDDIM_X3_NOPERSIST", &Options::x3_nopersist, nullptr},
    {"GDDIM_X3_NOATTN", &Options::x3_noattn, nullptr},
    {"GDDIM_X3_VARIANT", &Options::x3_variant, nullptr},
};

// every snapshot ever published stays alive (a few hundred bytes each): a reader that picked one up keeps a valid table whatever
// other threads reload meanwhile
std::deque<Options> g_history;
std::mutex g_mutex;
std::atomic<const Options *> g_current{nullptr};
thread_local const Options *t_scope = nullptr;

bool is_boolean(const Entry &e) {  // switches (anything set = on) as opposed to numbers (must parse)
    static const char *const numeric[] = {"GD_LANES", "GD_PROF_BY_LABEL", "GDDIM_ATTN_Q", "GDDIM_WINO2_KS", "GDDIM_WINO2_MINBLOCKS",
                                          "GDDIM_WINO42_MINBLOCKS", "GDDIM_WINO42_GRID", "GDDIM_WINO42_NW", "GDDIM_IGEMM_VARIANT",
                                          "GDDIM_IGEMM_BIGMIN", "GDDIM_IGEMM_MIDMIN", "GDDIM_IGEMM_KS", "GDDIM_IGEMM_PERS_MAXK",
                                          "GDDIM_IGEMM_DIAG", "GDDIM_X3_FUSE_MAXN", "GDDIM_X3_VARIANT"};
    for (const char *n : numeric)
        if (strcmp(n, e.env) == 0) return false;
    return true;
}

const Options *parse() {
    Options o;
    if (const char *pm = getenv("GDDIM_PRECISION")) o.precision = strcmp(pm, "bf16x3") == 0 ? 1 : 0;
    for (const Entry &e : kEntries) {
        const char *v = getenv(e.env);
        if (!v) continue;
        char *end = nullptr;
        long long x = strtoll(v, &end, 10);
        if (end == v) {
            if (is_boolean(e)) {
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
    // the switches nest: no Winograd at all > no two-dimensional forms > no hybrid form; a nested variable that is SET keeps its own value
    if (o.no_wino && !getenv("GDDIM_NOWINO2")) o.no_wino2 = 1;
    if (o.no_wino2 && !getenv("GDDIM_NOWINO42")) o.no_wino42 = 1;
    if (o.no_wino2 && !getenv("GDDIM_NOWINO2_SPLITK")) o.no_wino2_splitk = 1;
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

void options_reload() { parse(); }

OptScope::OptScope(const Options *o) : prev_(t_scope), set_(o != nullptr) {
    if (set_) t_scope = o;
}

OptScope::~OptScope() {
    if (set_) t_scope = prev_;
}

std::string options_json() {
    const Options &o = opts();
    std::ostringstream s;
    s << "{\"GDDIM_PRECISION\":\"" << (o.precision ? "bf16x3" : "fp32") << "\"";
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
