// The one place the native library reads its environment (options.h).
#include "options.h"

#include <cstdlib>
#include <cstring>
#include <mutex>
#include <sstream>

#include "../../include/gddim_amd.h"

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

Options g_opts;
std::once_flag g_once;

void parse() {
    Options o;
    if (const char *pm = getenv("GDDIM_PRECISION")) o.precision = strcmp(pm, "bf16x3") == 0 ? 1 : 0;
    for (const Entry &e : kEntries) {
        const char *v = getenv(e.env);
        if (!v) continue;
        // a variable set to nothing (or to a non-number) switches its entry on: `GDDIM_NO_EGN= cmd` and `GDDIM_NO_EGN=1 cmd` are the same
        char *end = nullptr;
        long long x = strtoll(v, &end, 10);
        if (end == v) x = 1;
        if (e.i) o.*(e.i) = (int)x; else o.*(e.ll) = x;
    }
    if (o.lanes < 1) o.lanes = 1;
    if (o.no_wino) o.no_wino2 = 1;     // the switches nest: no Winograd at all > no two-dimensional forms > no hybrid form
    if (o.no_wino2) o.no_wino42 = 1;
    if (o.no_wino2 && !getenv("GDDIM_NOWINO2_SPLITK")) o.no_wino2_splitk = 1;
    g_opts = o;
}

}  // namespace

const Options &opts() {
    std::call_once(g_once, parse);
    return g_opts;
}

void options_reload() {
    std::call_once(g_once, parse);
    parse();
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

extern "C" int gd_options_reload(void) {
    gd::options_reload();
    return 0;
}
