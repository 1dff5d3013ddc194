#!/usr/bin/env python
"""Copy the summaries tools/collect_profiles.sh left under gpurun_out/prof into profiles/ (tracked) and derive the per-launch HBM
traffic table bench.py reads: python tools/update_profiles.py [round tag, default r01]"""
import json
import os
import shutil
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
DST = os.path.join(REPO, "profiles")
tag = sys.argv[1] if len(sys.argv) > 1 else "r01"
SRC = os.path.join(REPO, "gpurun_out", {"r01": "prof", "r02": "prof2"}.get(tag, "prof" + tag[1:].lstrip("0")))

if tag != "r01":
    # tools/collect_profiles_r2.sh layout: every small file of gpurun_out/prof2 is evidence; logs stay behind
    import glob
    for f in sorted(glob.glob(os.path.join(SRC, "*"))):
        base = os.path.basename(f)
        if os.path.isfile(f) and (base.endswith(".json") or base.endswith(".csv") or base in ("parity_abs_tests.log", "split6_layers.log")) and os.path.getsize(f) > 0:
            shutil.copy(f, os.path.join(DST, "%s_%s" % (tag, base)))
    os.replace(os.path.join(DST, "%s_pmc_FETCH_SIZE_summary.json" % tag), os.path.join(DST, "%s_pmc_fetch_summary.json" % tag))
    os.replace(os.path.join(DST, "%s_pmc_WRITE_SIZE_summary.json" % tag), os.path.join(DST, "%s_pmc_write_summary.json" % tag))
    for a, b in (("pmc_FETCH_SIZE_summary.json", "pmc_fetch_summary.json"), ("pmc_WRITE_SIZE_summary.json", "pmc_write_summary.json"),
                 ("bench_trace_summary.json", "trace_summary.json")):
        if os.path.exists(os.path.join(SRC, a)):
            shutil.copy(os.path.join(SRC, a), os.path.join(SRC, b))

for name in ("bench_kernel_stats.csv", "bench_bf16x3_kernel_stats.csv", "trace_summary.json", "pmc_fetch_summary.json",
             "pmc_write_summary.json", "pmc_sq_summary.json", "trace_bench_line.json", "pmc_fetch_x3_summary.json",
             "pmc_write_x3_summary.json"):
    p = os.path.join(SRC, name)
    if os.path.exists(p):
        shutil.copy(p, os.path.join(DST, "%s_%s" % (tag, name)))
line = os.path.join(REPO, "gpurun_out", "bench_line.json")
if tag == "r01" and os.path.exists(line):
    shutil.copy(line, os.path.join(DST, "%s_bench_line.json" % tag))

fetch = json.load(open(os.path.join(SRC, "pmc_fetch_summary.json")))["kernels"]
write = json.load(open(os.path.join(SRC, "pmc_write_summary.json")))["kernels"]
trace = json.load(open(os.path.join(SRC, "trace_summary.json")))["kernels"]
out = {"note": "per-launch HBM traffic = (2*FETCH_SIZE + WRITE_SIZE)*1024 B, separate --pmc passes of `bench.py --nfe 3 --steps 1 "
               "--warmup 0 --no-graph` (same B=256 shapes as the headline run); see profiles/README.md", "kernels": {}}
for k, v in fetch.items():
    fs = v.get("counters_mean_per_dispatch", {}).get("FETCH_SIZE")
    ws = write.get(k, {}).get("counters_mean_per_dispatch", {}).get("WRITE_SIZE")
    if fs is None or ws is None:
        continue
    out["kernels"][k] = {"fetch_size_kib": fs, "write_size_kib": ws, "hbm_bytes_per_launch": (2 * fs + ws) * 1024,
                         "avg_us_in_trace": trace.get(k, {}).get("avg_us")}
# kernels that only run in the split-precision mode
fx, wx = os.path.join(SRC, "pmc_fetch_x3_summary.json"), os.path.join(SRC, "pmc_write_x3_summary.json")
if os.path.exists(fx) and os.path.exists(wx):
    f3, w3 = json.load(open(fx))["kernels"], json.load(open(wx))["kernels"]
    for k, v in f3.items():
        fs = v.get("counters_mean_per_dispatch", {}).get("FETCH_SIZE")
        ws = w3.get(k, {}).get("counters_mean_per_dispatch", {}).get("WRITE_SIZE")
        if fs is None or ws is None or k in out["kernels"]:
            continue
        out["kernels"][k] = {"fetch_size_kib": fs, "write_size_kib": ws, "hbm_bytes_per_launch": (2 * fs + ws) * 1024,
                             "avg_us_in_trace": v.get("avg_us"), "mode": "bf16x3"}
json.dump(out, open(os.path.join(DST, "%s_hbm_traffic.json" % tag), "w"), indent=1)
print("updated", DST)
