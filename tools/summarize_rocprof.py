#!/usr/bin/env python
"""Summarise rocprofv3 CSV output (kernel trace and/or PMC counter collection) per kernel name.

  python tools/summarize_rocprof.py <rocprof output dir> [--out profiles/xyz.json]
Kernel-trace CSVs give launches / total / average duration per kernel; counter-collection CSVs give the mean
counter value per dispatch.  Template arguments are kept, argument lists are stripped."""
import csv
import glob
import json
import os
import re
import sys
from collections import defaultdict


def short(name):
    name = re.sub(r"\(.*$", "", name)          # drop the argument list
    name = name.replace("void ", "").replace("gd::", "")
    return name.strip()


def col(row, *cands):
    low = {k.lower(): k for k in row}
    for c in cands:
        if c.lower() in low:
            return row[low[c.lower()]]
    return None


def main():
    root = sys.argv[1]
    out = sys.argv[sys.argv.index("--out") + 1] if "--out" in sys.argv else None
    trace = defaultdict(lambda: [0, 0.0])
    ctr = defaultdict(lambda: defaultdict(lambda: [0, 0.0]))
    for f in glob.glob(os.path.join(root, "**", "*.csv"), recursive=True):
        base = os.path.basename(f)
        with open(f, newline="") as fh:
            rd = csv.DictReader(fh)
            if "kernel_trace" in base:
                for r in rd:
                    n = short(col(r, "Kernel_Name") or "?")
                    t0, t1 = col(r, "Start_Timestamp"), col(r, "End_Timestamp")
                    if t0 is None or t1 is None:
                        continue
                    trace[n][0] += 1
                    trace[n][1] += (int(t1) - int(t0)) * 1e-6
            elif "counter_collection" in base:
                for r in rd:
                    n = short(col(r, "Kernel_Name") or "?")
                    c, v = col(r, "Counter_Name"), col(r, "Counter_Value")
                    if c is None or v is None:
                        continue
                    ctr[n][c][0] += 1
                    ctr[n][c][1] += float(v)
    res = {"kernels": {}}
    tot = sum(v[1] for v in trace.values())
    for n, (cnt, ms) in sorted(trace.items(), key=lambda kv: -kv[1][1]):
        res["kernels"][n] = {"launches": cnt, "total_ms": round(ms, 3), "avg_us": round(1e3 * ms / cnt, 3),
                             "share": round(ms / tot, 4) if tot else None}
    for n, cs in ctr.items():
        d = res["kernels"].setdefault(n, {})
        d["counters_mean_per_dispatch"] = {c: (v[1] / v[0]) for c, v in cs.items()}
        d["counter_dispatches"] = max(v[0] for v in cs.values())
    txt = json.dumps(res, indent=1)
    if out:
        with open(out, "w") as fh:
            fh.write(txt + "\n")
    print(txt[:6000])


if __name__ == "__main__":
    main()
