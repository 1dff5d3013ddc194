"""Key figures of a bench.py JSON line: python tools/show_line.py FILE"""
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
r = d.get("roofline", {})
print("fp32   %.2f images/s  median %.1f ms  frac %.3f  dominant launch %.1f us" % (d["value"], d.get("ms_per_step_median", d["ms_per_step"]), r.get("frac", 0), 1e3 * r.get("avg_launch_ms", 0)))
for k in ("bf16x3", "split6"):
    if k in d:
        rr = d[k].get("roofline", {})
        print("%-6s %.2f images/s  %.1f ms  dominant launch %.1f us  frac %.3f" % (k, d[k]["value"], d[k]["ms_per_step"], 1e3 * rr.get("avg_launch_ms", 0), rr.get("frac", 0)))
if "cpu_baseline" in d:
    print("cpu   ", json.dumps(d["cpu_baseline"])[:300])
