#!/bin/bash
# In-box A/B of two builds of the library (boxes differ by +-10%, one box repeats to +-0.3%):
#   bash tools/ab_libs.sh gddim_amd/libgddim_prev.so [bench args]
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OTHER=$1; shift
for rep in 1 2; do
  for lib in "$REPO/$OTHER" ""; do
    GDDIM_LIB=$lib python $REPO/bench.py --no-cpu-baseline "$@" 2>/dev/null | tail -1 > /tmp/ab.json
    python - <<PY
import json
d=json.loads(open("/tmp/ab.json").read())
b=d.get("bf16x3") or {}
k=d["kernel_breakdown"][0]
print("%-28s fp32 %.2f img/s  [%s %.1f TF]   bf16x3 %s img/s %s" % ("$lib"[-24:] or "current", d["value"], k["name"], k["tflops"] or 0,
      ("%.2f" % b["value"]) if b else "-", ("[%.1f us]" % (1e3*b["roofline"]["avg_launch_ms"])) if b else ""))
PY
  done
done
