#!/bin/bash
# In-box A/B of the bf16x3 mode with / without a kernel feature (edit the variable: GDDIM_X3_NOHALO, GDDIM_X3_NOFUSE, GDDIM_X3_FUSE_MAXN, GDDIM_X3_NOATTN): bash tools/variants_x3.sh
for v in 1 0 1 0; do
  GDDIM_X3_NOATTN=$v python bench.py --precision bf16x3 --no-cpu-baseline --steps 2 --warmup 1 2>/dev/null | tail -1 > /tmp/v.json
  python - <<PY
import json
d=json.loads(open("/tmp/v.json").read())
print("noattn=$v: %.2f img/s" % d["value"], [(k["name"][-30:], round(k["ms"]), round(k["tflops"] or 0,1), k["launches"]) for k in d["kernel_breakdown"][:6]])
PY
done
