#!/bin/bash
# in-box A/B of bench.py between library builds: bash tools/r3_ab_bench.sh gddim_amd/libgddim_<tag>.so [more libs]  ("" = the working tree's)
cd ${GRAFT_REPO_ROOT:-.}
for rep in 1 2; do
for lib in "" "$@"; do
  GDDIM_LIB=$lib timeout 900 python bench.py --steps 3 --warmup 1 --no-alt --no-cpu-baseline 2>/dev/null | tail -1 > gpurun_out/r3_bench_tmp.json
  python - <<PY
import json
d=json.loads(open("gpurun_out/r3_bench_tmp.json").read())
k=d["kernel_breakdown"][0]
print("rep $rep lib=${lib:-current}  %.2f img/s   %s %.1f ms %.1f TF issued %.3f" % (d["value"], k["name"], k["ms"], k["tflops"], k["issued_frac"]))
PY
done
done
