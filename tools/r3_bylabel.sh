#!/bin/bash
# per-plan-label kernel times of one sampler call (GD_PROF_BY_LABEL=1): where the non-convolution time sits
cd ${GRAFT_REPO_ROOT:-.}
mkdir -p gpurun_out
GD_PROF_BY_LABEL=1 timeout 900 python bench.py --steps 1 --warmup 1 --no-alt --no-cpu-baseline $BENCH_ARGS 2>/dev/null | tail -1 > gpurun_out/r3_bylabel.json
python - <<PY
import json
d=json.loads(open("gpurun_out/r3_bylabel.json").read())
print("%.2f img/s" % d["value"])
tot=sum(k["ms"] for k in d["kernel_breakdown"])
for k in d["kernel_breakdown"]:
    print("%-50s %8.2f ms %5.2f%% %5d launches  %7.1f us/launch  %s TF  %s GB/s" % (k["name"], k["ms"], 100*k["ms"]/tot, k["launches"], 1e3*k["ms"]/max(1,k["launches"]), k["tflops"] and round(k["tflops"],1), k.get("gbs") and round(k["gbs"])))
PY
