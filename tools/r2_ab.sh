#!/bin/bash
# round-2 A/B runs on one box: split-K of the 4x4 level, GroupNorm pass order, by-label breakdown
mkdir -p gpurun_out
B="python bench.py --steps 3 --warmup 1 --no-alt --no-cpu-baseline"
$B > gpurun_out/ab_default.json 2> gpurun_out/ab_default.err
GDDIM_IGEMM_KS=1 $B > gpurun_out/ab_ks1.json 2>/dev/null
GDDIM_IGEMM_KS=2 $B > gpurun_out/ab_ks2.json 2>/dev/null
GDDIM_GN_FORWARD=1 $B > gpurun_out/ab_gnfwd.json 2>/dev/null
GD_PROF_BY_LABEL=1 $B > gpurun_out/ab_bylabel.json 2>/dev/null
for f in default ks1 ks2 gnfwd; do python - <<PY
import json
d=json.loads([l for l in open("gpurun_out/ab_$f.json") if l.startswith("{")][-1])
kb={k["name"]:k for k in d["kernel_breakdown"]}
print("$f", round(d["value"],2), "img/s;", {n:(round(k["ms"],1), round(k["tflops"],1) if k["tflops"] else round(k["gbs"] or 0)) for n,k in kb.items() if any(s in n for s in ("64,64,2,2","gn_apply","128,64,4,2"))})
PY
done
