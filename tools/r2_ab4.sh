#!/bin/bash
# epilogue GroupNorm A/B + full GPU test suite timing
mkdir -p gpurun_out
(time python -m pytest tests -x -q -m gpu) > gpurun_out/r2_gpu_suite.log 2>&1; tail -4 gpurun_out/r2_gpu_suite.log
B="python bench.py --steps 3 --warmup 1 --no-alt --no-cpu-baseline"
$B > gpurun_out/ab4_default.json 2>/dev/null
GDDIM_NO_EGN=1 $B > gpurun_out/ab4_noegn.json 2>/dev/null
for f in default noegn; do python - <<PY
import json
d=json.loads([l for l in open("gpurun_out/ab4_$f.json") if l.startswith("{")][-1])
kb={k["name"]:k for k in d["kernel_breakdown"]}
print("$f", round(d["value"],2), "img/s; gn_apply", round(kb["gn_apply"]["ms"],1), kb["gn_apply"]["launches"], "; wino64", round(kb.get("conv3x3_wino_f32_kernel<64,64>",{}).get("ms",0),1), "; ks4", round(kb.get("igemm_f32_fast_kernel<64,64,2,2,false,false,4>",{}).get("ms",0),1), "total", round(sum(k["ms"] for k in d["kernel_breakdown"]),1))
PY
done
