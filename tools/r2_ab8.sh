#!/bin/bash
# epilogue GroupNorm at the 16x16 level (two-dimensional Winograd kernel): production-batch parity tests, then the sampler line with / without
mkdir -p gpurun_out
python -m pytest tests/test_gpu_bf16x3.py tests/test_gpu_parity_abs.py tests/test_gpu_unet.py -x -q -m gpu 2>&1 | tail -4
B="python bench.py --steps 2 --warmup 1 --no-alt --no-cpu-baseline"
$B > gpurun_out/ab8_default.json 2>/dev/null
GDDIM_NO_EGN=1 $B > gpurun_out/ab8_noegn.json 2>/dev/null
for f in default noegn; do python - <<PY
import json
d=json.loads([l for l in open("gpurun_out/ab8_$f.json") if l.startswith("{")][-1])
kb={k["name"]:k for k in d["kernel_breakdown"]}
print("$f", round(d["value"],2), "img/s; gn_apply", round(kb["gn_apply"]["ms"],1), kb["gn_apply"]["launches"], "; wino2d", round(kb["conv3x3_wino2d_f32_kernel"]["ms"],1), "total", round(sum(k["ms"] for k in d["kernel_breakdown"]),1))
PY
done
