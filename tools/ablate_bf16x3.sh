#!/bin/bash
# Ablation timing of igemm_bf16x3_kernel<256,128,4,2> inside one GPU-box call: bash tools/ablate_bf16x3.sh
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p $REPO/gpurun_out
for d in 0 128 129 130 136 144 160 192 176 240 248 128 0; do
  GDDIM_IGEMM_DIAG=$d python $REPO/bench.py --precision bf16x3 --no-cpu-baseline --steps 1 --warmup 1 --nfe 10 2>/dev/null | tail -1 > /tmp/abl.json
  python - <<PY
import json
d=json.loads(open("/tmp/abl.json").read())
for k in d["kernel_breakdown"]:
    if "bf16x3_kernel<256,128,4,2,pre" in k["name"]:
        print("diag=%-4s %s  avg %.1f us  (%d launches)  step %.1f ms" % ("$d", k["name"], 1e3*k["ms"]/k["launches"], k["launches"], d["ms_per_step"]))
PY
done
