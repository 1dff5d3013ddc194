#!/bin/bash
# Compile-time ablation timing of igemm_bf16x3_kernel<256,128,4,2,pre> inside one GPU-box call.  The variants are separate
# builds of kernels/igemm_bf16x3.hip with -DX3_ABL=<mask> (1 no epilogue, 2 no global loads, 4 no LDS stores, 8 no fragment
# reads, 16 no MFMA) linked into gddim_amd/libgddim_abl<mask>.so; results are wrong by construction, only the time matters.
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
for d in 0 1 2 6 14 30 16 0; do
  lib=""; [ $d != 0 ] && lib=$REPO/gddim_amd/libgddim_abl$d.so
  GDDIM_LIB=$lib python $REPO/bench.py --precision bf16x3 --no-cpu-baseline --steps 1 --warmup 1 --nfe 10 2>/dev/null | tail -1 > /tmp/abl.json
  python - <<PY
import json
d=json.loads(open("/tmp/abl.json").read())
for k in d["kernel_breakdown"]:
    if "bf16x3_kernel<256,128,4,2,pre" in k["name"]:
        print("abl=%-3s avg %.1f us  (%d launches)  step %.1f ms" % ("$d", 1e3*k["ms"]/k["launches"], k["launches"], d["ms_per_step"]))
PY
done
