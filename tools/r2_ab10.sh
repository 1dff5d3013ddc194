#!/bin/bash
# tile choice for the 8x8-level 1x1 GEMMs (M = 16 384, N = 256, K = 256 / 512): 128x64 (default) vs 64x64 (+ split-K) vs 128x128
mkdir -p gpurun_out
B="python bench.py --steps 2 --warmup 1 --no-alt --no-cpu-baseline"
for v in default "GDDIM_IGEMM_MIDMIN=1000000" "GDDIM_IGEMM_BIGMIN=256"; do
  if [ "$v" = default ]; then $B 2>/dev/null > gpurun_out/ab10.json; else env $v $B 2>/dev/null > gpurun_out/ab10.json; fi
  python - "$v" <<PY
import json,sys
d=json.loads([l for l in open("gpurun_out/ab10.json") if l.startswith("{")][-1])
print(sys.argv[1], round(d["value"],2), "img/s;", "; ".join("%s %.1f ms/%d @%.0f TF" % (k["name"].replace("igemm_f32_","").replace("_kernel",""), k["ms"], k["launches"], k["tflops"] or 0) for k in d["kernel_breakdown"] if "igemm" in k["name"] and k["ms"] > 5), "| total", round(sum(k["ms"] for k in d["kernel_breakdown"]),1))
PY
done
