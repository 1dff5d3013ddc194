#!/bin/bash
# round-2 A/B: persistent short-K GEMM; whole-loop graph at small batch
mkdir -p gpurun_out
B="python bench.py --steps 3 --warmup 1 --no-alt --no-cpu-baseline"
$B > gpurun_out/ab2_default.json 2> gpurun_out/ab2_default.err
GDDIM_IGEMM_NOPERSIST=1 $B > gpurun_out/ab2_nopers.json 2>/dev/null
for f in default nopers; do python - <<PY
import json
d=json.loads([l for l in open("gpurun_out/ab2_$f.json") if l.startswith("{")][-1])
print("$f", round(d["value"],2), "img/s;", [(k["name"], round(k["ms"],1), round(k["tflops"],1)) for k in d["kernel_breakdown"] if k["tflops"] and "igemm" in k["name"]][:6])
PY
done
# small batch (BASELINE config 1 shape: B=16, NFE=10): eager vs per-step graph vs whole-loop graph
for g in none step loop; do
python bench.py --batch 16 --nfe 10 --steps 20 --warmup 3 --graph $g --no-alt --no-cpu-baseline --no-profile 2>/dev/null | tail -1 > gpurun_out/small_b16_$g.json
python - <<PY
import json
d=json.loads(open("gpurun_out/small_b16_$g.json").read())
print("B=16 NFE=10 graph=$g", round(d["value"],1), "img/s", round(d["ms_per_step"],2), "ms/call")
PY
done
