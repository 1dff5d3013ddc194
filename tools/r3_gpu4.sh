#!/bin/bash
# round 3, GPU call 4: folded attention projections (Plan::attn) -- operator + U-Net + sampler parity tests, then A/B of the bench line
# against GDDIM_NO_ATTN_FOLD=1 (same library, same box)
cd ${GRAFT_REPO_ROOT:-.}
mkdir -p gpurun_out
timeout 1500 python -m pytest tests/test_gpu_ops.py tests/test_gpu_unet.py -x -q -k "attention or unet or net" > gpurun_out/r3_fold_tests.log 2>&1
echo "tests rc=$?" >> gpurun_out/r3_fold_tests.log
tail -8 gpurun_out/r3_fold_tests.log
for rep in 1 2; do
for nf in 1 ""; do
  GDDIM_NO_ATTN_FOLD=$nf timeout 900 python bench.py --steps 3 --warmup 1 --no-alt --no-cpu-baseline 2>/dev/null | tail -1 > gpurun_out/r3_bench_tmp.json
  python - <<PY
import json
d=json.loads(open("gpurun_out/r3_bench_tmp.json").read())
print("rep $rep nofold=${nf:-0}  %.2f img/s" % d["value"])
for k in d["kernel_breakdown"][:8]:
    print("   %-50s %8.1f ms %5d launches %s TF issued_frac %s" % (k["name"], k["ms"], k["launches"], k["tflops"] and round(k["tflops"],1), k.get("issued_frac") and round(k["issued_frac"],3)))
PY
done
done 2>&1 | tee gpurun_out/r3_fold_ab.log
