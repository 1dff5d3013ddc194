#!/bin/bash
# round 3, GPU call 5: GroupNorm of the 32x32 level in the hybrid Winograd kernel's epilogue (conv3x3_wino42_gn32_f32_kernel) -- tests, then
# A/B of the bench line with GDDIM_EGN32=0 / 1 (same library, same box)
cd ${GRAFT_REPO_ROOT:-.}
mkdir -p gpurun_out
timeout 1500 python -m pytest tests/test_gpu_unet.py -x -q -s -k "epilogue_groupnorm or batch_sizes or accr" > gpurun_out/r3_egn32_tests.log 2>&1
echo "tests rc=$?" >> gpurun_out/r3_egn32_tests.log
tail -12 gpurun_out/r3_egn32_tests.log
for rep in 1 2; do
for on in 0 1; do
  GDDIM_EGN32=$on timeout 900 python bench.py --steps 3 --warmup 1 --no-alt --no-cpu-baseline 2>/dev/null | tail -1 > gpurun_out/r3_bench_tmp.json
  python - <<PY
import json
d=json.loads(open("gpurun_out/r3_bench_tmp.json").read())
print("rep $rep egn32=$on  %.2f img/s" % d["value"])
for k in d["kernel_breakdown"][:7]:
    print("   %-50s %8.1f ms %5d launches %s TF issued_frac %s" % (k["name"], k["ms"], k["launches"], k["tflops"] and round(k["tflops"],1), k.get("issued_frac") and round(k["issued_frac"],3)))
PY
done
done 2>&1 | tee gpurun_out/r3_egn32_ab.log
