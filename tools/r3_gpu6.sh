#!/bin/bash
# round 3, GPU call 6: the GroupNorm affine pair travels with the chunk in the persistent GEMM's side-GroupNorm loader -- tests, then A/B
# against the previous build (gddim_amd/libgddim_base.so), same box
cd ${GRAFT_REPO_ROOT:-.}
mkdir -p gpurun_out
timeout 1500 python -m pytest tests/test_gpu_unet.py tests/test_gpu_parity_abs.py -x -q > gpurun_out/r3_sidegn_tests.log 2>&1
echo "tests rc=$?" >> gpurun_out/r3_sidegn_tests.log
tail -4 gpurun_out/r3_sidegn_tests.log
for rep in 1 2; do
for lib in gddim_amd/libgddim_base.so ""; do
  GDDIM_LIB=$lib timeout 900 python bench.py --steps 3 --warmup 1 --no-alt --no-cpu-baseline 2>/dev/null | tail -1 > gpurun_out/r3_bench_tmp.json
  python - <<PY
import json
d=json.loads(open("gpurun_out/r3_bench_tmp.json").read())
print("rep $rep lib=${lib:-current}  %.2f img/s" % d["value"])
for k in d["kernel_breakdown"][:5]:
    print("   %-50s %8.1f ms %5d launches %s TF issued_frac %s" % (k["name"], k["ms"], k["launches"], k["tflops"] and round(k["tflops"],1), k.get("issued_frac") and round(k["issued_frac"],3)))
PY
done
done 2>&1 | tee gpurun_out/r3_sidegn_ab.log
