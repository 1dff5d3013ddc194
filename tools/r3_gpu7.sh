#!/bin/bash
# round 3, GPU call 7: the shortcut convolution of the up-sampling blocks before the FIR (Plan::resblock) -- tests, then A/B against GDDIM_NO_UPSWAP=1
cd ${GRAFT_REPO_ROOT:-.}
mkdir -p gpurun_out
timeout 1500 python -m pytest tests/test_gpu_unet.py tests/test_gpu_parity_abs.py -x -q > gpurun_out/r3_upswap_tests.log 2>&1
echo "tests rc=$?" >> gpurun_out/r3_upswap_tests.log
tail -3 gpurun_out/r3_upswap_tests.log
for rep in 1 2; do
for off in 1 0; do
  GDDIM_NO_UPSWAP=$off timeout 900 python bench.py --steps 3 --warmup 1 --no-alt --no-cpu-baseline 2>/dev/null | tail -1 > gpurun_out/r3_bench_tmp.json
  python - <<PY
import json
d=json.loads(open("gpurun_out/r3_bench_tmp.json").read())
f=sum(k["ms"] for k in d["kernel_breakdown"] if k["name"].startswith("fir"))
g=sum(k["ms"] for k in d["kernel_breakdown"] if k["name"].startswith("igemm"))
print("rep $rep no_upswap=$off  %.2f img/s  fir %.1f ms  igemm* %.1f ms" % (d["value"], f, g))
PY
done
done 2>&1 | tee gpurun_out/r3_upswap_ab.log
