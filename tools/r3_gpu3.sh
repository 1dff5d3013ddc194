#!/bin/bash
# round 3, GPU call 3: persistent (cross-tile pipelined) hybrid Winograd kernel vs the one-workgroup-per-tile build (same box)
cd ${GRAFT_REPO_ROOT:-.}
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_ops.py -x -q -k "winograd42" > gpurun_out/r3_w42_tests.log 2>&1
echo "tests rc=$?" >> gpurun_out/r3_w42_tests.log
tail -6 gpurun_out/r3_w42_tests.log
for lib in gddim_amd/libgddim_w42np.so ""; do
  echo "== lib=${lib:-current}"
  GDDIM_LIB=$lib timeout 600 python tools/bench_wino2d.py --variants wino42 --iters 10 2>&1 | sed -e 's/wino42: vs direct .* |//'
done
for lib in gddim_amd/libgddim_w42np.so ""; do
  GDDIM_LIB=$lib timeout 900 python bench.py --steps 3 --warmup 1 --no-alt --no-cpu-baseline 2>/dev/null | tail -1 > gpurun_out/r3_bench_tmp.json
  python - <<PY
import json
d=json.loads(open("gpurun_out/r3_bench_tmp.json").read())
print("lib=${lib:-current}  %.2f img/s" % d["value"])
for k in d["kernel_breakdown"][:6]:
    print("   %-50s %8.1f ms %5d launches %s TF issued_frac %s" % (k["name"], k["ms"], k["launches"], k["tflops"] and round(k["tflops"],1), k.get("issued_frac") and round(k["issued_frac"],3)))
PY
done
