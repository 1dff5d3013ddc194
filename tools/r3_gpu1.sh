#!/bin/bash
# round 3, GPU call 1: the new hybrid Winograd kernel (unit tests, per-layer bench), attention remap, in-box A/B of the sampler
cd ${GRAFT_REPO_ROOT:-.}
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_ops.py -x -q -k "winograd42 or attention" -s > gpurun_out/r3_w42_tests.log 2>&1
echo "tests rc=$?" >> gpurun_out/r3_w42_tests.log
tail -5 gpurun_out/r3_w42_tests.log
timeout 600 python tools/bench_wino2d.py --variants wino2d,wino42 --iters 10 > gpurun_out/r3_w42_layers.log 2>&1
cat gpurun_out/r3_w42_layers.log
for v in 1 0; do
  GDDIM_NOWINO42=$v timeout 900 python bench.py --steps 3 --warmup 1 --no-alt --no-cpu-baseline 2>gpurun_out/r3_bench_nw42_$v.err | tail -1 > gpurun_out/r3_bench_nw42_$v.json
  python - <<PY
import json
d=json.loads(open("gpurun_out/r3_bench_nw42_$v.json").read())
print("NOWINO42=$v  %.2f img/s" % d["value"])
for k in d["kernel_breakdown"][:8]:
    print("   %-50s %8.1f ms %5d launches %s TF issued_frac %s" % (k["name"], k["ms"], k["launches"], k["tflops"] and round(k["tflops"],1), k.get("issued_frac") and round(k["issued_frac"],3)))
PY
done
