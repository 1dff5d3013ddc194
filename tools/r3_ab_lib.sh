#!/bin/bash
# in-box A/B of the working tree against another build of the library: bash tools/r3_ab_lib.sh gddim_amd/libgddim_<tag>.so
cd ${GRAFT_REPO_ROOT:-.}
OTHER=$1
timeout 900 python -m pytest tests/test_gpu_ops.py -x -q -k "winograd42" 2>&1 | tail -2
for rep in 1 2; do
for lib in $OTHER ""; do
  echo "rep $rep lib=${lib:-current}: $(GDDIM_LIB=$lib python tools/bench_wino2d.py --variants wino42 2>/dev/null | grep -E 'H=32 128->128|H=32 256->128|H=16 256->256|H=16 512->256|H= 8 256|per U-Net' | sed -e 's/wino42: vs direct .* |//' -e 's/  wino42  //' | tr '\n' '|')"
done
done
for lib in $OTHER ""; do
  GDDIM_LIB=$lib timeout 900 python bench.py --steps 3 --warmup 1 --no-alt --no-cpu-baseline 2>/dev/null | tail -1 > gpurun_out/r3_bench_tmp.json
  python - <<PY
import json
d=json.loads(open("gpurun_out/r3_bench_tmp.json").read())
k=d["kernel_breakdown"][0]
print("lib=${lib:-current}  %.2f img/s   %s %.1f ms %.1f TF issued %.3f" % (d["value"], k["name"], k["ms"], k["tflops"], k["issued_frac"]))
PY
done
