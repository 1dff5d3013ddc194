#!/bin/bash
# persistent (working tree) vs one-workgroup-per-tile (GDDIM_LIB=libgddim_w2head.so) 2-D Winograd kernel, same box
mkdir -p gpurun_out
python -m pytest tests/test_gpu_ops.py -x -q -m gpu -k "wino" 2>&1 | tail -3
echo "--- persistent"; python tools/bench_wino2d.py --only2d 2>/dev/null | tee gpurun_out/ab5_w2_pers.log
echo "--- head"; GDDIM_LIB=gddim_amd/libgddim_w2head.so python tools/bench_wino2d.py --only2d 2>/dev/null | tee gpurun_out/ab5_w2_head.log
echo "--- persistent, NOPERSIST=1"; GDDIM_WINO2_NOPERSIST=1 python tools/bench_wino2d.py --only2d 2>/dev/null | tail -1
B="python bench.py --steps 2 --warmup 1 --no-alt --no-cpu-baseline"
$B > gpurun_out/ab5_pers.json 2>/dev/null
GDDIM_LIB=gddim_amd/libgddim_w2head.so $B > gpurun_out/ab5_head.json 2>/dev/null
for f in pers head; do python - <<PY
import json
d=json.loads([l for l in open("gpurun_out/ab5_$f.json") if l.startswith("{")][-1])
kb={k["name"]:k for k in d["kernel_breakdown"]}
print("$f", round(d["value"],2), "img/s; wino2d", round(kb["conv3x3_wino2d_f32_kernel"]["ms"],1), round(kb["conv3x3_wino2d_f32_kernel"]["tflops"],1), "total", round(sum(k["ms"] for k in d["kernel_breakdown"]),1))
PY
done
