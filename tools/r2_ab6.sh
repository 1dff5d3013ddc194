#!/bin/bash
# working-tree 2-D Winograd kernel: operator tests, layer timings, sampler line
mkdir -p gpurun_out
python -m pytest tests/test_gpu_ops.py -x -q -m gpu -k "wino" 2>&1 | tail -3
python tools/bench_wino2d.py --only2d 2>/dev/null | tee gpurun_out/ab6_w2.log
python bench.py --steps 2 --warmup 1 --no-alt --no-cpu-baseline > gpurun_out/ab6.json 2>/dev/null
python - <<PY
import json
d=json.loads([l for l in open("gpurun_out/ab6.json") if l.startswith("{")][-1])
kb={k["name"]:k for k in d["kernel_breakdown"]}
print(round(d["value"],2), "img/s; wino2d", round(kb["conv3x3_wino2d_f32_kernel"]["ms"],1), round(kb["conv3x3_wino2d_f32_kernel"]["tflops"],1), "total", round(sum(k["ms"] for k in d["kernel_breakdown"]),1))
PY
