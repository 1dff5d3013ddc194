#!/bin/bash
# persistent 2-D Winograd kernel (working tree) vs the committed one-workgroup-per-tile kernel (libgddim_w2head.so), same box
mkdir -p gpurun_out
python -m pytest tests/test_gpu_ops.py -x -q -m gpu -k "wino" 2>&1 | tail -3
for lib in "" gddim_amd/libgddim_w2head.so; do
  echo "lib=${lib:-default}"
  GDDIM_LIB=$lib python tools/bench_wino2d.py --only2d 2>/dev/null | grep -E "H=32 128->128|H=16 512->256|H= 8 256|per U-Net"
  GDDIM_LIB=$lib python bench.py --steps 2 --warmup 1 --no-alt --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1])
kb={k['name']:k for k in d['kernel_breakdown']}
print(round(d['value'],2), 'img/s; wino2d', round(kb['conv3x3_wino2d_f32_kernel']['ms'],1), round(kb['conv3x3_wino2d_f32_kernel']['tflops'],1), 'total', round(sum(k['ms'] for k in d['kernel_breakdown']),1))"
done
echo "NOPERSIST=1:"; GDDIM_WINO2_NOPERSIST=1 python tools/bench_wino2d.py --only2d 2>/dev/null | tail -1
