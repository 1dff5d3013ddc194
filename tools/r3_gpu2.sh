#!/bin/bash
# round 3, GPU call 2: 16-channel chunks vs the 8-channel-chunk build of the hybrid Winograd kernel (same box)
cd ${GRAFT_REPO_ROOT:-.}
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_ops.py -x -q -k "winograd42" > gpurun_out/r3_w42_tests.log 2>&1
echo "tests rc=$?" >> gpurun_out/r3_w42_tests.log
tail -4 gpurun_out/r3_w42_tests.log
for lib in gddim_amd/libgddim_w42c8.so ""; do
  echo "== lib=${lib:-current}"
  GDDIM_LIB=$lib timeout 600 python tools/bench_wino2d.py --variants wino42 --iters 10 2>&1 | sed -e 's/wino42: vs direct .* |//'
done
