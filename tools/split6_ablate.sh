#!/bin/bash
# Timing ablations of the split6 prototype (results wrong by construction), built beforehand with
#   for a in 1 2 4 8 16 32 64 63 18 51; do python tools/build_variant.py s6a$a kernels/conv3x3_wino2d_split6.hip -fno-slp-vectorize -DS6_ABL=$a; done
#   for a in 63 51 16 18 1 2 4 32 64 39; do python tools/build_variant.py s6t$a kernels/conv3x3_wino2d_split6.hip -fno-slp-vectorize -DS6_TRACE -DS6_ABL=$a; done
# S6_ABL bits: 1 no patch loads in the loop, 2 no V stores, 4 no A fragment reads, 8 no barrier, 16 no transform / split arithmetic,
# 32 no U loads in the loop, 64 no MFMAs.   bash tools/split6_ablate.sh  -> gpurun_out/r06c/ablation.log, ablation_trace.log
cd ${GRAFT_REPO_ROOT:-.}
mkdir -p gpurun_out/r06c
for a in "" 1 2 4 8 16 32 64 63 18 51; do
  if [ -z "$a" ]; then lib=""; else lib=$PWD/gddim_amd/libgddim_s6a$a.so; fi
  [ -n "$lib" ] && [ ! -e "$lib" ] && continue
  echo "ABL=$a $(GDDIM_LIB=$lib python tools/bench_split6.py 20 2>/dev/null | grep '^[0-9]' | python -c "
import sys, json
for l in sys.stdin:
    k, d = l.split(' ', 1); d = json.loads(d); print(k, '%.1f us' % d['split6_us'], end='  ')
")"
done | tee gpurun_out/r06c/ablation.log
for a in trace t63 t51 t39 t16 t18 t1 t2 t4 t32 t64; do
  lib=$PWD/gddim_amd/libgddim_s6$a.so
  [ ! -e "$lib" ] && continue
  echo "== $a"; GDDIM_LIB=$lib python tools/split6_trace.py 2>/dev/null | grep "wg601 wave0" 
done | tee gpurun_out/r06c/ablation_trace.log
