#!/bin/bash
# in-box A/B of variant builds of the hybrid Winograd kernel: bash tools/r3_ab_variants.sh tag1 tag2 ... (libgddim_w42<tag>.so), per-layer bench
cd ${GRAFT_REPO_ROOT:-.}
for rep in 1 2; do
for t in "" "$@"; do
  lib=""; [ -n "$t" ] && lib=gddim_amd/libgddim_w42$t.so
  echo "rep $rep ${t:-default}: $(GDDIM_LIB=$lib python tools/bench_wino2d.py --variants wino42 2>/dev/null | grep -E 'H=32 128->128|H=16 256->256|H=16 512->256|H= 8 256|per U-Net' | sed -e 's/wino42: vs direct .* |//' -e 's/  wino42  //' | tr '\n' '|')"
done
done
