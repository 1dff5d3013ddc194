#!/bin/bash
# timing ablations of the hybrid Winograd kernel (tools/build_variant.py w42abl<mask> ... -DW42_ABL=<mask>), same box
cd ${GRAFT_REPO_ROOT:-.}
for t in "" ${ABLS:-1 2 4 8 16 5 31}; do
  lib=""; [ -n "$t" ] && lib=gddim_amd/libgddim_w42abl$t.so
  echo "ABL=${t:-0}: $(GDDIM_LIB=$lib python tools/bench_wino2d.py --variants wino42 2>/dev/null | grep -E 'H=32 128->128|H=16 512->256|per U-Net' | sed -e 's/wino42: vs direct .* |//' | tr '\n' '|')"
done
