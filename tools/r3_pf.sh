#!/bin/bash
cd ${GRAFT_REPO_ROOT:-.}
python tools/bench_wino2d.py --variants wino42 >/dev/null 2>&1  # warm the box
for rep in 1 2; do
for t in "" ${TAGS}; do
  lib=""; [ -n "$t" ] && lib=gddim_amd/libgddim_$t.so
  echo "${t:-base}: $(GDDIM_LIB=$lib python tools/bench_wino2d.py --variants wino42 2>/dev/null | grep -E 'H=32 128->128|H=32 256->128|H=16 256->256|H=16 512->256|per U-Net' | sed -e 's/wino42: vs direct .* |//' -e 's/  wino42  //' | tr '\n' '|')"
done
done
