#!/bin/bash
# timing ablations of the 2-D Winograd kernel (tools/build_variant.py w2abl<mask> ... -DW2_ABL=<mask>), same box
for t in "" ${ABLS:-1 2 3 4 8 16 31}; do
  lib=""; [ -n "$t" ] && lib=gddim_amd/libgddim_w2abl$t.so
  echo "ABL=${t:-0}: $(GDDIM_LIB=$lib python tools/bench_wino2d.py --only2d 2>/dev/null | grep -E 'H=32 128->128|H=16 512->256|per U-Net' | sed -e 's/2d vs direct rel max diff [0-9.e+-]*//' | tr '\n' '|')"
done
