#!/bin/bash
# gn_apply variants: loads in flight, nontemporal loads / stores
for lib in "" gnu8 gnu2 gnnt1 gnnt2 gnnt3; do
  L=""; [ -n "$lib" ] && L=gddim_amd/libgddim_$lib.so
  GDDIM_LIB=$L python bench.py --steps 2 --warmup 1 --no-alt --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1])
kb={k['name']:k for k in d['kernel_breakdown']}
print('${lib:-default}', round(d['value'],2), 'img/s; gn_apply', round(kb['gn_apply']['ms'],1), 'ms', round(kb['gn_apply']['gbs']), 'GB/s; total', round(sum(k['ms'] for k in d['kernel_breakdown']),1))"
done
