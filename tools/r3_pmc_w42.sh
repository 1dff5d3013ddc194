#!/bin/bash
# L1 / L2 traffic of the hybrid Winograd kernel on two layer shapes (separate PMC passes, kernel-trace only)
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/pmc_w42
mkdir -p $OUT; cd /tmp; export TMPDIR=/tmp
rocprofv3 -L 2>/dev/null | grep -o "TCP_[A-Z_]*\|TCC_[A-Z_0-9]*sum\|TCC_EA0_RDREQ[A-Z_0-9]*" | sort -u | head -80 > $OUT/counters.txt
for shape in "32 128 128" "16 512 256"; do
  tag=$(echo $shape | tr ' ' '_')
  for set in "TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum" "TCC_HIT_sum TCC_MISS_sum" "FETCH_SIZE" "TCC_REQ_sum TCC_READ_sum"; do
    st=$(echo $set | tr ' ' '+')
    rocprofv3 --kernel-trace --output-format csv --pmc $set -d $OUT/p_${tag}_$st -o b -- python $REPO/tools/w42_one.py $shape > $OUT/p_${tag}_$st.log 2>&1
    python $REPO/tools/summarize_rocprof.py $OUT/p_${tag}_$st --out $OUT/s_${tag}_$st.json > /dev/null 2>&1
    rm -rf $OUT/p_${tag}_$st
  done
done
python - <<PY
import json,glob,os
for f in sorted(glob.glob("$OUT/s_*.json")):
    d=json.load(open(f))
    for k,v in d.get("kernels",d).items():
        if "wino42_f32" in k: print(os.path.basename(f), k, json.dumps(v)[:400])
PY
