#!/bin/bash
# where the waves of the hybrid Winograd kernel spend their cycles (SQ wait / active counters, separate PMC passes, kernel-trace only)
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/pmc_w42_wait
mkdir -p $OUT; cd /tmp; export TMPDIR=/tmp
for shape in "32 128 128" "16 256 256"; do
  tag=$(echo $shape | tr ' ' '_')
  for set in "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM" "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_VMEM_RD" "SQ_WAIT_INST_LDS SQ_INST_CYCLES_VMEM SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE"; do
    st=$(echo $set | tr ' ' '+' | cut -c1-60)
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
        if "wino42_f32" in k: print(os.path.basename(f)[:12], json.dumps(v.get("counters_mean_per_dispatch",v))[:600])
PY
