#!/bin/bash
# PMC passes over one igemm shape (GPU box): bash tools/pmc_igemm.sh <shape index> <variant>
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/pmc_igemm_$1_$2
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
CMD=${PMC_CMD:-"python $REPO/tools/bench_igemm.py --only $1 --iters 5"}  # PMC_CMD: profile another command (e.g. a short bench.py run)
export GDDIM_IGEMM_VARIANT=$2
i=0
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM" \
           "SQ_VALU_MFMA_BUSY_CYCLES SQ_VALU_MFMA_COEXEC_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_WAIT_INST_LDS" \
           "SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_VMEM_RD SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_IFETCH" \
           "TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCC_HIT_sum TCC_MISS_sum" \
           "GRBM_GUI_ACTIVE TCP_TCC_READ_REQ_LATENCY_sum TCP_PENDING_STALL_CYCLES_sum TA_BUSY_avr"; do
  i=$((i+1))
  rocprofv3 --kernel-trace --output-format csv --pmc $set -d $OUT/p$i -o b -- $CMD > $OUT/p$i.log 2>&1
done
python $REPO/tools/summarize_rocprof.py $OUT --out $OUT/summary.json > /dev/null 2>&1
find $OUT -name "*.db" -delete; find $OUT -type f -size +1M -delete
python - <<PY
import json
d=json.load(open("$OUT/summary.json"))
for k,v in d["kernels"].items():
    if "igemm" in k or "wino" in k:
        print(k, "avg_us", v.get("avg_us"))
        for c,x in sorted(v.get("counters_mean_per_dispatch",{}).items()): print("   %-34s %.4g"%(c,x))
PY
tail -3 $OUT/p4.log | cut -c1-300
