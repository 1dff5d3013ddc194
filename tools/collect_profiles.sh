#!/bin/bash
# Run on the GPU box (via gpurun) from the repo root: rocprofv3 kernel-trace stats of the default bench command and
# separate PMC passes (never combined with trace domains other than --kernel-trace), summarised under gpurun_out/.
set -u
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/prof
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
BENCH="python $REPO/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-alt"
SHORT="python $REPO/bench.py --nfe 3 --steps 1 --warmup 0 --no-profile --no-cpu-baseline --no-graph --no-alt"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o bench -- $BENCH > $OUT/trace_bench.log 2>&1
grep -a "^{\"metric" $OUT/trace_bench.log | tail -1 > $OUT/trace_bench_line.json  # the JSON line of the traced process itself
python $REPO/tools/summarize_rocprof.py $OUT/trace --out $OUT/trace_summary.json > /dev/null 2>&1
rocprofv3 --kernel-trace --output-format csv --pmc FETCH_SIZE -d $OUT/pmc_fetch -o b -- $SHORT > $OUT/pmc_fetch.log 2>&1
python $REPO/tools/summarize_rocprof.py $OUT/pmc_fetch --out $OUT/pmc_fetch_summary.json > /dev/null 2>&1
rocprofv3 --kernel-trace --output-format csv --pmc WRITE_SIZE -d $OUT/pmc_write -o b -- $SHORT > $OUT/pmc_write.log 2>&1
python $REPO/tools/summarize_rocprof.py $OUT/pmc_write --out $OUT/pmc_write_summary.json > /dev/null 2>&1
rocprofv3 --kernel-trace --output-format csv --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE GRBM_GUI_ACTIVE -d $OUT/pmc_sq -o b -- $SHORT > $OUT/pmc_sq.log 2>&1
python $REPO/tools/summarize_rocprof.py $OUT/pmc_sq --out $OUT/pmc_sq_summary.json > /dev/null 2>&1
# split-precision mode: kernel statistics and HBM counters of the same workload
X3="python $REPO/bench.py --precision bf16x3 --steps 1 --warmup 1 --no-cpu-baseline --no-profile"
X3SHORT="python $REPO/bench.py --precision bf16x3 --nfe 3 --steps 1 --warmup 0 --no-profile --no-cpu-baseline --no-graph"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_x3 -o bench -- $X3 > $OUT/trace_x3.log 2>&1
find $OUT/trace_x3 -name "*kernel_stats.csv" -exec cp {} $OUT/bench_bf16x3_kernel_stats.csv \;
rocprofv3 --kernel-trace --output-format csv --pmc FETCH_SIZE -d $OUT/pmc_fetch_x3 -o b -- $X3SHORT > $OUT/pmc_fetch_x3.log 2>&1
python $REPO/tools/summarize_rocprof.py $OUT/pmc_fetch_x3 --out $OUT/pmc_fetch_x3_summary.json > /dev/null 2>&1
rocprofv3 --kernel-trace --output-format csv --pmc WRITE_SIZE -d $OUT/pmc_write_x3 -o b -- $X3SHORT > $OUT/pmc_write_x3.log 2>&1
python $REPO/tools/summarize_rocprof.py $OUT/pmc_write_x3 --out $OUT/pmc_write_x3_summary.json > /dev/null 2>&1
# keep only the small files (raw traces are large)
find $OUT/trace -name "*kernel_stats.csv" -exec cp {} $OUT/ \; 2>/dev/null
find $OUT -name "*_kernel_trace.csv" -size +2M -delete 2>/dev/null
find $OUT -name "*_counter_collection.csv" -size +2M -delete 2>/dev/null
find $OUT -name "*.db" -delete 2>/dev/null; find $OUT -type f -size +3M -delete 2>/dev/null; du -sh $OUT; ls -la $OUT $OUT/trace/* | head -40
tail -5 $OUT/trace_bench.log; tail -5 $OUT/pmc_sq.log
