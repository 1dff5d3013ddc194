#!/bin/bash
# Per-round evidence, run on the GPU box (via gpurun) from the repo root:  bash tools/collect_profiles.sh 6   (-> gpurun_out/prof6).  rocprofv3 kernel-trace statistics of the default bench
# command, separate PMC passes (never combined with trace domains other than --kernel-trace), and warmed bench lines (+ rocprofv3
# statistics) of BASELINE configs 3 and 5, the batch-size curve and the full-size parity log.  Summaries land under gpurun_out/prof<N>;
# `python tools/update_profiles.py r0<N>` copies them to profiles/.
set -u
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
ROUND=${1:-5}
OUT=$REPO/gpurun_out/prof$ROUND
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
PY="python $REPO/bench.py"
stats() {  # $1 = tag, rest = bench flags: rocprofv3 --kernel-trace --stats of one bench command, keep the small files
    local tag=$1; shift
    rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_$tag -o bench -- $PY "$@" > $OUT/trace_$tag.log 2>&1
    grep -a "^{\"metric" $OUT/trace_$tag.log | tail -1 > $OUT/${tag}_traced_line.json
    find $OUT/trace_$tag -name "*kernel_stats.csv" -exec cp {} $OUT/${tag}_kernel_stats.csv \;
    python $REPO/tools/summarize_rocprof.py $OUT/trace_$tag --out $OUT/${tag}_trace_summary.json > /dev/null 2>&1
    rm -rf $OUT/trace_$tag
}
# ---- headline (config 2): default line, traced line, PMC passes
$PY > $OUT/bench_line.json 2> $OUT/bench_line.err
stats bench --steps 1 --warmup 1 --no-cpu-baseline --no-alt
SHORT="$PY --nfe 3 --steps 1 --warmup 0 --no-profile --no-cpu-baseline --no-graph --no-alt"
for c in FETCH_SIZE WRITE_SIZE; do
    rocprofv3 --kernel-trace --output-format csv --pmc $c -d $OUT/pmc_$c -o b -- $SHORT > $OUT/pmc_$c.log 2>&1
    python $REPO/tools/summarize_rocprof.py $OUT/pmc_$c --out $OUT/pmc_${c}_summary.json > /dev/null 2>&1
    rm -rf $OUT/pmc_$c
done
rocprofv3 --kernel-trace --output-format csv --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE GRBM_GUI_ACTIVE -d $OUT/pmc_sq -o b -- $SHORT > $OUT/pmc_sq.log 2>&1
python $REPO/tools/summarize_rocprof.py $OUT/pmc_sq --out $OUT/pmc_sq_summary.json > /dev/null 2>&1
rm -rf $OUT/pmc_sq
# L1 <-> L2 traffic of the same command (what the weight fragments that bypass LDS cost): 128-byte requests L1 -> L2, L2 hits / misses
rocprofv3 --kernel-trace --output-format csv --pmc TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum -d $OUT/pmc_tcp -o b -- $SHORT > $OUT/pmc_tcp.log 2>&1
python $REPO/tools/summarize_rocprof.py $OUT/pmc_tcp --out $OUT/pmc_tcp_summary.json > /dev/null 2>&1
rm -rf $OUT/pmc_tcp
rocprofv3 --kernel-trace --output-format csv --pmc TCC_HIT_sum TCC_MISS_sum -d $OUT/pmc_tcc -o b -- $SHORT > $OUT/pmc_tcc.log 2>&1
python $REPO/tools/summarize_rocprof.py $OUT/pmc_tcc --out $OUT/pmc_tcc_summary.json > /dev/null 2>&1
rm -rf $OUT/pmc_tcc
# ---- config 3: blur gDDIM order0 NFE=50, B=256 (warmed, 3 timed rounds) + its rocprofv3 statistics
$PY --workload blur --steps 3 --warmup 1 --no-cpu-baseline > $OUT/config3_blur_line.json 2>/dev/null
stats config3_blur --workload blur --steps 1 --warmup 1 --no-cpu-baseline --no-alt
# ---- config 5: CLD NFE in {10, 20, 50, 100}, B=512, 64x64x3 (warmed, 2 timed rounds each) + rocprofv3 statistics of NFE=10
for n in 10 20 50 100; do
    $PY --image-size 64 --batch 512 --nfe $n --steps 2 --warmup 1 --no-cpu-baseline --no-alt > $OUT/config5_nfe${n}_line.json 2>/dev/null
done
stats config5_nfe10 --image-size 64 --batch 512 --nfe 10 --steps 1 --warmup 1 --no-cpu-baseline --no-alt
# ---- small batch (config 1 shape on the GPU): eager vs per-step graph vs one graph per call
for g in none step loop; do
    $PY --batch 16 --nfe 10 --steps 20 --warmup 3 --graph $g --no-alt --no-cpu-baseline --no-profile 2>/dev/null | tail -1 > $OUT/small_b16_nfe10_graph_$g.json
done
# ---- the driver's own flags, and the split6 prototype against the hybrid kernel: hipEvent times, then rocprofv3's per-kernel averages
$PY --steps 20 --warmup 5 > $OUT/bench_line_driver_flags.json 2>/dev/null
python $REPO/tools/bench_split6.py 20 2>/dev/null | tail -1 > $OUT/split6_bench.json
python $REPO/tools/bench_split6.py 100 2>/dev/null | tail -1 > $OUT/split6_bench_100_launches.json
python $REPO/tools/bench_split6.py 20 res 2>/dev/null | tail -1 > $OUT/split6_bench_residual.json
(cd $REPO && python tools/split6_layers.py 2>/dev/null | grep -v amdgpu.ids > $OUT/split6_layers.log)
(cd $REPO && bash tools/power_modes.sh $OUT/power > /dev/null 2>&1; cp $OUT/power/power_modes.json $OUT/power_modes.json)
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_split6 -o s6 -- python $REPO/tools/bench_split6.py 20 > $OUT/trace_split6.log 2>&1
find $OUT/trace_split6 -name "*kernel_stats.csv" -exec cp {} $OUT/split6_kernel_stats.csv \;
rm -rf $OUT/trace_split6
# ---- the batch-size curve (per-level kernels) and the full-size parity checks with their printed errors
python $REPO/tools/batch_sweep.py --batches 16,32,64,128,192,256,320,512,1024 --out $OUT/batch_sweep.json > $OUT/batch_sweep.log 2>&1
(cd $REPO && python -m pytest tests/test_gpu_parity_abs.py -m gpu -q -s 2>&1 | grep -a -E "max\|dx\||passed|failed|share of the flops" > $OUT/parity_abs_tests.log)
find $OUT -type f -size +3M -delete 2>/dev/null; du -sh $OUT; ls $OUT
