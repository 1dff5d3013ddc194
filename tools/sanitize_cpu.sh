#!/bin/bash
# AddressSanitizer + UndefinedBehaviorSanitizer on the HOST code of the native library: engine.cpp, plan.cpp,
# options.cpp and host_math.cpp are compiled with -fsanitize=address,undefined (host pass only: -fno-gpu-sanitize; GPU ASan is not
# available on this pool), linked with the regular kernel objects into gddim_amd/_build/libgddim_amd_asan.so, and the CPU tests that
# exercise that code without a GPU -- the ABI surface, plan-only engines (planner, arena aliasing, option snapshots, parameter packing
# and folding), the native R(t) scans and coefficient tables -- run against it with the sanitizer runtime preloaded into python.
#   tools/sanitize_cpu.sh [log]        (default log: profiles/r06_sanitize_cpu.log)
set -u
ROOT="$(cd "$(dirname "$0")/.." && pwd)"
LOG="${1:-$ROOT/profiles/r06_sanitize_cpu.log}"
B="$ROOT/gddim_amd/_build"
HIPCC="${HIPCC:-/opt/rocm/bin/hipcc}"
RT="$(ls /opt/rocm/lib/llvm/lib/clang/*/lib/linux/libclang_rt.asan-x86_64.so | head -1)"
mkdir -p "$B/asan"
python -c "import sys; sys.path.insert(0, '$ROOT'); from gddim_amd import build; build.build()" || exit 1   # the kernel objects
pids=()
for f in engine plan options host_math; do
  "$HIPCC" -x hip --offload-arch=gfx950 -O1 -g -std=c++17 -fPIC -fsanitize=address,undefined -fno-gpu-sanitize -fno-omit-frame-pointer \
     -fno-sanitize-recover=undefined -c "$ROOT/gddim_amd/csrc/$f.cpp" -o "$B/asan/$f.o" & pids+=($!)
done
for p in "${pids[@]}"; do wait "$p" || exit 1; done
"$HIPCC" --offload-arch=gfx950 -shared -fPIC -fsanitize=address,undefined -shared-libsan -o "$B/libgddim_amd_asan.so" \
   "$B"/asan/{engine,plan,options,host_math}.o \
   "$B"/kernels_{igemm,igemm_bf16x3,conv3x3_halo_bf16x3,conv3x3_wino2d_f32,conv3x3_wino42_f32,conv3x3_wino2d_split6,attn_fused_f32,norm_act,update}.hip.o || exit 1
{
  echo "# $(date -u +%FT%TZ)  host ASan + UBSan build of gddim_amd (engine.cpp plan.cpp options.cpp host_math.cpp), runtime $RT"
  echo "# GDDIM_LIB=$B/libgddim_amd_asan.so LD_PRELOAD=<asan rt> pytest tests/test_abi_cpu.py tests/test_plan_cpu.py tests/test_oracle_cld.py tests/test_golden.py tests/test_checkpoints_cpu.py"
  cd "$ROOT" && GDDIM_LIB="$B/libgddim_amd_asan.so" LD_PRELOAD="$RT" \
    ASAN_OPTIONS="detect_leaks=0:halt_on_error=1:abort_on_error=0:detect_odr_violation=0" UBSAN_OPTIONS="print_stacktrace=1:halt_on_error=1" \
    python -m pytest -m "not gpu" tests/test_abi_cpu.py tests/test_plan_cpu.py tests/test_oracle_cld.py tests/test_golden.py tests/test_checkpoints_cpu.py -x -q -p no:cacheprovider 2>&1 | grep -v "^$" | tail -40
  echo "# exit code of pytest: ${PIPESTATUS[0]}"
} > "$LOG" 2>&1
tail -15 "$LOG"
