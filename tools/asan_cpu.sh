#!/bin/bash
# Host-side AddressSanitizer + UBSan run of the CPU test tier (SURVEY.md §5 "race detection / sanitizers"): builds
# pybnesian_amd/libpbn_hip_asan.so (`make -f Makefile.asan`: host code instrumented, device code as usual) and runs `pytest -m "not gpu"`
# against it.  CPU container only - GPU ASan / XNACK are not available on the pool.   bash tools/asan_cpu.sh [log]
set -e
cd "$(dirname "$0")/.."
LOG=${1:-profiles/r3/asan_cpu.log}
make -C pybnesian_amd/csrc -f Makefile.asan -j8 > /dev/null
RT=$(/opt/rocm/lib/llvm/bin/clang -print-file-name=libclang_rt.asan-x86_64.so)
# detect_leaks=0: the interpreter itself is not leak-clean; everything else (heap overflows, use after free, UB) aborts the run
export ASAN_OPTIONS=detect_leaks=0:abort_on_error=1:halt_on_error=1 UBSAN_OPTIONS=print_stacktrace=1:halt_on_error=1
export PBN_LIB=$PWD/pybnesian_amd/libpbn_hip_asan.so
{ echo "# LD_PRELOAD=$RT PBN_LIB=$PBN_LIB python -m pytest tests -m 'not gpu' -q   ($(date -u +%F))"
  LD_PRELOAD=$RT python -m pytest tests -m "not gpu" -q -p no:cacheprovider 2>&1 | tail -25; } | tee "$LOG"
