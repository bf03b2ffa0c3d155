#!/bin/bash
# tools/run_cpu_tests_sanitized.sh [pytest args] -- the CPU test suite over the AddressSanitizer + UBSan builds of the oracle,
# of the real-CMSIS / real-dsp_if.c harnesses (oracle/_san/, `make -C oracle SAN=1`) and of the host-side C examples' helpers.
# python itself is not instrumented, so libasan is preloaded; leak checking is off (the interpreter "leaks" by design).
# CPU only -- the GPU build is never sanitized.  Exit code: pytest's; any sanitizer report aborts the process (non-zero).
set -e
R=$(cd "$(dirname "$0")/.." && pwd)
make -s -C "$R/oracle" SAN=1
ASAN=$(gcc -print-file-name=libasan.so)
export LD_PRELOAD="$ASAN${LD_PRELOAD:+:$LD_PRELOAD}"
export ASAN_OPTIONS="detect_leaks=0:abort_on_error=1:halt_on_error=1"
export UBSAN_OPTIONS="halt_on_error=1:abort_on_error=1:print_stacktrace=1"
export SELENITE_ORACLE_SAN=1
cd "$R"
if [ $# -eq 0 ]; then set -- tests -m "not gpu" -q -x -p no:cacheprovider; fi
exec python3 -m pytest "$@"
