#!/bin/bash
# tools/gpu.sh [--timeout S] '<command>' -- rebuild the library, then run <command> on an MI355X box through gpurun
set -e
cd "$(dirname "$0")/.."
make -s -j8 -C selenite-lite_amd 2>&1 | grep -E "error|Error" && exit 1
T=1500
if [ "$1" = "--timeout" ]; then T=$2; shift 2; fi
exec /usr/local/graft/bin/gpurun --timeout $T -- "$1"
