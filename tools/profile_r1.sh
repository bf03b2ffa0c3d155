#!/bin/bash
# tools/profile_r1.sh <tag> [bench args...] -- run on the GPU box (through gpurun).
# Pass 1: rocprofv3 --kernel-trace --stats (per-kernel durations) around bench.py.
# Pass 2/3: --pmc counters ALONE (no trace domains), FETCH_SIZE and WRITE_SIZE in separate passes
#           (TCC slots: FETCH_SIZE 3, WRITE_SIZE 2 -- MI355X_MICROARCH.md "rocprofv3 PMC slots").
# Results land in gpurun_out/prof_<tag>/ ; summaries worth keeping are copied to profiles/ by hand.
set -u
TAG=$1; shift
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$R/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o t -- python3 $R/bench.py --main-only "$@" > $OUT/bench_trace.json 2> $OUT/trace.err
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -o f -- python3 $R/bench.py --main-only "$@" --spinup-ms 0 --steps 3 --warmup 1 > $OUT/bench_pmc_fetch.json 2> $OUT/pmc_fetch.err
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -o w -- python3 $R/bench.py --main-only "$@" --spinup-ms 0 --steps 3 --warmup 1 > $OUT/bench_pmc_write.json 2> $OUT/pmc_write.err
find $OUT -name "*.csv" | head -20
