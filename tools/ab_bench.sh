#!/bin/bash
# tools/ab_bench.sh <outdir> [bench args] -- bench.py --main-only once per library in selenite-lite_amd/variants/
# (plus the product library), same box, same run; prints one line per variant.
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
O=$R/gpurun_out/$1; shift
mkdir -p $O
for rep in 1 2; do
for lib in $R/selenite-lite_amd/libselenite_rx.so $R/selenite-lite_amd/variants/lib_*.so; do
  n=$(basename $lib .so)
  SELENITE_RX_LIB=$lib python3 $R/bench.py --main-only "$@" > $O/$n.$rep.json 2> $O/$n.$rep.err
  python3 -c "import json; d=json.load(open('$O/$n.$rep.json')); print('%-28s rep$rep %9.1f Gs/s  %.4f ms  frac %.4f' % ('$n', d['value']/1e3, d['roofline']['launch_ms_hip_events'], d['roofline']['frac']))" 2>/dev/null || echo "$n rep$rep FAILED: $(tail -1 $O/$n.$rep.err)"
done
done
