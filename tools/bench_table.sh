#!/bin/bash
# tools/bench_table.sh <outfile> -- bench.py --main-only for every workload / option of interest on ONE box:
# "<args> | Msamples/s  ms per launch  roofline.frac  kernel  [nco path]"
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$1
: > $OUT
run() {
  python3 $R/bench.py --main-only "$@" 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read())
print('%-34s | %10.1f  %.4f  %.4f  %s  [%s]' % ('$*' or '(default)', d['value'], d['roofline']['launch_ms_hip_events'], d['roofline']['frac'], d['config']['kernel'], d['config']['nco']))" >> $OUT
}
run
run --arith split16
run --nco shared_table
run --nco per_channel_grid --arith split16
run --nco per_channel --arith split16
run --nco per_channel_grid_wide
run --nco per_channel_grid_wide --arith cmsis
run --io q15
run --global-gain
run --global-gain --arith split16
run --arith fma
run --arith cmsis
run --workload cfg2
run --workload cfg2 --arith split16
run --workload cfg2 --io q15
run --workload cfg2_192
run --workload cfg5
run --workload cfg5 --arith split16
run --workload cfg4
run --workload cfg4 --io q15
cat $OUT
