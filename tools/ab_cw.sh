#!/bin/bash
# tools/ab_cw.sh [<variant> ...] -- cfg4 (f32 and int16 slots) on the product library and on A/B libraries of it
# (selenite-lite_amd/variants/lib_<variant>.so: tools/variants/build.sh with UNIT=rx_cw, or a kept copy of an older build), interleaved, three rounds
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}; cd "$R"
V=$R/selenite-lite_amd/variants
[ $# -gt 0 ] || set -- cw_r5
run() { local name=$1 lib=$2; shift 2; echo "$name: $(SELENITE_RX_LIB=$lib python3 bench.py --main-only --steps 200 "$@" | python3 -c 'import json,sys; d=json.loads(sys.stdin.read()); print(d["ms_per_step"], d["roofline"]["launch_ms_median"], d["value"], d["roofline"]["frac"])')"; }
for rep in 1 2 3; do
  run product $R/selenite-lite_amd/libselenite_rx.so --workload cfg4
  for v in "$@"; do run $v $V/lib_$v.so --workload cfg4; done
  if [ -z "$AB_NO_Q15" ]; then
    run product_q15 $R/selenite-lite_amd/libselenite_rx.so --workload cfg4 --io q15
    for v in "$@"; do run ${v}_q15 $V/lib_$v.so --workload cfg4 --io q15; done
  fi
done
