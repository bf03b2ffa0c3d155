#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}; cd "$R"
V=$R/selenite-lite_amd/variants
run() { local name=$1 lib=$2; shift 2; echo "$name: $(SELENITE_RX_LIB=$lib python3 bench.py --main-only --steps 200 "$@" | python3 -c 'import json,sys; d=json.loads(sys.stdin.read()); print(d["ms_per_step"], d["roofline"]["launch_ms_median"], d["value"], d["roofline"]["frac"])')"; }
for rep in 1 2 3; do
  run cw_new $R/selenite-lite_amd/libselenite_rx.so --workload cfg4
  run cw_old $V/lib_cw_r5.so --workload cfg4
  run cw_new_q15 $R/selenite-lite_amd/libselenite_rx.so --workload cfg4 --io q15
  run cw_old_q15 $V/lib_cw_r5.so --workload cfg4 --io q15
done
