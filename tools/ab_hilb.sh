#!/bin/bash
# tools/ab_hilb.sh <variant> ... -- cfg5 and cfg2 (f32 and int16 slots) on the product library and on A/B libraries (selenite-lite_amd/variants/lib_<variant>.so), interleaved, three rounds
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}; cd "$R"
V=$R/selenite-lite_amd/variants
run() { local name=$1 lib=$2; shift 2; echo "$name: $(SELENITE_RX_LIB=$lib python3 bench.py --main-only --steps 200 "$@" | python3 -c 'import json,sys; d=json.loads(sys.stdin.read()); print(d["ms_per_step"], d["roofline"]["launch_ms_median"], d["value"], d["roofline"]["frac"])')"; }
for rep in 1 2 3; do
  for w in "--workload cfg5" "--workload cfg5 --io q15" "--workload cfg2" "--workload cfg2 --io q15"; do
    run "product $w" $R/selenite-lite_amd/libselenite_rx.so $w
    for v in "$@"; do run "$v $w" $V/lib_$v.so $w; done
  done
done
