#!/bin/bash
# tools/ab_hilb.sh [old library] -- the no-decimator shapes (k_hilb_split16: cfg2 literal, cfg5 shard; f32 and int16 slots) on the product
# library and on an older build of it, interleaved rounds on one box
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}; cd "$R"
OLD=${1:-$R/selenite-lite_amd/variants/lib_main.so}
run() { local name=$1 lib=$2; shift 2; echo "$name: $(SELENITE_RX_LIB=$lib python3 bench.py --main-only --steps 200 "$@" | python3 -c 'import json,sys; d=json.loads(sys.stdin.read()); print(d["ms_per_step"], d["roofline"]["launch_ms_median"], d["value"], d["roofline"]["frac"])')"; }
for rep in 1 2 3; do
  for w in cfg2 cfg5; do
    run $w/new $R/selenite-lite_amd/libselenite_rx.so --workload $w
    run $w/old $OLD --workload $w
    run $w/new/q15 $R/selenite-lite_amd/libselenite_rx.so --workload $w --io q15
    run $w/old/q15 $OLD --workload $w --io q15
  done
done
