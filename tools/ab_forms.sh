#!/bin/bash
# tools/ab_forms.sh [rounds] -- SELENITE_ARITH_AUTO in one launch or three on the no-decimator shapes (k_hilb_split16 recomputes a channel
# it guarded itself: FusedArgs::inl), interleaved rounds inside one call (one box): ms per step from bench.py --main-only.
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd "$R"
run() { # name args...
  local name=$1; shift
  echo "$name: $(python3 bench.py --main-only --steps 200 "$@" | python3 -c 'import json,sys; d=json.loads(sys.stdin.read()); print(d["ms_per_step"], d["roofline"]["launch_ms_median"], d["value"])')"
}
for rep in $(seq 1 ${1:-3}); do
  for w in cfg5 cfg2; do
    run $w/1   --workload $w --auto-launches 1
    run $w/3   --workload $w --auto-launches 3
    run $w/raw --workload $w --arith split16
  done
done
