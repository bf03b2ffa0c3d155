#!/bin/bash
# tools/ab_r3.sh [bench args] -- this build against the round-3 build (a copy under selenite-lite_amd/variants/r3: its own bench.py, binding and library), interleaved
cd $GRAFT_REPO_ROOT
for rep in 1 2 3; do
  echo "r4: $(python bench.py --main-only --steps 200 "$@" | python -c 'import json,sys; d=json.loads(sys.stdin.read()); print(d["ms_per_step"], d["value"], d["roofline"]["launch_ms_median"])')"
  echo "r3: $(python selenite-lite_amd/variants/r3/bench.py --main-only --steps 200 "$@" | python -c 'import json,sys; d=json.loads(sys.stdin.read()); print(d["ms_per_step"], d["value"], d["roofline"]["launch_ms_median"])')"
done
