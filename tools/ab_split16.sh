#!/bin/bash
# tools/ab_split16.sh <variants...> -- raw split16 on the bench workload for each A/B library (tools/variants/build.sh), interleaved
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}; cd "$R"
for rep in 1 2 3; do
for v in "$@"; do
  echo "$v: $(SELENITE_RX_LIB=$R/selenite-lite_amd/variants/lib_$v.so python bench.py --main-only --steps 200 --arith split16 | python -c 'import json,sys; d=json.loads(sys.stdin.read()); print(d["ms_per_step"], d["value"])')"
done; done
