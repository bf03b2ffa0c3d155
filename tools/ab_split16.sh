#!/bin/bash
# tools/ab_split16.sh <variants...> -- raw split16 on the bench workload for each A/B library (tools/build_variants.sh) and the round-3 copy, interleaved
cd $GRAFT_REPO_ROOT
for rep in 1 2 3; do
for v in "$@"; do
  if [ "$v" = r3 ]; then
    echo "r3: $(python selenite-lite_amd/variants/r3/bench.py --main-only --steps 200 --arith split16 | python -c 'import json,sys; d=json.loads(sys.stdin.read()); print(d["ms_per_step"], d["value"])')"
  else
    echo "$v: $(SELENITE_RX_LIB=$GRAFT_REPO_ROOT/selenite-lite_amd/variants/lib_$v.so python bench.py --main-only --steps 200 --arith split16 | python -c 'import json,sys; d=json.loads(sys.stdin.read()); print(d["ms_per_step"], d["value"])')"
  fi
done; done
