#!/bin/bash
# tools/ab_q15.sh <variants...> -- raw split16 with int16 slots on the bench workload for each A/B library of tools/build_variants_q15.sh,
# three interleaved rounds: "name: ms per step  Msamples/s"
cd $GRAFT_REPO_ROOT
for rep in 1 2 3; do
for v in "$@"; do
  echo "$v: $(SELENITE_RX_LIB=$GRAFT_REPO_ROOT/selenite-lite_amd/variants/lib_q15_$v.so python bench.py --main-only --steps 200 --arith split16 --io q15 2>/dev/null | python -c 'import json,sys; d=json.loads(sys.stdin.read()); print(d["ms_per_step"], d["value"])')"
done; done
