#!/bin/bash
# tools/ab_exact.sh <variant names...> -- time the bit-exact kernel of each A/B library (tools/build_exact_variants.sh; "main" = the regular
# build) on the cfg3 workload, interleaved, three rounds
cd $GRAFT_REPO_ROOT
for rep in 1 2 3; do
for v in "$@"; do
  lib=$GRAFT_REPO_ROOT/selenite-lite_amd/variants/lib_$v.so
  [ "$v" = main ] && lib=$GRAFT_REPO_ROOT/selenite-lite_amd/libselenite_rx.so
  echo "$v: $(SELENITE_RX_LIB=$lib python bench.py --main-only ${AB_ARGS:---arith cmsis} --steps 100 | python -c 'import json,sys; d=json.loads(sys.stdin.read()); print(d["ms_per_step"], d["value"], d["roofline"]["launch_ms_median"])')"
done; done
