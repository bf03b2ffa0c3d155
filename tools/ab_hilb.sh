#!/bin/bash
# tools/ab_hilb.sh [library ...] -- the no-decimator shapes (k_hilb_split16: cfg2 literal, cfg5 shard; f32 and int16 slots) on the product
# library (`new`) and on other builds of it (named by their file: variants/lib_<name>.so), interleaved rounds on one box
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}; cd "$R"
[ $# -gt 0 ] || set -- "$R/selenite-lite_amd/variants/lib_main.so"
run() { local name=$1 lib=$2; shift 2; echo "$name: $(SELENITE_RX_LIB=$lib python3 bench.py --main-only --steps 200 "$@" | python3 -c 'import json,sys; d=json.loads(sys.stdin.read()); print(d["ms_per_step"], d["roofline"]["launch_ms_median"], d["value"], d["roofline"]["frac"])')"; }
for rep in 1 2 3; do
  for w in ${WORKLOADS:-cfg2 cfg5}; do
    for io in f32 q15; do
      run $w/new/$io $R/selenite-lite_amd/libselenite_rx.so --workload $w --io $io
      for lib in "$@"; do n=$(basename $lib .so); run $w/${n#lib_}/$io $lib --workload $w --io $io; done
    done
  done
done
