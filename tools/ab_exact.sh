#!/bin/bash
# tools/ab_exact.sh <variant> ... -- the bit-exact arithmetic (k_ssb_fused<0,...>, --arith cmsis) on the product library and on A/B libraries
# of rx_fused_exact.hip (UNIT=rx_fused_exact tools/variants/build.sh), interleaved, three rounds; then AUTO on the stop-band-heavy workload
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}; cd "$R"
V=$R/selenite-lite_amd/variants
run() { local name=$1 lib=$2; shift 2; echo "$name: $(SELENITE_RX_LIB=$lib python3 bench.py --main-only --steps 100 "$@" | python3 -c 'import json,sys; d=json.loads(sys.stdin.read()); print(d["ms_per_step"], d["roofline"]["launch_ms_median"], d["value"], d["roofline"]["frac"], d["config"]["kernel"])')"; }
for rep in 1 2 3; do
  run product $R/selenite-lite_amd/libselenite_rx.so --arith cmsis
  for v in "$@"; do run $v $V/lib_$v.so --arith cmsis; done
done
run product_stopband $R/selenite-lite_amd/libselenite_rx.so --nco per_channel_grid_wide
for v in "$@"; do run ${v}_stopband $V/lib_$v.so --nco per_channel_grid_wide; done
