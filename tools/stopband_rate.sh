#!/bin/bash
# tools/stopband_rate.sh -- the rerun pass of SELENITE_ARITH_AUTO on the stop-band-heavy workload for several launch shapes
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}; cd "$R"
for g in 2048 4096 8192 16384 65536; do
  echo "rerun grid $g: $(SELENITE_RX_RERUN_GRID=$g python bench.py --main-only --nco per_channel_grid_wide --steps 100 | python -c 'import json,sys; d=json.loads(sys.stdin.read()); print(d["ms_per_step"], d["value"])')"
done
echo "headline: $(python bench.py --main-only --steps 200 | python -c 'import json,sys; d=json.loads(sys.stdin.read()); print(d["ms_per_step"], d["value"], d["roofline"]["launch_ms_median"])')"
echo "exact direct, same NCO flavour: $(python bench.py --main-only --arith cmsis --nco per_channel_grid_wide --steps 50 | python -c 'import json,sys; d=json.loads(sys.stdin.read()); print(d["ms_per_step"], d["value"])')"
