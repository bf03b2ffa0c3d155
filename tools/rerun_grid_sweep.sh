#!/bin/bash
# tools/rerun_grid_sweep.sh -- workgroups of AUTO's rerun pass (SELENITE_RX_RERUN_GRID): the stop-band-heavy workload (80 % of the
# channels on the list) and the headline (empty list), ms per step / median launch, next to the all-bit-exact call
one() { python3 bench.py --main-only "$@" 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['roofline']['launch_ms_median'])"; }
for g in 2048 3072 4096 6144 8192 16384; do echo "stop-band, grid $g: $(SELENITE_RX_RERUN_GRID=$g one --nco per_channel_grid_wide)"; done
echo "stop-band, default (2048, or 16384 when the last list held more than an eighth of the channels): $(one --nco per_channel_grid_wide)"
echo "stop-band, all bit-exact (--arith cmsis): $(one --nco per_channel_grid_wide --arith cmsis)"
for r in 1 2 3; do for g in 2048 6144 8192 16384; do echo "headline, grid $g: $(SELENITE_RX_RERUN_GRID=$g one)"; done; echo "headline, default: $(one)"; done
