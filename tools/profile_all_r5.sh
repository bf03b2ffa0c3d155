#!/bin/bash
# tools/profile_all_r5.sh -- every rocprofv3 pass behind profiles/r5/ (run through gpurun, then tools/summarize_profiles.py r5): kernel trace +
# separate FETCH_SIZE / WRITE_SIZE passes (counters only) for the headline (AUTO, f32 and int16 slots), the exact arithmetic, and the other
# BASELINE shapes -- cfg2 at its literal 48 000 samples, cfg4 (f32 and int16 slots), the cfg5 shard --; SQ counter passes of the headline
# kernel, of k_cw_fused and of k_hilb_split16; the AUTO-over-SPLIT16 decomposition; one default bench line.
set -u
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}; cd "$R"
bash tools/profile_run.sh cfg3_auto > /dev/null 2>&1
bash tools/profile_run.sh cfg3_q15_auto --io q15 > /dev/null 2>&1
bash tools/profile_run.sh cfg3_cmsis --arith cmsis > /dev/null 2>&1
bash tools/profile_run.sh cfg4 --workload cfg4 > /dev/null 2>&1
bash tools/profile_run.sh cfg4_q15 --workload cfg4 --io q15 > /dev/null 2>&1
bash tools/profile_run.sh cfg2_auto --workload cfg2 > /dev/null 2>&1
bash tools/profile_run.sh cfg5_auto --workload cfg5 > /dev/null 2>&1
bash tools/profile_run.sh cfg2_q15_auto --workload cfg2 --io q15 > /dev/null 2>&1
bash tools/pmc_sq2.sh cfg3_auto > gpurun_out/sq2_cfg3_auto.txt 2>&1
bash tools/pmc_sq2.sh cfg4 --workload cfg4 > gpurun_out/sq2_cfg4.txt 2>&1
bash tools/pmc_sq2.sh cfg2 --workload cfg2 > gpurun_out/sq2_cfg2.txt 2>&1
bash tools/auto_overhead.sh > gpurun_out/auto_overhead_r5.txt 2>&1
python3 bench.py --steps 20 --warmup 5 > gpurun_out/bench_cfg3_default_r5.json 2> gpurun_out/bench_cfg3_default_r5.err
bash tools/power_clocks.sh gpurun_out/power_clocks_r5.txt "cfg3 AUTO (headline)" --
bash tools/power_clocks.sh gpurun_out/power_clocks_r5.txt "cfg4 (CW)" -- --workload cfg4
ls gpurun_out/ | head -60
