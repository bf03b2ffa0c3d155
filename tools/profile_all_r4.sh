#!/bin/bash
# tools/profile_all_r4.sh -- every rocprofv3 pass behind profiles/r4/ (run through gpurun, then tools/summarize_profiles.py r4):
# kernel trace + separate FETCH_SIZE / WRITE_SIZE passes for the default (AUTO) arithmetic in both slot formats, the stop-band-heavy
# AUTO workload (split16 on the kept channels, k_hist_exact, the rerun pass of the bit-exact kernel over the held ones), the other
# arithmetic modes, the NCO flavours and the other BASELINE shapes (cfg2 at its literal 48 000 samples); SQ counters of the headline
# kernel, of the bit-exact kernel and of k_cw_fused.
set -u
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}; cd "$R"
bash tools/profile_run.sh cfg3_auto > /dev/null 2>&1
bash tools/profile_run.sh cfg3_q15_auto --io q15 > /dev/null 2>&1
bash tools/profile_run.sh cfg3_auto_stopband --nco per_channel_grid_wide > /dev/null 2>&1
bash tools/profile_run.sh cfg3_nco1 --nco per_channel --arith split16 > /dev/null 2>&1
bash tools/profile_run.sh cfg3_nco2 --nco shared_table --arith split16 > /dev/null 2>&1
bash tools/profile_run.sh cfg3_nco4 --nco per_channel_grid --arith split16 > /dev/null 2>&1
bash tools/profile_run.sh cfg3_fma --arith fma > /dev/null 2>&1
bash tools/profile_run.sh cfg3_cmsis --arith cmsis > /dev/null 2>&1
bash tools/profile_run.sh cfg4 --workload cfg4 --arith cmsis > /dev/null 2>&1
bash tools/profile_run.sh cfg2_auto --workload cfg2 > /dev/null 2>&1
bash tools/profile_run.sh cfg2_192_auto --workload cfg2_192 > /dev/null 2>&1
bash tools/profile_run.sh cfg5_auto --workload cfg5 > /dev/null 2>&1
bash tools/pmc_sq2.sh cfg3_auto > gpurun_out/sq2_cfg3_auto.txt 2>&1
bash tools/pmc_sq2.sh cfg3_cmsis --arith cmsis > gpurun_out/sq2_cfg3_cmsis.txt 2>&1
bash tools/pmc_sq2.sh cfg4 --workload cfg4 --arith cmsis > gpurun_out/sq2_cfg4.txt 2>&1
ls gpurun_out/ | head -80
