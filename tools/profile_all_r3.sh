#!/bin/bash
# tools/profile_all_r3.sh -- every rocprofv3 pass behind profiles/r3/ (run through gpurun, then tools/summarize_profiles.py r3):
# kernel trace + separate FETCH_SIZE / WRITE_SIZE passes for the default (AUTO) arithmetic in both slot formats, the three
# general NCO flavours, the other arithmetic modes and the other BASELINE shapes; SQ counters of the f32 and q15 headline kernels.
set -u
cd $GRAFT_REPO_ROOT
bash tools/profile_run.sh cfg3_auto > /dev/null 2>&1
bash tools/profile_run.sh cfg3_q15_auto --io q15 > /dev/null 2>&1
bash tools/profile_run.sh cfg3_nco1 --nco per_channel --arith split16 > /dev/null 2>&1
bash tools/profile_run.sh cfg3_nco2 --nco shared_table --arith split16 > /dev/null 2>&1
bash tools/profile_run.sh cfg3_nco4 --nco per_channel_grid --arith split16 > /dev/null 2>&1
bash tools/profile_run.sh cfg3_fma --arith fma > /dev/null 2>&1
bash tools/profile_run.sh cfg3_cmsis --arith cmsis > /dev/null 2>&1
bash tools/profile_run.sh cfg4 --workload cfg4 --arith cmsis > /dev/null 2>&1
bash tools/profile_run.sh cfg2_auto --workload cfg2 > /dev/null 2>&1
bash tools/profile_run.sh cfg5_auto --workload cfg5 > /dev/null 2>&1
bash tools/pmc_sq2.sh cfg3_auto > gpurun_out/sq2_cfg3_auto.txt 2>&1
bash tools/pmc_sq2.sh cfg3_q15_auto --io q15 > gpurun_out/sq2_cfg3_q15_auto.txt 2>&1
ls gpurun_out/ | head -80
