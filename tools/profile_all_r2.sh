#!/bin/bash
# tools/profile_all_r2.sh -- every rocprofv3 pass behind profiles/r2/ (run through gpurun, then tools/summarize_profiles.py r2)
set -u
cd $GRAFT_REPO_ROOT
bash tools/profile_run.sh cfg3_split16 > /dev/null 2>&1
bash tools/profile_run.sh cfg3_fma --arith fma > /dev/null 2>&1
bash tools/profile_run.sh cfg3_cmsis --arith cmsis > /dev/null 2>&1
bash tools/profile_run.sh cfg4 --workload cfg4 --arith cmsis > /dev/null 2>&1
bash tools/profile_run.sh cfg2_split16 --workload cfg2 > /dev/null 2>&1
bash tools/profile_run.sh cfg5_split16 --workload cfg5 > /dev/null 2>&1
bash tools/pmc_sq2.sh cfg3_split16 > gpurun_out/sq2_cfg3_split16.txt 2>&1
python3 bench.py > gpurun_out/bench_default_r2.json 2> gpurun_out/bench_default_r2.err
ls gpurun_out/
