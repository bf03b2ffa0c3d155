#!/bin/bash
# tools/profile_all_r6.sh -- every rocprofv3 pass behind profiles/r6/ (run through gpurun, then tools/summarize_profiles.py r6): kernel trace +
# separate FETCH_SIZE / WRITE_SIZE passes (counters only) for the headline (AUTO, f32 and int16 slots), the exact arithmetic, and the other
# BASELINE shapes -- cfg2 at its literal 48 000 samples, cfg4 (f32 and int16 slots: the round-6 k_cw_fused), the cfg5 shard --; SQ counter passes of
# the headline kernel, of k_cw_fused (both slot formats) and of the bit-exact k_ssb_fused; package power and clocks under the
# headline, cfg4 and the bit-exact kernel; the pattern roof of cfg4 from the library next to the kernel; one default bench line.
set -u
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}; cd "$R"
rm -rf gpurun_out/prof_* gpurun_out/sq2_*
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
bash tools/pmc_sq2.sh cfg4_q15 --workload cfg4 --io q15 > gpurun_out/sq2_cfg4_q15.txt 2>&1
bash tools/pmc_sq2.sh cfg3_cmsis --arith cmsis > gpurun_out/sq2_cfg3_cmsis.txt 2>&1
python3 tools/cw_pattern_roof_lib.py > gpurun_out/cw_pattern_roof_lib_r6.txt 2>&1
python3 tools/cw_pattern_roof_lib.py --io q15 > gpurun_out/cw_pattern_roof_lib_q15_r6.txt 2>&1
python3 bench.py --steps 20 --warmup 5 > gpurun_out/bench_cfg3_default_r6.json 2> gpurun_out/bench_cfg3_default_r6.err
rm -f gpurun_out/power_clocks_r6.txt
bash tools/power_clocks.sh gpurun_out/power_clocks_r6.txt "cfg3 AUTO (headline)" --
bash tools/power_clocks.sh gpurun_out/power_clocks_r6.txt "cfg4 (CW)" -- --workload cfg4
bash tools/power_clocks.sh gpurun_out/power_clocks_r6.txt "cfg4 (CW), int16 slots" -- --workload cfg4 --io q15
bash tools/power_clocks.sh gpurun_out/power_clocks_r6.txt "cfg3 bit-exact (k_ssb_fused)" -- --arith cmsis
ls gpurun_out/ | head -60
