set -u
cd $GRAFT_REPO_ROOT
bash tools/profile_r1.sh cfg3_split16 > /dev/null 2>&1
bash tools/profile_r1.sh cfg3_fma --arith fma > /dev/null 2>&1
bash tools/profile_r1.sh cfg3_cmsis --arith cmsis > /dev/null 2>&1
bash tools/profile_r1.sh cfg4 --workload cfg4 --arith cmsis > /dev/null 2>&1
bash tools/profile_r1.sh cfg2_split16 --workload cfg2 > /dev/null 2>&1
bash tools/profile_r1.sh cfg2_fma --workload cfg2 --arith fma > /dev/null 2>&1
bash tools/profile_r1.sh cfg5_split16 --workload cfg5 > /dev/null 2>&1
bash tools/profile_r1.sh cfg5_fma --workload cfg5 --arith fma > /dev/null 2>&1
ls gpurun_out/
