#!/bin/bash
# tools/pmc_sq.sh <tag> [bench args] -- SQ counter passes (counters only, no trace domains)
set -u
TAG=$1; shift
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$R/gpurun_out/sq_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
i=0
for set in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS" \
           "SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES" \
           "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_INSTS_WAVE32_LDS GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  rocprofv3 --pmc $set --output-format csv -d $OUT/p$i -o c -- python3 $R/bench.py --main-only --spinup-ms 0 --steps 2 --warmup 1 "$@" > /dev/null 2> $OUT/p$i.err
done
python3 - "$OUT" <<'PY'
import csv, glob, sys, collections
agg=collections.defaultdict(list)
for f in glob.glob(sys.argv[1]+"/p*/*counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        if any(t in r["Kernel_Name"] for t in ("k_ssb", "k_cw", "k_hilb", "k_tx")):
            agg[(r["Kernel_Name"].split("(")[0][-40:], r["Counter_Name"])].append(float(r["Counter_Value"]))
for k in sorted(agg): print(k[0], k[1], sum(agg[k])/len(agg[k]))
PY
