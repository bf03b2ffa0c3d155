#!/bin/bash
# tools/pmc_sq2.sh <tag> [bench args] -- SQ / SQC counter passes for the kernel bench.py times (counters only, no
# trace domains).  SELENITE_RX_LIB selects an A/B library.  Prints per-wave and per-CU figures.
set -u
TAG=$1; shift
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$R/gpurun_out/sq2_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
i=0
for set in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS" \
           "SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_VALU_MFMA_COEXEC_CYCLES" \
           "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_LDS_ADDR_CONFLICT SQ_LDS_CMD_FIFO_FULL SQ_LDS_DATA_FIFO_FULL SQ_LDS_UNALIGNED_STALL GRBM_GUI_ACTIVE" \
           "SQ_IFETCH SQ_IFETCH_LEVEL SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_BUSY_CYCLES SQC_ICACHE_INPUT_VALID_READYB SQC_TC_INST_REQ" \
           "SQ_INST_LEVEL_LDS SQ_INST_LEVEL_VMEM SQ_LEVEL_WAVES SQ_BUSY_CU_CYCLES SQ_THREAD_CYCLES_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_VMEM" \
           "SQ_INST_CYCLES_SALU SQ_INST_CYCLES_VMEM_RD SQ_INST_CYCLES_VMEM_WR SQ_INSTS_VALU_CVT SQ_CYCLES SQ_INSTS SQ_INSTS_BRANCH SQ_INSTS_SMEM"; do
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
out = open(sys.argv[1] + "/summary.txt", "w")
for k in sorted(agg):
    line = "%s %s %.6g" % (k[0], k[1], sum(agg[k])/len(agg[k]))
    print(line); out.write(line + "\n")
PY
