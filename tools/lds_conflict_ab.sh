#!/bin/bash
# tools/lds_conflict_ab.sh <outdir> [bench args] -- LDS counters (one --pmc pass, counters only) of the bench kernel for the
# product library and every A/B library in selenite-lite_amd/variants/: which access of k_ssb_split16 owns the bank conflicts.
set -u
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
O=$R/gpurun_out/$1; shift
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for lib in $R/selenite-lite_amd/libselenite_rx.so $R/selenite-lite_amd/variants/lib_*.so; do
  n=$(basename $lib .so)
  SELENITE_RX_LIB=$lib rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL SQ_WAVES \
      --output-format csv -d $O/$n -o c -- python3 $R/bench.py --main-only --spinup-ms 0 --steps 2 --warmup 1 "$@" > /dev/null 2> $O/$n.err
done
python3 - "$O" <<'PY'
import csv, glob, os, sys, collections
out = open(sys.argv[1] + "/summary.txt", "w")
for d in sorted(glob.glob(sys.argv[1] + "/lib*")):
    if not os.path.isdir(d):
        continue
    agg = collections.defaultdict(list)
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if any(t in r["Kernel_Name"] for t in ("k_ssb", "k_hilb", "k_cw", "k_tx")):
                agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
    m = {k: sum(v) / len(v) for k, v in agg.items()}
    w = m.get("SQ_WAVES", 0) or 1
    line = "%-28s " % os.path.basename(d) + " ".join("%s=%.4g" % (k.replace("SQ_", ""), m[k]) for k in sorted(m))
    print(line); out.write(line + "\n")
PY
