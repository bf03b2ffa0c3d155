#!/bin/bash
# tools/auto_overhead.sh -- what SELENITE_ARITH_AUTO costs over raw SELENITE_ARITH_SPLIT16 on the headline workload, three interleaved rounds on
# one box, and the kernels of one AUTO run (rocprofv3 --kernel-trace --stats)
set -u
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd "$R"
for r in 1 2 3; do
  echo "auto: $(python3 bench.py --main-only --steps 400 2>/dev/null | python3 -c 'import json,sys; d=json.loads(sys.stdin.read()); print(d["ms_per_step"], d["roofline"]["launch_ms_median"])')"
  echo "split16: $(python3 bench.py --main-only --steps 400 --arith split16 2>/dev/null | python3 -c 'import json,sys; d=json.loads(sys.stdin.read()); print(d["ms_per_step"], d["roofline"]["launch_ms_median"])')"
done
cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kt -o t -- python3 "$R/bench.py" --main-only --spinup-ms 100 --steps 200 --warmup 20 > /dev/null 2>&1
python3 -c "import csv,glob; [print(r['Name'][:40], r['Calls'], r['AverageNs'], r['MinNs']) for f in glob.glob('/tmp/kt/*kernel_stats.csv') for r in csv.DictReader(open(f))]"
