#!/bin/bash
# tools/power_clocks.sh <outfile> <label> -- [bench args]: sample rocm-smi (package power, sclk) while bench.py runs a long
# timed region; prints the median of the samples taken while the kernel was running.
OUT=$1; LABEL=$2; shift 2; [ "$1" == "--" ] && shift
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
python3 $R/bench.py --main-only --steps 12000 "$@" > /tmp/pc_bench.json 2>/tmp/pc_bench.err &
BP=$!
sleep 3.5
S=""
for i in 1 2 3 4 5 6; do
  S="$S$(rocm-smi --showpower --showclocks 2>/dev/null | grep -E 'sclk clock level|Package Power|fclk clock|mclk clock' | tr '\n' ' ')\n"
  sleep 0.4
done
wait $BP
echo "== $LABEL: $(python3 -c "import json; d=json.load(open('/tmp/pc_bench.json')); print(d['ms_per_step'], 'ms/step', d['value'], 'Ms/s')")" >> $OUT
echo -e "$S" | sed 's/GPU\[0\]\s*://g; s/\s\+/ /g' >> $OUT
