#!/bin/bash
# tools/mfma_energy.sh -- tools/mfma_energy for each variant with rocm-smi sampled beside it
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
for v in 0 1 2 3; do
  $R/tools/mfma_energy $v > /tmp/me_$v.txt &
  P=$!
  sleep 1.8
  for i in 1 2 3; do rocm-smi --showpower --showclocks 2>/dev/null | grep -E 'sclk clock level|Package Power' | tr '\n' ' ' | sed 's/GPU\[0\]\s*://g; s/\s\+/ /g'; echo; sleep 0.4; done
  wait $P
  cat /tmp/me_$v.txt
done
