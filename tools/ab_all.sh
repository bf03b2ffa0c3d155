#!/bin/bash
# tools/ab_all.sh <outdir> -- the product library against selenite-lite_amd/variants/lib_*.so on every bench workload / arithmetic, one box
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
for args in "" "--arith split16" "--io q15" "--arith fma" "--arith cmsis" "--workload cfg2" "--workload cfg5" "--workload cfg4" "--global-gain --arith split16" "--nco per_channel --arith split16"; do
  echo "== bench.py --main-only $args"
  bash $R/tools/ab_bench.sh $1 --steps 200 $args 2>&1 | grep rep2
done
echo "== TX (tools/bench_tx.py)"
for lib in $R/selenite-lite_amd/libselenite_rx.so $R/selenite-lite_amd/variants/lib_*.so; do
  echo "$(basename $lib): $(SELENITE_RX_LIB=$lib python3 $R/tools/bench_tx.py 2>/dev/null | tail -3 | tr '\n' ' ')"
done
