#!/bin/bash
# tools/build_exact_variants.sh name:"-DFLAG=.. -DFLAG=.." ...  -- A/B libraries of the bit-exact kernel k_ssb_fused<0,..,256,4,63>.
# Each variant recompiles csrc/rx_fused_exact.hip alone (cfg3 shape, f32 slots only) and links it with the objects of the
# regular build into selenite-lite_amd/variants/lib_<name>.so; run with SELENITE_RX_LIB=<that file>.
set -e
cd "$(dirname "$0")/../selenite-lite_amd"
make -s -j8 libselenite_rx.so
mkdir -p variants
OTHERS=$(ls build/*.o | grep -v 'rx_fused_exact.hip.o')
for spec in "$@"; do
  name=${spec%%:*}; flags=${spec#*:}
  ( /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC -ffp-contract=off -Wall -Wno-unused-function --offload-arch=gfx950 \
      -DSRX_EXACT_BENCH_ONLY $flags -c csrc/rx_fused_exact.hip -o variants/$name.o 2>/dev/null && \
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o variants/lib_$name.so variants/$name.o $OTHERS 2>/dev/null && echo built $name ) &
done
wait
