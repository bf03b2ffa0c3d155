#!/bin/bash
# tools/build_variants_q15.sh name:"-DFLAG .." ...  -- A/B libraries of the int16-slot split16 kernels (csrc/rx_split16_q15.hip recompiled
# alone with the timing knobs SRX_X_* of rx_split16_kernels.h, linked with the objects of the regular build into
# selenite-lite_amd/variants/lib_q15_<name>.so); run with SELENITE_RX_LIB=<that file>.  Results of such builds are WRONG on purpose.
set -e
cd "$(dirname "$0")/../selenite-lite_amd"
make -s -j8 libselenite_rx.so
mkdir -p variants
OTHERS=$(ls build/*.o | grep -v 'rx_split16_q15.hip.o')
for spec in "$@"; do
  name=${spec%%:*}; flags=${spec#*:}
  ( /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC -ffp-contract=off -Wall -Wno-unused-function --offload-arch=gfx950 --offload-compress \
      $flags -c csrc/rx_split16_q15.hip -o variants/q15_$name.o 2>/dev/null && \
    /opt/rocm/bin/hipcc --offload-arch=gfx950 --offload-compress -shared -fPIC -o variants/lib_q15_$name.so variants/q15_$name.o $OTHERS 2>/dev/null && echo built $name ) &
done
wait
