#!/bin/bash
# tools/variants/build.sh <name>[:<patch>[:<extra hipcc flags>]] ...  -- A/B libraries of the split16 kernels.
#
# An experiment is a PATCH against selenite-lite_amd/csrc (tools/variants/<name>.patch unless given), never a switch inside
# the product sources: the sources are copied to selenite-lite_amd/variants/src_<name>/, patched there, and
# rx_split16.hip is compiled from the copy (bench kernel only: bench_only_rx_split16.patch goes on first; -DSRX_DIAG) with the flags of the Makefile and linked
# with the objects of the regular build into selenite-lite_amd/variants/lib_<name>.so.  Run with SELENITE_RX_LIB=<that file>.
# `main` (no patch) is the unpatched copy built the same way: the A side of every comparison.
# UNIT=<translation unit without .hip> (default rx_split16) picks the unit the patch touches, e.g. UNIT=rx_cw for k_cw_fused.
set -eu
UNIT=${UNIT:-rx_split16}
UDEF="-DSRX_DIAG"      # A/B libraries read the experiment knobs of csrc/rx_diag.h from the environment; the product build does not
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
cd "$R/selenite-lite_amd"
make -s -j8 libselenite_rx.so
FLAGS=$(make -s print-flags)
mkdir -p variants
OTHERS=$(ls build/*.o | grep -v "$UNIT.hip.o")
for spec in "$@"; do
  IFS=: read -r name patch extra <<<"$spec::"
  [ -n "$patch" ] || patch="$R/tools/variants/$name.patch"
  (
    d=variants/src_$name; rm -rf "$d"; mkdir -p "$d"; cp csrc/*.h csrc/*.hip csrc/*.cpp "$d"/
    # (tools/variants/bench_only_<unit>.patch, where the unit has one: the copy instantiates only the kernel bench.py times -- seconds instead of minutes)
    if [ -f "$R/tools/variants/bench_only_$UNIT.patch" ] && [ -z "${FULL_UNIT:-}" ]; then patch -s -p3 -d "$d" < "$R/tools/variants/bench_only_$UNIT.patch"; fi
    if [ "$name" != main ]; then patch -s -p3 -d "$d" < "$patch"; fi
    # (the sources include ../../include/selenite_rx.h relative to csrc: the copy sits one level deeper)
    sed -i 's#"../../include/#"../../../include/#' "$d"/*.h "$d"/*.hip "$d"/*.cpp
    /opt/rocm/bin/hipcc $FLAGS $UDEF $extra -c "$d/$UNIT.hip" -o variants/$name.o 2>variants/$name.log &&
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o variants/lib_$name.so variants/$name.o $OTHERS 2>>variants/$name.log && echo "built $name" || { echo "FAILED $name (variants/$name.log)"; exit 1; }
  ) &
done
wait
