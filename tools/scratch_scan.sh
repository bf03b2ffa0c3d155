#!/bin/bash
# tools/scratch_scan.sh -- kernels of the built objects that use scratch (private segment > 0): a kernel with scratch pays ~12 us per
# dispatch on this stack (measured: k_ssb_split16 with 12 bytes of scratch, 0.5385 -> 0.552 ms per launch), so the hot ones must have none
cd "$(dirname "$0")/../selenite-lite_amd"
for src in csrc/rx_split16.hip csrc/rx_split16_q15.hip csrc/rx_fused.hip csrc/rx_fused_exact.hip csrc/rx_cw.hip csrc/rx_generic.hip csrc/tx_fused.hip csrc/tx.hip csrc/ring.hip csrc/rx_synth.hip; do
  /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC -ffp-contract=off --offload-arch=gfx950 -S --cuda-device-only -o /tmp/scan.s $src 2>/dev/null
  python3 - "$src" <<'PY'
import re, sys
t = open('/tmp/scan.s').read()
n = 0
for m in re.finditer(r'\.amdhsa_kernel (\S+)(.*?)\.end_amdhsa_kernel', t, re.S):
    ps = re.search(r'\.amdhsa_private_segment_fixed_size (\d+)', m.group(2))
    n += 1
    if ps and int(ps.group(1)) > 0:
        print("  scratch %4s B  %s" % (ps.group(1), m.group(1)[:150]))
print("%s: %d kernels" % (sys.argv[1], n))
PY
done
