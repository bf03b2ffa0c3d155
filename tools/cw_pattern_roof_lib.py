#!/usr/bin/env python3
"""tools/cw_pattern_roof_lib.py [--io q15] -- cfg4 on the product library next to k_cw_roof (selenite_rx_time_pattern_roof_device): the kernel's own
bursts / stores / launch shape / residency with no DSP, and with 256 ... 2048 dependent vector instructions per chunk where the biquad steps are.
One process, one box: median of 60 launches each, three rounds."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "selenite-lite_amd"))
import selenite_rx as sr
from selenite_rx import chain as ch

q15 = "--io" in sys.argv and sys.argv[sys.argv.index("--io") + 1] == "q15"
cfg_name, channels, bs = ch.WORKLOADS["cfg4"]
spec = ch.baseline_spec(cfg_name, channels, sr.ARITH_AUTO)
rx = sr.Rx(spec.config())
d_in = sr.DeviceBuffer(channels * bs * (4 if q15 else 8))
d_out = sr.DeviceBuffer(channels * bs * 4)
if not q15:
    rx.synth_device(d_in.ptr, 0, channels, 0, bs, ch.SEED)
rx.sync()
alg, _ = rx.algorithmic_bytes(bs)
if q15:
    alg -= channels * (4 * bs + 2 * bs)
for _ in range(40):
    for _ in range(16):
        (rx.process_q15_device if q15 else rx.process_device)(d_in.ptr, d_out.ptr, bs)
    rx.sync()
print("# cfg4 %s slots: %d channels x %d samples, %.3f GB algorithmic per launch; ms (median of 60 launches), frac of 8 TB/s" % ("int16" if q15 else "f32", channels, bs, alg / 1e9))
for rnd in range(3):
    med = float(np.median(rx.time_process_each(d_in.ptr, d_out.ptr, bs, 60, q15)))
    print("%s: %.4f ms  %.3f" % (rx.kernel_name(), med, alg / med / 1e6 / 8000))
    for work in (0, 256, 512, 768, 1024, 1280, 1400, 1536, 1792, 2048):
        m = float(np.median(rx.time_pattern_roof(d_in.ptr, d_out.ptr, bs, 60, q15, work)))
        print("  k_cw_roof work %4d: %.4f ms  %.3f   kernel / roof %.3f" % (work, m, alg / m / 1e6 / 8000, med / m))
rx.close()
