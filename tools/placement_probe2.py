#!/usr/bin/env python3
"""tools/placement_probe2.py <pattern> [workload] -- cfg2 (or another bench workload) timed on buffers allocated in different ways (one process per pattern):
A separate hipMalloc of input and output (what bench.py does); B a 3 GB allocation first; C one allocation holding both; D output first;
E as A after allocating and freeing 6 GB; F input and output each inside its own larger (4 GB) allocation"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "selenite-lite_amd"))
import numpy as np  # noqa: E402
import selenite_rx as sr  # noqa: E402
from selenite_rx import chain as ch  # noqa: E402

pat = sys.argv[1]
wl = sys.argv[2] if len(sys.argv) > 2 else "cfg2"
name, channels, bs = ch.WORKLOADS[wl]
spec = ch.baseline_spec(name, channels, sr.ARITH_AUTO)
nin, nout = channels * bs * 8, channels * (bs // spec.decim) * 4
rx = sr.Rx(spec.config())
keep = []
if pat == "B":
    keep.append(sr.DeviceBuffer(3 << 30))
if pat == "E":
    t = [sr.DeviceBuffer(2 << 30) for _ in range(3)]
    for b in t:
        b.free()
if pat == "C":
    big = sr.DeviceBuffer(nin + nout)
    d_in, d_out = big.ptr, big.ptr + nin
elif pat == "D":
    o = sr.DeviceBuffer(nout); i = sr.DeviceBuffer(nin)
    d_in, d_out = i.ptr, o.ptr
elif pat == "F":
    i = sr.DeviceBuffer(4 << 30); o = sr.DeviceBuffer(4 << 30)
    d_in, d_out = i.ptr, o.ptr
else:
    i = sr.DeviceBuffer(nin); o = sr.DeviceBuffer(nout)
    d_in, d_out = i.ptr, o.ptr
rx.synth_device(d_in, 0, channels, 0, bs, ch.SEED)
for _ in range(300):
    rx.process_device(d_in, d_out, bs)
rx.sync()
ms = np.sort(rx.time_process_each(d_in, d_out, bs, 100, False))
roof = np.sort(rx.time_streaming_roof(d_in, d_out, bs, 60, False))
print("%s %s  in %#x out %#x  kernel %.4f ms  copy %.4f ms" % (pat, wl, d_in, d_out, ms[50], roof[30]), flush=True)
