"""tools/fw_geometry_rate.py -- the firmware's own DSP-block geometry at scale: DSP blocks of 96 I/Q frames (Core/Inc/dsp_if.h:69-73,
24 audio samples behind the /4 decimator), cfg3 chain, 65536 channels, calls of 10 .. 80 slots.  Passes of 240 audio samples:
k_ssb_split16 (AUTO / split16) against the bit-exact k_ssb_fused (cmsis) and against the 256-frame DSP block of the headline.
Run on the GPU box:  python tools/fw_geometry_rate.py > profiles/r3/fw_geometry.txt"""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "selenite-lite_amd"))
import selenite_rx as sr
from selenite_rx import chain as ch

def measure(C_, bs, block, arith):
    spec = ch.ChainSpec(C_, block, 4, 256, 63, 0, sr.MODE_USB, arith, nco=True, nco_step_all=0x01000000)
    rx = sr.Rx(spec.config())
    d_in, d_out = sr.DeviceBuffer(C_ * bs * 8), sr.DeviceBuffer(C_ * (bs // 4) * 4)
    rx.synth_device(d_in.ptr, 0, C_, 0, bs, ch.SEED)
    rx.sync()
    t = time.perf_counter()
    while (time.perf_counter() - t) < 0.3:
        for _ in range(8):
            rx.process_device(d_in.ptr, d_out.ptr, bs)
        rx.sync()
    ms = np.sort(rx.time_process_each(d_in.ptr, d_out.ptr, bs, 100))
    g = rx.guard_stats()
    name = rx.kernel_name()
    rx.close(); d_in.free(); d_out.free()
    return float(ms[50]), float(ms[0]), name, g

print("# cfg3 chain, 65536 channels, 1x MI355X; median / min ms per call of 100 (own spin-up), Gsamples/s at the median")
print("# %-6s %-8s %-8s | %8s %8s %8s | %s" % ("block", "samples", "arith", "median", "min", "Gs/s", "kernel"))
C_ = 65536
for block, bs in ((96, 960), (96, 1920), (96, 7680), (96, 8160), (192, 7680), (384, 7680), (256, 8192)):
    for arith, nm in ((sr.ARITH_AUTO, "auto"), (sr.ARITH_SPLIT16, "split16"), (sr.ARITH_CMSIS, "cmsis")):
        med, mn, name, g = measure(C_, bs, block, arith)
        print("  %-6d %-8d %-8s | %8.4f %8.4f %8.1f | %s  guard %s" % (block, bs, nm, med, mn, C_ * bs / med / 1e6, name, g if arith == sr.ARITH_AUTO else ""))
        sys.stdout.flush()
