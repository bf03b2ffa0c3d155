"""How long does the GPU take to reach its sustained clocks?  Times consecutive batches of cfg3
calls from a cold (idle, 157 MHz) device.  Diagnostic:  python tools/clock_ramp.py [arith]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "selenite-lite_amd"))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import rxcommon as rc  # noqa: E402
import selenite_rx as sr  # noqa: E402

arith = {"split16": rc.ARITH_SPLIT16, "fma": rc.ARITH_FMA, "cmsis": rc.ARITH_CMSIS}[sys.argv[1] if len(sys.argv) > 1 else "split16"]
C_, BS = 65536, 4096
rx = sr.Rx(rc.baseline_spec("cfg3", C_, arith).config())
d_in, d_out = sr.DeviceBuffer(C_ * BS * 8), sr.DeviceBuffer(C_ * (BS // 4) * 4)
rx.synth_device(d_in.ptr, 0, C_, 0, BS, rc.SEED)
rx.sync()
time.sleep(2.0)                      # let the device fall back to idle clocks
t0 = time.perf_counter()
for batch in (5, 5, 10, 20, 40, 80, 160, 320, 640, 1280, 1280, 1280):
    ms = rx.time_process(d_in.ptr, d_out.ptr, BS, batch)
    print("t=%7.1f ms  %5d calls  %.4f ms/call  %.1f Gs/s" % ((time.perf_counter() - t0) * 1e3, batch, ms, C_ * BS / ms / 1e6))
