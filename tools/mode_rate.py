"""tools/mode_rate.py -- the cfg3 chain (NCO + 256-tap /4 + 63-tap pair + AGC, 65536 channels x 4096 samples per call) in every
demodulator mode DSP_Set_Mode can select (Core/Inc/rxtx_if.h:33-43) and every arithmetic: median ms per call of 100 (own spin-up),
Gsamples/s, kernel.  Run on the GPU box:  python tools/mode_rate.py > profiles/r3/mode_rate.txt"""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "selenite-lite_amd"))
import selenite_rx as sr
from selenite_rx import chain as ch

C_, bs = 65536, 4096
MODES = (("USB", sr.MODE_USB), ("LSB", sr.MODE_LSB), ("AM", sr.MODE_AM), ("FM", sr.MODE_FM), ("DIG", sr.MODE_DIG), ("PKT", sr.MODE_PKT))
ARITH = (("auto", sr.ARITH_AUTO), ("split16", sr.ARITH_SPLIT16), ("fma", sr.ARITH_FMA), ("cmsis", sr.ARITH_CMSIS))
d_in, d_out = sr.DeviceBuffer(C_ * bs * 8), sr.DeviceBuffer(C_ * (bs // 4) * 4)
print("# cfg3 chain, %d channels x %d samples per call, 1x MI355X; median ms per call of 100 | Gsamples/s | kernel" % (C_, bs))
for an, arith in ARITH:
    rx = sr.Rx(ch.ChainSpec(C_, 256, 4, 256, 63, 0, sr.MODE_USB, arith, nco=True, nco_step_all=0x01000000).config())
    rx.synth_device(d_in.ptr, 0, C_, 0, bs, ch.SEED)
    rx.sync()
    for mn, mode in MODES:
        assert rx.set_mode(mode) == 0
        t = time.perf_counter()
        while time.perf_counter() - t < 0.3:
            for _ in range(8):
                rx.process_device(d_in.ptr, d_out.ptr, bs)
            rx.sync()
        ms = np.sort(rx.time_process_each(d_in.ptr, d_out.ptr, bs, 100))
        print("  %-8s %-4s | %8.4f | %7.1f | %s" % (an, mn, ms[50], C_ * bs / ms[50] / 1e6, rx.kernel_name()))
        sys.stdout.flush()
    rx.close()
