"""tools/shape_rate.py -- the fused shapes WITHOUT a split-precision kernel of their own (64-tap decimators, decimation by 8) next to
the headline shape: 65536 channels x 4096 samples, NCO on the fs/256 grid + Hilbert SSB + AGC, every arithmetic; median ms of 100 calls,
Gsamples/s, fraction of the 8 TB/s roof on the shape's own algorithmic bytes, kernel.  Run on the GPU box:  python tools/shape_rate.py"""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "selenite-lite_amd"))
import selenite_rx as sr
from selenite_rx import chain as ch

C_, bs = 65536, 4096
d_in = sr.DeviceBuffer(C_ * bs * 8)
print("# %d channels x %d samples per call, 1x MI355X; median ms per call of 100 | Gsamples/s | frac of 8 TB/s | kernel" % (C_, bs))
for shape in ((256, 4, 63), (64, 4, 63), (256, 8, 63), (128, 8, 63), (64, 8, 63), (64, 2, 63), (256, 2, 63)):
    nd, M, nh = shape
    d_out = sr.DeviceBuffer(C_ * (bs // M) * 4)
    for an, arith in (("auto", sr.ARITH_AUTO), ("fma", sr.ARITH_FMA), ("cmsis", sr.ARITH_CMSIS)):
        rx = sr.Rx(ch.ChainSpec(C_, 256, M, nd, nh, 0, sr.MODE_USB, arith, nco=True, nco_step_all=0x01000000).config())
        rx.synth_device(d_in.ptr, 0, C_, 0, bs, ch.SEED)
        rx.sync()
        t = time.perf_counter()
        while time.perf_counter() - t < 0.3:
            for _ in range(8):
                rx.process_device(d_in.ptr, d_out.ptr, bs)
            rx.sync()
        ms = np.sort(rx.time_process_each(d_in.ptr, d_out.ptr, bs, 100))
        alg = C_ * (8 * bs + 4 * bs // M + 2 * (4 * (2 * (nd - 1) + 2 * (nh - 1)) + 8))
        print("  %-12s %-6s | %8.4f | %7.1f | %.3f | %s" % (shape, an, ms[50], C_ * bs / ms[50] / 1e6, alg / (ms[50] * 1e-3) / 8e12, rx.kernel_name()))
        sys.stdout.flush()
        rx.close()
    d_out.free()
