"""tools/short_call_probe.py -- calls too short for the matrix kernel in SELENITE_ARITH_AUTO (they run on the bit-exact kernel from a
repaired history): median us per call next to the plain bit-exact and raw split16 arithmetic, and the per-channel AUTO words afterwards."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.environ.get("PROBE_PKG") or os.path.join(ROOT, "selenite-lite_amd"))      # PROBE_PKG: another build (package + library) for an A/B
import selenite_rx as sr
from selenite_rx import chain as ch
C_ = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
for bs, block in ((96, 96), (256, 256), (512, 256), (1024, 256), (2048, 256), (4096, 256)):
    for an, arith in (("auto", sr.ARITH_AUTO), ("cmsis", sr.ARITH_CMSIS), ("split16", sr.ARITH_SPLIT16)):
        rx = sr.Rx(ch.ChainSpec(C_, block, 4, 256, 63, 0, sr.MODE_USB, arith, nco=True, nco_step_all=0x01000000).config())
        d_in, d_out = sr.DeviceBuffer(C_ * bs * 8), sr.DeviceBuffer(C_ * (bs // 4) * 4)
        rx.synth_device(d_in.ptr, 0, C_, 0, bs, ch.SEED)
        rx.sync()
        t = time.perf_counter()
        while time.perf_counter() - t < 0.15:
            for _ in range(32):
                rx.process_device(d_in.ptr, d_out.ptr, bs)
            rx.sync()
        ms = np.sort(rx.time_process_each(d_in.ptr, d_out.ptr, bs, 200))
        w = rx.auto_words() if arith == sr.ARITH_AUTO and hasattr(rx, "auto_words") else None
        hist = "" if w is None else " words&0xff: " + str({int(k): int(v) for k, v in zip(*np.unique(w & 0xff, return_counts=True))})
        print("%6d x %5d %-8s | %8.1f us | %s%s" % (C_, bs, an, ms[100] * 1e3, rx.kernel_name(), hist))
        sys.stdout.flush()
        rx.close(); d_in.free(); d_out.free()
