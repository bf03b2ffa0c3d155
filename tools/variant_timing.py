"""Times cfg3-shaped variants (NCO off / per-channel NCO / AGC off) to see what each stage costs.
Diagnostic only:  python tools/variant_timing.py"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "selenite-lite_amd"))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import rxcommon as rc  # noqa: E402
import selenite_rx as sr  # noqa: E402

C_, BS = 65536, 4096
d_in = sr.DeviceBuffer(C_ * BS * 8)
d_out = sr.DeviceBuffer(C_ * (BS // 4) * 4)
first = True
for label, kw, env in [("shared LO (default)", {}, {}),
                       ("NCO off", {"nco": False}, {}),
                       ("per-channel NCO", {}, {"SELENITE_RX_NO_SHARED_LO": "1"}),
                       ("AGC off", {"agc": False}, {})]:
    for a, name in ((rc.ARITH_SPLIT16, "split16"), (rc.ARITH_FMA, "fma")):
        for k, v in env.items():
            os.environ[k] = v
        spec = rc.ChainSpec(C_, 256, 4, 256, 63, 0, rc.MODE_USB, a, nco=kw.get("nco", True),
                            nco_step_all=0x01000000, agc=kw.get("agc", True))
        rx = sr.Rx(spec.config())
        if first:
            rx.synth_device(d_in.ptr, 0, C_, 0, BS, rc.SEED)
            first = False
        rx.time_process(d_in.ptr, d_out.ptr, BS, 3)
        ms = rx.time_process(d_in.ptr, d_out.ptr, BS, 20)
        print("%-22s %-8s %-28s %.4f ms  %.1f Gs/s" % (label, name, rx.kernel_name(), ms, C_ * BS / ms / 1e6))
        for k in env:
            os.environ.pop(k, None)
