import sys
sys.path.insert(0, "selenite-lite_amd")
import numpy as np
import selenite_rx as sr
from selenite_rx import chain as ch
C_, BS = 65536, 4096
d_in = sr.DeviceBuffer(C_ * BS * 8); d_out = sr.DeviceBuffer(C_ * BS * 4)
steps = (np.arange(C_, dtype=np.uint64) * 0x9E3779B1 % (1 << 26) + 0x00800000).astype(np.uint32)
for name, ar in (("cfg3", ch.ARITH_FMA), ("cfg3", ch.ARITH_CMSIS), ("cfg4", ch.ARITH_CMSIS)):
    rx = sr.Rx(ch.baseline_spec(name, C_, ar, nco_steps=steps).config())
    rx.synth_device(d_in.ptr, 0, C_, 0, BS, ch.SEED)
    rx.time_process(d_in.ptr, d_out.ptr, BS, 50)
    ms = rx.time_process(d_in.ptr, d_out.ptr, BS, 100)
    print("%s arith %d per-channel NCO %-22s %.3f ms %.0f Gs/s" % (name, ar, rx.kernel_name(), ms, C_ * BS / ms / 1e6))
    rx.close()
