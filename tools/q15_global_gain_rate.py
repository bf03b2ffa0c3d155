import os, sys
import numpy as np
sys.path.insert(0, "selenite-lite_amd"); sys.path.insert(0, "tests")
import selenite_rx as sr
from selenite_rx import chain as ch
C_, BS = 65536, 4096
for arith, nm in ((sr.ARITH_SPLIT16, "split16"), (sr.ARITH_AUTO, "auto"), (sr.ARITH_CMSIS, "cmsis")):
    rx = sr.Rx(ch.baseline_spec("cfg3", C_, arith, agc_global=True).config())
    d_f = sr.DeviceBuffer(C_ * BS * 8); d_in = sr.DeviceBuffer(C_ * BS * 4); d_out = sr.DeviceBuffer(C_ * (BS // 4) * 2)
    rx.synth_device(d_f.ptr, 0, C_, 0, BS, ch.SEED); rx.sync()
    chunk = 2048
    for c0 in range(0, C_, chunk):
        f = np.empty((chunk, BS, 2), np.float32)
        sr.lib().selenite_rx_memcpy_d2h(f.ctypes.data, d_f.ptr + c0 * BS * 8, f.nbytes)
        q = np.clip(np.trunc(f * np.float32(32768.0)), -32768, 32767).astype(np.int16)
        sr.lib().selenite_rx_memcpy_h2d(d_in.ptr + c0 * BS * 4, q.ctypes.data, q.nbytes)
    d_f.free()
    rx.time_process_q15(d_in.ptr, d_out.ptr, BS, 30)
    ms = np.sort(rx.time_process_each(d_in.ptr, d_out.ptr, BS, 60, True))
    print("int16 slots + global gain, %-8s %-40s median %.4f ms  %.1f Gs/s" % (nm, rx.kernel_name()[:40], ms[30], C_ * BS / ms[30] / 1e6))
    rx.close(); d_in.free(); d_out.free()
