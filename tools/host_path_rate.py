"""PCIe-inclusive rate of the host-buffer entry points selenite_rx_process_f32 / _q15 -- the literal drop-in
call of the slot: chunked H2D || kernels || D2H pipeline inside the library (rx_api.hip: process_host).  Measured
with pageable caller memory (staged through the library's pinned buffers) and with page-locked caller memory
(selenite_rx_host_alloc), next to the raw PCIe rate of one big pinned hipMemcpy.  DESIGN.md section 6 quotes this."""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "selenite-lite_amd"))
import selenite_rx as sr  # noqa: E402
from selenite_rx import chain as ch  # noqa: E402

C_, BS = 8192, 4096
rx = sr.Rx(ch.baseline_spec("cfg3", C_, sr.ARITH_SPLIT16).config())
iq = sr.synth_iq_host(0, C_, 0, BS, ch.SEED)
q = np.clip(np.trunc(iq * 32768.0), -32768, 32767).astype(np.int16)

# raw PCIe: one pinned buffer up, one down
pin = sr.pinned_array(iq.shape, np.float32)
pin[...] = iq
dev = sr.DeviceBuffer(iq.nbytes)
L = sr.lib()
for _ in range(2):
    L.selenite_rx_memcpy_h2d(dev.ptr, pin.ctypes.data, pin.nbytes)
t0 = time.perf_counter()
for _ in range(5):
    L.selenite_rx_memcpy_h2d(dev.ptr, pin.ctypes.data, pin.nbytes)
h2d = pin.nbytes * 5 / (time.perf_counter() - t0) / 1e9
t0 = time.perf_counter()
for _ in range(5):
    L.selenite_rx_memcpy_d2h(pin.ctypes.data, dev.ptr, pin.nbytes)
d2h = pin.nbytes * 5 / (time.perf_counter() - t0) / 1e9
print("raw PCIe, one pinned %d MB hipMemcpy: H2D %.1f GB/s, D2H %.1f GB/s" % (pin.nbytes >> 20, h2d, d2h))

for name, fn, arg, odt in (("f32", rx.process, iq, np.float32), ("q15", rx.process_q15, q, np.int16)):
    for mem in ("pageable", "pinned"):
        if mem == "pinned":
            a = sr.pinned_array(arg.shape, arg.dtype)
            a[...] = arg
            o = sr.pinned_array((C_, BS // 4), odt)
        else:
            a, o = arg, np.empty((C_, BS // 4), odt)
        fn(a, o)
        t0 = time.perf_counter()
        n = 8
        for _ in range(n):
            fn(a, o)
        dt = (time.perf_counter() - t0) / n
        print("host buffers, %s slots, %s caller memory: %.2f ms per call of %d x %d samples = %.2f Gsamples/s; input over PCIe at "
              "%.1f GB/s = %.0f %% of the raw H2D rate" % (name, mem, dt * 1e3, C_, BS, C_ * BS / dt / 1e9, a.nbytes / dt / 1e9,
                                                         100.0 * a.nbytes / dt / 1e9 / h2d))
