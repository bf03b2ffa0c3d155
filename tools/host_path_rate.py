"""PCIe-inclusive rate of the host-buffer entry point selenite_rx_process_f32 (copy in, kernel,
copy out, synchronous) -- the number DESIGN.md section 6 quotes next to the resident-data `value`."""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "selenite-lite_amd"))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import rxcommon as rc  # noqa: E402
import selenite_rx as sr  # noqa: E402

C_, BS = 8192, 4096
rx = sr.Rx(rc.baseline_spec("cfg3", C_, rc.ARITH_SPLIT16).config())
iq = sr.synth_iq_host(0, C_, 0, BS, rc.SEED)
q = np.clip(np.trunc(iq * 32768.0), -32768, 32767).astype(np.int16)
for name, fn, arg in (("f32", rx.process, iq), ("q15", rx.process_q15, q)):
    fn(arg)
    t0 = time.perf_counter()
    n = 5
    for _ in range(n):
        fn(arg)
    dt = (time.perf_counter() - t0) / n
    print("host buffers, %s slots: %.1f ms per call of %d x %d samples = %.2f Gsamples/s (%.1f GB/s over PCIe, pageable memory)"
          % (name, dt * 1e3, C_, BS, C_ * BS / dt / 1e9, (arg.nbytes + arg.nbytes // 8) / dt / 1e9))
