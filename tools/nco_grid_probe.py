"""tools/nco_grid_probe.py -- per-channel NCO steps on the fs/256 grid: the register-resident LO (NCO == 4) against the per-sample
NCO (NCO == 1) on the SAME steps, and small arbitrary steps, raw split16 / fma.  Diagnostic."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "selenite-lite_amd"))
import selenite_rx as sr
from selenite_rx import chain as ch
C_, BS = 65536, 4096
d_in = sr.DeviceBuffer(C_ * BS * 8)
d_out = sr.DeviceBuffer(C_ * (BS // 4) * 4)
steps_g = ((np.arange(C_, dtype=np.uint64) * 0x9E3779B1 >> 7) % 256 << 24).astype(np.uint32)
steps_s = (np.arange(C_, dtype=np.uint64) * 0x9E3779B1 % (1 << 26) + 0x00800000).astype(np.uint32)
steps_1 = np.full(C_, 0x01000000, np.uint32); steps_1[0] = 0x02000000          # nearly the headline steps, but per channel
first = True
for label, steps, env in (("grid steps, NCO4", steps_g, {}), ("grid steps, NCO1", steps_g, {"SELENITE_RX_NO_PERIODIC_LO": "1"}),
                          ("small steps, NCO1", steps_s, {}), ("headline-like steps, NCO4", steps_1, {}),
                          ("headline-like steps, NCO1", steps_1, {"SELENITE_RX_NO_PERIODIC_LO": "1"})):
    for a, nm in ((sr.ARITH_SPLIT16, "split16"), (sr.ARITH_FMA, "fma")):
        os.environ.update(env)
        rx = sr.Rx(ch.baseline_spec("cfg3", C_, a, nco_steps=steps).config())
        for k in env: os.environ.pop(k)
        if first:
            rx.synth_device(d_in.ptr, 0, C_, 0, BS, ch.SEED); rx.sync(); first = False
        rx.time_process(d_in.ptr, d_out.ptr, BS, 30)
        ms = np.sort(rx.time_process_each(d_in.ptr, d_out.ptr, BS, 60))
        st = rx.guard_stats()
        print("%-28s %-8s %-34s median %.4f ms  %.1f Gs/s  guarded ch/call %.0f" % (label, nm, rx.kernel_name()[:34], ms[30], C_ * BS / ms[30] / 1e6, st["channel_calls"] / 90.0))
        rx.close()
