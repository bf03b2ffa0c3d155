"""tools/mode_rate.py -- the cfg3 chain (NCO + 256-tap /4 + 63-tap pair + AGC, 65536 channels x 4096 samples per call) in every
demodulator mode DSP_Set_Mode can select (Core/Inc/rxtx_if.h:33-43) and every arithmetic: median ms per call of 100 (own spin-up),
Gsamples/s, kernel, and what the parity guard did in the last call.  FM runs twice: on the synthetic bench signal (a CW tone plus
noise: narrow deviation -- SELENITE_ARITH_AUTO guards it and the bit-exact kernel holds the channels) and on an FM carrier
(`FM*`: 0.7 amplitude, peak deviation 0.24 cycles per audio sample).  Run on the GPU box:  python tools/mode_rate.py > profiles/r4/mode_rate.txt"""
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
d_fm = sr.DeviceBuffer(C_ * bs * 8)
# an FM carrier per channel at fs/256 (the chain's LO): 256 channels with their own audio phase, repeated over the batch; the audio tone
# has 4096 / 64 whole periods per call, so the same buffer is a continuous signal call after call
t = np.arange(bs, dtype=np.float64)
ph0 = np.random.default_rng(3).uniform(0, 2 * np.pi, (256, 1))
z = 0.7 * np.exp(1j * (2 * np.pi * t[None, :] / 256.0 + (0.06 / (1.0 / 64)) * np.sin(2 * np.pi * t[None, :] / 64.0 + ph0)))
fm_host = np.ascontiguousarray(np.stack([z.real, z.imag], axis=-1).astype(np.float32))
for c0 in range(0, C_, 256):
    sr.lib().selenite_rx_memcpy_h2d(d_fm.ptr + c0 * bs * 8, fm_host.ctypes.data, fm_host.nbytes)
MODES = MODES + (("FM*", sr.MODE_FM),)
print("# cfg3 chain, %d channels x %d samples per call, 1x MI355X; median ms per call of 100 | Gsamples/s | kernel | rerun fraction of the last call" % (C_, bs))
for an, arith in ARITH:
    rx = sr.Rx(ch.ChainSpec(C_, 256, 4, 256, 63, 0, sr.MODE_USB, arith, nco=True, nco_step_all=0x01000000).config())
    rx.synth_device(d_in.ptr, 0, C_, 0, bs, ch.SEED)
    rx.sync()
    for mn, mode in MODES:
        assert rx.set_mode(mode) == 0
        src = d_fm.ptr if mn == "FM*" else d_in.ptr
        t = time.perf_counter()
        while time.perf_counter() - t < 0.3:
            for _ in range(8):
                rx.process_device(src, d_out.ptr, bs)
            rx.sync()
        ms = np.sort(rx.time_process_each(src, d_out.ptr, bs, 100))
        rx.guard_clear()
        rx.process_device(src, d_out.ptr, bs)
        print("  %-8s %-4s | %8.4f | %7.1f | %s | %.3f" % (an, mn, ms[50], C_ * bs / ms[50] / 1e6, rx.kernel_name(), rx.guard_stats()["rerun_channel_calls"] / C_))
        sys.stdout.flush()
    rx.close()
