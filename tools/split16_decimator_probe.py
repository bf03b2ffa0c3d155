"""tools/split16_decimator_probe.py -- the decimator of the split16 kernel on its own: Q = 0, NCO off, AGC off, USB, so
that audio = the decimated I rail behind the unit delay of the Hilbert pair.  Compared with the f64 FIR of the same
f32 input; the fma and cmsis kernels beside it."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
import rxcommon as rc
import selenite_rx as sr

nch, bs = 64, 8192
rng = np.random.default_rng(5)
n = np.arange(bs)
iq = np.zeros((nch, bs, 2), np.float32)
for c in range(nch):
    f = rng.uniform(0.15, 0.5, 3) if c % 2 == 0 else np.r_[rng.uniform(0.0, 0.08), rng.uniform(0.15, 0.5, 2)]
    x = sum(a * np.cos(2 * np.pi * ff * n + rng.uniform(0, 6)) for a, ff in zip((0.4, 0.2, 0.1), f)) + rng.uniform(-0.05, 0.05, bs)
    iq[c, :, 0] = x.astype(np.float32) * (1.0 if c % 4 < 2 else 1e-3 * (1 + c))
spec0 = rc.ChainSpec(nch, 256, 4, 256, 63, 0, rc.MODE_USB, rc.ARITH_CMSIS, nco=False, agc=False)
h = np.asarray(spec0.dec, np.float64)
delay = (63 - 1) // 2
x64 = iq[:, :, 0].astype(np.float64)
xp = np.concatenate([np.zeros((nch, 255)), x64], axis=1)
win = np.lib.stride_tricks.sliding_window_view(xp, 256, axis=1)[:, 0::4]       # output j uses x[4j-255 .. 4j] (arm_fir_decimate_f32.c: the oldest numTaps of the state)
dec = win @ h[::-1]
exact = np.concatenate([np.zeros((nch, delay)), dec[:, :bs // 4 - delay]], axis=1)
for name, ar in (("cmsis", rc.ARITH_CMSIS), ("fma", rc.ARITH_FMA), ("split16", rc.ARITH_SPLIT16)):
    g = sr.Rx(rc.ChainSpec(nch, 256, 4, 256, 63, 0, rc.MODE_USB, ar, nco=False, agc=False).config())
    y = g.process(iq).astype(np.float64)
    e = np.abs(y - exact).max(axis=1) / np.abs(x64).max(axis=1)
    ratio = np.abs(exact).max(axis=1) / np.abs(x64).max(axis=1)
    print("%-8s %-22s input-referred error: stop-band channels max %.2e  in-band channels max %.2e   (out/in %.3f / %.3f)" % (
        name, g.kernel_name(), e[0::2].max(), e[1::2].max(), np.median(ratio[0::2]), np.median(ratio[1::2])))
    if name == "split16":
        print("   per channel (stop-band):", " ".join("%.1e" % v for v in e[0::2][:16]))
