"""tools/split16_chain_probe.py -- input-referred error of the cfg3 chain (AGC off) in the three arithmetic modes against an
f64 evaluation of the same filters on the same f32 samples; channels whose energy sits in the decimator's stop band and
channels with an in-band tone.  NCO off: complex input straight into the decimator.  (tools/split16_decimator_probe.py
does the I rail alone.)"""
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
    z = sum(a * np.exp(2j * np.pi * ff * n + 1j * rng.uniform(0, 6)) for a, ff in zip((0.4, 0.2, 0.1), f))
    z = z + rng.uniform(-0.05, 0.05, bs) + 1j * rng.uniform(-0.05, 0.05, bs)
    iq[c, :, 0] = z.real; iq[c, :, 1] = z.imag
NCO = sys.argv[1] if len(sys.argv) > 1 else "off"       # off | per_channel | table | periodic
kw = dict(nco=False)
if NCO != "off":
    steps = rng.integers(0, 1 << 32, nch, dtype=np.uint64).astype(np.uint32)
    if NCO == "table":
        steps[:] = 0x01000100
    if NCO == "periodic":
        steps[:] = 0x01000000
    kw = dict(nco=True, nco_steps=steps) if NCO == "per_channel" else dict(nco=True, nco_step_all=int(steps[0]))
    # the mixed samples exactly as the chain forms them (f32, CMSIS operation order), then f64 filters on those
    L = rc.oracle_lib()
    L.orc_nco_lo.argtypes = [rc.C.c_void_p, rc.C.c_uint32, rc.C.c_void_p]
    L.orc_nco_lo.restype = None
    mixed = np.empty_like(iq)
    for c in range(nch):
        ph = (np.arange(bs, dtype=np.uint64) * int(steps[c])).astype(np.uint32)
        lo = np.empty((bs, 2), np.float32)
        L.orc_nco_lo(ph.ctypes.data, bs, lo.ctypes.data)
        a, b, cc, d = iq[c, :, 0], iq[c, :, 1], lo[:, 0], lo[:, 1]
        mixed[c, :, 0] = (a * cc).astype(np.float32) - (b * d).astype(np.float32)
        mixed[c, :, 1] = (a * d).astype(np.float32) + (b * cc).astype(np.float32)
    src = mixed
else:
    src = iq
spec0 = rc.ChainSpec(nch, 256, 4, 256, 63, 0, rc.MODE_USB, rc.ARITH_CMSIS, agc=False, **kw)
h = np.asarray(spec0.dec, np.float64); hb = np.asarray(spec0.hilb, np.float64); dl = np.asarray(spec0.delay, np.float64)


def dec(x):
    xp = np.concatenate([np.zeros((nch, 255)), x.astype(np.float64)], axis=1)
    return np.lib.stride_tricks.sliding_window_view(xp, 256, axis=1)[:, 0::4] @ h


def fir(c, d):
    dp = np.concatenate([np.zeros((nch, len(c) - 1)), d], axis=1)
    return np.lib.stride_tricks.sliding_window_view(dp, len(c), axis=1) @ c


di, dq = dec(src[:, :, 0]), dec(src[:, :, 1])
exact = fir(dl, di) - fir(hb, dq)
im = np.abs(iq).reshape(nch, -1).max(axis=1)
for name, ar in (("cmsis", rc.ARITH_CMSIS), ("fma", rc.ARITH_FMA), ("split16", rc.ARITH_SPLIT16)):
    g = sr.Rx(rc.ChainSpec(nch, 256, 4, 256, 63, 0, rc.MODE_USB, ar, agc=False, **kw).config())
    y = g.process(iq).astype(np.float64)
    e = np.abs(y - exact).max(axis=1) / im
    ratio = np.abs(exact).max(axis=1) / im
    print("nco=%s" % NCO, g.nco_path())
    print("%-8s %-24s input-referred error: stop-band channels max %.2e  in-band channels max %.2e   (out/in %.3f / %.3f)" % (
        name, g.kernel_name(), e[0::2].max(), e[1::2].max(), np.median(ratio[0::2]), np.median(ratio[1::2])))
    # where in the call is the worst error of the worst stop-band channel
    c = 2 * int(np.argmax(e[0::2]))
    d = np.abs(y[c] - exact[c]).reshape(-1, 64).max(axis=1) / im[c]
    print("      worst stop-band channel %d, per DSP block:" % c, " ".join("%.0e" % v for v in d))
