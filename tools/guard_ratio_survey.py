"""tools/guard_ratio_survey.py -- why the parity guard sits at 0.25 (-12 dB): raw SELENITE_ARITH_SPLIT16 against the CMSIS oracle with
every channel on its own random NCO step (most pass bands empty), per DSP block: relative error d / max|ref| binned by the block's
envelope relative to the largest |mixed sample component| of its pass (history included) -- the quantity the kernel compares."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
import rxcommon as rc
import selenite_rx as sr

nch, bs = 512, 4096
rng = np.random.default_rng(7)
steps = rng.integers(0, 1 << 32, nch, dtype=np.uint64).astype(np.uint32)
L = rc.oracle_lib()
L.orc_nco_lo.argtypes = [rc.C.c_void_p, rc.C.c_uint32, rc.C.c_void_p]
L.orc_nco_lo.restype = None
for agc in (False, True):
    kw = dict(nco_steps=steps, agc=agc)
    g = sr.Rx(rc.baseline_spec("cfg3", nch, rc.ARITH_SPLIT16, **kw).config())
    o = rc.CpuChain(rc.baseline_spec("cfg3", nch, rc.ARITH_CMSIS, **kw), "orc")
    oe = rc.CpuChain(rc.baseline_spec("cfg3", nch, rc.ARITH_CMSIS, nco_steps=steps, agc=False), "orc")     # envelopes before the gain
    ratios, errs = [], []
    for call in range(4):
        iq = rc.synth_iq(0, nch, call * bs, bs)
        hist = o.state()["dec_state"]
        yg, yo, ye = g.process(iq), o.process(iq, 8), oe.process(iq, 8)
        if call == 0:
            continue                                          # start-up transient
        mixed = np.empty((nch, bs, 2), np.float32)
        for c in range(nch):
            ph = ((np.arange(bs, dtype=np.uint64) + call * bs) * int(steps[c])).astype(np.uint32)
            lo = np.empty((bs, 2), np.float32)
            L.orc_nco_lo(ph.ctypes.data, bs, lo.ctypes.data)
            a, b = iq[c, :, 0], iq[c, :, 1]
            mixed[c, :, 0] = a * lo[:, 0] - b * lo[:, 1]
            mixed[c, :, 1] = a * lo[:, 1] + b * lo[:, 0]
        full = np.concatenate([np.stack([hist[:, 0], hist[:, 1]], axis=2), mixed], axis=1)
        pm = np.stack([np.abs(full[:, 1024 * p: 1024 * p + 1279]).reshape(nch, -1).max(axis=1) for p in range(4)], axis=1)
        env = np.abs(ye).reshape(nch, -1, 64).max(axis=2)
        d = np.abs(yg.astype(np.float64) - yo).reshape(nch, -1, 64).max(axis=2)
        m = np.abs(yo).reshape(nch, -1, 64).max(axis=2)
        ratios.append((env / np.repeat(pm, 4, axis=1)).ravel()); errs.append((d / np.maximum(m, 1e-30)).ravel())
    r, e = np.concatenate(ratios), np.concatenate(errs)
    print("AGC %s: %d blocks, %d channels, every channel its own random NCO step" % ("on" if agc else "off", r.size, nch))
    print("  envelope / pass maximum   blocks    worst d/max|ref|   99.9 %%     median")
    edges = [0.0, 0.02, 0.05, 0.1, 0.175, 0.25, 0.35, 0.5, 0.75, 1e9]
    for lo_, hi_ in zip(edges[:-1], edges[1:]):
        k = (r >= lo_) & (r < hi_)
        if k.any():
            print("  [%5.3f, %5s)          %7d    %.2e           %.2e   %.2e" % (lo_, "%.3f" % hi_ if hi_ < 10 else "inf", k.sum(), e[k].max(), np.percentile(e[k], 99.9), np.median(e[k])))
    for thr in (0.1, 0.175, 0.25, 0.35):
        k = r >= thr
        print("  blocks at or above %.3f (not guarded at that ratio): %7d of %d, worst %.2e" % (thr, k.sum(), r.size, e[k].max()))
    g.close()
