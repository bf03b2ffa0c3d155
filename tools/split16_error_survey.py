"""tools/split16_error_survey.py -- per-DSP-block error of the split16 and fma arithmetic against the CMSIS oracle when every
channel has its own random NCO step (most channels then have little in-band signal: the decimator output is the
stop-band residue of a full-scale input, the worst case for an output-relative error measure)."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
import rxcommon as rc
import selenite_rx as sr

nch = 256
rng = np.random.default_rng(7)
steps = rng.integers(0, 1 << 32, nch, dtype=np.uint64).astype(np.uint32)
ph = rng.integers(0, 1 << 32, nch, dtype=np.uint64).astype(np.uint32)
res = {}
for agc in (True, False):
    o = rc.CpuChain(rc.baseline_spec("cfg3", nch, rc.ARITH_CMSIS, nco_steps=steps, agc=agc), "orc")
    gs = {n: sr.Rx(rc.baseline_spec("cfg3", nch, a, nco_steps=steps, agc=agc).config()) for n, a in (("split16", rc.ARITH_SPLIT16), ("fma", rc.ARITH_FMA))}
    st = o.state(); st["nco_phase"] = ph
    o.L.orc_rx_set_state(o.h, rc.C.byref(rc.state_view(st)))
    for g in gs.values():
        g.set_state(st)
    errs = {n: [] for n in gs}; ratio = []
    for call in range(3):
        iq = rc.synth_iq(0, nch, call * 4096, 4096)
        yo = o.process(iq)
        inmax = np.abs(iq).reshape(nch, -1).max(axis=1)
        for n, g in gs.items():
            yg = g.process(iq)
            d = np.abs(yg.astype(np.float64) - yo).reshape(nch, -1, 64).max(axis=2)
            m = np.abs(yo).reshape(nch, -1, 64).max(axis=2)
            errs[n].append(d / m)
        if not agc:
            ratio.append(np.abs(yo).reshape(nch, -1, 64).max(axis=2) / inmax[:, None])
    for n in gs:
        e = np.concatenate([x.ravel() for x in errs[n]])
        print("agc=%d %-8s per-block rel err: median %.2e  99%% %.2e  max %.2e  blocks > 1e-5: %d of %d" % (agc, n, np.median(e), np.percentile(e, 99), e.max(), (e > 1e-5).sum(), e.size))
    if not agc:
        r = np.concatenate([x.ravel() for x in ratio])
        e = np.concatenate([x.ravel() for x in errs["split16"]])
        print("output block max / input max: median %.3f min %.4f; split16 err x ratio (input-referred): max %.2e" % (np.median(r), r.min(), (e * r).max()))
        ef = np.concatenate([x.ravel() for x in errs["fma"]])
        print("fma err x ratio (input-referred): max %.2e" % (ef * r).max())
