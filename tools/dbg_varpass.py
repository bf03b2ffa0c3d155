"""debug: k_ssb_split16 with 240-output passes vs the oracle, per DSP block"""
import sys, os
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "tests"))
import numpy as np
import rxcommon as rc
from rxcommon import CpuChain, synth_iq
import selenite_rx as sr
np.set_printoptions(linewidth=200, precision=2)
shape, block = (256, 4, 127), 192
if len(sys.argv) > 1: shape, block = tuple(int(x) for x in sys.argv[1].split(",")), int(sys.argv[2])
nd, M, nh = shape
nch, na = 21, block // M
kw = dict(nco=True, nco_step_all=0x00c00000, agc=True)
for arith in (rc.ARITH_SPLIT16, rc.ARITH_AUTO):
    g = sr.Rx(rc.ChainSpec(nch, block, M, nd, nh, 0, rc.MODE_LSB, arith, **kw).config())
    o = CpuChain(rc.ChainSpec(nch, block, M, nd, nh, 0, rc.MODE_LSB, rc.ARITH_CMSIS, **kw), "orc")
    pos = 0
    for bs in [block, 2 * block, 10 * block, 11 * block, block, 50 * block]:
        iq = synth_iq(0, nch, pos, bs); pos += bs
        yg, yo = g.process(iq), o.process(iq)
        d = np.abs(yg.astype(np.float64) - yo).reshape(nch, -1, na).max(axis=2)
        m = np.abs(yo).reshape(nch, -1, na).max(axis=2)
        r = d / np.maximum(m, 1e-30)
        print(arith, bs, "worst", r.max(), "at", np.unravel_index(r.argmax(), r.shape), g.kernel_name(), g.guard_stats())
        sg, so = g.state(), o.state()
        for k in sg:
            if np.asarray(sg[k]).size == 0: continue
            a, b = np.asarray(sg[k], dtype=np.float64).reshape(nch, -1), np.asarray(so[k], dtype=np.float64).reshape(nch, -1)
            dd = np.abs(a - b).max(axis=1) / np.maximum(np.abs(b).max(axis=1), 1e-30)
            print("    state", k, "worst rel", dd.max(), "channel", dd.argmax(), "guarded now:", g.guard_channels()[:nch].tolist() if k == "gain" else "")
        if r.max() > 1e-5:
            c = np.unravel_index(r.argmax(), r.shape)[0]
            print("  channel", c, "rel per block:", r[c])
            print("  block max:", m[c])
            print("  in max:", np.abs(iq[c]).max())
    print(g.guard_stats())
    g.close()
