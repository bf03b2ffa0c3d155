import os, sys
import numpy as np
ROOT = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "selenite-lite_amd"))
import rxcommon as rc
import selenite_rx as sr
from rxcommon import CpuChain, synth_iq
def per_block(yg, yo, na):
    nch = yo.shape[0]
    d = np.abs(yg.astype(np.float64) - yo).reshape(nch, -1, na).max(axis=2)
    m = np.abs(yo).reshape(nch, -1, na).max(axis=2)
    return d, m
for (nd, nh, block) in ((256, 63, 256), (128, 63, 512), (64, 63, 256), (200, 127, 1024)):
    for arith in (rc.ARITH_SPLIT16, rc.ARITH_AUTO):
        nch = 32
        kw = dict(nco=True, nco_step_all=0x01000000)
        g = sr.Rx(rc.ChainSpec(nch, block, 8, nd, nh, 0, rc.MODE_USB, arith, **kw).config())
        o = CpuChain(rc.ChainSpec(nch, block, 8, nd, nh, 0, rc.MODE_USB, rc.ARITH_CMSIS, **kw), "orc")
        pos = 0; worst = 0.0
        for bs in (4096, 8192, 3072, 4096 + 2048):
            iq = synth_iq(0, nch, pos, bs); pos += bs
            yg, yo = g.process(iq), o.process(iq)
            d, m = per_block(yg, yo, block // 8)
            worst = max(worst, float((d / np.maximum(m, 1e-30)).max()))
        sg, so = g.state(), o.state()
        st_ok = all((sg[k].view(np.uint32) == so[k].view(np.uint32)).all() for k in ("dec_state", "nco_phase"))
        fs = float(np.abs(sg["fir_state"] - so["fir_state"]).max() / np.abs(so["fir_state"]).max())
        print("nd %3d nh %3d block %4d arith %d | %-60s | worst rel/block %.3g | dec_state+phase exact %s | fir_state rel %.2g | guard %s" % (nd, nh, block, arith, g.kernel_name(), worst, st_ok, fs, g.guard_stats()))
        g.close(); o.close()
