import os, sys
import numpy as np
ROOT = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "selenite-lite_amd"))
import rxcommon as rc
import selenite_rx as sr
from rxcommon import CpuChain, synth_iq
for block in (256, 128, 512, 2048):
  for arith in (rc.ARITH_AUTO, rc.ARITH_SPLIT16, rc.ARITH_CMSIS):
    nch = 40
    kw = dict(nco=True, nco_step_all=0x01000000, agc_global=True)
    g = sr.Rx(rc.ChainSpec(nch, block, 8, 256, 63, 0, rc.MODE_USB, arith, **kw).config())
    o = CpuChain(rc.ChainSpec(nch, block, 8, 256, 63, 0, rc.MODE_USB, rc.ARITH_CMSIS, **kw), "orc")
    worst = 0
    for call in range(3):
        iq = synth_iq(0, nch, 4096 * call, 4096)
        yo, _ = o.process_env(iq)
        yg = g.process(iq)
        na = block // 8
        d = np.abs(yg.astype(np.float64) - yo).reshape(nch, -1, na).max(axis=2); m = np.abs(yo).reshape(nch, -1, na).max(axis=2)
        worst = max(worst, float((d / np.maximum(m, 1e-30)).max()))
    print(block, arith, g.kernel_name(), worst)
    g.close(); o.close()
# AM / FM / other modes by 8 in AUTO
for mode in (rc.MODE_AM, rc.MODE_FM, rc.MODE_LSB, rc.MODE_DIG):
    nch = 40
    kw = dict(nco=True, nco_step_all=0x01000000)
    g = sr.Rx(rc.ChainSpec(nch, 256, 8, 256, 63, 0, mode, rc.ARITH_AUTO, **kw).config())
    o = CpuChain(rc.ChainSpec(nch, 256, 8, 256, 63, 0, mode, rc.ARITH_CMSIS, **kw), "orc")
    worst = 0
    for call in range(3):
        iq = synth_iq(0, nch, 4096 * call, 4096)
        yo = o.process(iq); yg = g.process(iq)
        d = np.abs(yg.astype(np.float64) - yo).reshape(nch, -1, 32).max(axis=2); m = np.abs(yo).reshape(nch, -1, 32).max(axis=2)
        worst = max(worst, float((d / np.maximum(m, 1e-30)).max()))
    print("mode %#x" % mode, g.kernel_name(), worst, g.guard_stats())
    g.close(); o.close()
