"""tools/cw_stage_timing.py -- k_cw_fused with 2 / 4 / 8 biquad stages, 65 536 channels x 4096 samples, shared LO; with
SELENITE_RX_CW_ONE=1 the one-channel-per-lane kernel for 4 / 8 stages."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "selenite-lite_amd"))
import selenite_rx as sr
from selenite_rx import chain as ch

C_, BS = 65536, 4096
d_in = sr.DeviceBuffer(C_ * BS * 8)
d_out = sr.DeviceBuffer(C_ * BS * 4)
for ns in (2, 4, 8):
    spec = ch.ChainSpec(C_, 256, 1, 0, 0, ns, ch.MODE_CW, ch.ARITH_CMSIS, nco=True, nco_step_all=0x00800000)
    rx = sr.Rx(spec.config())
    rx.synth_device(d_in.ptr, 0, C_, 0, BS, ch.SEED)
    rx.time_process(d_in.ptr, d_out.ptr, BS, 200)
    ms = rx.time_process(d_in.ptr, d_out.ptr, BS, 200)
    print("%d stages: %.4f ms  %.1f Gsamples/s  (%s)" % (ns, ms, C_ * BS / ms / 1e6, rx.kernel_name()))
    rx.close()
