"""Throughput of the TX chain (csrc/tx.hip): C channels x B audio samples per call -> C x 4B complex
output samples, data resident in HBM, steady-state clocks.  Prints one JSON line.
Algorithmic bytes per channel-call: 4*B audio in + 8*B*L I/Q out + state in and out."""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "selenite-lite_amd"))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np  # noqa: E402
import rxcommon as rc  # noqa: E402
import selenite_rx as sr  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--channels", type=int, default=65536)
ap.add_argument("--block-size", type=int, default=1024, help="audio samples per channel and call")
ap.add_argument("--arith", default="split16", choices=["split16", "fma", "cmsis"])
ap.add_argument("--iters", type=int, default=50)
a = ap.parse_args()
spec = rc.TxSpec(a.channels, arith={"fma": rc.ARITH_FMA, "cmsis": rc.ARITH_CMSIS, "split16": rc.ARITH_SPLIT16}[a.arith])
tx = sr.Tx(spec.config())
L = spec.interp
d_a, d_iq = sr.DeviceBuffer(a.channels * a.block_size * 4), sr.DeviceBuffer(a.channels * a.block_size * L * 8)
chunk = 4096
for c0 in range(0, a.channels, chunk):
    n = min(chunk, a.channels - c0)
    # audio = I rail of the library's own synthetic generator (the oracle is test infrastructure only)
    x = np.ascontiguousarray(sr.synth_iq_host(c0, n, 0, a.block_size, rc.SEED)[:, :, 0])
    sr.lib().selenite_rx_memcpy_h2d(d_a.ptr + c0 * a.block_size * 4, x.ctypes.data, x.nbytes)
t0 = time.perf_counter()
while time.perf_counter() - t0 < 0.3:
    tx.time_process(d_a.ptr, d_iq.ptr, a.block_size, 2)
ms = tx.time_process(d_a.ptr, d_iq.ptr, a.block_size, a.iters)
state = 4 * (2 * (spec.nh_taps - 1) + 2 * (spec.ni_taps // L - 1) + 2)
bytes_call = a.channels * (4 * a.block_size + 8 * a.block_size * L + 2 * state)
print(json.dumps({"metric": "TX chain, complex output Msamples/s", "value": round(a.channels * a.block_size * L / ms / 1e3, 1),
                  "channels": a.channels, "audio_samples_per_call": a.block_size, "interp": L, "arith": a.arith,
                  "ms_per_call": round(ms, 4), "algorithmic_GBps": round(bytes_call / ms / 1e6, 1), "hbm_peak_GBps": 8000,
                  "kernel": tx.kernel_name()}))
