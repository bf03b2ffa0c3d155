"""Throughput of the batched DSP ring buffers (csrc/ring.hip): one DSP_In_Buff_Write +
DSP_In_Buff_Read pair per step over C rings, data resident in HBM.  Prints one JSON line.
Algorithmic bytes per frame and pair: write = 4 B packet read + 4 B ring written, read = 4 B ring
read + 4 B packet written (+ the repeated frame and 5 B of pointers per ring and call)."""
import argparse
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "selenite-lite_amd"))
import selenite_rx as sr  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--channels", type=int, default=1 << 20)
ap.add_argument("--words", type=int, default=96, help="uint16 words per packet (96 = 1 ms @ 48 kHz, dsp_if.h:69-73)")
ap.add_argument("--iters", type=int, default=50)
a = ap.parse_args()
ring = sr.Ring(a.channels)
src = sr.DeviceBuffer(a.channels * a.words * 2)
dst = sr.DeviceBuffer(a.channels * a.words * 2)
src.upload(np.random.default_rng(1).integers(-32768, 32768, (a.channels, a.words), dtype=np.int64).astype(np.int16))
ring.time_pair(src.ptr, dst.ptr, a.words, 5)
ms = ring.time_pair(src.ptr, dst.ptr, a.words, a.iters)
frames = a.words // 2
bytes_pair = a.channels * (16 * frames + 4 + 2 * 5)
print(json.dumps({"metric": "ring in_write+in_read pairs", "channels": a.channels, "words": a.words,
                  "ms_per_pair": round(ms, 4), "Mframes_per_s": round(a.channels * frames / ms / 1e3, 1),
                  "algorithmic_GBps": round(bytes_pair / ms / 1e6, 1), "hbm_peak_GBps": 8000}))
