#!/usr/bin/env python3
"""tools/output_hashes.py -- SHA-256 of the audio and of the streaming state the library (SELENITE_RX_LIB, else the product) produces for a fixed set
of shapes, arithmetics and slot formats on seeded input: run it with two builds and diff the output to show that a kernel change that was
not meant to change a bit did not (`SELENITE_RX_LIB=old.so python3 tools/output_hashes.py > a; python3 tools/output_hashes.py > b; diff a b`)."""
import hashlib
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "selenite-lite_amd"))
import numpy as np  # noqa: E402
import selenite_rx as sr  # noqa: E402
from selenite_rx import chain as ch  # noqa: E402

SHAPES = [("cfg1", 256 * 4), ("cfg2", 256 * 5), ("cfg2_48k128", 128 * 7), ("cfg2_48k", 192 * 5), ("cfg3", 1024 * 3), ("cfg3_by8", 2048), ("cfg4", 256 * 4)]
for name, bs in SHAPES:
    for arith in (sr.ARITH_CMSIS, sr.ARITH_FMA, sr.ARITH_SPLIT16, sr.ARITH_AUTO):
        for q15 in (False, True):
            for mode in (sr.MODE_USB, sr.MODE_LSB, sr.MODE_AM):
                if name == "cfg4" and mode != sr.MODE_USB:
                    continue
                n = 96
                spec = ch.baseline_spec(name, n, arith)
                if name != "cfg4":
                    spec.mode = mode
                rx = sr.Rx(spec.config())
                h = hashlib.sha256()
                for call in range(3):
                    iq = sr.synth_iq_host(0, n, call * bs, bs, ch.SEED)
                    if q15:
                        y = rx.process_q15(np.clip(np.trunc(iq * 32768.0), -32768, 32767).astype(np.int16))
                    else:
                        y = rx.process(iq)
                    h.update(np.ascontiguousarray(y).tobytes())
                st = rx.state()
                for k in sorted(st):
                    h.update(np.ascontiguousarray(st[k]).tobytes())
                print(name, arith, "q15" if q15 else "f32", mode, rx.kernel_name(), h.hexdigest()[:24])
                rx.close()
