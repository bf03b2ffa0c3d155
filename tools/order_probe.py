#!/usr/bin/env python3
"""tools/order_probe.py -- does a memory-bound workload's time depend on what the GPU did before it?  cfg2 (k_hilb_split16, at 0.93-0.97 of its
streaming roof) timed three times in ONE process: first thing, right behind ten seconds of the power-capped cfg3 kernel, and after ten idle seconds.
Prints ms per launch and the streaming roof of the same bytes each time (bench.side_workload)."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "selenite-lite_amd"))
import bench  # noqa: E402
import selenite_rx as sr  # noqa: E402
from selenite_rx import chain as ch  # noqa: E402


def cfg2(tag):
    r = bench.side_workload("cfg2", False, sr.ARITH_AUTO, 300.0, parity=False)
    print("%-34s %.4f ms  roof %.4f ms  (%.3f of it)" % (tag, r["ms_per_step"], r["roofline"]["streaming_roof_ms"], r["roofline"]["frac_of_streaming_roof"]), flush=True)


def heat(seconds):
    name, channels, bs = ch.WORKLOADS["cfg3"]
    spec = ch.baseline_spec(name, channels, sr.ARITH_AUTO)
    rx = sr.Rx(spec.config())
    d_in, d_out = sr.DeviceBuffer(channels * bs * 8), sr.DeviceBuffer(channels * (bs // 4) * 4)
    rx.synth_device(d_in.ptr, 0, channels, 0, bs, ch.SEED)
    t = time.perf_counter()
    while time.perf_counter() - t < seconds:
        for _ in range(64):
            rx.process_device(d_in.ptr, d_out.ptr, bs)
        rx.sync()
    rx.close(); d_in.free(); d_out.free()


cfg2("first thing in the process")
cfg2("again")
heat(10.0)
cfg2("behind 10 s of the cfg3 kernel")
cfg2("again, at once")
time.sleep(10.0)
cfg2("after 10 idle seconds")
heat(3.0)
cfg2("behind 3 s of the cfg3 kernel")
