"""tools/latency_table.py -- the short-call regime (VERDICT r2 #6a): microseconds per call and Gsamples/s for
samples-per-call in {96, 192, 256, 512, 1024, 4096} x channels in {1, 64, 4096, 65536}, cfg3 chain (AUTO arithmetic):

  dev_stream   back-to-back selenite_rx_process_f32_device calls, HIP events around 200 of them (what a streaming host sees per call)
  dev_sync     one call + selenite_rx_sync, wall clock (what a host that needs the audio back on the device before going on sees)
  host_pinned  selenite_rx_process_f32 on page-locked host buffers, wall clock (PCIe both ways included)

96 / 192 are the firmware's own block geometry (Core/Inc/dsp_if.h:69-73: 96 I/Q frames per 1 ms I2S half-buffer,
dsp_if.c:50-67); a DSP block of 96 inputs is 24 audio samples.  Also BASELINE cfg1 ("ns/block": C = 1, one 256-sample block, USB,
63-tap Hilbert pair + AGC) on the GPU and on one host core (the reference's CMSIS-DSP when oracle/_ref is there).
Run on the GPU box:  python tools/latency_table.py > profiles/r3/latency.txt"""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "selenite-lite_amd"))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import selenite_rx as sr
from selenite_rx import chain as ch

def spin(rx, d_in, d_out, bs, ms=150.0):
    t = time.perf_counter()
    while (time.perf_counter() - t) * 1e3 < ms:
        for _ in range(32):
            rx.process_device(d_in.ptr, d_out.ptr, bs)
        rx.sync()

def measure(C_, bs, block):
    spec = ch.ChainSpec(C_, block, 4, 256, 63, 0, sr.MODE_USB, sr.ARITH_AUTO, nco=True, nco_step_all=0x01000000)
    rx = sr.Rx(spec.config())
    d_in, d_out = sr.DeviceBuffer(C_ * bs * 8), sr.DeviceBuffer(C_ * (bs // 4) * 4)
    rx.synth_device(d_in.ptr, 0, C_, 0, bs, ch.SEED)
    rx.sync()
    spin(rx, d_in, d_out, bs, 150.0 if C_ * bs >= 1 << 20 else 60.0)
    iters = 200 if C_ * bs < (1 << 26) else 50
    ms = np.sort(rx.time_process_each(d_in.ptr, d_out.ptr, bs, iters))
    stream_us = float(ms[len(ms) // 2]) * 1e3
    sync = []
    for _ in range(30):
        t = time.perf_counter(); rx.process_device(d_in.ptr, d_out.ptr, bs); rx.sync(); sync.append(time.perf_counter() - t)
    sync_us = float(np.median(sync)) * 1e6
    host_us = None
    if C_ * bs * 8 <= (512 << 20):
        hin, hout = sr.pinned_array((C_, bs, 2), np.float32), sr.pinned_array((C_, bs // 4), np.float32)
        hin[...] = 0.25
        rx.process(hin, hout)
        host = []
        for _ in range(10):
            t = time.perf_counter(); rx.process(hin, hout); host.append(time.perf_counter() - t)
        host_us = float(np.median(host)) * 1e6
    name = rx.kernel_name()
    rx.close(); d_in.free(); d_out.free()
    return stream_us, sync_us, host_us, name

print("# cfg3 chain (NCO + 256-tap /4 + 63-tap Hilbert SSB + AGC), SELENITE_ARITH_AUTO, 1x MI355X; us per call | Gsamples/s")
print("# %-9s %-8s %-6s | %-22s | %-22s | %-22s | %s" % ("channels", "samples", "block", "dev_stream", "dev_sync", "host_pinned", "kernel"))
for C_ in (1, 64, 4096, 65536):
    for bs in (96, 192, 256, 512, 1024, 4096):
        block = 96 if bs in (96, 192) else 256
        st, sy, ho, name = measure(C_, bs, block)
        f = lambda us: "%9.1f us %8.3f" % (us, C_ * bs / us / 1e3) if us else "%22s" % "-"
        print("  %-9d %-8d %-6d | %s | %s | %s | %s" % (C_, bs, block, f(st), f(sy), f(ho), name))
        sys.stdout.flush()

# BASELINE cfg1: one 256-sample block, one channel
import rxcommon as rc
spec = rc.baseline_spec("cfg1", 1, rc.ARITH_AUTO)
rx = sr.Rx(spec.config())
d_in, d_out = sr.DeviceBuffer(256 * 8), sr.DeviceBuffer(256 * 4)
rx.synth_device(d_in.ptr, 0, 1, 0, 256, ch.SEED); rx.sync()
spin(rx, d_in, d_out, 256, 60.0)
ms = np.sort(rx.time_process_each(d_in.ptr, d_out.ptr, 256, 500))
sync = []
for _ in range(200):
    t = time.perf_counter(); rx.process_device(d_in.ptr, d_out.ptr, 256); rx.sync(); sync.append(time.perf_counter() - t)
which = "ref" if rc.ref_available() else "orc"
cpu = rc.CpuChain(rc.baseline_spec("cfg1", 1, rc.ARITH_CMSIS), which)
iq = rc.synth_iq(0, 1, 0, 256)
for _ in range(200): cpu.process(iq)
t = time.perf_counter()
for _ in range(5000): cpu.process(iq)
cpu_ns = (time.perf_counter() - t) / 5000 * 1e9
print("\n# BASELINE cfg1 (C = 1, one 256-sample block, USB, 63-tap Hilbert pair + AGC), ns per block:")
print("  GPU %s: back-to-back device calls %.0f ns/block (median of 500), call + sync %.0f ns/block; one host core, %s (ctypes call included): %.0f ns/block"
      % (rx.kernel_name(), ms[250] * 1e6, np.median(sync) * 1e9, "CMSIS-DSP 1.5.3 (oracle/_ref)" if which == "ref" else "oracle restatement", cpu_ns))

# the same block in the other arithmetic contracts (one kernel launch per call instead of AUTO's two)
for nm, ar in (("split16", rc.ARITH_SPLIT16), ("fma", rc.ARITH_FMA), ("cmsis (bit-exact)", rc.ARITH_CMSIS)):
    rx2 = sr.Rx(rc.baseline_spec("cfg1", 1, ar).config())
    spin(rx2, d_in, d_out, 256, 60.0)
    ms2 = np.sort(rx2.time_process_each(d_in.ptr, d_out.ptr, 256, 500))
    print("  GPU %-18s %s: back-to-back device calls %.0f ns/block" % (nm, rx2.kernel_name(), ms2[250] * 1e6))
    rx2.close()
