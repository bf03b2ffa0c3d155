#!/usr/bin/env python3
"""Generate the committed TX fixtures from the REAL CMSIS-DSP 1.5.3 code (oracle/ref_tx.c composition
in oracle/_ref/libcmsis_ref.so, build container only; nothing of the reference is copied).

    python tests/golden/make_tx_golden.py

Writes tx_primitives.npz (arm_fir_interpolate_f32 / arm_negate_f32 / arm_offset_f32 known-answer and
seeded cases) and tx_chain_<name>.npz (IQ outputs + end-of-run state of small TX instances).  Chain
inputs are regenerated from the seed (rxcommon.synth_audio), so only outputs are stored."""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
import rxcommon as rc  # noqa: E402
from rxcommon import fptr  # noqa: E402

TX_CASES = {
    # name: (TxSpec kwargs, channels, ALC blocks per call, calls)
    "usb": (dict(block=64, interp=4, ni_taps=256, nh_taps=63, mode=rc.MODE_USB), 3, 4, 2),
    "lsb_small": (dict(block=32, interp=3, ni_taps=45, nh_taps=15, mode=rc.MODE_LSB, nco_step_all=0x01234567), 2, 5, 2),
    "am": (dict(block=64, interp=2, ni_taps=32, nh_taps=31, mode=rc.MODE_AM, nco_step_all=0x02000000), 2, 3, 2),
    "plain": (dict(block=16, interp=1, ni_taps=0, nh_taps=0, mode=rc.MODE_USB, nco=False, alc=False), 2, 2, 1),
}


def prim_interp_cases():
    return [(8, 2, 8), (256, 4, 64), (45, 3, 7), (12, 4, 5), (5, 5, 3), (64, 1, 16)]


def main():
    if not rc.ref_available():
        raise SystemExit("oracle/_ref/libcmsis_ref.so missing: run `make -C oracle` where /root/reference exists")
    R = rc.ref_lib()
    rng = np.random.default_rng(0x5E1E7A)
    prim = {}
    # known answer: taps b = {1,2,3,4} stored reversed, L = 2, impulse -> b[0], b[1], b[2], b[3], 0 ...
    co = np.array([4, 3, 2, 1], np.float32)
    st = np.zeros(1 + 4, np.float32)
    x = np.array([1, 0, 0, 0], np.float32)
    y = np.empty(8, np.float32)
    R.ref_fir_interpolate(fptr(co), 4, 2, fptr(st), fptr(x), fptr(y), 4)
    prim["interp_impulse_y"] = y.copy()
    for nt, L, blk in prim_interp_cases():
        c = rng.standard_normal(nt).astype(np.float32)
        s = np.zeros(nt // L - 1 + blk, np.float32)
        xs, ys = [], []
        for _ in range(3):
            xv = rng.standard_normal(blk).astype(np.float32)
            yv = np.empty(blk * L, np.float32)
            R.ref_fir_interpolate(fptr(c), nt, L, fptr(s), fptr(xv), fptr(yv), blk)
            xs.append(xv); ys.append(yv)
        k = "interp_%d_%d_%d" % (nt, L, blk)
        prim[k + "_c"], prim[k + "_x"], prim[k + "_y"] = c, np.stack(xs), np.stack(ys)
        prim[k + "_state"] = s[:nt // L - 1].copy()
    v = rng.standard_normal(37).astype(np.float32)
    v[3], v[4] = 0.0, -0.0
    n, o = np.empty_like(v), np.empty_like(v)
    import ctypes as C
    R.ref_offset.argtypes = [rc.f32p, C.c_float, rc.f32p, C.c_uint32]
    R.ref_negate(fptr(v), fptr(n), v.size)
    R.ref_offset(fptr(v), 0.5, fptr(o), v.size)
    prim["neg_x"], prim["neg_y"], prim["off_y"] = v, n, o
    np.savez_compressed(os.path.join(HERE, "tx_primitives.npz"), **prim)

    for name, (kw, ch, nblk, ncalls) in TX_CASES.items():
        spec = rc.TxSpec(ch, **kw)
        ref = rc.TxCpuChain(spec, "ref")
        assert ref.ok(), name
        bs = spec.block * nblk
        outs = [ref.process(rc.synth_audio(0, ch, k * bs, bs)) for k in range(ncalls)]
        st = ref.state()
        aq = np.clip(np.trunc(rc.synth_audio(0, ch, 0, bs) * 32768.0), -32768, 32767).astype(np.int16)
        q = rc.TxCpuChain(spec, "ref").process_q15(aq)
        np.savez_compressed(os.path.join(HERE, "tx_chain_%s.npz" % name), iq=np.stack(outs), q15=q, **st)
        print("tx_chain_%s: %d calls x %s" % (name, ncalls, outs[0].shape))


if __name__ == "__main__":
    main()
