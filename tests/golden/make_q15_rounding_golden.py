#!/usr/bin/env python3
"""Golden vectors for the ARM_MATH_ROUNDING build of arm_float_to_q15 (selenite_rx_config::q15_rounding = 1).

Build container only: oracle/_ref/libcmsis_ref.so holds SupportFunctions/arm_float_to_q15.c compiled a second time with
-DARM_MATH_ROUNDING as arm_float_to_q15_rounding (oracle/Makefile), and oracle/ref_chain.c calls it behind the chain when the
config asks for it.  Nothing of the reference is copied: inputs and the outputs the reference's code produced.

    python tests/golden/make_q15_rounding_golden.py      ->  tests/golden/q15_rounding.npz
"""
import ctypes as C
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
import rxcommon as rc  # noqa: E402
from rxcommon import fptr  # noqa: E402

# name: (baseline name or ChainSpec kwargs, channels, samples per call, calls)
CASES = {
    "cfg2": ("cfg2", 2, 1024, 2),
    "cfg3": ("cfg3", 2, 1024, 2),
    "cfg4": ("cfg4", 2, 1024, 2),
    "global_gain": (dict(block=64, decim=2, nd_taps=21, nh_taps=15, mode=rc.MODE_USB, nco=True, nco_step_all=0x02000000,
                         agc_global=True), 6, 256, 2),
    "am_full": (dict(block=64, decim=4, nd_taps=48, nh_taps=31, n_biquad=2, mode=rc.MODE_AM, nco=True, nco_step_all=0x01234567,
                     bp_f0=0.05, bp_q=3.0), 3, 320, 2),
}


def case_spec(arg, channels, rounding=True, **kw):
    if isinstance(arg, str):
        return rc.baseline_spec(arg, channels, q15_rounding=rounding, **kw)
    return rc.ChainSpec(channels, q15_rounding=rounding, **dict(arg, **kw))


# TX chain (int16 audio in, int16 I/Q out): name -> (TxSpec kwargs, channels, audio samples per call, calls)
TX_CASES = {
    "tx_usb": (dict(), 2, 256, 2),                                                    # the fused shape: L = 4, 256-tap interpolator, 63-tap Hilbert, ALC block 64
    "tx_lsb_small": (dict(block=32, interp=2, ni_taps=16, nh_taps=15, mode=rc.MODE_LSB, nco_step_all=0x01234567), 3, 96, 2),   # the generic kernels
}


def tx_spec(arg, channels, rounding=True, **kw):
    return rc.TxSpec(channels, q15_rounding=rounding, **dict(arg, **kw))


def tx_input(channels, call, bs):
    return np.clip(np.trunc(rc.synth_audio(0, channels, call * bs, bs) * 32768.0), -32768, 32767).astype(np.int16)


def case_input(channels, call, bs):
    iq = rc.synth_iq(0, channels, call * bs, bs)
    return np.clip(np.trunc(iq * 32768.0), -32768, 32767).astype(np.int16)


def primitive_input():
    rng = np.random.default_rng(0x0515)
    k = np.arange(-40, 41, dtype=np.float64)
    ties = np.concatenate([(k + 0.5) / 32768.0, (k + 0.49999) / 32768.0, (k + 0.50001) / 32768.0, k / 32768.0])
    edge = np.array([0.0, -0.0, 1.0, -1.0, 0.99998, 0.999985, 0.99999, -0.99999, -1.00001, 32766.5 / 32768.0, 32767.49 / 32768.0,
                     32767.5 / 32768.0, -32767.5 / 32768.0, -32768.5 / 32768.0, 1.5, -3.0, 1e-9, -1e-9, 1.5e-5, -1.5e-5, 1.6e-5, -1.6e-5])
    return np.concatenate([rng.uniform(-1.1, 1.1, 4096), ties, edge]).astype(np.float32)


def main():
    if not rc.ref_available():
        raise SystemExit("oracle/_ref/libcmsis_ref.so missing: run `make -C oracle` where /root/reference exists")
    R = rc.ref_lib()
    out = {}
    x = primitive_input()
    q = np.empty(x.size, np.int16)
    R.arm_float_to_q15_rounding(fptr(x), q.ctypes.data_as(rc.i16p), C.c_uint32(x.size))
    out.update(fq_in=x, fq_out=q)
    for name, (arg, channels, bs, ncalls) in CASES.items():
        ch = rc.CpuChain(case_spec(arg, channels), "ref")
        assert ch.ok(), name
        out["chain_" + name] = np.stack([ch.process_q15(case_input(channels, call, bs)) for call in range(ncalls)])
        ch.close()
    for name, (arg, channels, bs, ncalls) in TX_CASES.items():
        ch = rc.TxCpuChain(tx_spec(arg, channels), "ref")
        assert ch.ok(), name
        out["chain_" + name] = np.stack([ch.process_q15(tx_input(channels, call, bs)) for call in range(ncalls)])
    np.savez_compressed(os.path.join(HERE, "q15_rounding.npz"), **out)
    print("written", os.path.join(HERE, "q15_rounding.npz"))


if __name__ == "__main__":
    main()
