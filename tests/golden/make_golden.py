#!/usr/bin/env python3
"""Generate the committed golden fixtures from the REAL CMSIS-DSP 1.5.3 code.

Runs ONLY in the build container: it loads oracle/_ref/libcmsis_ref.so, which oracle/Makefile
compiles from the CMSIS sources where they lie under /root/reference (nothing of the reference is
copied).  The fixtures are data: seeds/parameters, inputs where they are not regenerable from a
seed, and the outputs + end-of-run filter state the reference produced.

    python tests/golden/make_golden.py

Files written next to this script:
  sintable_f32.bin         sinTable_f32[513] as the reference holds it (arm_common_tables.c:21895)
  primitives.npz           per-primitive known-answer vectors and seeded random cases
  chain_<name>.npz         chain outputs/state for small instances of BASELINE cfg1..cfg4 and of the
                           other demodulator modes, produced by oracle/ref_chain.c (CMSIS composition)
Inputs of the chain cases are regenerated from the seed by the synthetic generator
(oracle orc_synth_iq == library selenite_rx_synth_iq_host), so only outputs are stored.
"""
import ctypes as C
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
import rxcommon as rc  # noqa: E402
from rxcommon import fptr  # noqa: E402

CHAIN_CASES = {
    # name: (spec kwargs or baseline name, channels, nblocks per call, ncalls)
    "cfg1": ("baseline", "cfg1", 2, 2, 3),
    "cfg2": ("baseline", "cfg2", 2, 2, 2),
    "cfg3": ("baseline", "cfg3", 2, 4, 2),
    "cfg4": ("baseline", "cfg4", 2, 2, 2),
    "lsb_full": ("spec", dict(block=64, decim=4, nd_taps=48, nh_taps=31, n_biquad=2, mode=rc.MODE_LSB,
                              nco=True, nco_step_all=0x01234567, bp_f0=0.05, bp_q=3.0), 3, 5, 2),
    "cwr_full": ("spec", dict(block=64, decim=4, nd_taps=48, nh_taps=31, n_biquad=2, mode=rc.MODE_CWR,
                              nco=True, nco_step_all=0x01234567, bp_f0=0.05, bp_q=3.0), 3, 5, 2),
    "am_full": ("spec", dict(block=64, decim=4, nd_taps=48, nh_taps=31, n_biquad=2, mode=rc.MODE_AM,
                             nco=True, nco_step_all=0x01234567, bp_f0=0.05, bp_q=3.0), 3, 5, 2),
    "fm_full": ("spec", dict(block=64, decim=4, nd_taps=48, nh_taps=31, n_biquad=2, mode=rc.MODE_FM,
                             nco=True, nco_step_all=0x01234567, bp_f0=0.05, bp_q=3.0), 3, 5, 2),
    "fm_nodec": ("spec", dict(block=96, decim=1, nd_taps=0, nh_taps=3, mode=rc.MODE_FM, nco=False, agc=False), 2, 3, 3),
    "global_gain": ("spec", dict(block=64, decim=2, nd_taps=21, nh_taps=15, mode=rc.MODE_USB, nco=True,
                                 nco_step_all=0x02000000, agc_global=True), 6, 4, 3),
}


def chain_spec(kind, arg, channels):
    if kind == "baseline":
        return rc.baseline_spec(arg, channels)
    return rc.ChainSpec(channels, **arg)


def main():
    if not rc.ref_available():
        raise SystemExit("oracle/_ref/libcmsis_ref.so missing: run `make -C oracle` where /root/reference exists")
    R = rc.ref_lib()
    rng = np.random.default_rng(0x5E1E917E)

    tab = np.ctypeslib.as_array(R.ref_sin_table(), (513,)).copy()
    tab.tofile(os.path.join(HERE, "sintable_f32.bin"))

    out = {}
    # --- sin / cos: known answers + sweep incl. negative shortcut branch and large arguments ---
    x = np.concatenate([
        np.array([0.0, 1.0, 100.0, -1e-7, -1.9e-7, -2.0e-7, -1.0, 6.2831850, 6.2831855, 3.1415927, -100.0,
                  1e-8, 12345.678], np.float32),
        (rng.random(2000, dtype=np.float32) * np.float32(6.2831853)),
        ((rng.random(500, dtype=np.float32) - np.float32(0.5)) * np.float32(200.0)),
        (np.arange(0, 1 << 24, 40503, dtype=np.uint32).astype(np.float32) * np.float32(2 * np.pi / 2 ** 24)),
    ]).astype(np.float32)
    s, c = np.empty_like(x), np.empty_like(x)
    R.ref_sin_cos(fptr(x), fptr(s), fptr(c), x.size)
    out.update(sincos_x=x, sincos_sin=s, sincos_cos=c)

    # --- FIR decimator: impulse KAT + seeded cases over (taps, M, block) incl. ragged sizes ---
    cases = []
    for (nt, M, blk, ncall) in [(3, 1, 8, 2), (256, 4, 256, 3), (255, 4, 64, 3), (17, 3, 12, 4), (5, 5, 5, 3),
                                (64, 8, 32, 2), (1, 1, 7, 2), (33, 2, 2, 5)]:
        coeffs = (rng.standard_normal(nt) / max(nt, 1) ** 0.5).astype(np.float32)
        if nt == 3:
            coeffs = np.array([1, 2, 3], np.float32)
        state = np.zeros(nt - 1 + blk, np.float32)
        ys, xs = [], []
        for k in range(ncall):
            xin = rng.standard_normal(blk).astype(np.float32)
            if nt == 3 and k == 0:
                xin[:] = 0
                xin[0] = 1
            y = np.empty(blk // M, np.float32)
            R.ref_fir_decimate(fptr(coeffs), nt, M, fptr(state), fptr(xin), fptr(y), blk)
            xs.append(xin)
            ys.append(y)
        cases.append((nt, M, blk, ncall))
        tag = "dec_%d_%d_%d" % (nt, M, blk)
        out[tag + "_c"], out[tag + "_x"], out[tag + "_y"] = coeffs, np.concatenate(xs), np.concatenate(ys)
        out[tag + "_state"] = state[:max(nt - 1, 0)].copy()
    out["dec_cases"] = np.array(cases, np.int32)
    out["dec_init_status"] = np.array([[nt, M, blk, R.ref_fir_decimate_init_status(nt, M, blk)]
                                       for (nt, M, blk) in [(8, 4, 256), (8, 4, 255), (8, 3, 9), (8, 3, 10), (4, 1, 1)]],
                                      np.int32)

    # --- FIR ---
    cases = []
    for (nt, blk, ncall) in [(3, 8, 2), (63, 64, 3), (127, 256, 2), (5, 3, 4), (1, 9, 2), (31, 1, 6), (64, 17, 3)]:
        coeffs = (rng.standard_normal(nt) / max(nt, 1) ** 0.5).astype(np.float32)
        if nt == 3:
            coeffs = np.array([1, 2, 3], np.float32)
        state = np.zeros(nt - 1 + blk, np.float32)
        ys, xs = [], []
        for k in range(ncall):
            xin = rng.standard_normal(blk).astype(np.float32)
            if nt == 3 and k == 0:
                xin[:] = 0
                xin[0] = 1
            y = np.empty(blk, np.float32)
            R.ref_fir(fptr(coeffs), nt, fptr(state), fptr(xin), fptr(y), blk)
            xs.append(xin)
            ys.append(y)
        cases.append((nt, blk, ncall))
        tag = "fir_%d_%d" % (nt, blk)
        out[tag + "_c"], out[tag + "_x"], out[tag + "_y"] = coeffs, np.concatenate(xs), np.concatenate(ys)
        out[tag + "_state"] = state[:max(nt - 1, 0)].copy()
    out["fir_cases"] = np.array(cases, np.int32)

    # --- biquad cascade: KAT {1,0,0,0.5,0} impulse + seeded stable sections ---
    cases = []
    for (ns, blk, ncall) in [(1, 8, 2), (4, 256, 2), (2, 7, 3), (3, 1, 5)]:
        if ns == 1:
            coeffs = np.array([1, 0, 0, 0.5, 0], np.float32)
        else:
            coeffs = rc.design_bandpass(ns, 0.03 + 0.01 * ns, 2.0 + ns)
        state = np.zeros(4 * ns, np.float32)
        ys, xs = [], []
        for k in range(ncall):
            xin = rng.standard_normal(blk).astype(np.float32)
            if ns == 1 and k == 0:
                xin[:] = 0
                xin[0] = 1
            y = np.empty(blk, np.float32)
            R.ref_biquad(fptr(coeffs), ns, fptr(state), fptr(xin), fptr(y), blk)
            xs.append(xin)
            ys.append(y)
        cases.append((ns, blk, ncall))
        tag = "biq_%d_%d" % (ns, blk)
        out[tag + "_c"], out[tag + "_x"], out[tag + "_y"] = coeffs, np.concatenate(xs), np.concatenate(ys)
        out[tag + "_state"] = state.copy()
    out["biq_cases"] = np.array(cases, np.int32)

    # --- elementwise / statistics / conversion primitives on one seeded vector set ---
    n = 1027
    a = rng.standard_normal(2 * n).astype(np.float32)
    b = rng.standard_normal(2 * n).astype(np.float32)
    a[5], a[6], b[7] = 0.0, -0.0, -0.0
    f32p, u32p, i16p = rc.f32p, rc.u32p, rc.i16p
    for fn in ("arm_cmplx_mult_cmplx_f32", "arm_cmplx_mag_f32", "arm_cmplx_conj_f32", "arm_add_f32", "arm_sub_f32", "arm_abs_f32",
               "arm_scale_f32", "arm_max_f32", "arm_q15_to_float", "arm_float_to_q15"):
        getattr(R, fn).restype = None
    cm = np.empty(2 * n, np.float32)
    R.arm_cmplx_mult_cmplx_f32(fptr(a), fptr(b), fptr(cm), C.c_uint32(n))
    mag = np.empty(n, np.float32)
    R.arm_cmplx_mag_f32(fptr(a), fptr(mag), C.c_uint32(n))
    cj = np.empty(2 * n, np.float32)
    R.arm_cmplx_conj_f32(fptr(a), fptr(cj), C.c_uint32(n))
    add, sub, ab, sc = (np.empty(2 * n, np.float32) for _ in range(4))
    R.arm_add_f32(fptr(a), fptr(b), fptr(add), C.c_uint32(2 * n))
    R.arm_sub_f32(fptr(a), fptr(b), fptr(sub), C.c_uint32(2 * n))
    R.arm_abs_f32(fptr(a), fptr(ab), C.c_uint32(2 * n))
    R.arm_scale_f32(fptr(a), C.c_float(0.37), fptr(sc), C.c_uint32(2 * n))
    mx, mi = C.c_float(), C.c_uint32()
    tie = ab.copy()
    tie[100] = tie[900] = 77.0                          # duplicated maximum: first index wins
    R.arm_max_f32(fptr(tie), C.c_uint32(2 * n), C.byref(mx), C.byref(mi))
    q = rng.integers(-32768, 32768, 2 * n, dtype=np.int16)
    q[:4] = [-32768, 32767, 0, -1]
    qf = np.empty(2 * n, np.float32)
    R.arm_q15_to_float(q.ctypes.data_as(i16p), fptr(qf), C.c_uint32(2 * n))
    fq_in = np.concatenate([a * np.float32(0.6), np.array([1.0, -1.0, 0.99999, -1.00001, 1.5, -3.0, 3.05e-5, -3.05e-5],
                                                          np.float32)]).astype(np.float32)
    fq = np.empty(fq_in.size, np.int16)
    R.arm_float_to_q15(fptr(fq_in), fq.ctypes.data_as(i16p), C.c_uint32(fq_in.size))
    out.update(ew_a=a, ew_b=b, ew_cmul=cm, ew_mag=mag, ew_conj=cj, ew_add=add, ew_sub=sub, ew_abs=ab, ew_scale=sc,
               ew_max_in=tie, ew_max=np.array([mx.value], np.float32), ew_max_idx=np.array([mi.value], np.uint32),
               q15_in=q, q15_f=qf, fq_in=fq_in, fq_out=fq)
    np.savez_compressed(os.path.join(HERE, "primitives.npz"), **out)

    # --- chain cases (ref_chain.c = composition of the real CMSIS calls) ---
    for name, (kind, arg, channels, nblocks, ncalls) in CHAIN_CASES.items():
        spec = chain_spec(kind, arg, channels)
        ch = rc.CpuChain(spec, "ref")
        assert ch.ok(), name
        bs = spec.block * nblocks
        ys = []
        for call in range(ncalls):
            iq = rc.synth_iq(0, channels, call * bs, bs)
            ys.append(ch.process(iq))
        st = ch.state()
        np.savez_compressed(os.path.join(HERE, "chain_%s.npz" % name), audio=np.stack(ys),
                            meta=np.array([channels, nblocks, ncalls], np.int32), **st)
        ch.close()

    # q15 slot case on cfg3
    spec = rc.baseline_spec("cfg3", 2)
    ch = rc.CpuChain(spec, "ref")
    iq = rc.synth_iq(0, 2, 0, 1024)
    q = np.clip(np.trunc(iq * 32768.0), -32768, 32767).astype(np.int16)
    np.savez_compressed(os.path.join(HERE, "chain_cfg3_q15.npz"), audio=ch.process_q15(q))
    print("golden fixtures written to", HERE)


if __name__ == "__main__":
    main()
