"""CPU: the oracle (oracle/rx_oracle.c) against the committed golden fixtures.

The fixtures under tests/golden/ were produced by the REAL CMSIS-DSP 1.5.3 code compiled from
/root/reference (tests/golden/make_golden.py).  This file needs neither /root/reference nor a GPU,
so it pins the oracle on the GPU box as well.  Everything is bit-exact.
"""
import os

import numpy as np
import pytest

import rxcommon as rc
from rxcommon import ARITH_CMSIS as A0, bits_equal, fptr

G = rc.GOLDEN_DIR


@pytest.fixture(scope="module")
def prim():
    return np.load(os.path.join(G, "primitives.npz"))


def test_sin_table_matches_reference_table(oracle):
    want = np.fromfile(os.path.join(G, "sintable_f32.bin"), np.float32)
    got = np.ctypeslib.as_array(oracle.orc_sin_table(), (513,))
    assert want.size == 513 and bits_equal(got, want)
    assert np.signbit(want[512]) and want[512] == 0.0          # the reference's "-0.00000000f"


def test_sin_cos_known_answers_and_sweep(oracle, prim):
    x = prim["sincos_x"]
    s = np.array([oracle.orc_sin_f32(float(v), A0) for v in x], np.float32)
    c = np.array([oracle.orc_cos_f32(float(v), A0) for v in x], np.float32)
    assert bits_equal(s, prim["sincos_sin"]) and bits_equal(c, prim["sincos_cos"])
    # SURVEY.md 8c known answers
    assert np.float32(oracle.orc_sin_f32(1.0, A0)) == np.float32(0.841455162)
    assert np.float32(oracle.orc_cos_f32(1.0, A0)) == np.float32(0.540292084)
    assert np.float32(oracle.orc_sin_f32(100.0, A0)) == np.float32(-0.506359994)
    assert np.float32(oracle.orc_sin_f32(-1e-7, A0)) == np.float32(-1e-7)      # shortcut branch


def test_fir_decimate_vectors(oracle, prim):
    for nt, M, blk, ncall in prim["dec_cases"]:
        tag = "dec_%d_%d_%d" % (nt, M, blk)
        c, x, want = prim[tag + "_c"], prim[tag + "_x"], prim[tag + "_y"]
        state = np.zeros(nt - 1 + blk, np.float32)
        got = np.empty(want.size, np.float32)
        no = blk // M
        for k in range(ncall):
            xin = np.ascontiguousarray(x[k * blk:(k + 1) * blk])
            y = np.empty(no, np.float32)
            oracle.orc_fir_decimate_f32(fptr(c), nt, M, fptr(state), fptr(xin), fptr(y), blk, A0)
            got[k * no:(k + 1) * no] = y
        assert bits_equal(got, want), tag
        assert bits_equal(state[:nt - 1], prim[tag + "_state"]), tag
    # impulse through {1,2,3}: time-reversed taps come out
    assert list(prim["dec_3_1_8_y"][:4]) == [3.0, 2.0, 1.0, 0.0]


def test_fir_vectors(oracle, prim):
    for nt, blk, ncall in prim["fir_cases"]:
        tag = "fir_%d_%d" % (nt, blk)
        c, x, want = prim[tag + "_c"], prim[tag + "_x"], prim[tag + "_y"]
        state = np.zeros(nt - 1 + blk, np.float32)
        got = np.empty(want.size, np.float32)
        for k in range(ncall):
            xin = np.ascontiguousarray(x[k * blk:(k + 1) * blk])
            y = np.empty(blk, np.float32)
            oracle.orc_fir_f32(fptr(c), nt, fptr(state), fptr(xin), fptr(y), blk, A0)
            got[k * blk:(k + 1) * blk] = y
        assert bits_equal(got, want), tag
        assert bits_equal(state[:nt - 1], prim[tag + "_state"]), tag


def test_biquad_vectors(oracle, prim):
    for ns, blk, ncall in prim["biq_cases"]:
        tag = "biq_%d_%d" % (ns, blk)
        c, x, want = prim[tag + "_c"], prim[tag + "_x"], prim[tag + "_y"]
        state = np.zeros(4 * ns, np.float32)
        got = np.empty(want.size, np.float32)
        for k in range(ncall):
            xin = np.ascontiguousarray(x[k * blk:(k + 1) * blk])
            y = np.empty(blk, np.float32)
            oracle.orc_biquad_cascade_df1_f32(fptr(c), ns, fptr(state), fptr(xin), fptr(y), blk, A0)
            got[k * blk:(k + 1) * blk] = y
        assert bits_equal(got, want), tag
        assert bits_equal(state, prim[tag + "_state"]), tag
    assert list(prim["biq_1_8_y"][:4]) == [1.0, 0.5, 0.25, 0.125]          # feedback is ADDED


def test_elementwise_statistics_and_q15(oracle, prim):
    a, b = prim["ew_a"], prim["ew_b"]
    n2 = a.size
    n = n2 // 2
    out = np.empty(n2, np.float32)
    oracle.orc_cmplx_mult_cmplx_f32(fptr(a), fptr(b), fptr(out), n, A0)
    assert bits_equal(out, prim["ew_cmul"])
    mag = np.empty(n, np.float32)
    oracle.orc_cmplx_mag_f32(fptr(a), fptr(mag), n, A0)
    assert bits_equal(mag, prim["ew_mag"])
    oracle.orc_cmplx_conj_f32(fptr(a), fptr(out), n)
    assert bits_equal(out, prim["ew_conj"])                      # arm_cmplx_conj_f32 (FM discriminator)
    oracle.orc_add_f32(fptr(a), fptr(b), fptr(out), n2)
    assert bits_equal(out, prim["ew_add"])
    oracle.orc_sub_f32(fptr(a), fptr(b), fptr(out), n2)
    assert bits_equal(out, prim["ew_sub"])
    oracle.orc_abs_f32(fptr(a), fptr(out), n2)
    assert bits_equal(out, prim["ew_abs"])
    oracle.orc_scale_f32(fptr(a), 0.37, fptr(out), n2)
    assert bits_equal(out, prim["ew_scale"])
    import ctypes as C
    mx, mi = C.c_float(), C.c_uint32()
    tie = np.ascontiguousarray(prim["ew_max_in"])
    oracle.orc_max_f32(fptr(tie), n2, C.byref(mx), C.byref(mi))
    assert np.float32(mx.value) == prim["ew_max"][0] and mi.value == prim["ew_max_idx"][0] == 100
    q = np.ascontiguousarray(prim["q15_in"])
    qf = np.empty(q.size, np.float32)
    oracle.orc_q15_to_float(q.ctypes.data_as(rc.i16p), fptr(qf), q.size)
    assert bits_equal(qf, prim["q15_f"])
    fi = np.ascontiguousarray(prim["fq_in"])
    fq = np.empty(fi.size, np.int16)
    oracle.orc_float_to_q15(fptr(fi), fq.ctypes.data_as(rc.i16p), fi.size)
    assert np.array_equal(fq, prim["fq_out"])
    assert list(fq[-8:-2]) == [32767, -32768, 32767, -32768, 32767, -32768]   # saturation
    assert list(fq[-2:]) == [0, 0]                                            # truncation toward zero


def test_fm_arctangent_accuracy(oracle):
    """The build-defined arctangent of the FM discriminator (oracle/fm_atan.h = csrc/rx_device.h fm_atan2) against atan2 in double:
    the whole plane incl. the axes, the octant seams, tiny and huge magnitudes; <= 4e-7 rad."""
    rng = np.random.default_rng(7)
    pts = [(0.0, 0.0), (0.0, 1.0), (0.0, -1.0), (1.0, 0.0), (-1.0, 0.0), (1.0, 1.0), (-1.0, 1.0), (1.0, -1.0), (-1.0, -1.0),
           (1e-30, 1e-30), (1e30, -1e30), (1e-38, 1.0), (1.0, 1e-38), (3e-39, 2e-39)]
    th = rng.uniform(-np.pi, np.pi, 20000)
    r = 10.0 ** rng.uniform(-20, 20, th.size)
    pts += list(zip((r * np.sin(th)).astype(np.float32).tolist(), (r * np.cos(th)).astype(np.float32).tolist()))
    seam = np.linspace(-1e-3, 1e-3, 2001)
    pts += [(float(np.float32(np.sin(a + k * np.pi / 4))), float(np.float32(np.cos(a + k * np.pi / 4)))) for k in range(8) for a in seam]
    worst = 0.0
    for y, x in pts:
        y, x = float(np.float32(y)), float(np.float32(x))
        got = oracle.orc_fm_atan2_f32(y, x)
        want = np.arctan2(y, x) if (x != 0.0 or y != 0.0) else 0.0
        d = abs(got - want)
        if x < 0 and y == 0.0:
            d = min(d, abs(got - np.pi))          # (the sign of zero is not looked at: y = -0 gives +pi)
        worst = max(worst, d)
    assert worst <= 4e-7, worst


def test_fm_of_a_tone_is_its_frequency(oracle):
    """Known answer: a complex tone at f cycles per sample, no NCO, no decimator: audio = 2 f (phase step in half turns), from the
    second sample on (the first one sees the zero delay line: angle of z[0] * conj(0) = 0)."""
    nh = 3
    spec = rc.ChainSpec(1, 64, 1, 0, nh, 0, rc.MODE_FM, rc.ARITH_CMSIS, agc=False)
    for f in (0.01, -0.2, 0.25, 0.4999, -0.37):
        ch = rc.CpuChain(spec, "orc")
        n = np.arange(256)
        iq = np.stack([np.cos(2 * np.pi * f * n), np.sin(2 * np.pi * f * n)], axis=-1).astype(np.float32)[None]
        y = ch.process(iq)[0]
        assert y[0] == 0.0
        assert np.abs(y[1:] - 2 * f).max() < 1e-6, (f, np.abs(y[1:] - 2 * f).max())
        ch.close()


def _chain_cases():
    import importlib.util
    spec = importlib.util.spec_from_file_location("make_golden", os.path.join(G, "make_golden.py"))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m


@pytest.mark.parametrize("name", ["cfg1", "cfg2", "cfg3", "cfg4", "lsb_full", "cwr_full", "am_full", "fm_full", "fm_nodec", "global_gain"])
def test_chain_against_cmsis_composition(name):
    mg = _chain_cases()
    kind, arg, channels, nblocks, ncalls = mg.CHAIN_CASES[name]
    spec = mg.chain_spec(kind, arg, channels)
    gold = np.load(os.path.join(G, "chain_%s.npz" % name))
    ch = rc.CpuChain(spec, "orc")
    assert ch.ok()
    bs = spec.block * nblocks
    for call in range(ncalls):
        iq = rc.synth_iq(0, channels, call * bs, bs)
        assert bits_equal(ch.process(iq), gold["audio"][call]), "%s call %d" % (name, call)
    st = ch.state()
    for k in st:
        if st[k].dtype == np.float32:
            assert bits_equal(st[k], gold[k]), k
        else:
            assert np.array_equal(st[k], gold[k]), k


def test_chain_q15_slot():
    spec = rc.baseline_spec("cfg3", 2)
    ch = rc.CpuChain(spec, "orc")
    iq = rc.synth_iq(0, 2, 0, 1024)
    q = np.clip(np.trunc(iq * 32768.0), -32768, 32767).astype(np.int16)
    gold = np.load(os.path.join(G, "chain_cfg3_q15.npz"))
    assert np.array_equal(ch.process_q15(q), gold["audio"])


def test_oracle_threads_do_not_change_results():
    spec = rc.baseline_spec("cfg3", 9)
    iq = rc.synth_iq(0, 9, 0, 1024)
    a, b = rc.CpuChain(spec, "orc"), rc.CpuChain(spec, "orc")
    assert bits_equal(a.process(iq, nthreads=1), b.process(iq, nthreads=4))


def test_fma_mode_stays_within_north_star_tolerance():
    """ARITH_FMA fuses the FIR tap loops only; per DSP block it stays within 1e-5 of the CMSIS result."""
    for name in ("cfg1", "cfg2", "cfg3", "cfg4"):
        a = rc.CpuChain(rc.baseline_spec(name, 3, rc.ARITH_FMA), "orc")
        b = rc.CpuChain(rc.baseline_spec(name, 3, rc.ARITH_CMSIS), "orc")
        na = 256 // a.spec.decim
        for call in range(3):
            iq = rc.synth_iq(0, 3, call * 2048, 2048)
            ya, yb = a.process(iq), b.process(iq)
            for blk in range(8):
                assert rc.rel_err(ya[:, blk * na:(blk + 1) * na], yb[:, blk * na:(blk + 1) * na]) <= 1e-5
