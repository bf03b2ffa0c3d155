"""CPU: the ARM_MATH_ROUNDING variant of arm_float_to_q15 (selenite_rx_config::q15_rounding; arm_float_to_q15.c:90-101) -- the oracle
against the vectors the reference's own source file produced when built with that define (tests/golden/make_q15_rounding_golden.py),
and, in the build container, against that build itself on fresh inputs."""
import ctypes as C
import importlib.util
import os

import numpy as np
import pytest

import rxcommon as rc
from rxcommon import fptr

G = rc.GOLDEN_DIR


def _gen():
    spec = importlib.util.spec_from_file_location("make_q15_rounding_golden", os.path.join(G, "make_q15_rounding_golden.py"))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m


@pytest.fixture(scope="module")
def gold():
    return np.load(os.path.join(G, "q15_rounding.npz"))


def orc_round(x):
    L = rc.oracle_lib()
    L.orc_float_to_q15_rounding.argtypes = [rc.f32p, rc.i16p, C.c_uint32]
    L.orc_float_to_q15_rounding.restype = None
    q = np.empty(x.size, np.int16)
    L.orc_float_to_q15_rounding(fptr(x), q.ctypes.data_as(rc.i16p), x.size)
    return q


def test_primitive_vectors(gold):
    x = np.ascontiguousarray(gold["fq_in"])
    assert np.array_equal(_gen().primitive_input().view(np.uint32), x.view(np.uint32))     # the committed input is the generator's
    q = orc_round(x)
    assert np.array_equal(q, gold["fq_out"])
    # what the variant means: half an LSB away from zero in float before the truncation; saturating
    want = np.clip(np.trunc(x * np.float32(32768.0) + np.where(x * np.float32(32768.0) > 0, np.float32(0.5), np.float32(-0.5))),
                   -32768, 32767).astype(np.int16)
    assert np.array_equal(q, want)
    t = np.empty(x.size, np.int16)
    rc.oracle_lib().orc_float_to_q15(fptr(x), t.ctypes.data_as(rc.i16p), x.size)
    d = q.astype(np.int32) - t.astype(np.int32)
    assert set(np.unique(d)) <= {-1, 0, 1} and np.count_nonzero(d) > x.size // 3          # and it is not the truncating build


@pytest.mark.parametrize("name", ["cfg2", "cfg3", "cfg4", "global_gain", "am_full"])
def test_chain_vectors(gold, name):
    m = _gen()
    arg, channels, bs, ncalls = m.CASES[name]
    o = rc.CpuChain(m.case_spec(arg, channels), "orc")
    t = rc.CpuChain(m.case_spec(arg, channels, rounding=False), "orc")
    differs = 0
    for call in range(ncalls):
        q = m.case_input(channels, call, bs)
        y = o.process_q15(q)
        assert np.array_equal(y, gold["chain_" + name][call]), (name, call)
        differs += np.count_nonzero(y != t.process_q15(q))
    assert differs > 0
    o.close(); t.close()


@pytest.mark.skipif(not rc.ref_available(), reason="oracle/_ref not built (needs /root/reference)")
def test_oracle_against_the_rounding_build_of_the_reference():
    R = rc.ref_lib()
    rng = np.random.default_rng(31)
    x = np.concatenate([rng.uniform(-1.2, 1.2, 200000), rng.normal(0, 1e-4, 50000),
                        (rng.integers(-33000, 33000, 50000) + rng.choice([0.5, -0.5, 0.499999, 0.500001], 50000)) / 32768.0]).astype(np.float32)
    want = np.empty(x.size, np.int16)
    R.arm_float_to_q15_rounding(fptr(x), want.ctypes.data_as(rc.i16p), C.c_uint32(x.size))
    assert np.array_equal(orc_round(x), want)
    spec = rc.ChainSpec(3, block=64, decim=4, nd_taps=48, nh_taps=31, n_biquad=2, mode=rc.MODE_LSB, nco=True, nco_step_all=0x01234567,
                        bp_f0=0.05, bp_q=3.0, q15_rounding=True)
    o, r = rc.CpuChain(spec, "orc"), rc.CpuChain(spec, "ref")
    for call in range(3):
        q = _gen().case_input(3, call, 640)
        assert np.array_equal(o.process_q15(q), r.process_q15(q))
    o.close(); r.close()


@pytest.mark.parametrize("name", ["tx_usb", "tx_lsb_small"])
def test_tx_chain_vectors(gold, name):
    """selenite_tx_config::q15_rounding: the TX oracle against what the reference's arm_float_to_q15 built with ARM_MATH_ROUNDING left behind the
    real-CMSIS TX composition (oracle/ref_tx.c)"""
    m = _gen()
    arg, channels, bs, ncalls = m.TX_CASES[name]
    o = rc.TxCpuChain(m.tx_spec(arg, channels), "orc")
    t = rc.TxCpuChain(m.tx_spec(arg, channels, rounding=False), "orc")
    assert o.ok() and t.ok()
    differs = 0
    for call in range(ncalls):
        a = m.tx_input(channels, call, bs)
        y = o.process_q15(a)
        assert np.array_equal(y, gold["chain_" + name][call]), (name, call)
        differs += np.count_nonzero(y != t.process_q15(a))
    assert differs > 0
    bad = m.tx_spec(arg, channels)
    bad_cfg = bad.config()
    bad_cfg.q15_rounding = 2                                  # sits in former tail padding: garbage there is refused, not taken for a mode
    h = C.c_void_p()
    L = rc.oracle_lib()
    L.orc_tx_create.argtypes = [C.POINTER(C.c_void_p), C.POINTER(rc.TxConfig)]
    assert L.orc_tx_create(C.byref(h), C.byref(bad_cfg)) != 0
