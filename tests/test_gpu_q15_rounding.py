"""GPU: selenite_rx_config::q15_rounding = 1 -- the int16 audio words are those of arm_float_to_q15 built with ARM_MATH_ROUNDING
(arm_float_to_q15.c:90-101) -- on every kernel family that stores int16 audio, against the vectors the reference's own source
produced (tests/golden/q15_rounding.npz) and against the oracle on larger seeded inputs."""
import importlib.util
import os

import numpy as np
import pytest

import rxcommon as rc
from rxcommon import CpuChain

pytestmark = pytest.mark.gpu
G = rc.GOLDEN_DIR


def gpu_rx(spec):
    import selenite_rx as sr
    return sr.Rx(spec.config())


def _gen():
    spec = importlib.util.spec_from_file_location("make_q15_rounding_golden", os.path.join(G, "make_q15_rounding_golden.py"))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m


@pytest.mark.parametrize("name", ["cfg2", "cfg3", "cfg4", "global_gain", "am_full"])
@pytest.mark.parametrize("path", ["fused", "generic", "fma"])
def test_fixture(name, path):
    """bit-exact arithmetic (fused kernels and the generic per-stage kernels) and the fma kernels: with CMSIS arithmetic the words ARE the
    fixture's; the fma arithmetic moves the float audio by rounding errors, so at most one word in a few thousand sits across a boundary."""
    import selenite_rx as sr
    m = _gen()
    arg, channels, bs, ncalls = m.CASES[name]
    gold = np.load(os.path.join(G, "q15_rounding.npz"))["chain_" + name]
    spec = m.case_spec(arg, channels, arith=rc.ARITH_FMA if path == "fma" else rc.ARITH_CMSIS)
    with sr.plan_option(sr.OPT_FORCE_GENERIC, 1 if path == "generic" else 0):
        rx = sr.Rx(spec.config())
    for call in range(ncalls):
        y = rx.process_q15(m.case_input(channels, call, bs))
        if path == "fma":
            assert np.abs(y.astype(np.int32) - gold[call]).max() <= 1, (name, call, rx.kernel_name())
        else:
            assert np.array_equal(y, gold[call]), (name, call, rx.kernel_name())


SHAPES = {
    "cfg2": lambda n, **kw: rc.baseline_spec("cfg2", n, **kw),
    "cfg3": lambda n, **kw: rc.baseline_spec("cfg3", n, **kw),
    "cfg4": lambda n, **kw: rc.baseline_spec("cfg4", n, **kw),
    "cfg1": lambda n, **kw: rc.baseline_spec("cfg1", n, **kw),
    "dec8": lambda n, **kw: rc.ChainSpec(n, block=512, decim=8, nd_taps=64, nh_taps=63, mode=rc.MODE_LSB, nco=True, nco_step_all=0x00400000, **kw),
    "cw3": lambda n, **kw: rc.ChainSpec(n, block=256, decim=4, nd_taps=32, nh_taps=0, n_biquad=3, mode=rc.MODE_CW, nco=True, nco_step_all=0x01000000, **kw),
}


@pytest.mark.parametrize("shape", sorted(SHAPES))
@pytest.mark.parametrize("arith", [rc.ARITH_CMSIS, rc.ARITH_AUTO, rc.ARITH_SPLIT16])
def test_against_the_oracle_and_it_is_not_the_truncating_store(shape, arith):
    n = 70
    spec = SHAPES[shape](n, arith=arith, q15_rounding=True)
    bs = spec.block * 4
    g, o = gpu_rx(spec), CpuChain(SHAPES[shape](n, q15_rounding=True), "orc")
    t = gpu_rx(SHAPES[shape](n, arith=arith, q15_rounding=False))
    moved = 0
    for call in range(3):
        q = np.clip(np.trunc(rc.synth_iq(0, n, call * bs, bs) * 32768.0), -32768, 32767).astype(np.int16)
        y, want = g.process_q15(q), o.process_q15(q)
        if arith == rc.ARITH_CMSIS:
            assert np.array_equal(y, want), (shape, call, g.kernel_name())
        else:                                    # split-precision arithmetic: the float audio within 1e-5 of the block maximum -> one LSB (selenite_rx.h)
            assert np.abs(y.astype(np.int32) - want).max() <= 1, (shape, call, g.kernel_name())
        moved += np.count_nonzero(y != t.process_q15(q))
    assert moved > y.size // 4, (shape, g.kernel_name())      # about half of all words differ between the two builds of arm_float_to_q15


def test_f32_slots_do_not_care():
    spec_r, spec_t = rc.baseline_spec("cfg3", 5, q15_rounding=True), rc.baseline_spec("cfg3", 5)
    iq = rc.synth_iq(0, 5, 0, 2048)
    assert rc.bits_equal(gpu_rx(spec_r).process(iq), gpu_rx(spec_t).process(iq))


@pytest.mark.parametrize("name", ["tx_usb", "tx_lsb_small"])
@pytest.mark.parametrize("arith", [rc.ARITH_CMSIS, rc.ARITH_FMA, rc.ARITH_SPLIT16])
def test_tx_fixture(name, arith):
    """selenite_tx_config::q15_rounding on the TX kernels (k_tx_fused / k_tx_split16 / the generic ones): CMSIS arithmetic gives the fixture's words,
    the other arithmetics move the float I/Q by rounding errors -- at most one LSB"""
    import selenite_rx as sr
    m = _gen()
    arg, channels, bs, ncalls = m.TX_CASES[name]
    gold = np.load(os.path.join(G, "q15_rounding.npz"))["chain_" + name]
    tx = sr.Tx(m.tx_spec(arg, channels, arith=arith).config())
    tr = sr.Tx(m.tx_spec(arg, channels, rounding=False, arith=arith).config())
    moved = 0
    for call in range(ncalls):
        a = m.tx_input(channels, call, bs)
        y = tx.process_q15(a)
        if arith == rc.ARITH_CMSIS:
            assert np.array_equal(y, gold[call]), (name, call, tx.kernel_name())
        else:
            assert np.abs(y.astype(np.int32) - gold[call]).max() <= 1, (name, call, tx.kernel_name())
        moved += np.count_nonzero(y != tr.process_q15(a))
    assert moved > y.size // 4


def test_tx_init_refuses_other_values():
    import ctypes as C
    import selenite_rx as sr
    g = _gen().tx_spec({}, 2).config()
    g.q15_rounding = 7
    h = C.c_void_p()
    sr.lib().selenite_tx_init.argtypes = [C.POINTER(C.c_void_p), C.POINTER(sr.TxConfig)]
    assert sr.lib().selenite_tx_init(C.byref(h), C.byref(g)) == rc.ARGUMENT_ERROR


def test_version_1_caller_with_garbage_in_its_padding_truncates():
    """ABI version 2 (include/selenite_rx.h): q15_rounding sits where version 1's selenite_rx_config had tail padding.  A caller built against
    the version-1 header (struct_size 112) whose padding bytes are 0xFF initialises and gets the firmware's truncating arm_float_to_q15
    (arm_float_to_q15.c:117) -- the words of round 4 -- not an ARGUMENT_ERROR and not the rounding build."""
    import selenite_rx as sr
    spec = rc.baseline_spec("cfg4", 40)
    cfg = spec.config()
    cfg.struct_size = sr.CONFIG_SIZE_V1
    cfg.q15_rounding, cfg.abi_version, cfg.reserved = 0xFFFFFFFF, 0xFFFFFFFF, 0xFFFFFFFF
    rx = sr.Rx(cfg)
    o = CpuChain(spec, "orc")                                          # q15_rounding False: truncation
    orr = CpuChain(rc.baseline_spec("cfg4", 40, q15_rounding=True), "orc")
    differs = False
    for call in range(2):
        iq = rc.synth_iq(0, 40, 1024 * call, 1024)
        q = np.clip(np.trunc(iq * np.float32(32768.0)), -32768, 32767).astype(np.int16)
        y, yt, yr = rx.process_q15(q), o.process_q15(q), orr.process_q15(q)
        assert np.array_equal(y, yt), call
        differs = differs or not np.array_equal(yt, yr)
    assert differs                                                     # (the two builds of arm_float_to_q15 do differ on this input)
