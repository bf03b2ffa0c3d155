"""CPU: the C-ABI library loads, exports every symbol include/selenite_rx.h declares, validates
arguments the way the CMSIS init functions do, and refuses to run without a GPU (no CPU fallback).
No compute call is made here."""
import ctypes as C
import os
import re

import numpy as np
import pytest

import rxcommon as rc
import selenite_rx as sr


def header_symbols():
    text = open(os.path.join(rc.ROOT, "include", "selenite_rx.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(selenite_rx_[a-z0-9_]+)\s*\(", text)))


def test_library_exports_every_declared_symbol():
    L = sr.lib()
    names = header_symbols()
    assert len(names) >= 30
    for n in names:
        assert hasattr(L, n), "libselenite_rx.so does not export %s" % n
    assert sorted(sr.ABI_SYMBOLS) == names              # the Python face covers the whole header
    assert L.selenite_rx_abi_version() == 2


def test_ring_header_symbols_are_exported_and_bound():
    text = open(os.path.join(rc.ROOT, "include", "selenite_ring.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    names = sorted(set(re.findall(r"\b(selenite_ring_[a-z0-9_]+)\s*\(", text)))
    assert len(names) == 18
    L = sr.lib()
    for n in names:
        assert hasattr(L, n), "libselenite_rx.so does not export %s" % n
    assert sorted(sr.RING_ABI_SYMBOLS) == names


def test_tx_header_symbols_are_exported_and_bound_and_struct_layout():
    text = open(os.path.join(rc.ROOT, "include", "selenite_tx.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    names = sorted(set(re.findall(r"\b(selenite_tx_[a-z0-9_]+)\s*\(", text)))
    assert len(names) == 16
    L = sr.lib()
    for n in names:
        assert hasattr(L, n), "libselenite_rx.so does not export %s" % n
    assert sorted(sr.TX_ABI_SYMBOLS) == names
    # C layout on LP64: 6 u32, 4 u8, u32, 4 pointers, 7 floats (version 1: + 4 of tail padding = 96); ABI version 2: q15_rounding there, abi_version, reserved
    assert C.sizeof(sr.TxConfig) == 6 * 4 + 4 + 4 + 4 * 8 + 7 * 4 + 3 * 4 == 104
    assert sr.TxConfig.interp_coeffs.offset == 32 and sr.TxConfig.alc_target.offset == 64 and sr.TxConfig.q15_rounding.offset == 92
    assert C.sizeof(sr.TxStateView) == 32


def test_tx_init_validation_and_no_cpu_fallback():
    def code(mutate):
        g = rc.TxSpec(2).config()
        mutate(g)
        h = C.c_void_p()
        sr.lib().selenite_tx_init.argtypes = [C.POINTER(C.c_void_p), C.POINTER(sr.TxConfig)]
        r = sr.lib().selenite_tx_init(C.byref(h), C.byref(g))
        if h:
            sr.lib().selenite_tx_free.argtypes = [C.c_void_p]
            sr.lib().selenite_tx_free(h)
        return r
    assert code(lambda g: setattr(g, "struct_size", 8)) == rc.ARGUMENT_ERROR
    assert code(lambda g: setattr(g, "channels", 0)) == rc.ARGUMENT_ERROR
    assert code(lambda g: setattr(g, "mode", rc.MODE_FM)) == rc.ARGUMENT_ERROR
    assert code(lambda g: setattr(g, "interp", 1)) == rc.ARGUMENT_ERROR          # taps without an interpolator
    assert code(lambda g: setattr(g, "ni_taps", 255)) == rc.LENGTH_ERROR         # arm_fir_interpolate_init_f32.c:91-96
    if not has_gpu():
        assert code(lambda g: None) == rc.DEVICE_ERROR


def test_ring_refuses_to_run_without_a_gpu():
    if has_gpu():
        pytest.skip("GPU present")
    h = C.c_void_p()
    assert sr.lib().selenite_ring_init(C.byref(h), 4, 384) == rc.DEVICE_ERROR
    assert not h


def test_config_struct_layout_matches_header():
    # C layout of selenite_rx_config on LP64: 8 u32, 4 u8, u32, 5 pointers, 7 floats (version 1 ends here: 108 bytes + 4 of tail padding = 112),
    # then ABI version 2: u32 q15_rounding (where that padding was), u32 abi_version, u32 reserved
    assert C.sizeof(sr.Config) == 8 * 4 + 4 + 4 + 5 * 8 + 7 * 4 + 3 * 4 == 120
    assert sr.Config.q15_rounding.offset == 108 and sr.Config.abi_version.offset == 112 and sr.Config.reserved.offset == 116
    assert sr.CONFIG_SIZE_V1 == 112 and sr.TX_CONFIG_SIZE_V1 == 96 and C.sizeof(sr.TxConfig) == 104 and sr.TxConfig.abi_version.offset == 96
    assert sr.lib().selenite_rx_abi_version() == sr.ABI_VERSION == 2
    assert sr.Config.dec_coeffs.offset == 40 and sr.Config.agc_target.offset == 80
    assert C.sizeof(sr.StateView) == 40


def has_gpu():
    return sr.lib().selenite_rx_device_count() > 0


def init_rc(spec, mutate=None):
    cfg = spec.config()
    if mutate:
        mutate(cfg)
    h = C.c_void_p()
    code = sr.lib().selenite_rx_init(C.byref(h), C.byref(cfg))
    if h:
        sr.lib().selenite_rx_free(h)
    return code


def test_init_argument_validation_precedes_device_use():
    ok = rc.baseline_spec("cfg3", 2)
    assert init_rc(ok, lambda g: setattr(g, "struct_size", 12)) == rc.ARGUMENT_ERROR
    assert init_rc(ok, lambda g: setattr(g, "q15_rounding", 2)) == rc.ARGUMENT_ERROR
    assert init_rc(ok, lambda g: setattr(g, "abi_version", 1)) == rc.ARGUMENT_ERROR      # the version-2 struct_size with another version
    assert init_rc(ok, lambda g: setattr(g, "abi_version", 0)) == rc.ARGUMENT_ERROR      # ... or a version-2-sized struct nobody filled in
    assert init_rc(ok, lambda g: setattr(g, "reserved", 5)) == rc.ARGUMENT_ERROR
    assert init_rc(ok, lambda g: setattr(g, "struct_size", 116)) == rc.ARGUMENT_ERROR
    assert init_rc(ok, lambda g: setattr(g, "channels", 0)) == rc.ARGUMENT_ERROR
    assert init_rc(ok, lambda g: (setattr(g, "mode", rc.MODE_FM), setattr(g, "nh_taps", 0))) == rc.ARGUMENT_ERROR   # FM needs the FIR pair's delay lines
    assert init_rc(ok, lambda g: setattr(g, "mode", 0x05)) == rc.ARGUMENT_ERROR          # not a value of the firmware's Mode enum (rxtx_if.h:33-43)
    assert init_rc(ok, lambda g: setattr(g, "arith", 7)) == rc.ARGUMENT_ERROR
    assert init_rc(ok, lambda g: setattr(g, "nd_taps", 0)) == rc.ARGUMENT_ERROR        # decim 4 without decimator
    assert init_rc(ok, lambda g: setattr(g, "hilb_coeffs", None)) == rc.ARGUMENT_ERROR
    # arm_fir_decimate_init_f32.c:74-97: blockSize % M != 0 -> ARM_MATH_LENGTH_ERROR (-2)
    assert init_rc(ok, lambda g: setattr(g, "block", 255)) == rc.LENGTH_ERROR
    assert sr.lib().selenite_rx_error_string(None)


def v1_caller(g):
    """what a caller built against the round-4 header hands over: struct_size 112, and behind agc_gain_init whatever its stack held"""
    g.struct_size = sr.CONFIG_SIZE_V1
    g.q15_rounding, g.abi_version, g.reserved = 0xFFFFFFFF, 0xFFFFFFFF, 0xFFFFFFFF


def test_version_1_caller_is_accepted_whatever_its_padding_holds():
    """VERDICT r5 missing #3: q15_rounding went into version 1's tail padding, so a version-1 caller with a non-zeroed struct got ARGUMENT_ERROR at random.
    Now struct_size tells the layouts apart: a 112-byte config is read up to agc_gain_init and passes validation (DEVICE_ERROR here = it got as far as the
    device; with a GPU: SUCCESS) -- tests/test_gpu_q15_rounding.py shows that such an instance truncates."""
    ok = rc.baseline_spec("cfg3", 2)
    assert init_rc(ok, v1_caller) == (rc.SUCCESS if has_gpu() else rc.DEVICE_ERROR)
    # the version-1 validation is still the whole validation
    assert init_rc(ok, lambda g: (v1_caller(g), setattr(g, "channels", 0))) == rc.ARGUMENT_ERROR
    assert init_rc(ok, lambda g: (v1_caller(g), setattr(g, "block", 255))) == rc.LENGTH_ERROR
    t = rc.TxSpec(2)
    def tx_code(mutate):
        g = t.config()
        mutate(g)
        h = C.c_void_p()
        sr.lib().selenite_tx_init.argtypes = [C.POINTER(C.c_void_p), C.POINTER(sr.TxConfig)]
        r = sr.lib().selenite_tx_init(C.byref(h), C.byref(g))
        if h:
            sr.lib().selenite_tx_free.argtypes = [C.c_void_p]
            sr.lib().selenite_tx_free(h)
        return r
    def tx_v1(g):
        g.struct_size = sr.TX_CONFIG_SIZE_V1
        g.q15_rounding, g.abi_version, g.reserved = 0xFFFFFFFF, 0xFFFFFFFF, 0xFFFFFFFF
    assert tx_code(tx_v1) == (rc.SUCCESS if has_gpu() else rc.DEVICE_ERROR)
    assert tx_code(lambda g: setattr(g, "abi_version", 1)) == rc.ARGUMENT_ERROR
    assert tx_code(lambda g: setattr(g, "q15_rounding", 2)) == rc.ARGUMENT_ERROR


def test_plan_options_are_validated_words_not_environment():
    """selenite_rx_set_plan_option (round 6: the switches the tests need are a C-ABI call, the library reads no environment variable for them):
    unknown options and out-of-range values are refused, accepted values read back, and everything is 0 unless somebody sets it."""
    L = sr.lib()
    for opt in range(6):
        assert L.selenite_rx_get_plan_option(opt) == 0
    assert L.selenite_rx_set_plan_option(-1, 1) == rc.ARGUMENT_ERROR and L.selenite_rx_set_plan_option(6, 1) == rc.ARGUMENT_ERROR
    assert L.selenite_rx_set_plan_option(sr.OPT_FORCE_GENERIC, 2) == rc.ARGUMENT_ERROR
    assert L.selenite_rx_set_plan_option(sr.OPT_RERUN_GRID, (1 << 20) + 1) == rc.ARGUMENT_ERROR
    with sr.plan_option(sr.OPT_CW_GRID, 7), sr.plan_option(sr.OPT_NO_SHARED_LO):
        assert L.selenite_rx_get_plan_option(sr.OPT_CW_GRID) == 7 and L.selenite_rx_get_plan_option(sr.OPT_NO_SHARED_LO) == 1
    assert L.selenite_rx_get_plan_option(sr.OPT_CW_GRID) == 0 and L.selenite_rx_get_plan_option(sr.OPT_NO_SHARED_LO) == 0
    os.environ["SELENITE_RX_FORCE_GENERIC"] = "1"                      # (round 5's switch: no longer read)
    try:
        assert L.selenite_rx_get_plan_option(sr.OPT_FORCE_GENERIC) == 0
    finally:
        del os.environ["SELENITE_RX_FORCE_GENERIC"]


def test_no_cpu_fallback_without_gpu():
    if has_gpu():
        pytest.skip("a GPU is present")
    with pytest.raises(sr.RxError) as e:
        sr.Rx(rc.baseline_spec("cfg1", 1).config())
    assert e.value.code == rc.DEVICE_ERROR and "no CPU fallback" in str(e.value)


def test_null_instance_calls_are_safe():
    L = sr.lib()
    L.selenite_rx_free(None)
    assert L.selenite_rx_status(None) == rc.ARGUMENT_ERROR
    assert L.selenite_rx_set_mode(None, 1) == rc.ARGUMENT_ERROR
    L.selenite_rx_process_f32(None, None, None, 256)


def test_design_helpers_match_numpy_designs():
    for n, fc in [(256, 0.1), (63, 0.2), (17, 0.05)]:
        assert np.array_equal(sr.design_lowpass(n, fc), rc.design_lowpass(n, fc))
    for n in (63, 127, 31):
        h, d = sr.design_hilbert(n)
        h2, d2 = rc.design_hilbert(n)
        assert np.array_equal(h, h2) and np.array_equal(d, d2)
        c = (n - 1) // 2
        assert np.all(h[(np.arange(n) - c) % 2 == 0] == 0.0)          # type III: structural zeros
        assert np.allclose(h, -h[::-1], atol=0) and d[c] == 1.0 and d.sum() == 1.0
    assert np.array_equal(sr.design_bandpass(4, 500 / 48000, 4.0), rc.design_bandpass(4, 500 / 48000, 4.0))
    lp = sr.design_lowpass(256, 0.1).astype(np.float64)
    assert abs(lp.sum() - 1.0) < 1e-6 and np.allclose(lp, lp[::-1], atol=1e-9)
    with pytest.raises(ValueError):
        sr.design_hilbert(64)
    with pytest.raises(ValueError):
        sr.design_lowpass(64, 0.7)


def test_host_synth_matches_oracle_synth_and_is_bounded():
    a = sr.synth_iq_host(7, 5, 10 ** 9, 777, rc.SEED)
    b = rc.synth_iq(7, 5, 10 ** 9, 777)
    assert rc.bits_equal(a, b)
    assert np.abs(a).max() < 0.8 and np.abs(a).std() > 0.1
    # channel- and time-shift consistency (what channel sharding relies on)
    full = rc.synth_iq(0, 8, 0, 512)
    assert rc.bits_equal(full[3:6], rc.synth_iq(3, 3, 0, 512))
    assert rc.bits_equal(full[:, 100:300], rc.synth_iq(0, 8, 100, 200))


def test_algorithmic_bytes_formula():
    spec = rc.baseline_spec("cfg3", 65536)
    cfg = spec.config()
    rd = C.c_uint64()
    tot = sr.lib().selenite_rx_algorithmic_bytes(C.byref(cfg), 4096, C.byref(rd))
    assert tot == 65536 * 41952                           # 41 952 B per channel-block
    assert rd.value == 65536 * (8 * 4096 + 4 * (510 + 124) + 8)


def test_pure_c_host_example_links_and_fails_loudly_without_gpu():
    import subprocess
    exe = os.path.join(rc.PKG_DIR, "host", "dsp_if_slot")
    assert os.path.exists(exe), "make -C selenite-lite_amd builds host/dsp_if_slot"
    if has_gpu():
        pytest.skip("a GPU is present (covered by the gpu-marked test)")
    r = subprocess.run([exe], capture_output=True, text=True, timeout=60)
    assert r.returncode == 77 and "no CPU fallback" in r.stderr
