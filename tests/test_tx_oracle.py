"""CPU: the TX oracle (oracle/tx_oracle.c) against the real CMSIS-DSP 1.5.3 code -- live in the build
container (oracle/_ref, skipped elsewhere) and against the committed fixtures everywhere."""
import os

import numpy as np
import pytest

import rxcommon as rc
from rxcommon import ARITH_CMSIS as A0, bits_equal, fptr
from golden.make_tx_golden import TX_CASES, prim_interp_cases

GOLD = os.path.join(rc.ROOT, "tests", "golden")


def orc_interp(oracle, c, nt, L, state, x, arith=A0):
    import ctypes as C
    oracle.orc_fir_interpolate_f32.argtypes = [rc.f32p, C.c_uint32, C.c_uint32, rc.f32p, rc.f32p, rc.f32p, C.c_uint32, C.c_int]
    y = np.empty(x.size * L, np.float32)
    oracle.orc_fir_interpolate_f32(fptr(c), nt, L, fptr(state), fptr(x), fptr(y), x.size, arith)
    return y


def test_interpolator_known_answer_and_fixture(oracle):
    fx = np.load(os.path.join(GOLD, "tx_primitives.npz"))
    y = orc_interp(oracle, np.array([4, 3, 2, 1], np.float32), 4, 2, np.zeros(5, np.float32), np.array([1, 0, 0, 0], np.float32))
    assert list(y) == [1, 2, 3, 4, 0, 0, 0, 0] and bits_equal(y, fx["interp_impulse_y"])     # b[0..3] comes out in order
    for nt, L, blk in prim_interp_cases():
        k = "interp_%d_%d_%d" % (nt, L, blk)
        s = np.zeros(nt // L - 1 + blk, np.float32)
        for xv, yv in zip(fx[k + "_x"], fx[k + "_y"]):
            assert bits_equal(orc_interp(oracle, fx[k + "_c"], nt, L, s, np.ascontiguousarray(xv)), yv), k
        assert bits_equal(s[:nt // L - 1], fx[k + "_state"]), k
    import ctypes as C
    oracle.orc_negate_f32.argtypes = [rc.f32p, rc.f32p, C.c_uint32]
    oracle.orc_offset_f32.argtypes = [rc.f32p, C.c_float, rc.f32p, C.c_uint32]
    v = fx["neg_x"]
    n, o = np.empty_like(v), np.empty_like(v)
    oracle.orc_negate_f32(fptr(v), fptr(n), v.size)
    oracle.orc_offset_f32(fptr(v), 0.5, fptr(o), v.size)
    assert bits_equal(n, fx["neg_y"]) and bits_equal(o, fx["off_y"])          # incl. -(0.0) = -0.0


@pytest.mark.parametrize("nt,L,blk", [(256, 4, 64), (256, 4, 3), (45, 3, 7), (9, 9, 4), (128, 2, 33), (6, 1, 10)])
def test_interpolator_random_vs_real_cmsis(oracle, ref, nt, L, blk):
    rng = np.random.default_rng(nt * 31 + L)
    c = rng.standard_normal(nt).astype(np.float32)
    sa, sb = np.zeros(nt // L - 1 + blk, np.float32), np.zeros(nt // L - 1 + blk, np.float32)
    for _ in range(5):
        x = rng.standard_normal(blk).astype(np.float32)
        yb = np.empty(blk * L, np.float32)
        ref.ref_fir_interpolate(fptr(c), nt, L, fptr(sb), fptr(x), fptr(yb), blk)
        assert bits_equal(orc_interp(oracle, c, nt, L, sa, x), yb)
        assert bits_equal(sa[:nt // L - 1], sb[:nt // L - 1])


def test_interpolator_init_length_error_matches_cmsis(ref):
    assert ref.ref_fir_interpolate_init_status(256, 4, 64) == 0
    assert ref.ref_fir_interpolate_init_status(255, 4, 64) == rc.LENGTH_ERROR       # arm_fir_interpolate_init_f32.c:91-96
    bad = rc.TxSpec(2, ni_taps=256, interp=4)
    bad.ni_taps = 255
    assert rc.TxCpuChain(bad, "orc").rc == rc.LENGTH_ERROR


@pytest.mark.parametrize("name", sorted(TX_CASES))
def test_tx_chain_fixture(name):
    kw, ch, nblk, ncalls = TX_CASES[name]
    fx = np.load(os.path.join(GOLD, "tx_chain_%s.npz" % name))
    spec = rc.TxSpec(ch, **kw)
    o = rc.TxCpuChain(spec, "orc")
    assert o.ok()
    bs = spec.block * nblk
    for k in range(ncalls):
        assert bits_equal(o.process(rc.synth_audio(0, ch, k * bs, bs)), fx["iq"][k]), "call %d" % k
    st = o.state()
    for key in ("fir_state", "interp_state", "alc_gain"):
        assert bits_equal(st[key], fx[key]), key
    assert np.array_equal(st["nco_phase"], fx["nco_phase"])
    aq = np.clip(np.trunc(rc.synth_audio(0, ch, 0, bs) * 32768.0), -32768, 32767).astype(np.int16)
    assert np.array_equal(rc.TxCpuChain(spec, "orc").process_q15(aq), fx["q15"])


@pytest.mark.parametrize("mode", [rc.MODE_LSB, rc.MODE_USB, rc.MODE_CW, rc.MODE_CWR, rc.MODE_AM, rc.MODE_DIG, rc.MODE_PKT])
def test_tx_chain_all_modes_vs_real_cmsis(ref, mode):
    spec = rc.TxSpec(3, block=32, interp=4, ni_taps=64, nh_taps=31, mode=mode,
                     nco_steps=np.array([0x01000000, 0x00345678, 0xFF000000], np.uint32))
    o, r = rc.TxCpuChain(spec, "orc"), rc.TxCpuChain(spec, "ref")
    assert o.ok() and r.ok()
    for k in range(3):
        a = rc.synth_audio(5, 3, k * 96, 96)
        assert bits_equal(o.process(a), r.process(a))
    so, sr_ = o.state(), r.state()
    for key in so:
        assert np.array_equal(so[key].view(np.uint32), sr_[key].view(np.uint32)), key


def test_tx_then_rx_loopback_recovers_the_audio_band():
    """Sanity of the [build-defined] pair: SSB-modulate, demodulate with the RX oracle at the same LO,
    and the audio comes back (delayed, filtered) -- correlation, not parity."""
    n = 4096
    t = np.arange(n)
    audio = (0.4 * np.sin(2 * np.pi * 0.031 * t) + 0.2 * np.sin(2 * np.pi * 0.011 * t)).astype(np.float32)[None, :]
    tx = rc.TxCpuChain(rc.TxSpec(1, block=64, interp=4, ni_taps=256, nh_taps=63, alc=False), "orc")
    rxs = rc.ChainSpec(1, 256, 4, 256, 63, 0, rc.MODE_USB, A0, nco=True, nco_step_all=0x01000000, agc=False)
    iq = tx.process(audio)
    # the RX LO (cos, -sin) is already the conjugate of the TX LO (cos, +sin): same step on both sides
    y = rc.CpuChain(rxs, "orc").process(iq)[0]
    lag = 31 + 31 + (255 + 255) // 8            # Hilbert delays + the two 256-tap filters at the audio rate
    a, b = audio[0, :n - 2 * lag], y[lag:n - lag]
    best = max(abs(np.corrcoef(a[:len(b) - s], b[s:])[0, 1]) for s in range(0, 2 * lag))
    assert best > 0.98
