"""Randomised configurations of the fused kernels against the oracle (fixed seed): shape, arithmetic, channel count,
call lengths (whole passes, partial last passes, single DSP blocks), slot format, NCO flavour, AGC, mode.  The exact modes
must be bit-exact, output and state; SELENITE_ARITH_AUTO within the PLAIN north-star bar -- max|gpu - ref| <= 1e-5 max|ref|
per DSP block, AGC on or off, any NCO -- and raw SELENITE_ARITH_SPLIT16 within its documented input-referred bar
(1e-5 of the block maximum + 1e-6 of the input level: the blocks it counts as guarded are the ones that need the second
term); both with bit-exact mixed samples (decimator state).  SELENITE_FUZZ_CASES=<n> runs more cases."""
import os

import numpy as np
import pytest

import rxcommon as rc
from rxcommon import ARITH_CMSIS, ARITH_FMA, CpuChain, bits_equal, synth_iq

pytestmark = [pytest.mark.gpu, pytest.mark.usefixtures("auto_form")]      # every test in both forms of SELENITE_ARITH_AUTO (conftest.py)
SHAPES = [(256, 4, 63), (128, 4, 63), (256, 4, 127), (128, 4, 127), (256, 4, 31), (0, 1, 63), (0, 1, 127), (0, 1, 31),
          (256, 2, 63), (128, 2, 63), (256, 8, 63), (64, 4, 63), (128, 8, 63), (64, 2, 63), (64, 8, 63),
          (128, 4, 31), (256, 2, 127), (128, 2, 127), (256, 2, 31), (128, 2, 31)]
SPLIT_SHAPES = {(256, 4, 63), (128, 4, 63), (256, 4, 127), (128, 4, 127), (256, 4, 31), (128, 4, 31), (256, 2, 63), (128, 2, 63),
                (256, 2, 127), (128, 2, 127), (256, 2, 31), (128, 2, 31),
                (0, 1, 63), (0, 1, 127), (0, 1, 31)}


def one_case(rng, idx):
    import selenite_rx as sr
    nd, M, nh = SHAPES[int(rng.integers(len(SHAPES)))]
    odd_nd = False
    if nd and rng.integers(2):
        # round 4: ANY decimator length up to an instantiated one runs on the fused kernels (taps zero-padded in front: arm_fir_decimate_f32
        # with numTaps of its own, arm_fir_decimate_f32.c:129-284); the split-precision kernel takes the even ones
        nd = int(rng.integers(2, nd + 1))
        odd_nd = bool(nd & 1)
    arith = [ARITH_CMSIS, ARITH_FMA, rc.ARITH_SPLIT16, rc.ARITH_AUTO, rc.ARITH_AUTO][int(rng.integers(5))]
    nch = int(rng.choice([1, 2, 3, 15, 16, 17, 31, 33, 63, 64, 65, 100, 129]))
    q15 = bool(rng.integers(2))
    mode = int(rng.choice([rc.MODE_USB, rc.MODE_LSB, rc.MODE_AM, rc.MODE_DIG]))
    nco = ["off", "periodic", "table", "per_channel"][int(rng.integers(4))]
    agc = bool(rng.integers(4))                      # mostly on
    if arith == rc.ARITH_SPLIT16:
        # split16 is measured input-referred here (1e-5 of the block maximum + 1e-6 of the input level: a random LO leaves
        # most channels with an empty pass band and LSB cancels the synthetic signal's main tone -- tests/test_gpu_truth.py);
        # an AGC that rides such a residue up by tens of dB would blur that figure, and the AGC code is the exact modes'
        agc = False
    kw = dict(agc=agc)
    if nco == "periodic":
        k = int(rng.integers(1, 3)) if arith == rc.ARITH_SPLIT16 else int(rng.integers(1, 200))     # raw split16: keep the tone in band
        kw.update(nco=True, nco_step_all=k << 24)
    elif nco == "table":
        kw.update(nco=True, nco_step_all=int(rng.integers(1, 1 << 32)) | 1)
    elif nco == "per_channel":
        kw.update(nco=True, nco_steps=rng.integers(0, 1 << 32, nch, dtype=np.uint64).astype(np.uint32))
    spec_g = rc.ChainSpec(nch, 256, M, nd, nh, 0, mode, arith, **kw)
    g = sr.Rx(spec_g.config())
    tol_mode = arith in (rc.ARITH_SPLIT16, rc.ARITH_AUTO) and "split16" in g.kernel_name()
    auto = arith == rc.ARITH_AUTO
    has_split = (nd == 0 and (nd, M, nh) in SPLIT_SHAPES) or (nd >= 2 and M in (2, 4) and not odd_nd) or (nd >= 2 and M == 8 and nh == 63 and not odd_nd)      # (by 8: the by-4 matrix kernel, every second output)
    assert ("split16" in g.kernel_name()) == (arith in (rc.ARITH_SPLIT16, rc.ARITH_AUTO) and has_split), (g.kernel_name(), nd, M, nh)
    # AUTO without a matrix kernel of its own is the bit-exact kernel; raw split16 without one runs as fma
    ref_arith = ARITH_CMSIS if (tol_mode or auto) else (ARITH_FMA if arith == rc.ARITH_SPLIT16 else arith)
    o = CpuChain(rc.ChainSpec(nch, 256, M, nd, nh, 0, mode, ref_arith, **kw), "orc")
    desc = "case %d: shape %s arith %d nch %d q15 %d mode %#x nco %s agc %d kernel %s" % (idx, (nd, M, nh), arith, nch, q15, mode, nco, agc, g.kernel_name())
    assert g.kernel_name() != "generic", desc
    na = 256 // M
    pos = 0
    for _ in range(int(rng.integers(1, 4))):
        bs = 256 * int(rng.integers(1, 25))
        iq = synth_iq(0, nch, pos, bs)
        pos += bs
        if q15:
            iq16 = np.clip(np.round(iq * 20000.0), -32768, 32767).astype(np.int16)
            yg, yo = g.process_q15(iq16), o.process_q15(iq16)
            if tol_mode:
                assert np.abs(yg.astype(np.int32) - yo.astype(np.int32)).max() <= 1, desc
            else:
                assert np.array_equal(yg, yo), desc
        else:
            yg, yo = g.process(iq), o.process(iq)
            if tol_mode:
                im = np.abs(iq).reshape(nch, -1).max(axis=1)
                d = np.abs(yg.astype(np.float64) - yo).reshape(nch, -1, na).max(axis=2)
                m = np.abs(yo).reshape(nch, -1, na).max(axis=2)
                bar = 1e-5 * m if auto else 1e-5 * m + 1e-6 * im[:, None]
                assert (d <= bar).all(), desc + " worst %.3g" % (d / np.maximum(m, 1e-30)).max()
                WORST[0] = max(WORST[0], float((d / np.maximum(m, 1e-30)).max())) if auto else WORST[0]
            else:
                assert bits_equal(yg, yo), desc
    sg, so = g.state(), o.state()
    assert np.array_equal(sg["nco_phase"], so["nco_phase"]), desc
    if nd:
        assert bits_equal(sg["dec_state"], so["dec_state"]), desc
    if not tol_mode:
        for key in sg:
            assert (bits_equal(sg[key], so[key]) if sg[key].dtype == np.float32 else np.array_equal(sg[key], so[key])), desc + " " + key
    g.close()


WORST = [0.0]


def test_random_configurations_of_the_fused_kernels():
    rng = np.random.default_rng(20261002)
    for idx in range(int(os.environ.get("SELENITE_FUZZ_CASES", "150"))):
        one_case(rng, idx)
    print("worst per-block relative error of the SELENITE_ARITH_AUTO cases: %.3g (bar 1e-5)" % WORST[0])


def test_in_range_shapes_never_take_the_generic_kernels():
    """any FIR pair of up to 127 taps (dense or type-III, odd or even count, any delay FIR) behind any decimator of up
    to 256 taps by 2 / 4 / 8 (or none) runs on a fused kernel -- kernel_name is never "generic" -- and is bit-exact in the CMSIS
    arithmetic.  Random shapes and taps."""
    import selenite_rx as sr
    rng = np.random.default_rng(77)
    for idx in range(int(os.environ.get("SELENITE_FUZZ_CASES", "150")) // 3):
        M = int(rng.choice([1, 2, 4, 8]))
        nd = 0 if M == 1 else int(rng.integers(2, 257))
        nh = int(rng.integers(2, 128))
        nch = int(rng.choice([1, 3, 17, 64, 65]))
        spec = rc.ChainSpec(nch, 256, M, nd, nh | 1, 0, int(rng.choice([rc.MODE_USB, rc.MODE_LSB, rc.MODE_AM, rc.MODE_FM])), ARITH_CMSIS,
                            nco=True, nco_steps=rng.integers(0, 1 << 32, nch, dtype=np.uint64).astype(np.uint32), agc=True)
        kind = int(rng.integers(3))
        if kind == 0 and nh % 2 == 1:
            pass                                       # the design helper's type-III pair, at a length that may have no instantiation
        else:
            spec.nh_taps = nh
            spec.hilb = (rng.standard_normal(nh) * 0.2).astype(np.float32)
            spec.delay = np.zeros(nh, np.float32)
            if kind == 1:
                spec.delay[int(rng.integers(nh))] = 1.0
            else:
                spec.delay[:] = (rng.standard_normal(nh) * 0.2).astype(np.float32)
        if kind == 0 and nh % 2 == 1:
            spec2 = rc.ChainSpec(nch, 256, M, nd, nh, 0, spec.mode, ARITH_CMSIS, nco=True, nco_steps=spec.nco_steps, agc=True)
            spec = spec2
        g, o = sr.Rx(spec.config()), CpuChain(spec, "orc")
        desc = "case %d: nd %d M %d nh %d kind %d mode %#x kernel %s" % (idx, nd, M, spec.nh_taps, kind, spec.mode, g.kernel_name())
        assert g.kernel_name() != "generic", desc
        pos = 0
        for _ in range(2):
            bs = 256 * int(rng.integers(1, 12))
            iq = synth_iq(0, nch, pos, bs)
            pos += bs
            assert bits_equal(g.process(iq), o.process(iq)), desc
        sg, so = g.state(), o.state()
        for key in sg:
            assert (bits_equal(sg[key], so[key]) if sg[key].dtype == np.float32 else np.array_equal(sg[key], so[key])), desc + " " + key
        g.close()


def test_random_configurations_of_the_cw_kernel():
    """k_cw_fused: 2 / 4 / 8 biquad stages, any channel count, call lengths, slot formats, NCO flavours, CW / CW-R, both exact
    arithmetic modes: always bit-exact, output and state."""
    import selenite_rx as sr
    rng = np.random.default_rng(4242)
    for idx in range(int(os.environ.get("SELENITE_FUZZ_CASES", "150")) // 3):
        ns = int(rng.choice([2, 4, 8]))
        arith = int(rng.choice([ARITH_CMSIS, ARITH_FMA]))
        nch = int(rng.choice([1, 2, 7, 8, 15, 16, 17, 31, 32, 33, 64, 65, 100]))
        q15 = bool(rng.integers(2))
        mode = int(rng.choice([rc.MODE_CW, rc.MODE_CWR]))
        nco = ["off", "shared", "per_channel"][int(rng.integers(3))]
        kw = dict(agc=bool(rng.integers(4)))
        if nco == "shared":
            kw.update(nco=True, nco_step_all=int(rng.integers(1, 1 << 32)))
        elif nco == "per_channel":
            kw.update(nco=True, nco_steps=rng.integers(0, 1 << 32, nch, dtype=np.uint64).astype(np.uint32))
        spec = rc.ChainSpec(nch, 256, 1, 0, 0, ns, mode, arith, **kw)
        g, o = sr.Rx(spec.config()), CpuChain(spec, "orc")
        desc = "cw case %d: stages %d arith %d nch %d q15 %d mode %#x nco %s %s" % (idx, ns, arith, nch, q15, mode, nco, g.kernel_name())
        assert g.kernel_name().startswith("k_cw_fused<%d" % ns), desc
        pos = 0
        for _ in range(int(rng.integers(1, 4))):
            bs = 256 * int(rng.integers(1, 13))
            iq = synth_iq(0, nch, pos, bs)
            pos += bs
            if q15:
                iq16 = np.clip(np.round(iq * 20000.0), -32768, 32767).astype(np.int16)
                assert np.array_equal(g.process_q15(iq16), o.process_q15(iq16)), desc
            else:
                assert bits_equal(g.process(iq), o.process(iq)), desc
        sg, so = g.state(), o.state()
        for key in sg:
            assert (bits_equal(sg[key], so[key]) if sg[key].dtype == np.float32 else np.array_equal(sg[key], so[key])), desc + " " + key
        g.close()


def test_random_configurations_of_the_tx_chain():
    """TX mirror (BASELINE-like shape: ALC block 64, 256-tap interpolator by 4, 63-tap Hilbert pair): channel count, call
    lengths, slot formats, NCO flavours, ALC, modes.  Exact modes bit-exact; split16 within 1e-5 per 256 output samples
    (ALC off: same reasoning as the RX fuzz) and its ALC / NCO / Hilbert state exact."""
    import selenite_rx as sr
    rng = np.random.default_rng(777)
    for idx in range(int(os.environ.get("SELENITE_FUZZ_CASES", "150")) // 3):
        arith = [ARITH_CMSIS, ARITH_FMA, rc.ARITH_SPLIT16][int(rng.integers(3))]
        nch = int(rng.choice([1, 2, 15, 16, 17, 63, 64, 65, 100]))
        q15 = bool(rng.integers(2))
        mode = int(rng.choice([rc.MODE_USB, rc.MODE_LSB, rc.MODE_AM, rc.MODE_CW]))
        nco = ["off", "periodic", "table", "per_channel"][int(rng.integers(4))]
        kw = dict(alc=bool(rng.integers(4)) and arith != rc.ARITH_SPLIT16, mode=mode)
        if nco == "off":
            kw["nco"] = False
        elif nco == "periodic":
            kw["nco_step_all"] = int(rng.integers(1, 64)) << 24
        elif nco == "table":
            kw["nco_step_all"] = int(rng.integers(1, 1 << 32)) | 1
        else:
            kw["nco_steps"] = rng.integers(0, 1 << 32, nch, dtype=np.uint64).astype(np.uint32)
        g = sr.Tx(rc.TxSpec(nch, arith=arith, **kw).config())
        tol_mode = arith == rc.ARITH_SPLIT16 and "split16" in g.kernel_name()
        o = rc.TxCpuChain(rc.TxSpec(nch, arith=ARITH_CMSIS if tol_mode else (ARITH_FMA if arith == rc.ARITH_SPLIT16 else arith), **kw), "orc")
        desc = "tx case %d: arith %d nch %d q15 %d mode %#x nco %s alc %d %s" % (idx, arith, nch, q15, mode, nco, kw["alc"], g.kernel_name())
        pos = 0
        for _ in range(int(rng.integers(1, 4))):
            bs = 256 * int(rng.integers(1, 9))
            a = rc.synth_audio(0, nch, pos, bs)
            pos += bs
            if q15:
                a16 = np.clip(np.round(a * 20000.0), -32768, 32767).astype(np.int16)
                yg, yo = g.process_q15(a16), o.process_q15(a16)
                if tol_mode:
                    assert np.abs(yg.astype(np.int32) - yo.astype(np.int32)).max() <= 1, desc
                else:
                    assert np.array_equal(yg, yo), desc
            else:
                yg, yo = g.process(a), o.process(a)
                if tol_mode:
                    im = np.abs(a).max(axis=1)
                    d = np.abs(yg.astype(np.float64) - yo).reshape(nch, -1, 512).max(axis=2)      # 256 complex outputs
                    m = np.abs(yo).reshape(nch, -1, 512).max(axis=2)
                    assert (d <= 1e-5 * m + 1e-6 * im[:, None]).all(), desc + " worst %.3g" % (d / np.maximum(m, 1e-30)).max()
                else:
                    assert bits_equal(yg, yo), desc
        sg, so = g.state(), o.state()
        assert np.array_equal(sg["nco_phase"], so["nco_phase"]) and bits_equal(sg["fir_state"], so["fir_state"]), desc
        if not tol_mode:
            assert bits_equal(sg["interp_state"], so["interp_state"]) and bits_equal(sg["alc_gain"], so["alc_gain"]), desc
        g.close()
