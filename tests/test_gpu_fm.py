"""FM (`MODE_FM = 0x08`, Core/Inc/rxtx_if.h:41) -- the one value of the firmware's Mode enum DSP_Set_Mode could not be given
until round 3.  Build-defined chain: NCO mix and decimator as in every mode, then
audio[n] = angle(z[n] * conj(z[n-1])) / pi -- arm_cmplx_conj_f32, arm_cmplx_mult_cmplx_f32, the stated arctangent of
oracle/fm_atan.h, arm_scale_f32 -- with z[-1] taken from the delay lines of the FIR pair, which keep running.  The oracle is pinned
against the real-CMSIS composition (tests/test_oracle_vs_ref.py, tests/golden/chain_fm_*.npz); here the HIP kernels against the
oracle, bit for bit, output and state."""
import numpy as np
import pytest

import rxcommon as rc
from rxcommon import ARITH_AUTO, ARITH_CMSIS, ARITH_FMA, ARITH_SPLIT16, CpuChain, bits_equal, synth_iq

pytestmark = pytest.mark.gpu


def run(spec_g, spec_o, lengths, q15=False):
    import selenite_rx as sr
    g, o = sr.Rx(spec_g.config()), CpuChain(spec_o, "orc")
    assert o.ok()
    nch, pos = spec_g.channels, 0
    for bs in lengths:
        iq = synth_iq(0, nch, pos, bs)
        pos += bs
        if q15:
            iq16 = np.clip(np.round(iq * 20000.0), -32768, 32767).astype(np.int16)
            assert np.array_equal(g.process_q15(iq16), o.process_q15(iq16)), bs
        else:
            yg, yo = g.process(iq), o.process(iq)
            assert np.isfinite(yg).all()
            assert bits_equal(yg, yo), (bs, rc.rel_err(yg, yo))
    sg, so = g.state(), o.state()
    for key in sg:
        assert (bits_equal(sg[key], so[key]) if sg[key].dtype == np.float32 else np.array_equal(sg[key], so[key])), key
    name = g.kernel_name()
    g.close()
    return name


@pytest.mark.parametrize("shape", [(256, 4, 63), (0, 1, 127), (128, 2, 63), (256, 8, 63), (0, 1, 31), (128, 4, 127)])
@pytest.mark.parametrize("arith", [ARITH_CMSIS, ARITH_FMA, ARITH_SPLIT16, ARITH_AUTO])
@pytest.mark.parametrize("nco", ["shared", "grid", "per_channel", "off"])
def test_fm_on_the_fused_kernel(shape, arith, nco):
    """Every arithmetic mode runs the exact / fma kernel in FM (the discriminator divides by |z|: no parity bar holds on a split
    product): SPLIT16 = FMA, AUTO = CMSIS, bit-exact against the oracle of that arithmetic.  Calls of whole passes, partial passes,
    single DSP blocks."""
    nd, M, nh = shape
    nch = 23
    steps = (np.arange(nch, dtype=np.uint64) * 0x9E3779B1 % (1 << 32)).astype(np.uint32)
    kw = {"shared": dict(nco=True, nco_step_all=0x01234567), "grid": dict(nco=True, nco_steps=(steps & np.uint32(0xFF000000))),
          "per_channel": dict(nco=True, nco_steps=steps), "off": dict()}[nco]
    ref = {ARITH_SPLIT16: ARITH_FMA, ARITH_AUTO: ARITH_CMSIS}.get(arith, arith)
    mk = lambda a: rc.ChainSpec(nch, 256, M, nd, nh, 0, rc.MODE_FM, a, agc=True, **kw)
    name = run(mk(arith), mk(ref), [256, 1024, 768, 256, 4352])
    mfma = ref == ARITH_FMA and nd and M == 4                          # whole-pass calls of the fma arithmetic by 4: f32 matrix cores
    assert name == ("k_ssb_mfma<%d,%d,%d>" if mfma else "k_ssb_fused<%d,%d,%d>") % shape


@pytest.mark.parametrize("cfg", [(64, 4, 48, 31), (12, 3, 17, 9), (56, 1, 0, 5), (96, 4, 256, 63), (48, 1, 0, 127), (56, 7, 7, 3)])
@pytest.mark.parametrize("arith", [ARITH_CMSIS, ARITH_FMA])
@pytest.mark.parametrize("q15", [False, True])
def test_fm_on_the_generic_kernels_and_the_firmware_geometry(cfg, arith, q15):
    blk, M, nd, nh = cfg
    mk = lambda: rc.ChainSpec(7, blk, M, nd, nh, 0, rc.MODE_FM, arith, nco=True, nco_step_all=0x00c00000, agc=True)
    run(mk(), mk(), [blk, 3 * blk, 10 * blk, blk, 11 * blk], q15=q15)


def test_fm_of_a_tone_is_its_frequency_on_the_gpu():
    import selenite_rx as sr
    spec = rc.ChainSpec(4, 256, 1, 0, 63, 0, rc.MODE_FM, ARITH_CMSIS, agc=False)
    g = sr.Rx(spec.config())
    f = np.array([0.01, -0.2, 0.25, -0.37])
    n = np.arange(1024)
    iq = np.stack([np.cos(2 * np.pi * f[:, None] * n), np.sin(2 * np.pi * f[:, None] * n)], axis=-1).astype(np.float32)
    y = g.process(iq)
    assert (y[:, 0] == 0.0).all()                                     # z[-1] = 0: the delay line of a fresh instance
    assert np.abs(y[:, 1:] - 2 * f[:, None]).max() < 1e-6
    g.close()


def test_fm_with_a_global_gain_and_mode_switches():
    import selenite_rx as sr
    nch = 9
    kw = dict(nco=True, nco_step_all=0x01000000, agc_global=True)
    mk = lambda: rc.ChainSpec(nch, 256, 4, 256, 63, 0, rc.MODE_FM, ARITH_CMSIS, **kw)
    g, o = sr.Rx(mk().config()), CpuChain(mk(), "orc")
    pos = 0
    for mode, bs in ((rc.MODE_FM, 1024), (rc.MODE_USB, 768), (rc.MODE_FM, 256), (rc.MODE_AM, 512), (rc.MODE_FM, 2048)):
        assert g.set_mode(mode) == 0 and o.set_mode(mode) == 0
        iq = synth_iq(0, nch, pos, bs)
        pos += bs
        yo, _ = o.process_env(iq)
        assert bits_equal(g.process(iq), yo), (mode, bs)
    g.close()


def test_fm_needs_the_delay_lines_of_the_fir_pair():
    import ctypes as C
    import selenite_rx as sr
    cfg = rc.ChainSpec(2, 64, 1, 0, 0, 0, rc.MODE_FM).config()
    h = C.c_void_p()
    assert sr.lib().selenite_rx_init(C.byref(h), C.byref(cfg)) == rc.ARGUMENT_ERROR and not h
    g = sr.Rx(rc.ChainSpec(2, 64, 1, 0, 0, 0, rc.MODE_USB).config())
    assert g.set_mode(rc.MODE_FM) == rc.ARGUMENT_ERROR and g.set_mode(rc.MODE_AM) == 0
    g.close()


def test_fm_at_size():
    """16 384 channels x 4096 samples of the cfg3 chain in FM, AUTO (= the bit-exact kernel): sampled channels against the oracle."""
    import selenite_rx as sr
    nch, bs = 16384, 4096
    steps = (np.arange(nch, dtype=np.uint64) * 0x9E3779B1 % (1 << 32)).astype(np.uint32) & np.uint32(0xFF000000)
    spec = rc.ChainSpec(nch, 256, 4, 256, 63, 0, rc.MODE_FM, ARITH_AUTO, nco=True, nco_steps=steps, agc=True)
    g = sr.Rx(spec.config())
    d_in, d_out = sr.DeviceBuffer(nch * bs * 8), sr.DeviceBuffer(nch * (bs // 4) * 4)
    g.synth_device(d_in.ptr, 0, nch, 0, bs, rc.SEED)
    g.process_device(d_in.ptr, d_out.ptr, bs)
    g.sync()
    y = d_out.download((nch, bs // 4), np.float32)
    pick = np.array([0, 1, 63, 64, 4095, 8191, 8192, 12345, 16383])
    for c in pick:
        so = rc.ChainSpec(1, 256, 4, 256, 63, 0, rc.MODE_FM, ARITH_CMSIS, nco=True, nco_steps=steps[c:c + 1], agc=True)
        o = CpuChain(so, "orc")
        assert bits_equal(y[c:c + 1], o.process(synth_iq(int(c), 1, 0, bs))), c
    g.close(); d_in.free(); d_out.free()
