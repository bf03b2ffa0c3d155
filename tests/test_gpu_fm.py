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


def run(spec_g, spec_o, lengths, q15=False, bar=False):
    """bar: the plain north-star bar per DSP block instead of bit equality (AUTO on the matrix kernel)"""
    import selenite_rx as sr
    g, o = sr.Rx(spec_g.config()), CpuChain(spec_o, "orc")
    assert o.ok()
    nch, pos, na = spec_g.channels, 0, spec_g.block // spec_g.decim
    for bs in lengths:
        iq = synth_iq(0, nch, pos, bs)
        pos += bs
        if q15:
            iq16 = np.clip(np.round(iq * 20000.0), -32768, 32767).astype(np.int16)
            assert np.array_equal(g.process_q15(iq16), o.process_q15(iq16)), bs
        else:
            yg, yo = g.process(iq), o.process(iq)
            assert np.isfinite(yg).all()
            if bar:
                d = np.abs(yg.astype(np.float64) - yo).reshape(nch, -1, na).max(axis=2)
                m = np.abs(yo).reshape(nch, -1, na).max(axis=2)
                assert (d <= 1e-5 * m).all(), (bs, (d / np.maximum(m, 1e-30)).max())
            else:
                assert bits_equal(yg, yo), (bs, rc.rel_err(yg, yo))
    sg, so = g.state(), o.state()
    for key in sg:
        if bar and key in ("fir_state", "agc_gain"):
            assert rc.rel_err(sg[key], so[key]) <= 1e-5, key
            continue
        assert (bits_equal(sg[key], so[key]) if sg[key].dtype == np.float32 else np.array_equal(sg[key], so[key])), key
    if bar:
        assert g.guard_stats()["handover_blocks"] == 0
    name = g.kernel_name()
    g.close()
    return name


@pytest.mark.parametrize("shape", [(256, 4, 63), (0, 1, 127), (128, 2, 63), (256, 8, 63), (0, 1, 31), (128, 4, 127)])
@pytest.mark.parametrize("arith", [ARITH_CMSIS, ARITH_FMA, ARITH_SPLIT16, ARITH_AUTO])
@pytest.mark.parametrize("nco", ["shared", "grid", "per_channel", "off"])
def test_fm_on_the_fused_kernel(shape, arith, nco):
    """FM runs on the exact / fma kernels (the discriminator divides by |z|: no parity bar holds on a raw split product): SPLIT16 = FMA,
    bit-exact against the oracle of that arithmetic; AUTO = CMSIS bit for bit on the shapes without a split-precision decimator, and
    (round 4) the matrix kernel guarded on min|z| x max|audio| with the bit-exact rerun on the others: the plain bar on every block.
    Calls of whole passes, partial passes, single DSP blocks."""
    nd, M, nh = shape
    nch = 23
    steps = (np.arange(nch, dtype=np.uint64) * 0x9E3779B1 % (1 << 32)).astype(np.uint32)
    kw = {"shared": dict(nco=True, nco_step_all=0x01234567), "grid": dict(nco=True, nco_steps=(steps & np.uint32(0xFF000000))),
          "per_channel": dict(nco=True, nco_steps=steps), "off": dict()}[nco]
    ref = {ARITH_SPLIT16: ARITH_FMA, ARITH_AUTO: ARITH_CMSIS}.get(arith, arith)
    mk = lambda a: rc.ChainSpec(nch, 256, M, nd, nh, 0, rc.MODE_FM, a, agc=True, **kw)
    matrix = arith == ARITH_AUTO and nd and M in (2, 4, 8)              # AUTO on a shape with k_ssb_split16
    name = run(mk(arith), mk(ref), [256, 1024, 768, 256, 4352], bar=matrix)
    mfma = ref == ARITH_FMA and nd and M == 4                          # whole-pass calls of the fma arithmetic by 4: f32 matrix cores
    assert name.startswith(("k_ssb_split16<%d,%d,%d>" if matrix else "k_ssb_mfma<%d,%d,%d>" if mfma else "k_ssb_fused<%d,%d,%d>") % shape)


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


def fm_signal(nch, pos, n, dev=0.06, fa=0.004, amp=0.7, noise=0.0, seed=3):
    """nch complex FM carriers at fs/256 (the LO of the cfg3 chain puts them at 0), peak deviation `dev` cycles per INPUT sample, audio
    tone `fa` cycles per input sample with a phase of its own per channel, amplitude `amp`, optional white noise."""
    t = np.arange(pos, pos + n, dtype=np.float64)
    ph0 = np.random.default_rng(seed).uniform(0, 2 * np.pi, (nch, 1))
    phase = 2 * np.pi * t[None, :] / 256.0 + (dev / fa) * np.sin(2 * np.pi * fa * t[None, :] + ph0)
    z = amp * np.exp(1j * phase)
    if noise:
        r = np.random.default_rng(seed + 1 + pos)
        z = z + noise * (r.standard_normal(z.shape) + 1j * r.standard_normal(z.shape))
    return np.stack([z.real, z.imag], axis=-1).astype(np.float32)


@pytest.mark.parametrize("shape", [(256, 4, 63), (128, 2, 63), (256, 4, 127)])
def test_fm_in_auto_runs_a_real_fm_signal_on_the_matrix_kernel(shape):
    """Round 4: FM in SELENITE_ARITH_AUTO on the split-precision decimator.  A clean, wide FM carrier (min|z| x max|audio| well over a
    quarter of the pass maximum) stays on the matrix kernel -- nothing guarded after the start-up calls, the plain bar on every block
    against the CMSIS arithmetic; with the deviation turned down to a twentieth the discriminator's error bar (|dz| / (pi |z|) against
    1e-5 of a small audio maximum) cannot be met by a split product any more: every channel is guarded, recomputed and then HELD by
    the bit-exact kernel -- bit-exact audio (AGC off)."""
    import selenite_rx as sr
    nd, M, nh = shape
    nch, na = 24, 256 // M
    mk = lambda a: rc.ChainSpec(nch, 256, M, nd, nh, 0, rc.MODE_FM, a, nco=True, nco_step_all=0x01000000, agc=False)
    g, o = sr.Rx(mk(ARITH_AUTO).config()), CpuChain(mk(ARITH_CMSIS), "orc")
    assert g.kernel_name().startswith("k_ssb_split16<%d,%d,%d>" % shape)
    pos, bs = 0, 4096

    def call(dev):
        nonlocal pos
        iq = fm_signal(nch, pos, bs, dev=dev / M * 4)          # (the same deviation per OUTPUT sample for every decimation ratio)
        pos += bs
        yg, yo = g.process(iq), o.process(iq)
        d = np.abs(yg.astype(np.float64) - yo).reshape(nch, -1, na).max(axis=2)
        m = np.abs(yo).reshape(nch, -1, na).max(axis=2)
        assert (d <= 1e-5 * m).all(), (dev, (d / np.maximum(m, 1e-30)).max())
        return yg, yo
    for k in range(4):
        call(0.06)                                                      # start-up: recomputed, held for two clean calls, handed back
    g.guard_clear()
    for k in range(3):
        yg, yo = call(0.06)
    st = g.guard_stats()
    assert st == dict(blocks=0, channel_calls=0, rerun_channel_calls=0, handover_blocks=0), st
    assert not bits_equal(yg, yo) and np.abs(yo).max() > 0.3           # the matrix kernel, a real FM audio level (half turns per sample)
    for k in range(2):
        call(0.003)
    g.guard_clear()
    yg, yo = call(0.003)
    st = g.guard_stats()
    assert st["rerun_channel_calls"] == nch and st["handover_blocks"] == 0, st
    assert bits_equal(yg, yo)
    g.close()


def test_fm_at_size():
    """16 384 channels x 4096 samples of the cfg3 chain in FM, AUTO: sampled channels against the oracle (the synthetic signal's
    narrow deviation keeps most channels under the FM guard: recomputed by the bit-exact kernel; the others within the bar)."""
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
        yo = o.process(synth_iq(int(c), 1, 0, bs))
        d = np.abs(y[c:c + 1].astype(np.float64) - yo).reshape(-1, 64).max(axis=1)
        assert (d <= 1e-5 * np.abs(yo).reshape(-1, 64).max(axis=1)).all(), c
    g.close(); d_in.free(); d_out.free()
