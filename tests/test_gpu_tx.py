"""GPU: the TX chain (include/selenite_tx.h, csrc/tx.hip) through the C-ABI against the oracle
(oracle/tx_oracle.c, pinned against real CMSIS-DSP in tests/test_tx_oracle.py) and the committed
fixtures produced by the real-CMSIS composition.  Bit-exact in both arithmetic modes (the FMA mode
against the oracle's fmaf restatement), and within 1e-5 relative of the CMSIS arithmetic."""
import os

import numpy as np
import pytest

import rxcommon as rc
from rxcommon import ARITH_CMSIS, ARITH_FMA, bits_equal, rel_err
from golden.make_tx_golden import TX_CASES

pytestmark = pytest.mark.gpu
GOLD = os.path.join(rc.ROOT, "tests", "golden")
TOL = 1e-5   # north_star tolerance


def gpu_tx(spec):
    import selenite_rx as sr
    return sr.Tx(spec.config())


def assert_state_equal(g, o):
    sg, so = g.state(), o.state()
    for k in so:
        assert np.array_equal(sg[k].view(np.uint32), so[k].view(np.uint32)), "state %s differs" % k


@pytest.mark.parametrize("name", sorted(TX_CASES))
def test_tx_fixture_from_real_cmsis(name):
    kw, ch, nblk, ncalls = TX_CASES[name]
    fx = np.load(os.path.join(GOLD, "tx_chain_%s.npz" % name))
    spec = rc.TxSpec(ch, **kw)
    g = gpu_tx(spec)
    bs = spec.block * nblk
    for k in range(ncalls):
        assert bits_equal(g.process(rc.synth_audio(0, ch, k * bs, bs)), fx["iq"][k]), "call %d" % k
    st = g.state()
    for key in ("fir_state", "interp_state", "alc_gain"):
        assert bits_equal(st[key], fx[key]), key
    assert np.array_equal(st["nco_phase"], fx["nco_phase"])
    aq = np.clip(np.trunc(rc.synth_audio(0, ch, 0, bs) * 32768.0), -32768, 32767).astype(np.int16)
    assert np.array_equal(gpu_tx(spec).process_q15(aq), fx["q15"])


@pytest.mark.parametrize("arith", [ARITH_CMSIS, ARITH_FMA])
@pytest.mark.parametrize("mode", [rc.MODE_LSB, rc.MODE_USB, rc.MODE_CW, rc.MODE_CWR, rc.MODE_AM, rc.MODE_DIG, rc.MODE_PKT])
def test_tx_all_modes_bit_exact(mode, arith):
    spec = rc.TxSpec(37, block=64, interp=4, ni_taps=256, nh_taps=63, mode=mode, arith=arith,
                     nco_steps=(np.arange(37, dtype=np.uint64) * 0x00131313 + 0x00800000).astype(np.uint32))
    g, o = gpu_tx(spec), rc.TxCpuChain(spec, "orc")
    for k in range(3):
        a = rc.synth_audio(0, 37, k * 256, 256)
        assert bits_equal(g.process(a), o.process(a)), "call %d" % k
    assert_state_equal(g, o)


def test_tx_fma_mode_within_tolerance_of_cmsis_arithmetic():
    sf, sc = rc.TxSpec(16, arith=ARITH_FMA), rc.TxSpec(16, arith=ARITH_CMSIS)
    g, o = gpu_tx(sf), rc.TxCpuChain(sc, "orc")
    for k in range(2):
        a = rc.synth_audio(0, 16, k * 512, 512)
        yg, yo = g.process(a), o.process(a)
        for b in range(0, yo.shape[1], 256):
            assert rel_err(yg[:, b:b + 256], yo[:, b:b + 256]) <= TOL


@pytest.mark.parametrize("block,L,ni,nh", [(64, 1, 0, 0), (16, 2, 2, 1), (32, 3, 45, 15), (100, 5, 35, 3), (256, 4, 256, 127),
                                           (8, 8, 64, 129), (64, 4, 1024, 63)])
def test_tx_edge_shapes(block, L, ni, nh):
    spec = rc.TxSpec(5, block=block, interp=L, ni_taps=ni, nh_taps=nh, mode=rc.MODE_LSB, nco_step_all=0x0FEDCBA9)
    g, o = gpu_tx(spec), rc.TxCpuChain(spec, "orc")
    assert o.ok()
    for k in range(2):
        a = rc.synth_audio(3, 5, k * 3 * block, 3 * block)
        assert bits_equal(g.process(a), o.process(a))
    assert_state_equal(g, o)


def test_tx_set_mode_state_roundtrip_reset_and_errors():
    import selenite_rx as sr
    spec = rc.TxSpec(4)
    g, o = gpu_tx(spec), rc.TxCpuChain(spec, "orc")
    a0, a1 = rc.synth_audio(0, 4, 0, 128), rc.synth_audio(0, 4, 128, 128)
    for mode in (rc.MODE_USB, rc.MODE_AM, rc.MODE_LSB):
        assert g.set_mode(mode) == 0 and o.set_mode(mode) == 0
        assert bits_equal(g.process(a0), o.process(a0))
    assert g.set_mode(rc.MODE_FM) == rc.ARGUMENT_ERROR                  # instance keeps its mode
    snap = g.state()
    y = g.process(a1)
    g.set_state(snap)
    assert bits_equal(g.process(a1), y)                                 # checkpoint / resume
    g.reset()
    g.set_mode(rc.MODE_USB)
    assert bits_equal(g.process(a0), gpu_tx(spec).process(a0))
    with pytest.raises(sr.RxError) as e:
        g.process(rc.synth_audio(0, 4, 0, 100))                         # not a multiple of the ALC block
    assert e.value.code == rc.LENGTH_ERROR
    bad = rc.TxSpec(2)
    bad.ni_taps = 255                                                   # arm_fir_interpolate_init_f32.c:91-96
    with pytest.raises(sr.RxError) as e:
        gpu_tx(bad)
    assert e.value.code == rc.LENGTH_ERROR


def test_tx_large_batch_sampled_channels_and_rx_loopback():
    """16 384 channels on the device; sampled channels against the oracle; then the GPU RX chain
    demodulates what the GPU TX chain produced and the audio comes back."""
    import selenite_rx as sr
    nch, bs = 16384, 1024
    spec = rc.TxSpec(nch, alc=False)
    g = gpu_tx(spec)
    d_a, d_iq = sr.DeviceBuffer(nch * bs * 4), sr.DeviceBuffer(nch * bs * 4 * 8)
    chans = [0, 1, 63, 64, 8191, nch - 1]
    audio = np.zeros((nch, bs), np.float32)
    t = np.arange(bs)
    audio[:] = (0.4 * np.sin(2 * np.pi * 0.031 * t) + 0.2 * np.sin(2 * np.pi * 0.011 * t)).astype(np.float32)
    audio[chans] = rc.synth_audio(0, len(chans), 0, bs)
    d_a.upload(audio)
    g.process_device(d_a.ptr, d_iq.ptr, bs)
    g.sync(); g.check()
    iq = d_iq.download((nch, bs * 4, 2), np.float32)
    o = rc.TxCpuChain(rc.TxSpec(len(chans), alc=False), "orc")
    assert bits_equal(iq[chans], o.process(audio[chans]))
    rx = sr.Rx(rc.ChainSpec(nch, 256, 4, 256, 63, 0, rc.MODE_USB, ARITH_CMSIS, nco=True, nco_step_all=0x01000000, agc=False).config())
    d_y = sr.DeviceBuffer(nch * bs * 4)
    rx.process_device(d_iq.ptr, d_y.ptr, bs * 4)
    rx.sync(); rx.check()
    y = d_y.download((nch, bs), np.float32)[5000]
    a = audio[5000]
    best = max(abs(np.corrcoef(a[:bs - 300], y[s:s + bs - 300])[0, 1]) for s in range(60, 260))
    assert best > 0.98


@pytest.mark.parametrize("arith", [ARITH_CMSIS, ARITH_FMA])
@pytest.mark.parametrize("variant", ["shared_lo", "per_channel_nco", "no_nco", "no_alc", "am", "lsb"])
def test_tx_fused_kernel_bit_exact(variant, arith):
    """k_tx_fused (csrc/tx_fused.hip): every NCO flavour, ALC on and off, the sideband variants."""
    kw = dict(arith=arith)
    if variant == "per_channel_nco":
        kw["nco_steps"] = (np.arange(70, dtype=np.uint64) * 0x00100101 + 0x00400000).astype(np.uint32)
    if variant == "no_nco":
        kw["nco"] = False
    if variant == "no_alc":
        kw["alc"] = False
    if variant == "am":
        kw["mode"] = rc.MODE_AM
    if variant == "lsb":
        kw["mode"] = rc.MODE_LSB
    spec = rc.TxSpec(70, **kw)
    g, o = gpu_tx(spec), rc.TxCpuChain(spec, "orc")
    assert g.kernel_name() == "k_tx_fused<4,256,63>"
    for k, bs in enumerate((256, 1024, 512)):
        a = rc.synth_audio(0, 70, 2048 * k, bs)
        assert bits_equal(g.process(a), o.process(a)), "call %d" % k
    assert_state_equal(g, o)
    a = rc.synth_audio(0, 70, 9000, 192)                    # not a multiple of 256: generic kernel, same stream
    assert bits_equal(g.process(a), o.process(a))
    assert_state_equal(g, o)


def test_tx_fused_q15_and_against_generic_large():
    import selenite_rx as sr
    spec = rc.TxSpec(33)
    g, o = gpu_tx(spec), rc.TxCpuChain(spec, "orc")
    for k in range(2):
        aq = np.clip(np.trunc(rc.synth_audio(0, 33, k * 512, 512) * 32768.0), -32768, 32767).astype(np.int16)
        assert np.array_equal(g.process_q15(aq), o.process_q15(aq))
    nch, bs = 4096, 1024
    big = rc.TxSpec(nch)
    fused = gpu_tx(big)
    import selenite_rx as sr
    with sr.plan_option(sr.OPT_TX_FORCE_GENERIC):
        gen = gpu_tx(big)
    assert fused.kernel_name() == "k_tx_fused<4,256,63>" and gen.kernel_name() == "k_tx_generic"
    a = np.concatenate([rc.synth_audio(c0, 512, 0, bs) for c0 in range(0, nch, 512)], axis=0)
    for _ in range(2):
        assert bits_equal(fused.process(a), gen.process(a))
    sf, sg = fused.state(), gen.state()
    for k in sf:
        assert np.array_equal(sf[k].view(np.uint32), sg[k].view(np.uint32)), k


@pytest.mark.parametrize("variant", ["shared_lo", "per_channel_nco", "no_nco", "no_alc", "am", "lsb"])
def test_tx_split16_interpolator_on_the_matrix_pipe(variant):
    """SELENITE_ARITH_SPLIT16 for TX (k_tx_split16): tolerance-based output (1e-5 relative per ALC block
    of 256 output samples against the CMSIS arithmetic); ALC gain, NCO phase and both filter states are
    computed from exact f32 data and must match the FMA-mode oracle / CMSIS chain as stated."""
    kw = {}
    if variant == "per_channel_nco":
        kw["nco_steps"] = (np.arange(70, dtype=np.uint64) * 0x00100101 + 0x00400000).astype(np.uint32)
    if variant == "no_nco":
        kw["nco"] = False
    if variant == "no_alc":
        kw["alc"] = False
    if variant == "am":
        kw["mode"] = rc.MODE_AM
    if variant == "lsb":
        kw["mode"] = rc.MODE_LSB
    g = gpu_tx(rc.TxSpec(70, arith=rc.ARITH_SPLIT16, **kw))
    o = rc.TxCpuChain(rc.TxSpec(70, arith=ARITH_CMSIS, **kw), "orc")
    of = rc.TxCpuChain(rc.TxSpec(70, arith=ARITH_FMA, **kw), "orc")
    assert g.kernel_name() == "k_tx_split16<4,256,63>"
    worst = 0.0
    for k, bs in enumerate((256, 1024, 512)):
        a = rc.synth_audio(0, 70, 2048 * k, bs)
        yg, yo = g.process(a), o.process(a)
        of.process(a)
        assert np.isfinite(yg).all()
        for b in range(0, yo.shape[1], 256):
            for ch in range(70):
                worst = max(worst, rel_err(yg[ch, b:b + 256], yo[ch, b:b + 256]))
    print("tx split16 %s worst rel_err %.3g" % (variant, worst))
    assert worst <= TOL, worst
    sg, so, sf = g.state(), o.state(), of.state()
    assert bits_equal(sg["alc_gain"], so["alc_gain"]) and np.array_equal(sg["nco_phase"], so["nco_phase"])
    assert bits_equal(sg["fir_state"], so["fir_state"])
    assert bits_equal(sg["interp_state"], sf["interp_state"])          # Hilbert pair runs fused (FMA) in this mode
    assert rel_err(sg["interp_state"], so["interp_state"]) <= TOL


@pytest.mark.parametrize("alc", [False, True])
@pytest.mark.parametrize("drop", [1e-1, 1e-2, 1e-3, 1e-4])
def test_tx_split16_holds_the_plain_bar_through_level_steps(alc, drop):
    """Why TX has no parity guard (the RX kernels do): its interpolator's output sits at the level of its input by
    construction and the block exponent follows the data pass by pass, so a level step inside a call -- the audio drops by 20 ... 80 dB
    after 300 samples, stays there, comes back -- leaves every ALC block of 256 output samples within the plain 1e-5 of the CMSIS
    arithmetic (measured <= 1e-6)."""
    g = gpu_tx(rc.TxSpec(16, arith=rc.ARITH_SPLIT16, alc=alc))
    o = rc.TxCpuChain(rc.TxSpec(16, arith=ARITH_CMSIS, alc=alc), "orc")
    for k in range(4):
        a = rc.synth_audio(0, 16, 1024 * k, 1024).copy()
        if k == 1:
            a[:, 300:] *= drop
        if k == 2:
            a *= drop
        yg, yo = g.process(a), o.process(a)
        nb = yo.shape[1] // 256
        d = np.abs(yg.astype(np.float64) - yo).reshape(16, nb, -1).max(axis=2)
        m = np.abs(yo).reshape(16, nb, -1).max(axis=2)
        assert (d <= TOL * m).all(), (k, (d / np.maximum(m, 1e-30)).max())
    g.close()


def test_tx_split16_q15_silence_and_full_scale():
    g = gpu_tx(rc.TxSpec(9, arith=rc.ARITH_SPLIT16))
    o = rc.TxCpuChain(rc.TxSpec(9), "orc")
    z = np.zeros((9, 512), np.float32)
    assert bits_equal(g.process(z), o.process(z))                      # silence stays exact zero
    full = np.ones((9, 512), np.float32)
    full[:, ::2] = -1.0
    assert rel_err(g.process(full), o.process(full)) <= TOL
    gq, oq = gpu_tx(rc.TxSpec(9, arith=rc.ARITH_SPLIT16)), rc.TxCpuChain(rc.TxSpec(9), "orc")
    aq = np.clip(np.trunc(rc.synth_audio(0, 9, 0, 512) * 32768.0), -32768, 32767).astype(np.int16)
    assert np.max(np.abs(gq.process_q15(aq).astype(np.int32) - oq.process_q15(aq).astype(np.int32))) <= 1


@pytest.mark.parametrize("q15", [False, True])
def test_tx_periodic_shared_lo_in_registers_equals_the_table_path_bit_for_bit(q15):
    """k_tx_split16 keeps a shared LO of period 256 (NCO step a multiple of 2^24: the default TX spec) in registers; the
    same LO values as the per-call table, so the same bits -- output and state, calls of several lengths, a non-zero
    start phase."""
    import os
    spec = rc.TxSpec(70, arith=rc.ARITH_SPLIT16)
    assert spec.nco_step_all & 0x00FFFFFF == 0
    import selenite_rx as sr
    reg = gpu_tx(spec)
    with sr.plan_option(sr.OPT_NO_PERIODIC_LO):                       # (taken over when the instance is created)
        tab = gpu_tx(spec)
    st = reg.state()
    st["nco_phase"][:] = 0x3456789A
    reg.set_state(st)
    tab.set_state(st)
    for k, bs in enumerate((256, 1024, 512, 2048)):
        a = rc.synth_audio(0, 70, 4096 * k, bs)
        if q15:
            a = np.clip(np.round(a * 20000.0), -32768, 32767).astype(np.int16)
        ya = reg.process_q15(a) if q15 else reg.process(a)
        yb = tab.process_q15(a) if q15 else tab.process(a)
        assert np.array_equal(ya.view(np.uint32) if not q15 else ya, yb.view(np.uint32) if not q15 else yb)
    sa, sb = reg.state(), tab.state()
    for key in sa:
        assert np.array_equal(sa[key].view(np.uint32) if sa[key].dtype == np.float32 else sa[key],
                              sb[key].view(np.uint32) if sb[key].dtype == np.float32 else sb[key]), key
