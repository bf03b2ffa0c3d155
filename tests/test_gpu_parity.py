"""GPU parity: HIP path (through the C-ABI of libselenite_rx.so) vs the CPU oracle on the same
seeded synthetic I/Q.  Sizes are what the oracle finishes in seconds.

Bars (BASELINE.json north_star: 1e-5 relative):
  * ARITH_CMSIS: bit-exact (0 ULP) vs the oracle, which is bit-exact vs CMSIS-DSP 1.5.3;
  * ARITH_FMA:   bit-exact vs the oracle's fmaf restatement AND rel_err <= 1e-5 vs the CMSIS
                 arithmetic (rel_err = max|a-b| / max|ref|).
"""
import os

import numpy as np
import pytest

import rxcommon as rc
from rxcommon import (ARITH_CMSIS, ARITH_FMA, MODE_AM, MODE_CW, MODE_CWR, MODE_DIG, MODE_LSB, MODE_PKT,
                      MODE_USB, ChainSpec, CpuChain, baseline_spec, bits_equal, rel_err, synth_iq)

pytestmark = pytest.mark.gpu

TOL = 1e-5   # north_star tolerance


def gpu_rx(spec):
    import selenite_rx as sr
    return sr.Rx(spec.config())


def run_pair(spec, nblocks=4, ncalls=3, first_channel=0):
    """Stream ncalls calls of nblocks DSP blocks through GPU and oracle; yield outputs per call."""
    g = gpu_rx(spec)
    o = CpuChain(spec, "orc")
    assert o.ok()
    bs = spec.block * nblocks
    outs = []
    for call in range(ncalls):
        iq = synth_iq(first_channel, spec.channels, call * bs, bs)
        outs.append((g.process(iq), o.process(iq)))
    return g, o, outs


def assert_state_equal(g, o):
    sg, so = g.state(), o.state()
    for k in sg:
        if sg[k].dtype == np.float32:
            assert bits_equal(sg[k], so[k]), "state %s differs" % k
        else:
            assert np.array_equal(sg[k], so[k]), "state %s differs" % k


@pytest.mark.parametrize("name", ["cfg1", "cfg2", "cfg3", "cfg4"])
@pytest.mark.parametrize("arith", [ARITH_CMSIS, ARITH_FMA])
def test_baseline_configs_bit_exact(name, arith):
    spec = baseline_spec(name, 5, arith)
    g, o, outs = run_pair(spec)
    for yg, yo in outs:
        assert np.isfinite(yg).all()
        assert bits_equal(yg, yo), "%s arith=%d rel_err=%g" % (name, arith, rel_err(yg, yo))
    assert_state_equal(g, o)


@pytest.mark.parametrize("name", ["cfg1", "cfg2", "cfg3", "cfg4"])
def test_fma_mode_within_tolerance_of_cmsis(name):
    spec_f = baseline_spec(name, 4, ARITH_FMA)
    spec_c = baseline_spec(name, 4, ARITH_CMSIS)
    g = gpu_rx(spec_f)
    o = CpuChain(spec_c, "orc")
    bs = spec_f.block * 8
    for call in range(3):
        iq = synth_iq(0, 4, call * bs, bs)
        yg, yo = g.process(iq), o.process(iq)
        for b in range(8):                       # per DSP block, as SURVEY.md 8d states the gate
            na = spec_f.block // spec_f.decim
            e = rel_err(yg[:, b * na:(b + 1) * na], yo[:, b * na:(b + 1) * na])
            assert e <= TOL, "%s block %d rel_err %g" % (name, b, e)


@pytest.mark.parametrize("mode", [MODE_LSB, MODE_USB, MODE_CW, MODE_CWR, MODE_AM, MODE_DIG, MODE_PKT])
@pytest.mark.parametrize("arith", [ARITH_CMSIS, ARITH_FMA])
def test_all_modes_full_chain(mode, arith):
    spec = ChainSpec(3, 64, decim=4, nd_taps=48, nh_taps=31, n_biquad=2, mode=mode, arith=arith,
                     nco=True, nco_step_all=0x01234567, agc=True, bp_f0=0.05, bp_q=3.0)
    g, o, outs = run_pair(spec, nblocks=5, ncalls=2)
    for yg, yo in outs:
        assert bits_equal(yg, yo), "mode %d arith %d rel_err %g" % (mode, arith, rel_err(yg, yo))
    assert_state_equal(g, o)


def test_set_mode_keeps_state_and_matches():
    spec = ChainSpec(2, 128, decim=2, nd_taps=33, nh_taps=63, n_biquad=4, mode=MODE_USB,
                     nco=True, nco_step_all=0x00400000, agc=True)
    g = gpu_rx(spec)
    o = CpuChain(spec, "orc")
    n0 = 0
    for mode in [MODE_USB, MODE_LSB, MODE_CW, MODE_AM, rc.MODE_FM, MODE_CWR, rc.MODE_FM, MODE_USB]:
        assert g.set_mode(mode) == 0 and o.set_mode(mode) == 0
        iq = synth_iq(0, 2, n0, 256)
        n0 += 256
        assert bits_equal(g.process(iq), o.process(iq)), "after set_mode(%d)" % mode
    assert g.set_mode(0x05) == rc.ARGUMENT_ERROR                # not a Mode value: the instance keeps its mode
    assert_state_equal(g, o)


def test_per_channel_nco_steps_and_ragged_sizes():
    steps = np.array([0, 1, 0x80000000, 0xFFFFFFFF, 0x12345678, 0x00010000, 77], np.uint32)
    spec = ChainSpec(7, 12, decim=3, nd_taps=17, nh_taps=9, mode=MODE_LSB, nco=True, nco_steps=steps, agc=True)
    g, o, outs = run_pair(spec, nblocks=3, ncalls=4)
    for yg, yo in outs:
        assert bits_equal(yg, yo)
    assert_state_equal(g, o)


@pytest.mark.parametrize("nd,M,nh", [(1, 1, 1), (2, 2, 3), (5, 1, 0), (0, 1, 5), (300, 4, 65), (64, 8, 33), (7, 7, 0)])
def test_edge_tap_counts(nd, M, nh):
    spec = ChainSpec(3, 56, decim=M, nd_taps=nd, nh_taps=nh, mode=MODE_USB, nco=False, agc=True)
    g, o, outs = run_pair(spec, nblocks=2, ncalls=3)
    for yg, yo in outs:
        assert bits_equal(yg, yo), "nd=%d M=%d nh=%d rel_err %g" % (nd, M, nh, rel_err(yg, yo))
    assert_state_equal(g, o)


def test_no_agc_no_nco_passthrough_shapes():
    spec = ChainSpec(2, 32, mode=MODE_USB, agc=False)       # audio = I rail
    g = gpu_rx(spec)
    iq = synth_iq(0, 2, 0, 64)
    y = g.process(iq)
    assert bits_equal(y, iq[:, :, 0])


def test_silence_and_full_scale_inputs():
    spec = baseline_spec("cfg3", 2)
    g, o = gpu_rx(spec), CpuChain(spec, "orc")
    z = np.zeros((2, 1024, 2), np.float32)
    assert bits_equal(g.process(z), o.process(z))            # env floor path of the AGC
    full = np.ones((2, 1024, 2), np.float32)
    full[:, ::2, :] = -1.0
    assert bits_equal(g.process(full), o.process(full))
    assert_state_equal(g, o)


def test_q15_slot_format():
    spec = baseline_spec("cfg3", 3)
    g, o = gpu_rx(spec), CpuChain(spec, "orc")
    for call in range(2):
        iq = synth_iq(0, 3, call * 1024, 1024)
        q = np.clip(np.trunc(iq * 32768.0), -32768, 32767).astype(np.int16)
        assert np.array_equal(g.process_q15(q), o.process_q15(q))
    assert_state_equal(g, o)


def test_global_gain_single_gpu():
    spec = ChainSpec(6, 64, decim=2, nd_taps=21, nh_taps=15, mode=MODE_USB, nco=True,
                     nco_step_all=0x02000000, agc=True, agc_global=True)
    g, o, outs = run_pair(spec, nblocks=4, ncalls=3)
    for yg, yo in outs:
        assert bits_equal(yg, yo)
    assert_state_equal(g, o)
    gains = g.state()["agc_gain"]
    assert np.all(gains == gains[0])


def test_state_roundtrip_and_reset():
    spec = baseline_spec("cfg3", 2)
    g = gpu_rx(spec)
    iq0, iq1 = synth_iq(0, 2, 0, 512), synth_iq(0, 2, 512, 512)
    g.process(iq0)
    snap = g.state()
    y1 = g.process(iq1)
    g.set_state(snap)
    assert bits_equal(g.process(iq1), y1)                     # checkpoint / resume
    g.reset()
    g2 = gpu_rx(spec)
    assert bits_equal(g.process(iq0), g2.process(iq0))


def test_block_partition_invariance_of_fir_state():
    """CMSIS FIRs carry exact state: one call of 8 blocks == 8 calls of 1 block."""
    spec = baseline_spec("cfg3", 2)
    a, b = gpu_rx(spec), gpu_rx(spec)
    iq = synth_iq(0, 2, 0, 256 * 8)
    ya = a.process(iq)
    yb = np.concatenate([b.process(iq[:, i * 256:(i + 1) * 256]) for i in range(8)], axis=1)
    assert bits_equal(ya, yb)


def test_bad_block_size_sets_sticky_status():
    import selenite_rx as sr
    spec = baseline_spec("cfg1", 1)
    g = gpu_rx(spec)
    iq = synth_iq(0, 1, 0, 100)
    with pytest.raises(sr.RxError) as e:
        g.process(iq)
    assert e.value.code == rc.LENGTH_ERROR


def test_device_synth_matches_host():
    import selenite_rx as sr
    spec = baseline_spec("cfg1", 1)
    g = gpu_rx(spec)
    nch, ns = 37, 1000
    buf = sr.DeviceBuffer(nch * ns * 8)
    g.synth_device(buf.ptr, 11, nch, 123456789, ns, rc.SEED)
    g.sync()
    d = buf.download((nch, ns, 2), np.float32)
    assert bits_equal(d, synth_iq(11, nch, 123456789, ns))


def test_generic_and_fused_paths_agree():
    """Every fused kernel is cross-checked against the generic kernels on the GPU itself, at a
    size the CPU oracle would not finish quickly."""
    import selenite_rx as sr
    for name in ["cfg2", "cfg3", "cfg4"]:
        spec = baseline_spec(name, 300)
        iq = synth_iq(0, 300, 0, 2048)
        fused = gpu_rx(spec)
        with sr.plan_option(sr.OPT_FORCE_GENERIC):
            gen = gpu_rx(spec)
        assert gen.kernel_name() == "generic"
        for _ in range(2):
            assert bits_equal(fused.process(iq), gen.process(iq)), name


def test_shared_lo_table_path_equals_per_channel_nco():
    """When every channel shares step and phase the LO is computed once per call (L2 table); it must
    be indistinguishable from the per-channel in-kernel NCO, also across set_state/reset."""
    spec = baseline_spec("cfg3", 5)
    shared = gpu_rx(spec)
    import selenite_rx as sr
    with sr.plan_option(sr.OPT_NO_SHARED_LO):
        per_ch = gpu_rx(spec)
    o = CpuChain(spec, "orc")
    for call in range(3):
        iq = synth_iq(0, 5, call * 1024, 1024)
        ya, yb, yo = shared.process(iq), per_ch.process(iq), o.process(iq)
        assert bits_equal(ya, yo) and bits_equal(yb, yo)
    assert_state_equal(shared, o)
    # make the phases differ per channel: the library must fall back to the per-channel NCO
    st = shared.state()
    st["nco_phase"] = (st["nco_phase"] + np.arange(5, dtype=np.uint32) * np.uint32(0x10000000)).astype(np.uint32)
    shared.set_state(st)
    o.L.orc_rx_set_state(o.h, rc.C.byref(rc.state_view(st)))
    iq = synth_iq(0, 5, 3072, 1024)
    assert bits_equal(shared.process(iq), o.process(iq))
    # and back to a common phase
    st = shared.state()
    st["nco_phase"][:] = 0x12345678
    shared.set_state(st)
    o.L.orc_rx_set_state(o.h, rc.C.byref(rc.state_view(st)))
    iq = synth_iq(0, 5, 4096, 2048)
    assert bits_equal(shared.process(iq), o.process(iq))
    assert_state_equal(shared, o)


@pytest.mark.parametrize("arith", [ARITH_CMSIS, ARITH_FMA])
@pytest.mark.parametrize("variant", ["shared_lo", "per_channel_nco", "no_nco", "no_agc", "cwr"])
def test_cw_fused_kernel_bit_exact(arith, variant):
    """k_cw_fused (systolic 16 channels x 4 stages per wavefront) against the oracle."""
    kw = dict(block=256, decim=1, nd_taps=0, nh_taps=0, n_biquad=4, mode=MODE_CW, arith=arith,
              nco=True, nco_step_all=0x00800000, agc=True)
    if variant == "per_channel_nco":
        kw["nco_steps"] = (np.arange(48, dtype=np.uint32) * np.uint32(0x00123457) + np.uint32(0x00400000))
    elif variant == "no_nco":
        kw["nco"] = False
    elif variant == "no_agc":
        kw["agc"] = False
    elif variant == "cwr":
        kw["mode"] = MODE_CWR
    spec = ChainSpec(48, **kw)
    g, o = gpu_rx(spec), CpuChain(spec, "orc")
    assert g.kernel_name() == "k_cw_fused<4,256>"
    for call in range(3):
        iq = synth_iq(0, 48, call * 768, 768)
        yg, yo = g.process(iq), o.process(iq)
        assert bits_equal(yg, yo), "%s call %d rel_err %g" % (variant, call, rel_err(yg, yo))
    assert_state_equal(g, o)


def test_cw_fused_q15_and_vs_generic_large():
    spec = baseline_spec("cfg4", 64)
    g, o = gpu_rx(spec), CpuChain(spec, "orc")
    iq = synth_iq(0, 64, 0, 1024)
    q = np.clip(np.trunc(iq * 32768.0), -32768, 32767).astype(np.int16)
    assert np.array_equal(g.process_q15(q), o.process_q15(q))
    # large: fused vs generic on the GPU
    spec = baseline_spec("cfg4", 1024)
    iq = synth_iq(0, 1024, 0, 2048)
    fused = gpu_rx(spec)
    import selenite_rx as sr
    with sr.plan_option(sr.OPT_FORCE_GENERIC):
        gen = gpu_rx(spec)
    assert fused.kernel_name() == "k_cw_fused<4,256>" and gen.kernel_name() == "generic"
    for _ in range(2):
        assert bits_equal(fused.process(iq), gen.process(iq))
    sf, sg = fused.state(), gen.state()
    for k in sf:
        assert np.array_equal(sf[k].view(np.uint32), sg[k].view(np.uint32)), k


def test_split16_mode_within_north_star_tolerance():
    """SELENITE_ARITH_SPLIT16: decimator as f16 hi/lo split MFMA product.  Not bit-exact by
    construction; the bar is the north star's: max|gpu-ref| <= 1e-5 * max|ref| per DSP block, against
    the CMSIS arithmetic.  The f32 streaming state of the decimator must still be bit-exact."""
    import selenite_rx as sr
    nch = 96
    spec_s = baseline_spec("cfg3", nch, rc.ARITH_SPLIT16)
    spec_c = baseline_spec("cfg3", nch, ARITH_CMSIS)
    g = gpu_rx(spec_s)
    assert g.kernel_name() == "k_ssb_split16<256,4,63>"
    o = CpuChain(spec_c, "orc")
    worst = 0.0
    na = spec_c.block // spec_c.decim
    for call in range(3):
        iq = synth_iq(0, nch, call * 4096, 4096)
        yg, yo = g.process(iq), o.process(iq)
        assert np.isfinite(yg).all()
        for ch in range(nch):
            for b in range(yo.shape[1] // na):
                e = rel_err(yg[ch, b * na:(b + 1) * na], yo[ch, b * na:(b + 1) * na])
                worst = max(worst, e)
    print("split16 worst per-channel-block rel_err = %.3g" % worst)
    assert worst <= TOL, worst
    sg, so = g.state(), o.state()
    assert bits_equal(sg["dec_state"], so["dec_state"])          # mixed samples are exact f32 in every mode
    assert np.array_equal(sg["nco_phase"], so["nco_phase"])
    assert rel_err(sg["fir_state"], so["fir_state"]) <= TOL
    assert np.allclose(sg["agc_gain"], so["agc_gain"], rtol=1e-5, atol=0)


def test_split16_full_scale_and_silence_and_q15():
    spec_s = baseline_spec("cfg3", 8, rc.ARITH_SPLIT16)
    spec_c = baseline_spec("cfg3", 8, ARITH_CMSIS)
    g, o = gpu_rx(spec_s), CpuChain(spec_c, "orc")
    z = np.zeros((8, 1024, 2), np.float32)
    assert bits_equal(g.process(z), o.process(z))
    full = np.ones((8, 2048, 2), np.float32)
    full[:, ::3, :] = -1.0
    yg, yo = g.process(full), o.process(full)
    assert rel_err(yg, yo) <= TOL
    g2, o2 = gpu_rx(spec_s), CpuChain(spec_c, "orc")
    iq = synth_iq(0, 8, 0, 2048)
    q = np.clip(np.trunc(iq * 32768.0), -32768, 32767).astype(np.int16)
    a, b = g2.process_q15(q).astype(np.int32), o2.process_q15(q).astype(np.int32)
    assert np.abs(a - b).max() <= 1                                   # one q15 LSB


def test_mfma_fma_kernel_is_exercised_and_bit_exact():
    """FMA mode with the f32 MFMA decimator (k_ssb_mfma): bit-exact vs the oracle's fmaf chain."""
    spec = baseline_spec("cfg3", 24, ARITH_FMA)
    g, o = gpu_rx(spec), CpuChain(spec, "orc")
    assert g.kernel_name() == "k_ssb_mfma<256,4,63>"
    for call in range(3):
        iq = synth_iq(0, 24, call * 2048, 2048)
        assert bits_equal(g.process(iq), o.process(iq))
    assert_state_equal(g, o)


# ---- global-gain variant (north_star (e): one envelope across every channel) on the fused kernels ----
@pytest.mark.gpu
@pytest.mark.parametrize("name,arith,kernel", [
    ("cfg3", ARITH_CMSIS, "k_ssb_fused<256,4,63>"),
    ("cfg3", ARITH_FMA, "k_ssb_mfma<256,4,63>"),
    ("cfg2", ARITH_CMSIS, "k_ssb_fused<0,1,127>"),
    ("cfg4", ARITH_CMSIS, "k_cw_fused<4,256>"),
    ("cfg4", ARITH_FMA, "k_cw_fused<4,256>"),
])
def test_global_gain_on_fused_kernels_bit_exact(name, arith, kernel):
    spec = baseline_spec(name, 80, arith, agc_global=True)
    g = gpu_rx(spec)
    assert g.kernel_name() == kernel
    o = CpuChain(spec, "orc")
    for call in range(3):
        iq = synth_iq(0, 80, call * 1024, 1024)
        assert bits_equal(g.process(iq), o.process(iq)), "call %d" % call
    assert_state_equal(g, o)
    gains = g.state()["agc_gain"]
    assert np.all(gains == gains[0])


@pytest.mark.gpu
def test_global_gain_q15_slots_and_split16():
    spec = baseline_spec("cfg3", 33, ARITH_CMSIS, agc_global=True)
    g, o = gpu_rx(spec), CpuChain(spec, "orc")
    for call, bs in enumerate((512, 1024, 2048)):
        iq = synth_iq(0, 33, 4096 * call, bs)
        q = np.clip(np.trunc(iq * 32768.0), -32768, 32767).astype(np.int16)
        assert np.array_equal(g.process_q15(q), o.process_q15(q))
    assert_state_equal(g, o)
    # split16: tolerance-based like the per-channel variant
    spec_s = baseline_spec("cfg3", 64, rc.ARITH_SPLIT16, agc_global=True)
    gs, oc = gpu_rx(spec_s), CpuChain(baseline_spec("cfg3", 64, ARITH_CMSIS, agc_global=True), "orc")
    assert gs.kernel_name() == "k_ssb_split16<256,4,63>"
    for call in range(2):
        iq = synth_iq(0, 64, call * 2048, 2048)
        yg, yo = gs.process(iq), oc.process(iq)
        for b in range(yo.shape[1] // 64):
            assert rel_err(yg[:, b * 64:(b + 1) * 64], yo[:, b * 64:(b + 1) * 64]) <= TOL


@pytest.mark.gpu
def test_global_gain_two_shards_on_one_gpu_match_unsharded_oracle():
    """Two instances own channels [0,40) and [40,80); the envelope exchange (an all-reduce MAX between
    ranks in bench.py) is done on the host here.  Result must equal the unsharded oracle bit for bit."""
    import selenite_rx as sr
    nch, half, bs = 80, 40, 1024
    spec_all = baseline_spec("cfg3", nch, ARITH_CMSIS, agc_global=True)
    o = CpuChain(spec_all, "orc")
    shards = [gpu_rx(baseline_spec("cfg3", half, ARITH_CMSIS, agc_global=True)) for _ in range(2)]
    nblk = bs // spec_all.block
    d_in = [sr.DeviceBuffer(half * bs * 8) for _ in range(2)]
    d_out = [sr.DeviceBuffer(half * (bs // 4) * 4) for _ in range(2)]
    d_env = [sr.DeviceBuffer(nblk * 4) for _ in range(2)]
    for call in range(3):
        iq = synth_iq(0, nch, call * bs, bs)
        for r in range(2):
            d_in[r].upload(iq[r * half:(r + 1) * half])
            shards[r].global_phase1(d_in[r].ptr, d_out[r].ptr, d_env[r].ptr, bs)
            shards[r].sync(); shards[r].check()
        env = np.maximum(d_env[0].download((nblk,), np.float32), d_env[1].download((nblk,), np.float32))
        ys = []
        for r in range(2):
            d_env[r].upload(env)
            shards[r].global_phase2(d_out[r].ptr, d_env[r].ptr, bs)
            shards[r].sync(); shards[r].check()
            ys.append(d_out[r].download((half, bs // 4), np.float32))
        assert bits_equal(np.concatenate(ys, axis=0), o.process(iq)), "call %d" % call


@pytest.mark.gpu
@pytest.mark.parametrize("name,arith,kernel", [
    ("cfg3", ARITH_CMSIS, "k_ssb_fused<256,4,63>"), ("cfg3", ARITH_FMA, "k_ssb_mfma<256,4,63>"),
    ("cfg2", ARITH_CMSIS, "k_ssb_fused<0,1,127>"), ("cfg1", ARITH_FMA, "k_ssb_fused<0,1,63>"),
])
def test_am_mode_on_fused_kernels_bit_exact_and_hilbert_state_untouched(name, arith, kernel):
    """DSP_Set_Mode(MODE_AM) (rxtx_if.h:38): arm_cmplx_mag_f32 of the decimated rails.  The fused
    kernels serve it; the Hilbert pair's state must stay exactly as the last SSB block left it, so
    that USB -> AM -> USB continues like the CMSIS composition."""
    spec = baseline_spec(name, 48, arith)
    g, o = gpu_rx(spec), CpuChain(spec, "orc")
    seq = [rc.MODE_USB, MODE_AM, MODE_AM, MODE_LSB, MODE_AM, MODE_USB]
    for k, mode in enumerate(seq):
        assert g.set_mode(mode) == 0 and o.set_mode(mode) == 0
        assert g.kernel_name() == kernel
        iq = synth_iq(0, 48, k * 1024, 1024)
        assert bits_equal(g.process(iq), o.process(iq)), "step %d mode %d" % (k, mode)
        assert_state_equal(g, o)
    gs, oc = gpu_rx(baseline_spec("cfg3", 48, rc.ARITH_SPLIT16)), CpuChain(baseline_spec("cfg3", 48, ARITH_CMSIS), "orc")
    assert gs.set_mode(MODE_AM) == 0 and oc.set_mode(MODE_AM) == 0
    assert gs.kernel_name() == "k_ssb_split16<256,4,63>"
    for k in range(2):
        iq = synth_iq(0, 48, k * 2048, 2048)
        yg, yo = gs.process(iq), oc.process(iq)
        for b in range(yo.shape[1] // 64):
            assert rel_err(yg[:, b * 64:(b + 1) * 64], yo[:, b * 64:(b + 1) * 64]) <= TOL


@pytest.mark.gpu
@pytest.mark.parametrize("name,kernel", [("cfg2", "k_hilb_split16<127>"), ("cfg1", "k_hilb_split16<63>")])
def test_hilbert_on_the_matrix_pipe_within_tolerance_state_exact(name, kernel):
    """SELENITE_ARITH_SPLIT16 on the no-decimator shapes: the Hilbert FIR as an f16 hi/lo split MFMA
    product (k_hilb_split16).  Tolerance-based (north star: 1e-5 relative per DSP block); the streaming
    state is written from the exact f32 samples and must match the CMSIS chain bit for bit."""
    nch = 80
    g = gpu_rx(baseline_spec(name, nch, rc.ARITH_SPLIT16))
    o = CpuChain(baseline_spec(name, nch, ARITH_CMSIS), "orc")
    assert g.kernel_name() == kernel
    worst = 0.0
    for call, bs in enumerate((1024, 256, 2048)):
        iq = synth_iq(0, nch, 4096 * call, bs)
        yg, yo = g.process(iq), o.process(iq)
        assert np.isfinite(yg).all()
        for b in range(bs // 256):
            for ch in range(nch):
                worst = max(worst, rel_err(yg[ch, 256 * b:256 * b + 256], yo[ch, 256 * b:256 * b + 256]))
    print("%s split16 worst per-channel-block rel_err = %.3g" % (name, worst))
    assert worst <= TOL, worst
    sg, so = g.state(), o.state()
    assert bits_equal(sg["fir_state"], so["fir_state"])
    assert np.allclose(sg["agc_gain"], so["agc_gain"], rtol=1e-5, atol=0)


@pytest.mark.gpu
def test_hilbert_split16_modes_q15_silence_full_scale():
    nch = 33
    for mode in (rc.MODE_LSB, MODE_AM, rc.MODE_USB):
        g = gpu_rx(baseline_spec("cfg2", nch, rc.ARITH_SPLIT16))
        o = CpuChain(baseline_spec("cfg2", nch, ARITH_CMSIS), "orc")
        assert g.set_mode(mode) == 0 and o.set_mode(mode) == 0
        for call in range(2):
            iq = synth_iq(0, nch, 512 * call, 512)
            yg, yo = g.process(iq), o.process(iq)
            for b in range(2):
                assert rel_err(yg[:, 256 * b:256 * b + 256], yo[:, 256 * b:256 * b + 256]) <= TOL, mode
        assert bits_equal(g.state()["fir_state"], o.state()["fir_state"]), mode
    g = gpu_rx(baseline_spec("cfg2", 4, rc.ARITH_SPLIT16))
    o = CpuChain(baseline_spec("cfg2", 4, ARITH_CMSIS), "orc")
    z = np.zeros((4, 512, 2), np.float32)
    assert bits_equal(g.process(z), o.process(z))                      # silence stays exact zero
    full = np.ones((4, 512, 2), np.float32)
    full[:, ::2, :] = -1.0
    yg, yo = g.process(full), o.process(full)
    assert rel_err(yg, yo) <= TOL
    iq = synth_iq(0, 4, 0, 512)
    q = np.clip(np.trunc(iq * 32768.0), -32768, 32767).astype(np.int16)
    gq, oq = gpu_rx(baseline_spec("cfg2", 4, rc.ARITH_SPLIT16)), CpuChain(baseline_spec("cfg2", 4, ARITH_CMSIS), "orc")
    assert np.max(np.abs(gq.process_q15(q).astype(np.int32) - oq.process_q15(q).astype(np.int32))) <= 1


@pytest.mark.gpu
@pytest.mark.parametrize("arith", [ARITH_CMSIS, rc.ARITH_SPLIT16])
def test_eight_way_channel_sharding_equals_the_unsharded_run_on_4096_channels(arith):
    """the result of a channel-sharded job (8 ranks, contiguous channel ranges from
    selenite_rx.shard.channel_range) equals the 1-GPU result bit for bit -- channels share nothing but
    read-only taps.  Eight instances on one device stand in for the eight ranks; also in the split16
    arithmetic, whose result depends on nothing but the channel's own samples."""
    import selenite_rx as sr
    from selenite_rx.shard import channel_range
    nch, bs, world = 4096, 2048, 8
    d_in, d_out = sr.DeviceBuffer(nch * bs * 8), sr.DeviceBuffer(nch * (bs // 4) * 4)
    whole = gpu_rx(baseline_spec("cfg3", nch, arith))
    shards = []
    for r in range(world):
        c0, cnt = channel_range(nch, r, world)
        shards.append((c0, c0 + cnt, gpu_rx(baseline_spec("cfg3", cnt, arith))))
    for call in range(2):
        whole.synth_device(d_in.ptr, 0, nch, call * bs, bs, rc.SEED)
        whole.process_device(d_in.ptr, d_out.ptr, bs)
        whole.sync(); whole.check()
        y = d_out.download((nch, bs // 4), np.float32)
        parts = []
        for c0, c1, g in shards:
            g.process_device(d_in.ptr + c0 * bs * 8, d_out.ptr + c0 * (bs // 4) * 4, bs)
            g.sync(); g.check()
        y2 = d_out.download((nch, bs // 4), np.float32)
        assert bits_equal(y, y2), "call %d" % call
    sw = whole.state()
    for c0, c1, g in shards:
        ss = g.state()
        for k in ss:
            assert np.array_equal(ss[k].view(np.uint32), sw[k][c0:c1].view(np.uint32)), k


@pytest.mark.gpu
@pytest.mark.parametrize("arith", [ARITH_CMSIS, ARITH_FMA, rc.ARITH_SPLIT16])
def test_call_lengths_that_are_not_whole_passes_split_into_fused_plus_generic(arith):
    """blockSize only has to be a multiple of cfg.block (256).  The decimating fused kernels work in
    passes of 1024 inputs; other lengths run the whole passes fused and the remaining DSP blocks on the
    generic kernels, on the same streaming state, addressing the caller's buffers with the full stride."""
    nch = 40
    g = gpu_rx(baseline_spec("cfg3", nch, arith))
    ref_arith = ARITH_CMSIS if arith == rc.ARITH_SPLIT16 else arith
    o = CpuChain(baseline_spec("cfg3", nch, ref_arith), "orc")
    pos = 0
    for bs in (1280, 256, 1024, 1792, 768, 2048 + 512):
        iq = synth_iq(0, nch, pos, bs)
        pos += bs
        yg, yo = g.process(iq), o.process(iq)
        if arith == rc.ARITH_SPLIT16:
            for b in range(bs // 256):
                assert rel_err(yg[:, 64 * b:64 * b + 64], yo[:, 64 * b:64 * b + 64]) <= TOL, bs
        else:
            assert bits_equal(yg, yo), bs
    sg, so = g.state(), o.state()
    assert bits_equal(sg["dec_state"], so["dec_state"]) and np.array_equal(sg["nco_phase"], so["nco_phase"])
    if arith != rc.ARITH_SPLIT16:
        assert bits_equal(sg["fir_state"], so["fir_state"]) and bits_equal(sg["agc_gain"], so["agc_gain"])
    q = np.clip(np.trunc(synth_iq(0, nch, 0, 1280) * 32768.0), -32768, 32767).astype(np.int16)
    gq, oq = gpu_rx(baseline_spec("cfg3", nch, ARITH_CMSIS)), CpuChain(baseline_spec("cfg3", nch, ARITH_CMSIS), "orc")
    assert np.array_equal(gq.process_q15(q), oq.process_q15(q))


@pytest.mark.gpu
@pytest.mark.parametrize("nd,nh", [(128, 63), (256, 127), (128, 127), (256, 31), (128, 31)])
@pytest.mark.parametrize("arith", [ARITH_CMSIS, ARITH_FMA, rc.ARITH_SPLIT16])
def test_neighbour_shapes_of_cfg3_run_the_fused_kernels(nd, nh, arith):
    """The decimating kernels are templates: besides the BASELINE shape the library instantiates a
    128-tap decimator and 31- / 127-tap Hilbert variants.  Same parity contract per arithmetic mode."""
    nch = 48
    kw = dict(nco=True, nco_step_all=0x01000000, agc=True)
    g = gpu_rx(rc.ChainSpec(nch, 256, 4, nd, nh, 0, rc.MODE_LSB, arith, **kw))
    ref_arith = ARITH_CMSIS if arith == rc.ARITH_SPLIT16 else arith
    o = CpuChain(rc.ChainSpec(nch, 256, 4, nd, nh, 0, rc.MODE_LSB, ref_arith, **kw), "orc")
    want = {ARITH_CMSIS: "k_ssb_fused", ARITH_FMA: "k_ssb_mfma", rc.ARITH_SPLIT16: "k_ssb_split16"}[arith]
    assert g.kernel_name() == "%s<%d,4,%d>" % (want, nd, nh)
    for call in range(2):
        iq = synth_iq(0, nch, 2048 * call, 2048)
        yg, yo = g.process(iq), o.process(iq)
        if arith == rc.ARITH_SPLIT16:
            for b in range(8):
                assert rel_err(yg[:, 64 * b:64 * b + 64], yo[:, 64 * b:64 * b + 64]) <= TOL
        else:
            assert bits_equal(yg, yo)
    assert bits_equal(g.state()["dec_state"], o.state()["dec_state"])
    if arith != rc.ARITH_SPLIT16:
        assert_state_equal(g, o)


@pytest.mark.gpu
@pytest.mark.parametrize("nch", [1, 15, 17, 37])
def test_cw_fused_kernel_any_channel_count(nch):
    """k_cw_fused packs 16 channels per wavefront; counts that are not a multiple of 16 run it too
    (clamped loads, masked stores in the last workgroup)."""
    spec = baseline_spec("cfg4", nch, ARITH_CMSIS)
    g, o = gpu_rx(spec), CpuChain(spec, "orc")
    assert g.kernel_name() == "k_cw_fused<4,256>"
    for call in range(3):
        iq = synth_iq(0, nch, 512 * call, 512)
        assert bits_equal(g.process(iq), o.process(iq))
    assert_state_equal(g, o)


@pytest.mark.gpu
@pytest.mark.parametrize("stages,nch,grid,q15,variant", [(4, 75, 2, False, "shared"), (4, 75, 2, True, "shared"), (4, 64, 4, False, "per_channel"),
                                                         (4, 81, 3, False, "none"), (2, 150, 2, False, "shared"), (8, 37, 2, False, "shared"), (4, 70, 1, True, "per_channel")])
def test_cw_fused_kernel_persistent_over_channel_groups(stages, nch, grid, q15, variant):
    """Round 6: a workgroup of k_cw_fused takes SEVERAL channel groups (b, b + grid, ...), requesting the first chunks of its next group during
    the last DSP block of the present one -- what the launcher does where the groups divide evenly over the device (cfg4: 4096 groups on 2048
    workgroups).  Pinned here to 1 / 2 / 3 / 4 workgroups (selenite_rx_set_plan_option(SELENITE_RX_OPT_CW_GRID)) so that small inputs cover uneven
    shares, a partial last group, every NCO flavour and both slot formats: audio and state bit for bit the oracle's, over several calls."""
    import selenite_rx as sr
    kw = dict(nco=variant != "none", agc=True)
    if variant == "per_channel":
        kw["nco_steps"] = (np.arange(nch, dtype=np.uint64) * 0x9E3779B1 % (1 << 27) + 0x00400000).astype(np.uint32)
    else:
        kw["nco_step_all"] = 0x00800000
    spec = rc.ChainSpec(nch, 256, 1, 0, 0, stages, MODE_CW, ARITH_CMSIS, **kw)
    with sr.plan_option(sr.OPT_CW_GRID, grid):
        g = gpu_rx(spec)
        o = CpuChain(spec, "orc")
        assert g.kernel_name() == "k_cw_fused<%d,256>" % stages
        for call, bs in enumerate((512, 256, 1024)):
            iq = synth_iq(0, nch, 2048 * call, bs)
            if q15:
                q = np.clip(np.trunc(iq * 32768.0), -32768, 32767).astype(np.int16)
                assert np.array_equal(g.process_q15(q), o.process_q15(q)), call
            else:
                assert bits_equal(g.process(iq), o.process(iq)), call
        assert_state_equal(g, o)


@pytest.mark.gpu
def test_the_library_reads_no_kernel_selection_switch_from_the_environment():
    """Round 6: the round-5 environment switches are gone from the default build -- set, they change nothing; the same
    selections through selenite_rx_set_plan_option do."""
    import selenite_rx as sr
    spec = baseline_spec("cfg4", 33)
    env = {"SELENITE_RX_FORCE_GENERIC": "1", "SELENITE_RX_NO_SHARED_LO": "1", "SELENITE_RX_NO_PERIODIC_LO": "1", "SELENITE_RX_CW_LDS_PAD": "20000",
           "SELENITE_RX_SPLIT16_GRID": "1", "SELENITE_RX_RERUN_GRID": "1", "SELENITE_RX_NO_HIST_EXT": "1", "SELENITE_RX_DEC2_PARITY": "2"}
    os.environ.update(env)
    try:
        g = gpu_rx(spec)
        h = gpu_rx(baseline_spec("cfg3", 33, rc.ARITH_SPLIT16))
        assert g.kernel_name() == "k_cw_fused<4,256>" and h.kernel_name() == "k_ssb_split16<256,4,63>" and "registers" in h.nco_path()
    finally:
        for k in env:
            del os.environ[k]
    with sr.plan_option(sr.OPT_FORCE_GENERIC), sr.plan_option(sr.OPT_NO_PERIODIC_LO):
        assert gpu_rx(spec).kernel_name() == "generic"


@pytest.mark.gpu
def test_instances_run_concurrently_on_their_own_streams():
    """Distinct instances are independent (SURVEY.md 8b "Threading"): each has its own stream; calls
    issued back to back without synchronisation in between must give what sequential runs give."""
    import selenite_rx as sr
    nch, bs = 512, 2048
    specs = [baseline_spec("cfg3", nch, ARITH_CMSIS), baseline_spec("cfg4", nch, ARITH_CMSIS),
             baseline_spec("cfg2", nch, ARITH_FMA), baseline_spec("cfg3", nch, ARITH_FMA)]
    inst = [gpu_rx(s) for s in specs]
    ring, oring = sr.Ring(nch), rc.OracleRing(nch)
    iq = synth_iq(0, nch, 0, bs)
    d_in = sr.DeviceBuffer(iq.nbytes)
    d_in.upload(iq)
    outs = [sr.DeviceBuffer(nch * (bs // s.decim) * 4) for s in specs]
    pkt = (np.arange(nch * 96, dtype=np.int64).reshape(nch, 96) * 131 % 65536 - 32768).astype(np.int16)
    d_pkt = sr.DeviceBuffer(pkt.nbytes)
    d_pkt.upload(pkt)
    for rep in range(3):                                   # three rounds, nothing synchronised inside a round
        for g, d_o in zip(inst, outs):
            g.process_device(d_in.ptr, d_o.ptr, bs)
        sr.lib().selenite_ring_in_write_device(ring.h, d_pkt.ptr, 96)
        for g in inst:
            g.sync(); g.check()
        oring.in_write(pkt)
    for g, s, d_o in zip(inst, specs, outs):
        o = CpuChain(s, "orc")
        want = None
        for rep in range(3):
            want = o.process(iq)
        assert bits_equal(d_o.download((nch, bs // s.decim), np.float32), want), g.kernel_name()
    ring.L.selenite_ring_sync(ring.h)
    sg, so = ring.state(), oring.state()
    assert all(np.array_equal(sg[k], so[k]) for k in so)


@pytest.mark.parametrize("arith", [ARITH_CMSIS, rc.ARITH_SPLIT16, rc.ARITH_AUTO])
def test_host_pointer_calls_are_pipelined_in_channel_chunks_and_bit_identical(arith):
    """selenite_rx_process_f32 / _q15 on host buffers (the literal drop-in signature): chunked H2D || kernels || D2H
    pipeline.  With a 1 MiB chunk the 300 channels below take five chunks (the last one ragged); the result and the
    streaming state must equal the un-chunked device call bit for bit, for pageable and for page-locked caller memory,
    over several streamed calls (the shared LO must advance once per call, not once per chunk)."""
    import subprocess, sys, textwrap
    code = textwrap.dedent("""
        import os, sys, numpy as np
        sys.path.insert(0, %r); sys.path.insert(0, %r)
        import rxcommon as rc, selenite_rx as sr
        nch, bs, arith = 300, 2048, %d
        spec = rc.baseline_spec("cfg3", nch, arith)
        if arith == rc.ARITH_AUTO:      # every channel its own NCO step: most pass bands empty, so most channels of every chunk are
            # under the parity guard and go through the rerun pass (flag words and state of a channel SUB-RANGE of the instance)
            spec = rc.baseline_spec("cfg3", nch, arith, nco_steps=(np.arange(nch, dtype=np.uint64) * 0x9E3779B1 %% (1 << 32)).astype(np.uint32))
        a, b, c = sr.Rx(spec.config()), sr.Rx(spec.config()), sr.Rx(spec.config())
        d_in, d_out = sr.DeviceBuffer(nch * bs * 8), sr.DeviceBuffer(nch * (bs // 4) * 4)
        for call in range(3):
            iq = rc.synth_iq(0, nch, call * bs, bs)
            d_in.upload(iq); a.process_device(d_in.ptr, d_out.ptr, bs); a.sync()
            want = d_out.download((nch, bs // 4), np.float32)
            got_pageable = b.process(iq)
            pin = sr.pinned_array(iq.shape, np.float32); pin[...] = iq
            out = sr.pinned_array((nch, bs // 4), np.float32)
            got_pinned = c.process(pin, out)
            assert rc.bits_equal(got_pageable, want) and rc.bits_equal(got_pinned, want), call
        if arith == rc.ARITH_AUTO:      # a sideband flip: channels the last call kept on the matrix kernel are rerun from a repaired history
            # (hist_ext rows and provenance words of a channel sub-range: DESIGN.md section 3, "across a call boundary")
            for g in (a, b, c):
                assert g.set_mode(rc.MODE_LSB) == 0
            iq = rc.synth_iq(0, nch, 3 * bs, bs)
            d_in.upload(iq); a.process_device(d_in.ptr, d_out.ptr, bs); a.sync()
            want = d_out.download((nch, bs // 4), np.float32)
            assert rc.bits_equal(b.process(iq), want)
            pin = sr.pinned_array(iq.shape, np.float32); pin[...] = iq
            assert rc.bits_equal(c.process(pin, sr.pinned_array((nch, bs // 4), np.float32)), want)
            o = rc.CpuChain(rc.baseline_spec("cfg3", nch, rc.ARITH_CMSIS, nco_steps=spec.nco_steps), "orc")
            for call in range(3):
                o.process(rc.synth_iq(0, nch, call * bs, bs))
            o.set_mode(rc.MODE_LSB)
            yo = o.process(iq)
            d = np.abs(want.astype(np.float64) - yo).reshape(nch, -1, 64).max(axis=2); m = np.abs(yo).reshape(nch, -1, 64).max(axis=2)
            assert (d <= 1e-5 * m).all(), (d / np.maximum(m, 1e-30)).max()
            assert a.guard_stats()["handover_blocks"] == 0 and b.guard_stats()["handover_blocks"] == 0
        sa, sb, sc = a.state(), b.state(), c.state()
        for k in sa:
            assert np.array_equal(sa[k].view(np.uint32), sb[k].view(np.uint32)) and np.array_equal(sa[k].view(np.uint32), sc[k].view(np.uint32)), k
        q = np.clip(np.trunc(rc.synth_iq(0, nch, 0, bs) * 32768.0), -32768, 32767).astype(np.int16)
        x, y = sr.Rx(spec.config()), sr.Rx(spec.config())
        d16i, d16o = sr.DeviceBuffer(q.nbytes), sr.DeviceBuffer(nch * (bs // 4) * 2)
        d16i.upload(q); x.process_q15_device(d16i.ptr, d16o.ptr, bs); x.sync()
        assert np.array_equal(y.process_q15(q), d16o.download((nch, bs // 4), np.int16))
        print("OK")
    """ % (os.path.join(rc.ROOT, "tests"), rc.PKG_DIR, arith))
    env = dict(os.environ, SELENITE_RX_HOST_CHUNK_MB="1")
    out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=600, env=env)
    assert out.returncode == 0 and out.stdout.strip().endswith("OK"), out.stderr[-3000:]


@pytest.mark.parametrize("arith", [ARITH_CMSIS, ARITH_FMA, rc.ARITH_SPLIT16])
@pytest.mark.parametrize("case", ["dense_hilbert", "delay_not_impulse", "negative_zero_taps", "dense_both"])
def test_dense_and_negative_zero_fir_pair_taps_take_the_generic_kernels_bit_exact(case, arith):
    """The type-III fused kernels assume what selenite_rx_design_hilbert makes: a unit-impulse delay FIR and a type-III
    Hilbert FIR (exact +0.0f taps at even distance from the centre, which they skip).  Anything else -- dense random
    taps, a delay FIR that is not an impulse, taps that are -0.0f (x + (-0.0 * y) is not x for x = -0) -- must not take that
    shortcut and still match the CMSIS-order oracle bit for bit (arm_fir_f32.c:640-936): round 3 sent them to the generic kernels,
    round 4 to the DENSE flavour of k_ssb_fused, which multiplies by every tap it is given."""
    rng = np.random.default_rng(hash(case) % 1000)
    nch = 6
    spec = baseline_spec("cfg3", nch, arith)
    nh = spec.nh_taps
    if case in ("dense_hilbert", "dense_both"):
        spec.hilb = (rng.standard_normal(nh) * 0.2).astype(np.float32)
    if case in ("delay_not_impulse", "dense_both"):
        d = np.zeros(nh, np.float32)
        d[nh // 2] = 0.75
        d[nh // 2 + 3] = 0.25
        spec.delay = d if case == "delay_not_impulse" else (rng.standard_normal(nh) * 0.2).astype(np.float32)
    if case == "negative_zero_taps":
        h = spec.hilb.copy()
        h[h == 0.0] = np.float32(-0.0)
        spec.hilb = h
        d = spec.delay.copy()
        d[d == 0.0] = np.float32(-0.0)
        spec.delay = d
    ref_arith = ARITH_FMA if arith == rc.ARITH_SPLIT16 else arith      # outside the fused decimator split16 runs as FMA
    spec_o = baseline_spec("cfg3", nch, ref_arith)
    spec_o.hilb, spec_o.delay = spec.hilb, spec.delay
    g, o = gpu_rx(spec), CpuChain(spec_o, "orc")
    assert g.kernel_name() == "k_ssb_fused<256,4,63> (dense FIR pair)", g.kernel_name()
    for call in range(2):
        iq = synth_iq(0, nch, call * 1024, 1024)
        iq[:, ::7, :] *= -1.0
        yg, yo = g.process(iq), o.process(iq)
        assert bits_equal(yg, yo), "%s arith=%d rel_err=%g" % (case, arith, rel_err(yg, yo))
    assert_state_equal(g, o)


@pytest.mark.parametrize("arith,kernel", [(ARITH_CMSIS, "k_ssb_fused<0,1,31>"), (ARITH_FMA, "k_ssb_fused<0,1,31>"),
                                          (rc.ARITH_SPLIT16, "k_hilb_split16<31>")])
def test_short_hilbert_pair_without_decimator(arith, kernel):
    """31-tap Hilbert pair, no decimator: the shortest instantiated shape of the no-decimator kernels."""
    nch = 20
    g = gpu_rx(rc.ChainSpec(nch, 256, 1, 0, 31, 0, rc.MODE_USB, arith, agc=True))
    ref_arith = ARITH_CMSIS if arith == rc.ARITH_SPLIT16 else arith
    o = CpuChain(rc.ChainSpec(nch, 256, 1, 0, 31, 0, rc.MODE_USB, ref_arith, agc=True), "orc")
    assert g.kernel_name() == kernel
    for call in range(3):
        iq = synth_iq(0, nch, 768 * call, 768)
        yg, yo = g.process(iq), o.process(iq)
        if arith == rc.ARITH_SPLIT16:
            for b in range(3):
                assert rel_err(yg[:, 256 * b:256 * b + 256], yo[:, 256 * b:256 * b + 256]) <= TOL
        else:
            assert bits_equal(yg, yo)
    assert bits_equal(g.state()["fir_state"], o.state()["fir_state"])


@pytest.mark.parametrize("stages", [2, 8])
@pytest.mark.parametrize("variant", ["shared_lo", "per_channel_nco", "no_nco_no_agc", "q15"])
@pytest.mark.parametrize("nch", [50, 64])
def test_cw_fused_kernel_two_and_eight_stages(stages, variant, nch):
    """The systolic CW kernel for 2 and 8 DF1 sections (32 / 8 channels per wavefront, 1 / 7 fill and drain steps):
    bit-exact against the oracle like the 4-stage BASELINE shape, ragged channel counts included
    (arm_biquad_cascade_df1_f32.c:165-407)."""
    kw = dict(block=256, decim=1, nd_taps=0, nh_taps=0, n_biquad=stages, mode=MODE_CW, arith=ARITH_CMSIS,
              nco=True, nco_step_all=0x00800000, agc=True, bp_q=2.0 if stages == 8 else 4.0)
    if variant == "per_channel_nco":
        kw["nco_steps"] = (np.arange(nch, dtype=np.uint32) * np.uint32(0x00123457) + np.uint32(0x00400000))
    elif variant == "no_nco_no_agc":
        kw["nco"] = False
        kw["agc"] = False
    spec = ChainSpec(nch, **kw)
    g, o = gpu_rx(spec), CpuChain(spec, "orc")
    assert g.kernel_name() == "k_cw_fused<%d,256>" % stages
    for call in range(3):
        iq = synth_iq(0, nch, call * 768, 768)
        if variant == "q15":
            q = np.clip(np.trunc(iq * 32768.0), -32768, 32767).astype(np.int16)
            assert np.array_equal(g.process_q15(q), o.process_q15(q)), call
        else:
            yg, yo = g.process(iq), o.process(iq)
            assert np.isfinite(yg).all() and bits_equal(yg, yo), "%s call %d rel_err %g" % (variant, call, rel_err(yg, yo))
    assert_state_equal(g, o)


@pytest.mark.gpu
@pytest.mark.parametrize("q15", [False, True])
def test_periodic_shared_lo_in_registers_equals_the_table_path_bit_for_bit(q15):
    """A common NCO step that is a multiple of 2^24 repeats the LO every 256 samples: k_ssb_split16 then keeps it
    in registers (selenite_rx_nco_path says so).  Same LO values, same arithmetic -> same bits as the per-call table,
    audio and state, for whole-pass calls, calls with a partial last pass, a non-zero common phase; a step off
    that grid must take the table path and meet the tolerance against the oracle."""
    nch = 70
    spec = baseline_spec("cfg3", nch, rc.ARITH_SPLIT16)           # step 0x01000000
    reg = gpu_rx(spec)
    import selenite_rx as sr
    with sr.plan_option(sr.OPT_NO_PERIODIC_LO):
        tab = gpu_rx(spec)
    assert "registers" in reg.nco_path() and reg.kernel_name() == "k_ssb_split16<256,4,63>"
    assert tab.nco_path() == "shared LO table per call"
    st = reg.state()
    st["nco_phase"][:] = 0x3456789A
    reg.set_state(st)
    tab.set_state(st)
    pos = 0
    for bs in (4096, 4352, 1024, 8192 + 256):
        iq = synth_iq(0, nch, pos, bs)
        pos += bs
        if q15:
            iq = np.clip(np.round(iq * 20000.0), -32768, 32767).astype(np.int16)
            ya, yb = reg.process_q15(iq), tab.process_q15(iq)
            assert np.array_equal(ya, yb)
        else:
            ya, yb = reg.process(iq), tab.process(iq)
            assert bits_equal(ya, yb)
    sa, sb = reg.state(), tab.state()
    for k in sa:
        assert np.array_equal(sa[k].view(np.uint32) if sa[k].dtype == np.float32 else sa[k],
                              sb[k].view(np.uint32) if sb[k].dtype == np.float32 else sb[k]), k
    # phases that differ per channel: back to the per-channel NCO
    st = reg.state()
    st["nco_phase"] = (st["nco_phase"] + np.arange(nch, dtype=np.uint32) * np.uint32(0x01234567)).astype(np.uint32)
    reg.set_state(st)
    assert reg.nco_path().startswith("per-channel")
    if q15:
        return
    # a step off the fs / 256 grid: table path, tolerance against the CMSIS oracle
    spec_o = baseline_spec("cfg3", nch, rc.ARITH_SPLIT16)
    spec_o.nco_step_all = 0x01000100
    spec_c = baseline_spec("cfg3", nch, ARITH_CMSIS)
    spec_c.nco_step_all = 0x01000100
    g, o = gpu_rx(spec_o), CpuChain(spec_c, "orc")
    assert g.nco_path() == "shared LO table per call"
    na = spec_c.block // spec_c.decim
    for call in range(2):
        iq = synth_iq(0, nch, call * 4096, 4096)
        yg, yo = g.process(iq), o.process(iq)
        for c in range(nch):
            for b in range(yo.shape[1] // na):
                assert rel_err(yg[c, b * na:(b + 1) * na], yo[c, b * na:(b + 1) * na]) <= TOL
    assert bits_equal(g.state()["dec_state"], o.state()["dec_state"])


@pytest.mark.gpu
@pytest.mark.parametrize("nd,M,nh", [(256, 2, 63), (256, 8, 63), (64, 4, 63), (128, 2, 63), (128, 8, 63), (64, 2, 63), (64, 8, 63),
                                     (256, 2, 127), (128, 2, 127), (256, 2, 31), (128, 2, 31)])
@pytest.mark.parametrize("arith", [ARITH_CMSIS, ARITH_FMA, rc.ARITH_SPLIT16])
@pytest.mark.parametrize("q15", [False, True])
def test_other_decimation_ratios_and_a_64_tap_decimator_run_the_fused_kernels(nd, M, nh, arith, q15):
    """arm_fir_decimate_f32 with M = 2 and M = 8 (and a 64-tap decimator by 4) on the fused kernels: bit-exact in
    the CMSIS and fma modes.  split16: decimation by 2 has its own k_ssb_split16 instantiations and a 64-tap decimator runs on the
    128-tap one (tolerance against CMSIS, mixed samples bit-exact); by 8 (round 4, late) runs the by-4 matrix kernel with every second
    output of the tile kept.  DSP blocks of 256 inputs = 128 / 32 / 64 audio samples: AGC groups of 32 / 8 / 16 lanes.
    Whole-pass calls and a call with a ragged tail."""
    nch = 37
    kw = dict(nco=True, nco_step_all=0x01234567, agc=True)
    g = gpu_rx(rc.ChainSpec(nch, 256, M, nd, nh, 0, rc.MODE_USB, arith, **kw))
    split = arith == rc.ARITH_SPLIT16 and M in (2, 4, 8)    # k_ssb_split16 (round 4: the 64-tap decimator on the 128-tap kernel, taps zero-padded in front; by 8: the by-4 product, every second output kept)
    ref_arith = ARITH_CMSIS if split else (ARITH_FMA if arith == rc.ARITH_SPLIT16 else arith)
    o = CpuChain(rc.ChainSpec(nch, 256, M, nd, nh, 0, rc.MODE_USB, ref_arith, **kw), "orc")
    want = "k_ssb_split16" if split else ("k_ssb_mfma" if (arith != ARITH_CMSIS and M == 4) else "k_ssb_fused")
    assert g.kernel_name() == "%s<%d,%d,%d>" % (want, nd, M, nh)
    na = 256 // M
    pos = 0
    for bs in (256 * M * 2, 256 * M + 256 * 3, 256 * M):
        iq = synth_iq(0, nch, pos, bs)
        pos += bs
        if q15:
            iq16 = np.clip(np.round(iq * 20000.0), -32768, 32767).astype(np.int16)
            yg, yo = g.process_q15(iq16), o.process_q15(iq16)
            if split:
                assert np.abs(yg.astype(np.int32) - yo.astype(np.int32)).max() <= 1
            else:
                assert np.array_equal(yg, yo)
        else:
            yg, yo = g.process(iq), o.process(iq)
            if split:
                for blk in range(yo.shape[1] // na):
                    assert rel_err(yg[:, blk * na:(blk + 1) * na], yo[:, blk * na:(blk + 1) * na]) <= TOL
            else:
                assert bits_equal(yg, yo)
    if split:
        assert bits_equal(g.state()["dec_state"], o.state()["dec_state"])
        assert np.array_equal(g.state()["nco_phase"], o.state()["nco_phase"])
    else:
        assert_state_equal(g, o)


def dense_spec(nch, block, M, nd, nh, mode, arith, seed, unit_delay=False, **kw):
    """A chain whose FIR pair has ARBITRARY taps on both rails: a dense 'Hilbert' FIR (no structural zeros) and a delay FIR that is not a
    unit impulse (a fractional-delay-like low-pass) -- legal arm_fir_f32 inputs (arm_fir_f32.c:553-979) no design helper makes."""
    spec = ChainSpec(nch, block, M, nd, nh | 1, 0, mode, arith, **kw)      # (the design helper wants an odd count; the taps are replaced)
    spec.nh_taps = nh
    rng = np.random.default_rng(seed)
    k = np.arange(nh) - (nh - 1) / 2.0
    win = np.hamming(nh) if nh > 2 else np.ones(nh)
    spec.hilb = np.ascontiguousarray((win * rng.uniform(-1, 1, nh) / np.sqrt(nh)).astype(np.float32))
    spec.delay = np.ascontiguousarray((win * np.sinc(k - 0.3)).astype(np.float32))
    if unit_delay:                       # a dense Hilbert design beside a unit-impulse delay (not at the centre: any index is legal)
        spec.delay = np.zeros(nh, np.float32)
        spec.delay[(nh - 1) // 3] = 1.0
    return spec


@pytest.mark.parametrize("nd,M,nh", [(256, 4, 63), (256, 4, 127), (100, 4, 64), (128, 2, 17), (37, 2, 40), (256, 8, 63), (64, 8, 125), (0, 1, 63), (0, 1, 2), (0, 1, 127)])
@pytest.mark.parametrize("arith", [ARITH_CMSIS, ARITH_FMA, rc.ARITH_SPLIT16, rc.ARITH_AUTO])
@pytest.mark.parametrize("q15,unit_delay", [(False, False), (True, False), (False, True)])
def test_a_fir_pair_with_arbitrary_taps_runs_on_the_fused_kernel(nd, M, nh, arith, q15, unit_delay):
    """Round 4 (VERDICT r3 missing 4): dense Hilbert taps, a delay FIR that is not a unit impulse, any tap count up to 127 (even ones
    too), any decimator up to 256 taps by 2 / 4 / 8 or none -- the DENSE flavour of k_ssb_fused instead of the generic kernels (59
    Gsamples/s on the cfg3 shape).  Bit-exact against the oracle, output and state, in the CMSIS and fma arithmetic; SPLIT16 runs as
    fma and AUTO as CMSIS there (the header's contract for shapes without a matrix kernel).  Every mode, whole and partial passes."""
    nch = 19
    steps = (np.arange(nch, dtype=np.uint64) * 0x9E3779B1 % (1 << 32)).astype(np.uint32)
    kw = dict(nco=True, nco_steps=steps, agc=True)
    ref = {rc.ARITH_SPLIT16: ARITH_FMA, rc.ARITH_AUTO: ARITH_CMSIS}.get(arith, arith)
    g = gpu_rx(dense_spec(nch, 256, M, nd, nh, MODE_USB, arith, 7, unit_delay, **kw))
    o = CpuChain(dense_spec(nch, 256, M, nd, nh, MODE_USB, ref, 7, unit_delay, **kw), "orc")
    assert g.kernel_name() == "k_ssb_fused<%d,%d,%d> (dense FIR pair)" % (nd, M, nh), g.kernel_name()
    pos = 0
    for mode, bs in ((MODE_USB, 256 * M * 2), (MODE_LSB, 256 * M + 768), (MODE_AM, 256), (rc.MODE_FM if nh >= 2 else MODE_USB, 512), (MODE_USB, 256 * M * 3)):
        assert g.set_mode(mode) == 0 and o.set_mode(mode) == 0
        iq = synth_iq(0, nch, pos, bs)
        pos += bs
        if q15:
            iq16 = np.clip(np.round(iq * 20000.0), -32768, 32767).astype(np.int16)
            assert np.array_equal(g.process_q15(iq16), o.process_q15(iq16)), (mode, bs)
        else:
            yg, yo = g.process(iq), o.process(iq)
            assert bits_equal(yg, yo), (mode, bs, rel_err(yg, yo))
    assert_state_equal(g, o)
