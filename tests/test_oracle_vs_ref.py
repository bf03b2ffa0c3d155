"""CPU, build container only: the oracle against the REAL CMSIS-DSP 1.5.3 code compiled from
/root/reference (oracle/_ref/libcmsis_ref.so), live and on more inputs than the committed
fixtures hold.  Skipped where the reference library cannot exist (the GPU box)."""
import numpy as np
import pytest

import rxcommon as rc
from rxcommon import ARITH_CMSIS as A0, bits_equal, fptr


def test_sin_table(oracle, ref):
    a = np.ctypeslib.as_array(oracle.orc_sin_table(), (513,))
    b = np.ctypeslib.as_array(ref.ref_sin_table(), (513,))
    assert bits_equal(a, b)


def test_sin_cos_dense_sweep(oracle, ref):
    rng = np.random.default_rng(1)
    x = np.concatenate([np.arange(0, 1 << 24, 257, dtype=np.uint32).astype(np.float32) * np.float32(2 * np.pi / 2 ** 24),
                        (rng.random(20000, dtype=np.float32) - np.float32(0.5)) * np.float32(1000.0),
                        -np.logspace(-9, -5, 200).astype(np.float32)]).astype(np.float32)
    s, c = np.empty_like(x), np.empty_like(x)
    ref.ref_sin_cos(fptr(x), fptr(s), fptr(c), x.size)
    so = np.array([oracle.orc_sin_f32(float(v), A0) for v in x], np.float32)
    co = np.array([oracle.orc_cos_f32(float(v), A0) for v in x], np.float32)
    assert bits_equal(so, s) and bits_equal(co, c)


@pytest.mark.parametrize("nt,M,blk", [(256, 4, 256), (255, 4, 256), (7, 1, 5), (100, 5, 50), (31, 2, 6), (4, 4, 4)])
def test_fir_decimate_random(oracle, ref, nt, M, blk):
    rng = np.random.default_rng(nt * 1000 + M)
    c = rng.standard_normal(nt).astype(np.float32)
    sa, sb = np.zeros(nt - 1 + blk, np.float32), np.zeros(nt - 1 + blk, np.float32)
    for _ in range(5):
        x = rng.standard_normal(blk).astype(np.float32)
        ya, yb = np.empty(blk // M, np.float32), np.empty(blk // M, np.float32)
        oracle.orc_fir_decimate_f32(fptr(c), nt, M, fptr(sa), fptr(x), fptr(ya), blk, A0)
        ref.ref_fir_decimate(fptr(c), nt, M, fptr(sb), fptr(x), fptr(yb), blk)
        assert bits_equal(ya, yb) and bits_equal(sa[:nt - 1], sb[:nt - 1])


@pytest.mark.parametrize("nt,blk", [(63, 64), (127, 256), (9, 5), (8, 8), (2, 1), (130, 33)])
def test_fir_random(oracle, ref, nt, blk):
    rng = np.random.default_rng(nt * 7 + blk)
    c = rng.standard_normal(nt).astype(np.float32)
    sa, sb = np.zeros(nt - 1 + blk, np.float32), np.zeros(nt - 1 + blk, np.float32)
    for _ in range(5):
        x = rng.standard_normal(blk).astype(np.float32)
        ya, yb = np.empty(blk, np.float32), np.empty(blk, np.float32)
        oracle.orc_fir_f32(fptr(c), nt, fptr(sa), fptr(x), fptr(ya), blk, A0)
        ref.ref_fir(fptr(c), nt, fptr(sb), fptr(x), fptr(yb), blk)
        assert bits_equal(ya, yb) and bits_equal(sa[:nt - 1], sb[:nt - 1])


@pytest.mark.parametrize("ns,blk", [(4, 256), (1, 3), (6, 10), (2, 4)])
def test_biquad_random(oracle, ref, ns, blk):
    rng = np.random.default_rng(ns * 31 + blk)
    c = rc.design_bandpass(ns, 0.02 + 0.01 * ns, 3.0)
    sa, sb = np.zeros(4 * ns, np.float32), np.zeros(4 * ns, np.float32)
    for _ in range(6):
        x = rng.standard_normal(blk).astype(np.float32)
        ya, yb = np.empty(blk, np.float32), np.empty(blk, np.float32)
        oracle.orc_biquad_cascade_df1_f32(fptr(c), ns, fptr(sa), fptr(x), fptr(ya), blk, A0)
        ref.ref_biquad(fptr(c), ns, fptr(sb), fptr(x), fptr(yb), blk)
        assert bits_equal(ya, yb) and bits_equal(sa, sb)


def test_decimate_init_status_codes(ref):
    assert ref.ref_fir_decimate_init_status(8, 4, 256) == rc.SUCCESS
    assert ref.ref_fir_decimate_init_status(8, 4, 255) == rc.LENGTH_ERROR     # arm_fir_decimate_init_f32.c:74-97
    o = rc.CpuChain(rc.ChainSpec(1, 255, decim=4, nd_taps=8), "orc")
    assert o.rc == rc.LENGTH_ERROR


@pytest.mark.parametrize("name", ["cfg1", "cfg2", "cfg3", "cfg4"])
def test_chain_streaming(name, ref):
    spec = rc.baseline_spec(name, 4)
    a, b = rc.CpuChain(spec, "orc"), rc.CpuChain(spec, "ref")
    for call in range(4):
        iq = rc.synth_iq(3, 4, call * 768, 768)
        assert bits_equal(a.process(iq), b.process(iq))
    sa, sb = a.state(), b.state()
    for k in sa:
        assert np.array_equal(sa[k].view(np.uint32), sb[k].view(np.uint32)), k


@pytest.mark.parametrize("mode", [rc.MODE_LSB, rc.MODE_USB, rc.MODE_CW, rc.MODE_CWR, rc.MODE_AM, rc.MODE_FM, rc.MODE_DIG, rc.MODE_PKT])
def test_chain_all_modes_and_mode_switch(mode, ref):
    spec = rc.ChainSpec(2, 96, decim=3, nd_taps=40, nh_taps=33, n_biquad=3, mode=mode, nco=True,
                        nco_steps=np.array([0x00123456, 0xFFF00000], np.uint32), bp_f0=0.04, bp_q=5.0)
    a, b = rc.CpuChain(spec, "orc"), rc.CpuChain(spec, "ref")
    for call, m2 in enumerate([mode, rc.MODE_USB, mode]):
        a.set_mode(m2), b.set_mode(m2)
        iq = rc.synth_iq(0, 2, call * 192, 192)
        assert bits_equal(a.process(iq), b.process(iq))


@pytest.mark.parametrize("blk,M,nd,nh", [(64, 4, 48, 31), (256, 4, 256, 63), (256, 1, 0, 127), (96, 4, 256, 63), (64, 2, 128, 3), (5, 1, 3, 3)])
def test_fm_chain_against_the_cmsis_composition(blk, M, nd, nh, ref):
    """FM (rxtx_if.h:41; build-defined chain): the real arm_cmplx_conj_f32 + arm_cmplx_mult_cmplx_f32 + arm_scale_f32
    and the real arm_fir_f32 moving the delay lines, around the stated arctangent -- against the restatement, output and state, across
    calls of different lengths (the sample in front of a block comes from the delay line)."""
    mk = lambda: rc.ChainSpec(5, blk, M, nd, nh, 0, rc.MODE_FM, rc.ARITH_CMSIS, nco=True, nco_step_all=0x01234567, agc=True)
    a, b = rc.CpuChain(mk(), "orc"), rc.CpuChain(mk(), "ref")
    assert a.ok() and b.ok()
    pos = 0
    for bs in (blk, 3 * blk, 8 * blk, blk):
        iq = rc.synth_iq(0, 5, pos, bs)
        pos += bs
        assert bits_equal(a.process(iq), b.process(iq)), bs
    sa, sb = a.state(), b.state()
    for k in sa:
        assert np.array_equal(sa[k].view(np.uint32), sb[k].view(np.uint32)), k


def test_fm_needs_the_delay_lines(ref):
    for which in ("orc", "ref"):
        assert not rc.CpuChain(rc.ChainSpec(1, 64, 1, 0, 0, 0, rc.MODE_FM), which).ok()
        assert not rc.CpuChain(rc.ChainSpec(1, 64, 1, 0, 1, 0, rc.MODE_FM), which).ok()
        ch = rc.CpuChain(rc.ChainSpec(1, 64, 1, 0, 0, 0, rc.MODE_USB), which)
        assert ch.ok() and ch.set_mode(rc.MODE_FM) == rc.ARGUMENT_ERROR and ch.set_mode(rc.MODE_AM) == 0


def test_chain_q15_and_global_gain(ref):
    spec = rc.baseline_spec("cfg3", 3, agc_global=True)
    a, b = rc.CpuChain(spec, "orc"), rc.CpuChain(spec, "ref")
    iq = rc.synth_iq(0, 3, 0, 2048)
    (ya, ea), (yb, eb) = a.process_env(iq), b.process_env(iq)
    assert bits_equal(ya, yb) and bits_equal(ea, eb)
    spec = rc.baseline_spec("cfg1", 2)
    a, b = rc.CpuChain(spec, "orc"), rc.CpuChain(spec, "ref")
    q = (np.random.default_rng(5).integers(-20000, 20000, (2, 512, 2))).astype(np.int16)
    assert np.array_equal(a.process_q15(q), b.process_q15(q))


@pytest.mark.parametrize("shape,block", [((256, 4, 63), 96), ((0, 1, 63), 96), ((128, 2, 63), 96), ((256, 4, 127), 192), ((0, 1, 127), 48)])
@pytest.mark.parametrize("q15", [False, True])
def test_chain_in_the_firmware_block_geometry(shape, block, q15, ref):
    """DSP blocks of 96 (192, 48) frames -- the I2S half-buffer of the firmware (Core/Inc/dsp_if.h:69-73), one slot per call and
    several: the restatement against the composition of the real CMSIS-DSP primitives, bit for bit.  The GPU tests of
    tests/test_gpu_short_calls.py compare the fused kernel with the restatement on exactly these configurations."""
    nd, M, nh = shape
    spec = rc.ChainSpec(5, block, M, nd, nh, 0, rc.MODE_LSB, rc.ARITH_CMSIS, nco=True, nco_step_all=0x00c00000, agc=True)
    a, b = rc.CpuChain(spec, "orc"), rc.CpuChain(spec, "ref")
    pos = 0
    for bs in (block, 2 * block, 10 * block, 11 * block, block):
        iq = rc.synth_iq(0, 5, pos, bs)
        pos += bs
        if q15:
            q = np.clip(np.round(iq * 20000.0), -32768, 32767).astype(np.int16)
            assert np.array_equal(a.process_q15(q), b.process_q15(q)), bs
        else:
            assert bits_equal(a.process(iq), b.process(iq)), bs
    sa, sb = a.state(), b.state()
    for k in sa:
        assert np.array_equal(sa[k].view(np.uint32), sb[k].view(np.uint32)), k
