"""Short calls and the firmware's own block geometry on the fused kernels (VERDICT r2 #4 / #6b).

The I2S half-buffer of the firmware carries 96 I/Q frames per millisecond (Core/Inc/dsp_if.h:69-73, the callbacks at
Core/Src/dsp_if.c:50-67): a DSP block of 96 inputs is 24 audio samples behind the /4 decimator -- not a divisor of the
256-output pass of the fused kernels, which therefore used to hand such configurations (and every call that was not a whole
number of passes, in the exact / fma arithmetic) to the generic kernels.  k_ssb_fused now runs passes of variable length
(the largest whole number of DSP blocks in 256 outputs; the call's last pass whatever is left).  Exact arithmetic: bit-exact
against the oracle, output and state, for any sequence of call lengths."""
import numpy as np
import pytest

import rxcommon as rc
from rxcommon import ARITH_AUTO, ARITH_CMSIS, ARITH_FMA, CpuChain, bits_equal, synth_iq

pytestmark = pytest.mark.gpu


def run_pair(spec_g, spec_o, lengths, q15=False, exact=True, na=None):
    import selenite_rx as sr
    g, o = sr.Rx(spec_g.config()), CpuChain(spec_o, "orc")
    nch, pos = spec_g.channels, 0
    for bs in lengths:
        iq = synth_iq(0, nch, pos, bs)
        pos += bs
        if q15:
            iq16 = np.clip(np.round(iq * 20000.0), -32768, 32767).astype(np.int16)
            yg, yo = g.process_q15(iq16), o.process_q15(iq16)
            assert np.array_equal(yg, yo) if exact else np.abs(yg.astype(np.int32) - yo).max() <= 1, bs
        else:
            yg, yo = g.process(iq), o.process(iq)
            if exact:
                assert bits_equal(yg, yo), (bs, rc.rel_err(yg, yo))
            else:
                d = np.abs(yg.astype(np.float64) - yo).reshape(nch, -1, na).max(axis=2)
                m = np.abs(yo).reshape(nch, -1, na).max(axis=2)
                bar = 1e-5 * m       # the PLAIN bar on every block (round 4: a call too short to keep the samples k_hist_exact needs runs on
                                     # the bit-exact kernel in AUTO, so no block ever starts from a Hilbert-pair history of split16 precision)
                assert (d <= bar).all(), (bs, (d / np.maximum(m, 1e-30)).max())
    sg, so = g.state(), o.state()
    if exact:
        for key in sg:
            assert (bits_equal(sg[key], so[key]) if sg[key].dtype == np.float32 else np.array_equal(sg[key], so[key])), key
    else:
        assert bits_equal(sg["dec_state"], so["dec_state"]) and np.array_equal(sg["nco_phase"], so["nco_phase"])
        assert g.guard_stats()["handover_blocks"] == 0
    name = g.kernel_name()
    g.close()
    return name


@pytest.mark.parametrize("shape", [(256, 4, 63), (128, 2, 63), (256, 8, 63), (0, 1, 127), (64, 4, 63), (128, 4, 31)])
@pytest.mark.parametrize("arith", [ARITH_CMSIS, ARITH_FMA])
@pytest.mark.parametrize("nco", ["shared", "per_channel"])
def test_any_call_length_on_the_exact_fused_kernels(shape, arith, nco):
    """Calls of 1, 2, 3, 5, 17 DSP blocks in any order (never a whole number of passes except by accident)."""
    nd, M, nh = shape
    nch = 37
    kw = dict(nco=True, nco_step_all=0x01234567) if nco == "shared" else \
        dict(nco=True, nco_steps=(np.arange(nch, dtype=np.uint64) * 0x9E3779B1 % (1 << 32)).astype(np.uint32))
    mk = lambda: rc.ChainSpec(nch, 256, M, nd, nh, 0, rc.MODE_USB, arith, **kw)
    name = run_pair(mk(), mk(), [256, 512, 768, 256, 1280, 4352, 256, 2048])
    assert name.startswith("k_ssb_fused") or name.startswith("k_ssb_mfma")


@pytest.mark.parametrize("shape,block", [((256, 4, 63), 96), ((0, 1, 63), 96), ((128, 2, 63), 96), ((256, 4, 127), 192), ((0, 1, 127), 48)])
@pytest.mark.parametrize("arith", [ARITH_CMSIS, ARITH_FMA, ARITH_AUTO])
@pytest.mark.parametrize("q15", [False, True])
def test_the_firmware_block_geometry_runs_on_the_fused_kernel(shape, block, arith, q15):
    """DSP blocks of 96 (192, 48) frames: AGC groups of 6 (12, 24, 48) lanes, passes of 240 (192) audio samples.  Calls of one
    slot (96 frames, dsp_if.c:50-67), two, ten, eleven, fifty.  Exact / fma arithmetic: bit-exact on k_ssb_fused.  AUTO: whole
    240- / 192-output passes on the matrix kernels (k_ssb_split16; round 4: k_hilb_split16 for the no-decimator shapes), plain bar."""
    nd, M, nh = shape
    nch = 21
    kw = dict(nco=True, nco_step_all=0x00c00000, agc=True)
    ref = ARITH_CMSIS if arith == ARITH_AUTO else arith
    matrix = arith == ARITH_AUTO
    name = run_pair(rc.ChainSpec(nch, block, M, nd, nh, 0, rc.MODE_LSB, arith, **kw),
                    rc.ChainSpec(nch, block, M, nd, nh, 0, rc.MODE_LSB, ref, **kw),
                    [block, 2 * block, 10 * block, 11 * block, block, 50 * block], q15=q15, exact=not matrix, na=block // M)
    assert name.startswith(("k_hilb_split16<%d>" % nh) if matrix and nd == 0 else ("k_ssb_split16<%d,%d,%d>" if matrix else "k_ssb_fused<%d,%d,%d>") % shape)


@pytest.mark.parametrize("shape,block", [((256, 4, 63), 96), ((256, 4, 63), 192), ((256, 4, 63), 384), ((128, 2, 63), 96),
                                         ((256, 4, 127), 192), ((128, 4, 31), 96)])
@pytest.mark.parametrize("arith", [rc.ARITH_SPLIT16, ARITH_AUTO])
@pytest.mark.parametrize("nco", ["shared", "grid", "per_channel", "off"])
def test_the_firmware_block_geometry_on_the_matrix_kernel(shape, block, arith, nco):
    """k_ssb_split16 with passes of 240 (192) audio samples: the largest whole number of 24- (48-, 96-) sample DSP blocks in a
    256-output tile; the tile is computed in full, its last outputs dropped, the histories advance by 960 (768) samples.  Calls
    of whole passes, of passes + a tail that holds a decimator history (same launch), + a shorter tail (cut off: k_ssb_fused),
    one-slot calls in between (bit-exact kernel).  Raw split16: input-referred bar; AUTO: plain bar.  Decimator state bit-exact."""
    import selenite_rx as sr
    nd, M, nh = shape
    nch, na = 29, block // M
    pq = 256 // na * na
    steps = (np.arange(nch, dtype=np.uint64) * 0x9E3779B1 % (1 << 32)).astype(np.uint32)
    kw = {"shared": dict(nco=True, nco_step_all=0x00c00000), "grid": dict(nco=True, nco_step_all=0x03000000),
          "per_channel": dict(nco=True, nco_steps=steps), "off": dict()}[nco]
    kw["agc"] = arith == ARITH_AUTO
    g = sr.Rx(rc.ChainSpec(nch, block, M, nd, nh, 0, rc.MODE_USB, arith, **kw).config())
    o = CpuChain(rc.ChainSpec(nch, block, M, nd, nh, 0, rc.MODE_USB, ARITH_CMSIS, **kw), "orc")
    assert g.kernel_name().startswith("k_ssb_split16<%d,%d,%d>" % shape)
    unit, pos = pq * M, 0
    for bs in (unit, 3 * unit, block, 2 * unit + (unit // block - 1) * block, unit + block, 5 * unit, 2 * block, 7 * unit + 4 * block):
        iq = synth_iq(0, nch, pos, bs)
        pos += bs
        yg, yo = g.process(iq), o.process(iq)
        d = np.abs(yg.astype(np.float64) - yo).reshape(nch, -1, na).max(axis=2)
        m = np.abs(yo).reshape(nch, -1, na).max(axis=2)
        bar = 1e-5 * m if arith == ARITH_AUTO else 1e-5 * m + 1e-6 * np.abs(iq).max()
        assert (d <= bar).all(), (bs, (d / np.maximum(m, 1e-30)).max())
        assert bits_equal(g.state()["dec_state"], o.state()["dec_state"]), bs
        assert np.array_equal(g.state()["nco_phase"], o.state()["nco_phase"]), bs
    g.close()


@pytest.mark.parametrize("q15", [False, True])
def test_auto_on_short_calls(q15):
    """256 / 512 / 768-sample calls of the cfg3 chain in SELENITE_ARITH_AUTO: a partial pass that holds a whole decimator history
    goes to the matrix kernel (+ rerun of guarded channels through the variable-length pass of k_ssb_fused); plain 1e-5 bar."""
    nch = 48
    rng = np.random.default_rng(2)
    steps = rng.integers(0, 1 << 32, nch, dtype=np.uint64).astype(np.uint32)
    kw = dict(nco=True, nco_steps=steps)
    mk = lambda a: rc.ChainSpec(nch, 256, 4, 256, 63, 0, rc.MODE_USB, a, **kw)
    run_pair(mk(ARITH_AUTO), mk(ARITH_CMSIS), [256, 512, 768, 256, 1024, 1280, 256, 256], q15=q15, exact=False, na=64)


def test_am_and_global_gain_with_partial_passes():
    import selenite_rx as sr
    nch = 19
    kw = dict(nco=True, nco_step_all=0x01000000)
    mk = lambda a, **k: rc.ChainSpec(nch, 256, 4, 256, 63, 0, rc.MODE_AM, a, **dict(kw, **k))
    run_pair(mk(ARITH_CMSIS), mk(ARITH_CMSIS), [768, 256, 1280])
    g = sr.Rx(rc.ChainSpec(nch, 256, 4, 256, 63, 0, rc.MODE_USB, ARITH_CMSIS, agc_global=True, **kw).config())
    o = CpuChain(rc.ChainSpec(nch, 256, 4, 256, 63, 0, rc.MODE_USB, ARITH_CMSIS, agc_global=True, **kw), "orc")
    pos = 0
    for bs in (768, 256, 2304):
        iq = synth_iq(0, nch, pos, bs)
        pos += bs
        yo, _ = o.process_env(iq)
        assert bits_equal(g.process(iq), yo), bs
    g.close()


@pytest.mark.parametrize("arith", [rc.ARITH_SPLIT16, ARITH_AUTO])
def test_a_tail_too_short_for_the_matrix_kernel_is_cut_off_and_runs_on_the_variable_length_kernel(arith):
    """DSP block 128 behind the 256-tap /4 decimator: a call of 1024 k + 128 samples ends in a partial pass that holds less than
    a decimator history -- the whole passes stay on k_ssb_split16, the tail goes to k_ssb_fused (fused_tail_split), same
    streaming state.  Raw split16: input-referred bar; AUTO: plain bar."""
    import selenite_rx as sr
    nch, na = 33, 32
    kw = dict(nco=True, nco_step_all=0x01000000, agc=arith == ARITH_AUTO)
    g = sr.Rx(rc.ChainSpec(nch, 128, 4, 256, 63, 0, rc.MODE_USB, arith, **kw).config())
    o = CpuChain(rc.ChainSpec(nch, 128, 4, 256, 63, 0, rc.MODE_USB, ARITH_CMSIS, **kw), "orc")
    assert g.kernel_name().startswith("k_ssb_split16<256,4,63>")
    pos = 0
    for bs in (1152, 128, 2176, 1024, 3200):
        iq = synth_iq(0, nch, pos, bs)
        pos += bs
        yg, yo = g.process(iq), o.process(iq)
        d = np.abs(yg.astype(np.float64) - yo).reshape(nch, -1, na).max(axis=2)
        m = np.abs(yo).reshape(nch, -1, na).max(axis=2)
        bar = 1e-5 * m if arith == ARITH_AUTO else 1e-5 * m + 1e-6 * np.abs(iq).max()
        assert (d <= bar).all(), (bs, (d / np.maximum(m, 1e-30)).max())
        assert bits_equal(g.state()["dec_state"], o.state()["dec_state"]), bs
    g.close()


@pytest.mark.parametrize("stages,block", [(2, 96), (2, 128), (2, 192), (4, 128), (4, 192), (4, 512), (8, 128), (8, 512)])
@pytest.mark.parametrize("q15", [False, True])
def test_cw_kernel_for_other_dsp_blocks(stages, block, q15):
    """k_cw_fused instantiated for DSP blocks of 96 (two stages: one firmware slot), 128, 192 and 512 samples beside the
    BASELINE 256: the systolic array needs whole input chunks of 1024 / (64 / stages) samples.  Bit-exact like cfg4
    (arm_biquad_cascade_df1_f32.c:165-407), ragged channel counts, per-channel NCO."""
    import selenite_rx as sr
    nch = 37
    kw = dict(nco=True, nco_steps=(np.arange(nch, dtype=np.uint32) * np.uint32(0x00123457) + np.uint32(0x00400000)), agc=True,
              bp_q=2.0 if stages == 8 else 4.0)
    spec = rc.ChainSpec(nch, block, 1, 0, 0, stages, rc.MODE_CW, ARITH_CMSIS, **kw)
    name = run_pair(spec, spec, [block, 3 * block, 2 * block, 7 * block], q15=q15)
    assert name == "k_cw_fused<%d,%d>" % (stages, block)


@pytest.mark.parametrize("nh,block", [(127, 192), (63, 192), (127, 96), (31, 160), (127, 128)])
@pytest.mark.parametrize("arith", [rc.ARITH_SPLIT16, ARITH_AUTO])
@pytest.mark.parametrize("nco", ["shared", "per_channel", "off"])
def test_the_hilbert_matrix_kernel_with_passes_of_fewer_than_256_outputs(nh, block, arith, nco):
    """k_hilb_split16 (no decimator) with DSP blocks that do not divide the 256-sample tile: passes of the largest whole number of
    blocks in it (192 for blocks of 192 frames -- BASELINE cfg2's literal 48 000 samples are 250 of them --, 192 = two blocks of 96,
    160, 256 = two of 128); the last pass of a call may be shorter (round 4, late: calls of one block, of passes + one block).
    Raw split16: input-referred bar; AUTO: plain bar.  The FIR pair's state (the mixed samples) bit-exact after every call."""
    import selenite_rx as sr
    nch, na = 27, block
    pq = 256 // na * na
    steps = (np.arange(nch, dtype=np.uint64) * 0x9E3779B1 % (1 << 32)).astype(np.uint32)
    kw = {"shared": dict(nco=True, nco_step_all=0x00c00000), "per_channel": dict(nco=True, nco_steps=steps), "off": dict()}[nco]
    kw["agc"] = True
    g = sr.Rx(rc.ChainSpec(nch, block, 1, 0, nh, 0, rc.MODE_USB, arith, **kw).config())
    o = CpuChain(rc.ChainSpec(nch, block, 1, 0, nh, 0, rc.MODE_USB, ARITH_CMSIS, **kw), "orc")
    assert g.kernel_name().startswith("k_hilb_split16<%d>" % nh), g.kernel_name()
    pos = 0
    for bs in (pq, 5 * pq, block, 3 * pq, 2 * pq + block if pq != block else 2 * pq, 25 * pq):
        iq = synth_iq(0, nch, pos, bs)
        pos += bs
        yg, yo = g.process(iq), o.process(iq)
        d = np.abs(yg.astype(np.float64) - yo).reshape(nch, -1, na).max(axis=2)
        m = np.abs(yo).reshape(nch, -1, na).max(axis=2)
        bar = 1e-5 * m if arith == ARITH_AUTO else 1e-5 * m + 1e-6 * np.abs(iq).max()
        assert (d <= bar).all(), (bs, (d / np.maximum(m, 1e-30)).max())
        assert bits_equal(g.state()["fir_state"], o.state()["fir_state"]), bs
        assert np.array_equal(g.state()["nco_phase"], o.state()["nco_phase"]), bs
    g.close()
