"""SELENITE_ARITH_AUTO in one launch or three (selenite_rx_set_auto_launches; rx_split16_kernels.h: k_hilb_split16, FusedArgs::inl).

One launch (default, the no-decimator shapes): the workgroup of the matrix kernel recomputes the channel it guarded -- or one the
bit-exact arithmetic holds -- itself, with the body of the bit-exact kernel.  Three launches: k_hist_exact (dense list) and the rerun
pass of the bit-exact kernel follow the matrix kernel (the decimating shapes always do).  Which arithmetic serves a channel is
decided by the channel's word alone, so audio, streaming state, channel words and guard counters must be the SAME BITS in both forms,
call after call -- on workloads where channels are flagged, held, released and flagged again, f32 and int16 slots."""
import numpy as np
import pytest

import rxcommon as rc
from rxcommon import ARITH_AUTO, ARITH_CMSIS, CpuChain, bits_equal, synth_iq

pytestmark = pytest.mark.gpu


def mixed_iq(nch, pos, n, call):
    """bench signal with level steps, which is what the guard looks at (a block 12 dB under the largest sample its pass, the pass before
    it or the call before it saw): every third channel drops 40 dB at the start of calls 1 and 5 and comes back in calls 3 and 6
    (flagged, held, released, flagged again); every seventh alternates call by call (held for good); every eleventh drops in the middle
    of call 3 (flagged inside the call); channel 5 is silent."""
    iq = synth_iq(0, nch, pos, n)
    ch = np.arange(nch)
    if call in (1, 2, 5):
        iq[ch % 3 == 1] *= np.float32(0.01)
    if call % 2 == 1:
        iq[ch % 7 == 3] *= np.float32(0.01)
    if call == 3:
        iq[ch % 11 == 4, n // 2:] *= np.float32(0.01)
    if call > 3:
        iq[ch % 11 == 4] *= np.float32(0.01)
    iq[5] = 0.0
    return iq


def run(spec, form, calls, q15):
    import selenite_rx as sr
    g = sr.Rx(spec.config())
    assert g.set_auto_launches(form) == 0
    outs, forms = [], []
    pos = 0
    for call, bs in enumerate(calls):
        iq = mixed_iq(spec.channels, pos, bs, call)
        pos += bs
        if q15:
            y = g.process_q15(np.clip(np.round(iq * 20000.0), -32768, 32767).astype(np.int16))
        else:
            y = g.process(iq)
        outs.append(y.copy())
        forms.append(g.auto_launches_last())
    st = g.state()
    words = g.auto_words()
    stats = g.guard_stats()
    gch = g.guard_channels()
    g.close()
    return outs, st, words, stats, gch, forms


# (shape, DSP block): k_hilb_split16 (whole, 192- and 128-sample passes); a decimating shape for the record (three launches either way)
CASES = [((0, 1, 127), 256), ((0, 1, 63), 192), ((0, 1, 31), 256), ((0, 1, 127), 128), ((256, 4, 63), 256)]


@pytest.mark.parametrize("nco", ["shared", "grid", "per_channel"])
@pytest.mark.parametrize("q15", [False, True])
@pytest.mark.parametrize("shape,block", CASES)
def test_one_launch_and_three_launches_give_the_same_bits(shape, block, q15, nco):
    nd, M, nh = shape
    nch = 203
    if nco == "shared":
        kw = dict(nco=True, nco_steps=np.full(nch, 0x0B000000, np.uint32))
    elif nco == "grid":
        kw = dict(nco=True, nco_steps=((np.arange(nch, dtype=np.uint64) * 7 % 200 + 1) << 24).astype(np.uint32))
    else:
        kw = dict(nco=True, nco_steps=(np.arange(nch, dtype=np.uint64) * 0x9E3779B1 % (1 << 32)).astype(np.uint32))
    spec = rc.ChainSpec(nch, block, M, nd, nh, 0, rc.MODE_USB, ARITH_AUTO, **kw)
    unit = 1024 if M > 1 else 768
    calls = (4 * unit, 2 * unit, 2 * unit, 4 * unit, 2 * unit, 2 * unit, 2 * unit)
    a = run(spec, 1, calls, q15)
    b = run(spec, 3, calls, q15)
    assert set(a[5]) == ({1} if nd == 0 else {3}) and set(b[5]) == {3}, (a[5], b[5])      # the forms really were the two
    for k, (ya, yb) in enumerate(zip(a[0], b[0])):
        assert np.array_equal(ya.view(np.uint32 if ya.dtype == np.float32 else np.int16), yb.view(np.uint32 if yb.dtype == np.float32 else np.int16)), k
    for key in a[1]:
        assert np.array_equal(np.ascontiguousarray(a[1][key]).view(np.uint8), np.ascontiguousarray(b[1][key]).view(np.uint8)), key
    assert np.array_equal(a[2], b[2])
    assert a[3] == b[3]
    assert np.array_equal(a[4], b[4])
    assert a[3]["rerun_channel_calls"] > 0 and a[3]["handover_blocks"] == 0
    assert (a[2] & 32).any()                                            # some channel is held at the end


def test_the_one_launch_form_holds_the_bar_and_a_recomputed_channel_is_cmsis_bit_for_bit():
    """against the oracle: every DSP block within 1e-5 of its maximum; a channel the matrix kernel's workgroup recomputed itself -- or
    holds -- carries the bit-exact chain's audio (AGC off: the gain a matrix-kernel call left is ~1e-6 off the reference's)."""
    import selenite_rx as sr
    nch, block = 96, 256
    kw = dict(nco=True, nco_steps=np.full(nch, 0x0B000000, np.uint32), agc=False)
    g = sr.Rx(rc.ChainSpec(nch, block, 1, 0, 127, 0, rc.MODE_USB, ARITH_AUTO, **kw).config())
    assert g.set_auto_launches(1) == 0
    o = CpuChain(rc.ChainSpec(nch, block, 1, 0, 127, 0, rc.MODE_USB, ARITH_CMSIS, **kw), "orc")
    pos, seen_held = 0, 0
    for call in range(7):
        iq = mixed_iq(nch, pos, 3072, call)
        pos += 3072
        yg, yo = g.process(iq), o.process(iq)
        assert g.auto_launches_last() == 1
        d = np.abs(yg.astype(np.float64) - yo).reshape(nch, -1, block).max(axis=2)
        m = np.abs(yo).reshape(nch, -1, block).max(axis=2)
        assert (d <= 1e-5 * m).all(), (call, (d / np.maximum(m, 1e-30)).max())
        held = (g.auto_words() & 32) != 0
        seen_held += int(held.sum())
        if held.any():
            assert bits_equal(yg[held], yo[held]), call                 # recomputed in this call: the reference's bits
    assert seen_held > 0
    st = g.guard_stats()
    assert st["handover_blocks"] == 0 and st["rerun_channel_calls"] > 0
    g.close()
