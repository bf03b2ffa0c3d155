"""GPU: batched DSP ring buffers (include/selenite_ring.h, csrc/ring.hip) against the oracle
(oracle/ring_oracle.c) -- integer work, so every read, every pointer, every flag and every ring
slot must be identical.  Reference semantics: Core/Src/dsp_if.c:83-340."""
import os

import numpy as np
import pytest

import rxcommon as rc
from rxcommon import OracleRing
from test_ring_oracle import packet, random_ops

pytestmark = pytest.mark.gpu


def gpu_ring(channels, frames=384):
    import selenite_rx as sr
    return sr.Ring(channels, frames)


def assert_same_state(g, o):
    sg, so = g.state(), o.state()
    for k in so:
        assert np.array_equal(sg[k], so[k]), k


@pytest.mark.parametrize("channels,frames", [(1, 384), (5, 384), (257, 384), (7, 8), (64, 33), (3, 32767)])
def test_random_traffic_matches_oracle(channels, frames):
    rng = np.random.default_rng(channels * 100003 + frames)
    g, o = gpu_ring(channels, frames), OracleRing(channels, frames)
    st = o.state()
    st["rd_ptr"][:] = rng.integers(0, frames, channels)
    st["wr_ptr"][:] = rng.integers(0, frames, channels)
    st["buff_enable"][:] = rng.integers(0, 2, channels)
    st["i"][...] = rng.integers(-32768, 32768, st["i"].shape)
    st["q"][...] = rng.integers(-32768, 32768, st["q"].shape)
    o.set_state(st)
    g.set_state(st)
    for kind, nfr, base in random_ops(rng, 60, min(frames, 400)):
        if kind.endswith("write"):
            pkt = rng.integers(-32768, 32768, (channels, 2 * nfr), dtype=np.int64).astype(np.int16)
            getattr(g, kind)(pkt)
            getattr(o, kind)(pkt)
        elif kind == "in_read":
            assert np.array_equal(g.in_read(4 * nfr), o.in_read(4 * nfr))
        else:
            assert np.array_equal(g.out_read(2 * nfr), o.out_read(2 * nfr))
    assert_same_state(g, o)


def test_committed_fixture_and_mute():
    fx = np.load(os.path.join(rc.ROOT, "tests", "golden", "ring_trace.npz"))
    from golden.make_ring_golden import replay
    g = gpu_ring(int(fx["channels"]), int(fx["frames"]))
    outs, state = replay(g)
    assert np.array_equal(np.concatenate([x.ravel() for x in outs]), fx["reads"])
    for k in ("i", "q", "buff_enable", "rd_ptr", "wr_ptr"):
        assert np.array_equal(state[k], fx[k]), k
    g.mute()
    s = g.state()
    assert not s["i"].any() and not s["q"].any()
    assert np.array_equal(s["wr_ptr"], fx["wr_ptr"]) and np.array_equal(s["rd_ptr"], fx["rd_ptr"])


def test_hand_traces_of_the_reference_source_on_the_gpu():
    g = gpu_ring(1)
    g.out_write(packet(1, 48, 1))
    g.out_read(96)
    for base in (101, 201, 301, 401):
        g.out_write(packet(1, 48, base))
    s = g.state()                                            # same numbers as test_ring_oracle's slip trace
    assert s["wr_ptr"][0] == 47 and s["i"][0, 383] == 401 and s["i"][0, 0] == 402 and s["i"][0, 47] == 448


def test_bad_sizes_are_refused_with_sticky_status():
    import selenite_rx as sr
    g = gpu_ring(2)
    with pytest.raises(sr.RxError) as e:
        g.in_read(6)                                         # 1.5 frames
    assert e.value.code == rc.LENGTH_ERROR
    h = sr.C.c_void_p()
    assert sr.lib().selenite_ring_init(sr.C.byref(h), 4, 40000) == rc.LENGTH_ERROR     # uint16 gap would wrap


def test_slot_roundtrip_ring_to_rx_chain_to_ring_full_size():
    """The callback pair of dsp_if.c:50-67 at batch scale: IN ring -> q15 RX chain -> OUT ring, all
    on the device; sampled channels checked against oracle ring + oracle chain."""
    import selenite_rx as sr
    nch, nblk = 8192, 3
    spec = rc.baseline_spec("cfg1", nch)
    rx, ring_in, ring_out = sr.Rx(spec.config()), gpu_ring(nch), gpu_ring(nch)
    chans = [0, 1, 63, 64, 4095, nch - 1]
    o_rx = rc.CpuChain(rc.baseline_spec("cfg1", len(chans)), "orc")
    o_in, o_out = OracleRing(len(chans)), OracleRing(len(chans))
    d_pkt, d_aud = sr.DeviceBuffer(nch * 512 * 2), sr.DeviceBuffer(nch * 256 * 2)
    L = sr.lib()
    for b in range(nblk):
        iq = rc.synth_iq(0, nch, b * 256, 256)
        q = np.clip(np.trunc(iq * 32768.0), -32768, 32767).astype(np.int16).reshape(nch, 512)
        ring_in.in_write(q)
        L.selenite_ring_in_read_device(ring_in.h, d_pkt.ptr, 1024)
        ring_in.L.selenite_ring_sync(ring_in.h)
        rx.process_q15_device(d_pkt.ptr, d_aud.ptr, 256)
        rx.sync()
        # audio is mono: 256 words = 128 "frames" for the OUT ring
        L.selenite_ring_out_write_device(ring_out.h, d_aud.ptr, 512)
        got = ring_out.out_read(256)
        ring_in.check(); ring_out.check(); rx.check()
        o_in.in_write(q[chans])
        aud = o_rx.process_q15(o_in.in_read(1024).reshape(len(chans), 256, 2))
        o_out.out_write(aud)
        assert np.array_equal(got[chans], o_out.out_read(256)), "block %d" % b
