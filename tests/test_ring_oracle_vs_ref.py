"""CPU, build container: the ring-buffer restatement (oracle/ring_oracle.c) against the reference's OWN code --
Core/Src/dsp_if.c compiled from /root/reference (oracle/Makefile -> oracle/_ref/libdsp_if_ref.so, harness
oracle/ref_ring.c).  Same calls, same packets, every returned word and the whole ring state compared.  This pins
SURVEY.md 8 row f2; the committed fixture tests/golden/ring_trace.npz is generated from the reference by
tests/golden/make_ring_golden.py and checked in test_ring_oracle.py / test_gpu_ring.py where /root/reference is absent."""
import os
import sys

import numpy as np
import pytest

import rxcommon as rc
from rxcommon import OracleRing, RefRing

sys.path.insert(0, os.path.join(rc.ROOT, "tests", "golden"))
pytestmark = pytest.mark.skipif(not rc.ring_ref_available(), reason="oracle/_ref/libdsp_if_ref.so not built (no /root/reference)")


def same(a, b):
    sa, sb = a.state(), b.state()
    return all(np.array_equal(sa[k], sb[k]) for k in sa)


def test_firmware_ring_geometry():
    r = RefRing(1)
    assert r.frames == 768 and r.L.ref_ring_audio_freq() == 96000       # usbd_audio.h:46 wins over dsp_if.h:55-57


def test_fixture_trace_reference_equals_restatement():
    from make_ring_golden import CHANNELS, FRAMES, replay
    ro, so = replay(OracleRing(CHANNELS, FRAMES))
    rr, sr_ = replay(RefRing(CHANNELS, FRAMES))
    assert len(ro) == len(rr) and all(np.array_equal(x, y) for x, y in zip(ro, rr))
    assert all(np.array_equal(so[k], sr_[k]) for k in so)
    fx = np.load(os.path.join(rc.GOLDEN_DIR, "ring_trace.npz"))
    assert np.array_equal(fx["reads"], np.concatenate([o.ravel() for o in rr]))      # the committed fixture IS the reference's output


@pytest.mark.parametrize("seed", range(12))
def test_random_traffic_reference_equals_restatement(seed):
    """Random op sequences from random pointer positions: unprimed rings, both slip directions (gap > 3/4 and < 1/4 of
    the ring), the reset-instead-of-wrap of the first read (dsp_if.c:320-323), mute, odd packet sizes."""
    rng = np.random.default_rng(1000 + seed)
    C_ = 6
    o, r = OracleRing(C_, 768), RefRing(C_)
    st = o.state()
    st["rd_ptr"][:] = rng.integers(0, 768, C_)
    st["wr_ptr"][:] = rng.integers(0, 768, C_)
    st["buff_enable"][:] = rng.integers(0, 2, C_)
    st["i"][...] = rng.integers(-32768, 32768, st["i"].shape)
    st["q"][...] = rng.integers(-32768, 32768, st["q"].shape)
    o.set_state(st); r.set_state(st)
    for step in range(120):
        kind = int(rng.integers(0, 9))
        nfr = int(rng.choice([1, 2, 47, 48, 49, 96, 97, 191, 192, 400]))
        pkt = rng.integers(-32768, 32768, (C_, 2 * nfr)).astype(np.int16)
        if kind in (0, 1):
            o.in_write(pkt); r.in_write(pkt)
        elif kind in (2, 3):
            assert np.array_equal(o.in_read(4 * nfr), r.in_read(4 * nfr)), (seed, step)
        elif kind in (4, 5):
            o.out_write(pkt); r.out_write(pkt)
        elif kind in (6, 7):
            assert np.array_equal(o.out_read(2 * nfr), r.out_read(2 * nfr)), (seed, step)
        else:
            o.mute(); r.mute()
        assert same(o, r), (seed, step, kind, nfr)
