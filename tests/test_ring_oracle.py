"""CPU: the DSP ring-buffer oracle (oracle/ring_oracle.c) against traces derived by hand from the
reference source (Core/Src/dsp_if.c:83-340; ring of 384 frames = the 48 kHz fallback of dsp_if.h:55-57,81-84 -- the
firmware build itself has 768), against an independent frame-at-a-time Python model written from the same source,
and against the committed fixture, which is the output of the reference's own dsp_if.c
(tests/golden/make_ring_golden.py; live comparison: tests/test_ring_oracle_vs_ref.py)."""
import os

import numpy as np
import pytest

import rxcommon as rc
from rxcommon import OracleRing

N = 384


def packet(channels, frames, base):
    """frame k of channel c = (I, Q) = (base + k + 1000*c, -(base + k + 1000*c)) -- easy to recognise."""
    v = (base + np.arange(frames)[None, :] + 1000 * np.arange(channels)[:, None]).astype(np.int16)
    return np.stack([v, -v], axis=2).reshape(channels, 2 * frames)


def test_in_ring_unprimed_writer_skips_a_slot_per_packet_and_leaves_the_repeat():
    r = OracleRing(1, N)
    r.in_write(packet(1, 48, 1))            # dsp_if.c:252,277-285: gap stays 0 -> wr 0 -> 1
    s = r.state()
    assert s["wr_ptr"][0] == 49 and s["rd_ptr"][0] == 0 and s["buff_enable"][0] == 0
    assert s["i"][0, 0] == 0 and s["i"][0, 1] == 1 and s["i"][0, 48] == 48
    assert s["i"][0, 49] == 48 and s["q"][0, 49] == -48      # repeated last frame (:292-294)
    r.in_write(packet(1, 48, 101))
    s = r.state()
    assert s["wr_ptr"][0] == 98                              # 49 -> 50, 48 frames, filler at 98
    assert s["i"][0, 49] == 48 and s["i"][0, 50] == 101 and s["i"][0, 97] == 148 and s["i"][0, 98] == 148


def test_in_ring_first_read_parks_half_a_ring_behind_then_gap_is_live():
    r = OracleRing(1, N)
    r.in_write(packet(1, 48, 1))
    r.in_write(packet(1, 48, 101))                            # wr = 98
    out = r.in_read(192)                                      # 48 frames; rd = 98 + 192 = 290 (:316-326)
    assert np.all(out == 0)
    s = r.state()
    assert s["buff_enable"][0] == 1 and s["rd_ptr"][0] == 338
    r.in_write(packet(1, 48, 201))                            # gap = 98 + 384 - 338 = 144: no shift
    s = r.state()
    assert s["wr_ptr"][0] == 146 and s["i"][0, 98] == 201 and s["i"][0, 145] == 248 and s["i"][0, 146] == 248


def test_in_ring_first_read_resets_to_zero_instead_of_wrapping():
    r = OracleRing(1, N)
    s = r.state()
    s["wr_ptr"][0] = 200
    r.set_state(s)
    r.in_read(4)                                              # rd = 200 + 192 = 392 >= 384 -> 0 (sic, :320-323)
    assert r.state()["rd_ptr"][0] == 1


def test_out_ring_priming_slip_when_writer_is_faster():
    r = OracleRing(1, N)
    r.out_write(packet(1, 48, 1))                             # wr = rd + 192 (:124-134); frames 192..239
    s = r.state()
    assert s["buff_enable"][0] == 1 and s["wr_ptr"][0] == 240 and s["i"][0, 192] == 1 and s["i"][0, 240] == 48
    out = r.out_read(96)                                      # slots 0..47 (zeros), rd = 48
    assert np.all(out == 0) and r.state()["rd_ptr"][0] == 48
    r.out_write(packet(1, 48, 101))                           # gap 192
    r.out_write(packet(1, 48, 201))                           # gap 240
    r.out_write(packet(1, 48, 301))                           # gap 288: not > 288; wraps: filler at 0, wr = 0
    s = r.state()
    assert s["wr_ptr"][0] == 0 and s["i"][0, 383] == 348 and s["i"][0, 0] == 348
    r.out_write(packet(1, 48, 401))                           # gap = 0 + 384 - 48 = 336 > 288: wr 0 -> 383
    s = r.state()
    assert s["wr_ptr"][0] == 47
    assert s["i"][0, 383] == 401 and s["i"][0, 0] == 402      # frame 348 at slot 383 was dropped (slip)
    assert s["i"][0, 46] == 448 and s["i"][0, 47] == 448


def test_out_ring_repeat_when_reader_is_faster():
    r = OracleRing(1, N)
    r.out_write(packet(1, 48, 1))                             # wr 240, rd 0
    for _ in range(3):
        r.out_read(96)                                        # rd = 144: gap = 96, not < 96
    r.out_write(packet(1, 48, 101))                           # no shift: frames 240..287
    assert r.state()["wr_ptr"][0] == 288
    r.out_read(96); r.out_read(2)                             # rd = 193: gap = 95 < 96 -> wr 288 -> 289
    r.out_write(packet(1, 48, 201))
    s = r.state()
    assert s["wr_ptr"][0] == 337
    assert s["i"][0, 288] == 148 and s["i"][0, 289] == 201    # slot 288 keeps the repeated frame 148
    r.mute()                                                  # contents only (:188-195)
    s2 = r.state()
    assert not s2["i"].any() and not s2["q"].any() and s2["wr_ptr"][0] == 337 and s2["rd_ptr"][0] == s["rd_ptr"][0]


# ---- independent model: one frame at a time, Python ints reduced mod 2^16 like the C fields ----
class PyRing:
    def __init__(self, n):
        self.n, self.i, self.q = n, [0] * n, [0] * n
        self.en = self.rd = self.wr = 0

    def _push(self, a, b):
        self.i[self.wr], self.q[self.wr] = a, b
        self.wr += 1
        if self.wr == self.n:
            self.wr = 0

    def _gap(self):
        g = self.wr + (self.n if self.rd > self.wr else 0)
        return (g - self.rd) & 0xFFFF

    def _write(self, gap, buf):
        if gap > 3 * self.n // 4:
            if self.wr < 1:
                self.wr += self.n
            self.wr -= 1
        if gap < self.n // 4:
            self.wr += 1
            if self.wr >= self.n:
                self.wr -= self.n
        for k in range(0, len(buf), 2):
            self._push(buf[k], buf[k + 1])
        self._push(buf[-2], buf[-1])
        if self.wr < 1:
            self.wr += self.n
        self.wr -= 1

    def in_write(self, buf):
        self._write(self._gap() if self.en else 0, buf)

    def out_write(self, buf):
        if not self.en:
            self.wr = self.rd + self.n // 2
            if self.wr >= self.n:
                self.wr -= self.n
            self.en = 1
        self._write(self._gap(), buf)

    def _read(self, words):
        out = []
        for _ in range(0, words, 2):
            out += [self.i[self.rd], self.q[self.rd]]
            self.rd += 1
            if self.rd >= self.n:
                self.rd = 0
        return out

    def out_read(self, words):
        return self._read(words)

    def in_read(self, nbytes):
        if not self.en:
            self.rd = self.wr + self.n // 2
            if self.rd >= self.n:
                self.rd = 0
            self.en = 1
        return self._read(nbytes // 2)


def random_ops(rng, nops, n):
    ops = []
    for _ in range(nops):
        kind = rng.choice(["in_write", "in_read", "out_write", "out_read"])
        frames = int(rng.choice([1, 2, 47, 48, 49, 96, n - 1, n, n + 1, 2 * n + 3]))
        ops.append((kind, frames, int(rng.integers(1, 30000))))
    return ops


@pytest.mark.parametrize("n", [384, 8, 33])
def test_oracle_matches_independent_python_model_on_random_traffic(n):
    rng = np.random.default_rng(n)
    C_ = 3
    orc = OracleRing(C_, n)
    py = [PyRing(n) for _ in range(C_)]
    st = orc.state()
    st["rd_ptr"][:] = rng.integers(0, n, C_)
    st["wr_ptr"][:] = rng.integers(0, n, C_)
    orc.set_state(st)
    for c in range(C_):
        py[c].rd, py[c].wr = int(st["rd_ptr"][c]), int(st["wr_ptr"][c])
    for kind, frames, base in random_ops(rng, 120, n):
        if kind.endswith("write"):
            pkt = packet(C_, frames, base)
            getattr(orc, kind)(pkt)
            for c in range(C_):
                getattr(py[c], kind)([int(x) for x in pkt[c]])
        else:
            got = orc.in_read(4 * frames) if kind == "in_read" else orc.out_read(2 * frames)
            for c in range(C_):
                want = py[c].in_read(4 * frames) if kind == "in_read" else py[c].out_read(2 * frames)
                assert list(got[c]) == want
        s = orc.state()
        for c in range(C_):
            assert (s["rd_ptr"][c], s["wr_ptr"][c], s["buff_enable"][c]) == (py[c].rd, py[c].wr, py[c].en)
            assert list(s["i"][c]) == py[c].i and list(s["q"][c]) == py[c].q


def test_oracle_reproduces_committed_fixture():
    fx = np.load(os.path.join(rc.ROOT, "tests", "golden", "ring_trace.npz"))
    from golden.make_ring_golden import replay
    outs, state = replay(OracleRing(int(fx["channels"]), int(fx["frames"])))
    assert np.array_equal(np.concatenate([o.ravel() for o in outs]), fx["reads"])
    for k in ("i", "q", "buff_enable", "rd_ptr", "wr_ptr"):
        assert np.array_equal(state[k], fx[k]), k
