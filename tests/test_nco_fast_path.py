"""The split16 kernels compute the per-channel LO with a restated sin/cos sequence (csrc/rx_device.h: nco_lo_pair):
v_fract_f32 in place of `in - (float)(int)in` and of `findex - (float)index`, no negative-argument branches, packed
f32 arithmetic.  CPU part: a numpy float32 model of exactly that sequence against the oracle (arm_sin_f32.c:72-119,
arm_cos_f32.c:70-111) for ALL 2^24 values of phase >> 8.  GPU part: the kernels' mixed samples (decimator state, f32
exact in every arithmetic mode) against the oracle on phases that sweep the whole table."""
import numpy as np
import pytest

import rxcommon as rc
from rxcommon import CpuChain, baseline_spec, bits_equal

f32 = np.float32
K = f32(float.fromhex("0x1.921fb6p-22"))
INV2PI = f32(0.159154943092)


def model_lo(u, table):
    """the operation sequence of nco_lo_pair, one rounding per line as on the GPU"""
    x = u.astype(f32) * K
    ps = x * INV2PI
    pc = ps + f32(0.25)
    out = []
    for p in (ps, pc):
        i = p - np.floor(p)                     # v_fract_f32
        f = i * f32(512.0)
        idx = f.astype(np.uint32)               # v_cvt_u32_f32 (truncation, f in [0, 512))
        r = f - np.floor(f)                     # v_fract_f32
        a, b = table[idx], table[idx + 1]
        w = f32(1.0) - r
        out.append(w * a + r * b)
    s, c = out
    return c, -s


def test_restated_sin_cos_sequence_equals_the_oracle_for_every_phase():
    L = rc.oracle_lib()
    table = np.ctypeslib.as_array(L.orc_sin_table(), (513,)).astype(f32)
    L.orc_nco_lo.argtypes = [rc.C.c_void_p, rc.C.c_uint32, rc.C.c_void_p]
    L.orc_nco_lo.restype = None
    CH = 1 << 20
    for base in range(0, 1 << 24, CH):
        u = np.arange(base, base + CH, dtype=np.uint32)
        ph = (u << np.uint32(8)) | np.uint32(0xA5)              # the low 8 bits never matter
        ref = np.empty((CH, 2), f32)
        L.orc_nco_lo(ph.ctypes.data, CH, ref.ctypes.data)
        c, ms = model_lo(u, table)
        assert c.dtype == f32 and ms.dtype == f32
        assert bits_equal(c, ref[:, 0]) and bits_equal(ms, ref[:, 1]), "phase block %#x" % base


@pytest.mark.gpu
@pytest.mark.parametrize("name", ["cfg3", "cfg2"])
def test_per_channel_nco_in_the_split16_kernels_is_bit_exact_over_the_whole_table(name):
    """every channel its own step and start phase, chosen so that the phases of a call sweep the sine table densely
    (steps of a few table cells per sample, every lerp fraction); mixed samples = decimator / Hilbert input"""
    nch = 128
    rng = np.random.default_rng(7)
    steps = (rng.integers(0, 1 << 32, nch, dtype=np.uint64)).astype(np.uint32)
    steps[:8] = [0, 0x100, 0xFFFFFFFF, 0x80000000, 0x7FFFFF00, 0x00800000, 0xFFFFFF00, 1]
    spec_s = baseline_spec(name, nch, rc.ARITH_SPLIT16, nco_steps=steps, **({"nco": True} if name != "cfg3" else {}))
    spec_c = baseline_spec(name, nch, rc.ARITH_CMSIS, nco_steps=steps, **({"nco": True} if name != "cfg3" else {}))
    import selenite_rx as sr
    g, o = sr.Rx(spec_s.config()), CpuChain(spec_c, "orc")
    assert "split16" in g.kernel_name() and g.nco_path().startswith("per-channel")
    st = g.state()
    st["nco_phase"] = rng.integers(0, 1 << 32, nch, dtype=np.uint64).astype(np.uint32)
    g.set_state(st)
    o.L.orc_rx_set_state(o.h, rc.C.byref(rc.state_view(st)))
    bs = 4096 if name == "cfg3" else spec_s.block * 4
    key = "dec_state" if name == "cfg3" else "fir_state"
    for call in range(3):
        iq = rc.synth_iq(0, nch, call * bs, bs)
        yg, yo = g.process(iq), o.process(iq)
        sg, so = g.state(), o.state()
        assert np.array_equal(sg["nco_phase"], so["nco_phase"])
        assert bits_equal(sg[key][:, 0] if name != "cfg3" else sg[key], so[key][:, 0] if name != "cfg3" else so[key])
        # cfg3: the last 255 mixed samples of every channel, both rails; cfg2: the I rail's delay-line state
        # audio: 1e-5 of the DSP block's maximum plus the f32 noise floor of the input level -- random steps leave most
        # channels with an empty pass band, where two f32-class results differ by rounding noise (tests/test_gpu_truth.py)
        na = spec_c.block // spec_c.decim
        im = np.abs(iq).reshape(nch, -1).max(axis=1)
        d = np.abs(yg.astype(np.float64) - yo).reshape(nch, -1, na).max(axis=2)
        m = np.abs(yo).reshape(nch, -1, na).max(axis=2)
        assert (d <= 1e-5 * m + 4e-7 * im[:, None] * np.maximum(1.0, g.state()["agc_gain"][:, None])).all()
