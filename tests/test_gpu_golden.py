"""GPU: the HIP path (through the C-ABI) against the committed golden fixtures that the REAL
CMSIS-DSP composition produced in the build container (tests/golden/make_golden.py).  No oracle
and no /root/reference involved: fixture in, bit-exact fixture out."""
import importlib.util
import os

import numpy as np
import pytest

import rxcommon as rc
from rxcommon import bits_equal

pytestmark = pytest.mark.gpu
G = rc.GOLDEN_DIR


def _mg():
    spec = importlib.util.spec_from_file_location("make_golden", os.path.join(G, "make_golden.py"))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m


def test_library_sin_table_is_the_reference_table():
    """NCO parity hinges on the regenerated sinTable_f32: check it through the device synth, whose
    tones are pure table-lerp sin/cos, against the oracle pinned to the reference table."""
    import selenite_rx as sr
    want = np.fromfile(os.path.join(G, "sintable_f32.bin"), np.float32)
    got = np.ctypeslib.as_array(rc.oracle_lib().orc_sin_table(), (513,))
    assert bits_equal(got, want)
    rx = sr.Rx(rc.baseline_spec("cfg1", 1).config())
    buf = sr.DeviceBuffer(64 * 4096 * 8)
    rx.synth_device(buf.ptr, 0, 64, 0, 4096, rc.SEED)
    rx.sync()
    assert bits_equal(buf.download((64, 4096, 2), np.float32), rc.synth_iq(0, 64, 0, 4096))


@pytest.mark.parametrize("name", ["cfg1", "cfg2", "cfg3", "cfg4", "lsb_full", "cwr_full", "am_full", "fm_full", "fm_nodec", "global_gain"])
@pytest.mark.parametrize("generic", [False, True])
def test_chain_fixture(name, generic):
    import selenite_rx as sr
    mg = _mg()
    kind, arg, channels, nblocks, ncalls = mg.CHAIN_CASES[name]
    spec = mg.chain_spec(kind, arg, channels)
    gold = np.load(os.path.join(G, "chain_%s.npz" % name))
    with sr.plan_option(sr.OPT_FORCE_GENERIC, 1 if generic else 0):
        rx = sr.Rx(spec.config())
    bs = spec.block * nblocks
    for call in range(ncalls):
        iq = rc.synth_iq(0, channels, call * bs, bs)
        assert bits_equal(rx.process(iq), gold["audio"][call]), "%s call %d (%s)" % (name, call, rx.kernel_name())
    st = rx.state()
    for k in st:
        assert np.array_equal(st[k].view(np.uint32), gold[k].view(np.uint32)), k


def test_q15_fixture():
    import selenite_rx as sr
    rx = sr.Rx(rc.baseline_spec("cfg3", 2).config())
    iq = rc.synth_iq(0, 2, 0, 1024)
    q = np.clip(np.trunc(iq * 32768.0), -32768, 32767).astype(np.int16)
    gold = np.load(os.path.join(G, "chain_cfg3_q15.npz"))
    assert np.array_equal(rx.process_q15(q), gold["audio"])


def test_pure_c_host_slot_example_runs():
    """selenite-lite_amd/host/dsp_if_slot.c: C host code in the shape of the firmware's callback slot
    (dsp_if.c:50-67), linked against the C-ABI only."""
    import subprocess
    exe = os.path.join(rc.PKG_DIR, "host", "dsp_if_slot")
    r = subprocess.run([exe], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stdout + r.stderr
    assert r.stdout.count("k_ssb_fused<256,4,63>") == 3
