"""CPU: the test infrastructure itself is clean under AddressSanitizer + UndefinedBehaviorSanitizer (+ float-cast-overflow):
the oracle restatements, the harnesses around the REAL CMSIS-DSP sources (oracle/ref_chain.c, ref_tx.c) and around the real
Core/Src/dsp_if.c (ref_ring.c) are rebuilt with `make -C oracle SAN=1` into oracle/_san/ and the oracle test files run over
them in a child python with libasan preloaded (tools/run_cpu_tests_sanitized.sh; the whole CPU suite runs the same way by
hand: 111 tests, ~1 min).  CPU only -- the GPU build is never sanitized.  Plus the defined results of the conversions C
leaves undefined (float -> int16 of NaN / out-of-range values: arm_float_to_q15.c:117 on the firmware's FPU saturates and
maps NaN to 0)."""
import os
import shutil
import subprocess

import numpy as np
import pytest

import rxcommon as rc


def test_float_to_q15_of_values_c_leaves_undefined():
    L = rc.oracle_lib()
    x = np.array([np.nan, np.inf, -np.inf, 1e30, -1e30, 70000.0, -70000.0, 0.99999, -1.0, 1.0, 3.0517578125e-05, -3.05e-05, 0.0],
                 np.float32)
    q = np.empty(x.size, np.int16)
    L.orc_float_to_q15(rc.fptr(x), q.ctypes.data_as(rc.i16p), x.size)
    assert q.tolist() == [0, 32767, -32768, 32767, -32768, 32767, -32768, 32767, -32768, 32767, 1, 0, 0]


@pytest.mark.skipif(shutil.which("gcc") is None, reason="no gcc")
def test_oracle_and_reference_harnesses_are_clean_under_asan_and_ubsan():
    if rc.SAN:
        pytest.skip("already running sanitized")
    asan = subprocess.run(["gcc", "-print-file-name=libasan.so"], capture_output=True, text=True).stdout.strip()
    if not os.path.isabs(asan) or not os.path.exists(asan):
        pytest.skip("libasan not installed")
    files = ["tests/test_oracle_golden.py", "tests/test_ring_oracle.py", "tests/test_tx_oracle.py", "tests/test_oracle_vs_ref.py",
             "tests/test_ring_oracle_vs_ref.py", "tests/test_sanitizers.py::test_float_to_q15_of_values_c_leaves_undefined"]
    out = subprocess.run([os.path.join(rc.ROOT, "tools", "run_cpu_tests_sanitized.sh")] + files + ["-q", "-x", "-p", "no:cacheprovider"],
                         capture_output=True, text=True, timeout=900, cwd=rc.ROOT)
    assert out.returncode == 0, (out.stdout[-3000:], out.stderr[-3000:])
    assert "runtime error" not in out.stderr and "AddressSanitizer" not in out.stderr, out.stderr[-3000:]
    assert " passed" in out.stdout and "failed" not in out.stdout
