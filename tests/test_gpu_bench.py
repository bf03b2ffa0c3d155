"""GPU: bench.py keeps its contract -- one JSON line with the fields the driver reads, the roofline
object measured live, the cpu_baseline object, for the f32 and the int16-slot formats.  Small shapes
(the numbers are not asserted, only their presence and consistency)."""
import json
import os
import subprocess
import sys

import pytest

import rxcommon as rc

pytestmark = pytest.mark.gpu
BENCH = os.path.join(rc.ROOT, "bench.py")


def run_bench(*args):
    out = subprocess.run([sys.executable, BENCH, "--steps", "3", "--warmup", "1", "--spinup-ms", "0",
                          "--channels", "1024"] + list(args), check=True, capture_output=True, text=True, timeout=600)
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, out.stdout
    return json.loads(lines[0])


def test_bench_json_contract_default_shape():
    d = run_bench()
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
              "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert k in d, k
    assert d["n_gpus"] == 1 and d["steps"] == 3 and d["warmup"] == 1 and d["scaling"] == "weak"
    assert d["vs_baseline"] is None and d["higher_is_better"] is True and d["dtype"] == "f32"
    assert "workload" in d["config"] and "cfg3" in d["config"]["workload"] and "model" not in d["config"]
    r = d["roofline"]
    assert r["bound"] == "hbm" and r["unit"] == "GB/s" and r["peak"] == 8000.0
    assert abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-3
    assert r["algorithmic_bytes_per_launch"] == 1024 * 41952          # SURVEY.md 8d per channel-block figure
    derived = r["algorithmic_bytes_per_launch"] / (r["launch_ms_hip_events"] * 1e-3) / 1e9
    assert abs(r["achieved"] - derived) <= 0.02 * derived             # (launch_ms is rounded to 0.1 us in the JSON)
    c = d["cpu_baseline"]
    assert c["kind"] == "port" and c["cores"] >= 1 and c["value"] > 0 and "sample" in c
    assert d["value"] > 0 and d["ms_per_step"] > 0


def test_bench_q15_slots_cw_shape_and_global_gain():
    q = run_bench("--io", "q15", "--no-cpu-baseline")
    assert q["io"] == "q15" and q["roofline"]["traffic"] is None
    assert q["roofline"]["algorithmic_bytes_per_launch"] == 1024 * (41952 - 4 * 4096 - 2 * 1024)
    cw = run_bench("--workload", "cfg4", "--arith", "cmsis", "--no-cpu-baseline")
    assert cw["config"]["kernel"] == "k_cw_fused<4,256>"
    g = run_bench("--global-gain", "--no-cpu-baseline", "--main-only")
    assert g["config"]["agc"] == "global" and g["config"]["kernel"] == "k_ssb_split16<256,4,63>"


def test_pure_c_host_benchmark_agrees_with_the_python_driven_one():
    """selenite-lite_amd/host/bench_rx.c: the north star's "host code stays C" call pattern, measured."""
    exe = os.path.join(rc.PKG_DIR, "host", "bench_rx")
    out = subprocess.run([exe, "4096", "4096", "20"], check=True, capture_output=True, text=True, timeout=300)
    d = json.loads(out.stdout.strip().splitlines()[-1])
    assert d["host"] == "C" and d["kernel"] == "k_ssb_split16<256,4,63>" and d["channels"] == 4096
    derived = 4096 * 41952 / d["ms_per_call"] / 1e6
    assert d["msamples_per_s"] > 0 and abs(d["algorithmic_GBps"] - derived) <= 0.02 * derived
