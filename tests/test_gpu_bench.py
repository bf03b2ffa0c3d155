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
    out = subprocess.run([sys.executable, BENCH, "--steps", "3", "--warmup", "4", "--spinup-ms", "0",
                          "--channels", "1024"] + list(args), check=True, capture_output=True, text=True, timeout=600)
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, out.stdout
    return json.loads(lines[0])


def test_bench_json_contract_default_shape():
    d = run_bench()
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
              "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert k in d, k
    assert d["n_gpus"] == 1 and d["steps"] == 3 and d["warmup"] == 4 and d["scaling"] == "weak"
    assert d["vs_baseline"] is None and d["higher_is_better"] is True and d["dtype"].startswith("f32")
    assert "f16 hi+lo" in d["dtype"]                                   # the default arithmetic says what its multiplicands are
    assert "workload" in d["config"] and "cfg3" in d["config"]["workload"] and "model" not in d["config"]
    r = d["roofline"]
    assert r["bound"] == "hbm" and r["unit"] == "GB/s" and r["peak"] == 8000.0
    assert abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-3
    assert r["algorithmic_bytes_per_launch"] == 1024 * 41952          # SURVEY.md 8d per channel-block figure
    derived = r["algorithmic_bytes_per_launch"] / (d["ms_per_step"] * 1e-3) / 1e9      # round 4: from the wall clock the driver can check
    assert abs(r["achieved"] - derived) <= 0.02 * derived             # (ms_per_step is rounded to 0.1 us in the JSON)
    assert 0 < r["frac_hip_events"] < 1 and r["launch_ms_hip_events"] <= d["ms_per_step"] * 1.05
    sr_ = d["streaming_roof"]                                         # the no-arithmetic kernel of the same run
    assert sr_["kernel"] == "k_stream_roof" and 0 < sr_["ms_per_launch"] and 0 < sr_["frac_of_peak"] < 1
    assert abs(r["frac_of_streaming_roof"] - sr_["ms_per_launch"] / d["ms_per_step"]) < 5e-3 + 1.5e-4 / d["ms_per_step"]      # (both rounded to 0.1 us in the JSON)
    assert d["north_star_target"]["target"] == 0.60 and d["north_star_target"]["read_frac"] == r["read_frac"]
    c = d["cpu_baseline"]
    assert c["kind"] in ("reference", "port") and c["cores"] == os.cpu_count() and c["value"] > 0 and "sample" in c
    assert c["single_core"]["cores"] == 1 and 0 < c["single_core"]["value"] <= c["value"]
    assert d["other_nco_modes"]["per_channel"]["value"] > 0 and d["config"]["nco"].startswith("shared LO")
    # round 3: the default arithmetic is AUTO (plain 1e-5 bar on every block), the line says what the guard saw and how
    # the timed arithmetic compares with the CMSIS chain; side legs: own spin-up, >= 100 launches, median per launch
    assert d["config"]["kernel"].startswith("k_ssb_split16<256,4,63>+exact rerun") and d["config"]["arith"].startswith("auto")
    assert d["guard_blocks"] == d["guard"]["blocks"] == 0 and d["guard"]["rerun_channel_calls"] == 0      # steady state of the bench signal
    assert d["parity"]["worst_rel"] == d["parity_worst_rel"] <= 1e-5 and d["parity"]["blocks"] > 0 and d["parity"]["against"] in ("reference", "port")
    assert d["parity"]["bar"] == 1e-5 and d["parity"]["within_bar"] is True
    assert r["launch_ms_median"] > 0 and r["launch_ms_min"] <= r["launch_ms_median"] <= r["launch_ms_p90"]
    for grp, names in (("other_arith_modes", ("split16", "fma", "cmsis")), ("other_nco_modes", ("per_channel", "per_channel_grid", "shared_table"))):
        for n in names:
            e = d[grp][n]
            assert e["launches"] >= 100 and e["ms_min"] <= e["ms_per_step"] <= e["ms_p90"] and e["value"] > 0, (grp, n)
    assert d["other_nco_modes"]["per_channel_grid"]["nco"].startswith("per-channel LO, period 256")
    assert d["other_nco_modes"]["per_channel"]["nco"].startswith("per-channel arm_sin/cos_f32")
    assert d["other_nco_modes"]["shared_table"]["nco"] == "shared LO table per call"
    assert 0 < d["auto_stopband_cost"]["rerun_fraction"] <= 1 and d["auto_stopband_cost"]["value"] > 0
    assert d["fma_roof"]["flops_per_sample"] == 293.5                  # shared LO: 6 NCO flops per sample, not 20
    assert d["value"] > 0 and d["ms_per_step"] > 0
    # round 5: every other single-GPU BASELINE configuration in the same line -- median of >= 100 launches, roofline from
    # the SURVEY 8d bytes, parity on a 64-channel sample against the compiled reference
    w = d["workloads"]
    per_ch = {"cfg2": 8 * 48000 + 4 * 48000 + 2 * 4 * 2 * 126 + 8, "cfg4": 8 * 4096 + 4 * 4096 + 2 * (16 * 4) + 2 * 8,
              "cfg5": 8 * 1024 + 4 * 1024 + 2 * 4 * 2 * 126 + 8, "cfg3_q15": 41952 - 4 * 4096 - 2 * 1024}
    for n in ("cfg2", "cfg4", "cfg5", "cfg3_q15"):
        e = w[n]
        assert e["launches"] >= 100 and e["ms_min"] <= e["ms_per_step"] <= e["ms_p90"] and e["value"] > 0, n
        r2 = e["roofline"]
        assert r2["bound"] == "hbm" and 0 < r2["frac"] < 1 and abs(r2["frac"] - r2["achieved"] / 8000.0) < 1e-3, n
        assert r2["algorithmic_bytes_per_launch"] == 1024 * per_ch[n], (n, r2["algorithmic_bytes_per_launch"] / 1024)
        # the same bytes with no arithmetic, timed right behind the launches (the memory-bound shapes move with the box and with what ran before)
        # (both times are rounded to 0.1 us in the JSON: at this test's scaled-down sizes -- 10 us launches -- that alone is a percent of the ratio)
        assert 0 < r2["streaming_roof_ms"] and abs(r2["frac_of_streaming_roof"] - r2["streaming_roof_ms"] / e["ms_per_step"]) < 5e-3 + 1.5e-4 / e["ms_per_step"], n
        assert r2["frac_of_streaming_roof"] < 1.5, (n, r2)               # (a copy slower than the kernel by a few percent happens at 10 us per launch; not by half)
        p2 = e["parity"]
        assert p2["channels"] == 64 and p2["blocks"] > 0 and p2["against"] in ("reference", "port"), n
        # round 6: every parity figure carries its bar and the verdict against it -- int16 slots: per DSP block
        # |gpu - ref| <= 1 LSB + 1e-5 x block maximum (LSB), blocks at full scale counted apart (none on the bench signal)
        assert p2["bar"] is not None and p2["within_bar"] is True, (n, p2)
        if n == "cfg3_q15":
            assert p2["worst_lsb"] <= 1 and p2["worst_margin_lsb"] <= 0 and p2["blocks_at_full_scale"] == 0, p2
        else:
            assert p2["bar"] == 1e-5 and p2["worst_rel"] <= 1e-5, (n, p2)
    assert w["cfg4"]["kernel"] == "k_cw_fused<4,256>" and w["cfg4"]["parity"]["worst_rel"] == 0.0      # bit-exact in every arithmetic mode
    # round 6: the CW kernel against the roof of ITS fetch pattern (k_cw_roof: same bursts, stores, launch shape, residency; no DSP)
    pr = w["cfg4"]["roofline"]["pattern_roof"]
    assert 0 < pr["pattern_roof_ms"] <= pr["pattern_roof_with_arith_ms"] * 1.5 and pr["arith_vector_instructions_per_chunk"] > 1000
    assert pr["kernel"] == "k_cw_roof" and 0 < pr["pattern_roof_ms"] and abs(pr["frac_of_pattern_roof"] - pr["pattern_roof_ms"] / w["cfg4"]["ms_per_step"]) < 5e-3 + 1.5e-4 / w["cfg4"]["ms_per_step"]
    # round 6: cfg2 in both placements of its buffers, same run
    pl = w["cfg2"]["placement"]
    for k in ("separate_allocations", "one_allocation"):
        assert pl[k]["ms_per_step"] > 0 and 0 < pl[k]["frac"] < 1 and pl[k]["streaming_roof_ms"] > 0, k
    assert pl["separate_allocations"]["ms_per_step"] == w["cfg2"]["ms_per_step"]
    assert w["cfg2"]["kernel"].startswith("k_hilb_split16<127>") and w["cfg5"]["kernel"].startswith("k_hilb_split16<127>")
    assert "mall_note" in w["cfg5"] and w["cfg3_q15"]["io"] == "q15"


def test_bench_q15_slots_cw_shape_and_global_gain():
    q = run_bench("--io", "q15", "--no-cpu-baseline")
    assert q["io"] == "q15" and q["roofline"]["traffic"] is None
    assert q["roofline"]["algorithmic_bytes_per_launch"] == 1024 * (41952 - 4 * 4096 - 2 * 1024)
    cw = run_bench("--workload", "cfg4", "--arith", "cmsis", "--no-cpu-baseline")
    assert cw["config"]["kernel"] == "k_cw_fused<4,256>"
    g = run_bench("--global-gain", "--no-cpu-baseline", "--main-only")
    assert g["config"]["agc"] == "global" and g["config"]["kernel"].startswith("k_ssb_split16<256,4,63>")


def test_pure_c_host_benchmark_agrees_with_the_python_driven_one():
    """selenite-lite_amd/host/bench_rx.c: the north star's "host code stays C" call pattern, measured."""
    exe = os.path.join(rc.PKG_DIR, "host", "bench_rx")
    out = subprocess.run([exe, "4096", "4096", "20"], check=True, capture_output=True, text=True, timeout=300)
    d = json.loads(out.stdout.strip().splitlines()[-1])
    assert d["host"] == "C" and d["kernel"] == "k_ssb_split16<256,4,63>" and d["channels"] == 4096
    derived = 4096 * 41952 / d["ms_per_call"] / 1e6
    assert d["msamples_per_s"] > 0 and abs(d["algorithmic_GBps"] - derived) <= 0.02 * derived


def test_bench_launches_its_own_ranks_and_shards_the_channels():
    """`python bench.py --gpus 2` with no launcher: bench.py starts the two ranks itself (here over gloo, both on the
    one GPU of the test box) and reports the aggregate of both."""
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT")}
    env["SELENITE_BENCH_SHARE_GPU"] = "1"
    out = subprocess.run([sys.executable, BENCH, "--gpus", "2", "--dist-backend", "gloo", "--steps", "3", "--warmup", "1",
                          "--spinup-ms", "0", "--channels", "1024", "--main-only"], capture_output=True, text=True,
                         timeout=900, env=env)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, out.stdout
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["config"]["parallelism"].startswith("channels sharded x2")
    assert abs(d["per_gpu_msamples_s"] * 2 - d["value"]) <= 0.01 * d["value"]
    # what lets the driver verify the ranks: backend, world, one PCI bus id per rank, collectives per step, per-rank times
    ds = d["dist"]
    assert ds["backend"] == "gloo" and ds["world"] == 2 and len(ds["devices"]) == 2 and ds["collectives_per_step"] == 0
    assert all(":" in x for x in ds["devices"]) and len(ds["per_rank_ms_per_step"]) == 2
    assert max(ds["per_rank_ms_per_step"]) <= d["ms_per_step"] * 1.001 + 1e-3
    assert ds["comm_count"] == 2
    w5 = d["cfg5_weak_scaling"]                                        # the BASELINE cfg5 leg of every N > 1 line (scaled by --channels here)
    assert w5["n_gpus"] == 2 and abs(w5["per_gpu_msamples_s"] * 2 - w5["value"]) <= 0.01 * w5["value"]
    g = subprocess.run([sys.executable, BENCH, "--gpus", "2", "--dist-backend", "gloo", "--steps", "3", "--warmup", "1",
                        "--spinup-ms", "0", "--channels", "1024", "--main-only", "--global-gain"], capture_output=True,
                       text=True, timeout=900, env=env)
    assert g.returncode == 0, g.stderr[-2000:]
    gd = json.loads([l for l in g.stdout.splitlines() if l.startswith("{")][0])
    assert gd["config"]["agc"] == "global" and gd["dist"]["collectives_per_step"] == 1


def test_bench_under_the_nccl_backend_with_one_rank_carries_the_dist_block():
    """The RCCL process group of an N > 1 run, exercised with the one rank a one-GPU box allows (SELENITE_BENCH_FORCE_DIST):
    init over nccl, barriers, the MAX all-reduce of the times, the object gathers of the `dist` block, the global-gain
    all-reduce on the library's stream."""
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT")}
    env.update(SELENITE_BENCH_FORCE_DIST="1", MASTER_ADDR="127.0.0.1", MASTER_PORT="29611", HSA_ENABLE_IPC_MODE_LEGACY="0")
    for extra, coll in ((["--n1-ms", "0.5", "--n1-cfg5-ms", "0.25"], 0), (["--global-gain"], 1)):
        out = subprocess.run([sys.executable, BENCH, "--steps", "3", "--warmup", "4", "--spinup-ms", "0", "--channels", "1024",
                              "--main-only"] + extra, capture_output=True, text=True, timeout=900, env=env)
        assert out.returncode == 0, out.stderr[-2000:]
        d = json.loads([l for l in out.stdout.splitlines() if l.startswith("{")][0])
        ds = d["dist"]
        assert ds["backend"] == "nccl" and ds["world"] == 1 and ds["collectives_per_step"] == coll
        assert len(ds["devices"]) == 1 and ":" in ds["devices"][0] and len(ds["per_rank_ms_per_step"]) == 1
        # round 6: what an N > 1 line carries, asserted on the N > 1 code path a one-GPU box can run -- the communicator's size as
        # RCCL counted it equals the world, one distinct PCI bus id per rank, and (per-channel AGC) the cfg5 weak-scaling leg with its efficiency field
        assert ds["comm_count"] == ds["world"] == d["n_gpus"] and len(set(ds["devices"])) == ds["world"]
        if not coll:
            w5 = d["cfg5_weak_scaling"]
            assert w5["n_gpus"] == 1 and w5["scaling"] == "weak" and w5["kernel"].startswith("k_hilb_split16<127>") and w5["value"] > 0
            assert "2048 channels/GPU x 1024" in w5["workload"]                          # (--channels 1024 scales the leg: twice the headline's count, as in BASELINE)
            # (ms_per_step is rounded to 0.1 us in the JSON, the efficiency is formed from the unrounded time: at this test's 10-us steps that is up to half a percent)
            for e, n1 in ((w5, 0.25), (d, 0.5)):
                want = n1 / e["ms_per_step"]
                assert abs(e["weak_scaling_efficiency"] - want) <= want * (1e-3 + 6e-5 / e["ms_per_step"]) + 1e-4, (e["weak_scaling_efficiency"], want)
        else:
            assert "cfg5_weak_scaling" not in d


def test_global_gain_ranks_on_one_stream_match_the_unsharded_oracle():
    """Two gloo ranks on the one GPU run bench.py's GlobalGainStepper (phase 1 -> all-reduce -> phase 2 without host
    synchronisation) and compare with the unsharded oracle: tests/dist_global_gain_worker.py."""
    from selenite_rx import shard
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT")}
    rcode = shard.launch_ranks(2, os.path.join(rc.ROOT, "tests", "dist_global_gain_worker.py"), [], env)
    assert rcode == 0


def test_plain_c_host_runs_the_global_gain_path_over_rccl():
    """selenite-lite_amd/host/global_gain_rccl.c: one C process, every GPU of the box a rank, channels sharded,
    phase 1 -> ncclAllReduce(ncclMax) -> phase 2 without host synchronisation; sharded == unsharded bit for bit.
    (One GPU on the test box: the communicator has one rank; the driver's 8-GPU node runs the same binary.)"""
    exe = os.path.join(rc.PKG_DIR, "host", "global_gain_rccl")
    out = subprocess.run([exe, "0", "192", "2048", "3"], capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    d = json.loads(out.stdout.strip().splitlines()[-1])
    assert d["host"] == "C" and d["ranks"] >= 1 and d["sharded_equals_unsharded"] is True
    assert len(d["devices"]) == d["ranks"] and all(":" in x["pci_bus_id"] and x["nccl_comm_count"] == d["ranks"] and
                                                      x["nccl_user_rank"] == x["rank"] for x in d["devices"])
    assert sum(x["channels"] for x in d["devices"]) == d["channels"] and d["collectives_per_call"] == 1
