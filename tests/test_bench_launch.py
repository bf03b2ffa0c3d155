"""CPU: bench.py's own launcher for --gpus N > 1 (a plain `python bench.py --gpus 8` printed n_gpus 8
from ONE rank).  --selftest-launch runs the launcher / rendezvous / max-over-ranks plumbing with gloo and without
any GPU call; the full multi-rank bench (two gloo ranks sharing the one GPU of the test box) is in
test_gpu_bench.py."""
import json
import os
import subprocess
import sys

import rxcommon as rc

BENCH = os.path.join(rc.ROOT, "bench.py")


def run(args, env_extra=None, drop=()):
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT") + tuple(drop)}
    env.update(env_extra or {})
    return subprocess.run([sys.executable, BENCH] + args, capture_output=True, text=True, timeout=300, env=env)


def test_bench_starts_its_own_ranks_when_no_launcher_did():
    out = run(["--gpus", "2", "--dist-backend", "gloo", "--selftest-launch"])
    assert out.returncode == 0, out.stderr
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, out.stdout                                   # rank 0 only
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["world"] == 2
    assert d["config"]["parallelism"].startswith("channels sharded x2")
    assert abs(d["max_over_ranks_s"] - 0.002) < 1e-9                     # MAX over the ranks, not rank 0's own value
    # the block an N > 1 line carries so the driver can verify the ranks: gathered over the process group
    ds = d["dist"]
    assert ds == {"backend": "gloo", "world": 2, "comm_count": 2, "devices": ["cpu:0", "cpu:1"], "collectives_per_step": 0,
                  "per_rank_ms_per_step": [1.0, 2.0]}          # comm_count: what an all-reduce of ones over the group returned
    g = json.loads([l for l in run(["--gpus", "2", "--dist-backend", "gloo", "--selftest-launch", "--global-gain"]).stdout.splitlines()
                    if l.startswith("{")][0])
    assert g["dist"]["collectives_per_step"] == 1


def test_bench_refuses_a_world_size_that_differs_from_gpus():
    out = run(["--gpus", "2", "--selftest-launch"], {"WORLD_SIZE": "1", "RANK": "0", "LOCAL_RANK": "0"})
    assert out.returncode == 2 and "refusing" in out.stderr and not out.stdout.strip()


def test_shard_helpers_are_pure_python():
    """The launcher must run before anything touches the GPU: importing selenite_rx.shard loads neither torch nor
    the HIP library."""
    code = ("import sys; sys.path.insert(0, %r); import selenite_rx.shard as s, selenite_rx as sr; "
            "assert 'torch' not in sys.modules and sr._lib is None; print(s.channel_range(10, 1, 3))" % rc.PKG_DIR)
    out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=120)
    assert out.returncode == 0 and out.stdout.strip() == "(4, 3)", out.stderr


def test_a_hung_rank_is_named_and_the_launcher_returns_non_zero_within_its_timeout():
    """the first real N > 1 run must defend itself.  Rank 1 never reaches the collective behind the communicator
    check (test hook); the launcher's limit is 8 s: it kills both ranks (each in its own session), names the stuck ranks with the
    stage each reached, and exits 124 -- no JSON line, nothing left running."""
    import time
    t0 = time.monotonic()
    out = run(["--gpus", "2", "--dist-backend", "gloo", "--selftest-launch"],
              {"SELENITE_SELFTEST_HANG_RANK": "1", "SELENITE_LAUNCH_TIMEOUT_S": "8"})
    dt = time.monotonic() - t0
    assert out.returncode == 124, (out.returncode, out.stderr)
    assert dt < 60, dt
    assert "stuck" in out.stderr and "rank 1 at 'hung on purpose (selftest)'" in out.stderr, out.stderr
    assert "rank 0 at '" in out.stderr                                   # the peer waiting for it in the barrier is named too
    assert not [l for l in out.stdout.splitlines() if l.startswith("{")]


def test_a_rank_gives_up_on_its_own_when_its_peer_never_shows_up():
    """The same defence inside every rank (the driver starts the ranks with torch.distributed.run, not with launch_ranks): a rank whose
    rendezvous never completes says where it sits and exits 3 after SELENITE_RANK_TIMEOUT_S."""
    import socket
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    out = run(["--gpus", "2", "--dist-backend", "gloo", "--selftest-launch"],
              {"WORLD_SIZE": "2", "RANK": "1", "LOCAL_RANK": "1", "MASTER_ADDR": "127.0.0.1", "MASTER_PORT": str(port),
               "SELENITE_RANK_TIMEOUT_S": "5"})
    assert out.returncode == 3, (out.returncode, out.stderr[-400:])
    assert "rank 1 of 2 still at 'init_process_group(gloo)'" in out.stderr


def test_a_failing_rank_stops_the_others():
    out = run(["--gpus", "2", "--dist-backend", "gloo", "--selftest-launch"], {"SELENITE_SELFTEST_FAIL_RANK": "1", "SELENITE_LAUNCH_TIMEOUT_S": "60"})
    assert out.returncode == 7 and "rank 1 exited with 7" in out.stderr and "rank 0 at '" in out.stderr, out.stderr


def test_the_c_host_of_the_rccl_path_has_a_watchdog():
    """selenite-lite_amd/host/global_gain_rccl.c: a watchdog thread names the stage and leaves with _exit(3) -- here before any GPU call."""
    import pytest
    exe = os.path.join(rc.PKG_DIR, "host", "global_gain_rccl")
    if not os.path.exists(exe):
        pytest.skip("host/global_gain_rccl is built only where rccl.h / librccl exist")
    out = subprocess.run([exe], capture_output=True, text=True, timeout=60,
                         env=dict(os.environ, GLOBAL_GAIN_SELFTEST_HANG="1", GLOBAL_GAIN_TIMEOUT_S="2"))
    assert out.returncode == 3 and "still at 'hung on purpose (selftest)' after 2 s" in out.stderr


def test_a_sigterm_to_the_launcher_takes_the_ranks_along():
    """Advisor, round 4: the ranks sit in sessions of their own, so a SIGTERM aimed at the launcher (a driver's timeout) used to leave them
    behind until their own watchdog fired.  Rank 1 hangs on purpose; the launcher gets SIGTERM; within seconds no rank is left."""
    import signal
    import time
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT")}
    env.update(SELENITE_SELFTEST_HANG_RANK="1", SELENITE_LAUNCH_TIMEOUT_S="120", SELENITE_SELFTEST_TAG="sigterm-%d" % os.getpid())
    p = subprocess.Popen([sys.executable, BENCH, "--gpus", "2", "--dist-backend", "gloo", "--selftest-launch"], env=env,
                         stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)

    def ranks():            # the children of the launcher: processes whose environment carries this test's tag and a RANK
        found = []
        for pid in os.listdir("/proc"):
            if not pid.isdigit() or int(pid) == p.pid:
                continue
            try:
                e = open("/proc/%s/environ" % pid, "rb").read()
            except OSError:
                continue
            if env["SELENITE_SELFTEST_TAG"].encode() in e and b"\0RANK=" in b"\0" + e:
                found.append(int(pid))
        return found

    t0 = time.monotonic()
    while len(ranks()) < 2 and time.monotonic() - t0 < 60:
        time.sleep(0.2)
    assert len(ranks()) == 2, "the ranks never started"
    time.sleep(1.0)
    p.send_signal(signal.SIGTERM)
    try:
        p.wait(timeout=30)
    except subprocess.TimeoutExpired:
        p.kill()
        raise AssertionError("the launcher ignored SIGTERM")
    assert p.returncode == 128 + signal.SIGTERM, p.returncode
    t1 = time.monotonic()
    while ranks() and time.monotonic() - t1 < 15:
        time.sleep(0.2)
    assert ranks() == [], "ranks left behind: %s" % ranks()
