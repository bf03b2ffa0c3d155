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
    assert ds == {"backend": "gloo", "world": 2, "devices": ["cpu:0", "cpu:1"], "collectives_per_step": 0,
                  "per_rank_ms_per_step": [1.0, 2.0]}
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
