"""Channel sharding across ranks: one process per GPU, torch.distributed over RCCL / xGMI.

Channels are independent units: rank r of W owns a contiguous channel range and runs the
whole chain on it with NO data-path collective.  The only exchange the chain can need is the global-gain
AGC: per DSP block one float (max |audio| over the rank's channels) all-reduced with MAX -- exact, so
sharded == unsharded bit for bit.

This module holds the multi-GPU host logic that bench.py and the tests share:

  channel_range       the partition
  launch_ranks        start one rank per GPU with torch.distributed.run when no launcher did
  RankEnv             rank / local rank / world size from the launcher's environment, process-group set-up
  GlobalGainStepper   one global-gain process call: phase 1 -> all-reduce(MAX) -> phase 2, ordered on ONE stream
"""
import os
import socket
import subprocess
import sys


def channel_range(total_channels, rank, world):
    """Contiguous, balanced partition: returns (first_channel, count)."""
    base, extra = divmod(total_channels, world)
    first = rank * base + min(rank, extra)
    return first, base + (1 if rank < extra else 0)


def global_gain_call(phase1, allreduce_max, phase2):
    """One process call of a global-gain instance: phase1() leaves the per-block envelopes of this
    rank's channels in the exchange buffer, allreduce_max() makes them global, phase2() applies the
    gain law and scales."""
    phase1()
    allreduce_max()
    phase2()


def free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def launch_ranks(nproc, script, argv, env=None):
    """Run `script argv` as `nproc` ranks of one node under torch.distributed.run (rendezvous on 127.0.0.1) and
    return the launcher's exit code.  Called by a process that has NOT touched the GPU: the ranks are fresh child
    processes (never an exec of a process that initialised HIP)."""
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(nproc),
           "--master-addr", "127.0.0.1", "--master-port", str(free_port()), script] + list(argv)
    e = dict(os.environ if env is None else env)
    e.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")      # dmabuf IPC only on this driver (RCCL, tensor sharing)
    e.setdefault("OMP_NUM_THREADS", "1")
    return subprocess.run(cmd, env=e).returncode


class RankEnv:
    """What the launcher (torch.distributed.run, or nothing at N = 1) told this process."""

    def __init__(self, environ=None):
        environ = os.environ if environ is None else environ
        self.rank = int(environ.get("RANK", "0"))
        self.local_rank = int(environ.get("LOCAL_RANK", "0"))
        self.world = int(environ.get("WORLD_SIZE", "1"))
        self.launched = "WORLD_SIZE" in environ
        self.dist = None
        self.torch = None
        self.backend = None

    def init_process_group(self, backend, use_gpu=True):
        """torch FIRST: its bundled libamdhip64 (same soname) then serves libselenite_rx.so too, so the process
        holds exactly one HIP runtime."""
        import torch
        import torch.distributed as dist
        self.torch, self.dist, self.backend = torch, dist, backend
        if use_gpu:
            torch.cuda.set_device(self.local_rank)
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        dist.init_process_group(backend=backend, rank=self.rank, world_size=self.world)

    def barrier(self):
        if self.dist is not None:
            if self.backend == "nccl":
                self.torch.cuda.synchronize()
            self.dist.barrier()

    def max_over_ranks(self, seconds):
        if self.dist is None:
            return seconds
        dev = ("cuda:%d" % self.local_rank) if self.backend == "nccl" else "cpu"
        t = self.torch.tensor([seconds], dtype=self.torch.float64, device=dev)
        self.dist.all_reduce(t, op=self.dist.ReduceOp.MAX)
        return float(t.item())

    def gather_objects(self, obj):
        """[obj of rank 0, obj of rank 1, ...] on every rank (one small pickled message per rank; outside any timed region)."""
        if self.dist is None:
            return [obj]
        out = [None] * self.world
        try:
            self.dist.all_gather_object(out, obj)
        except Exception as e:                      # diagnostics must never cost the measurement
            sys.stderr.write("RankEnv.gather_objects: %s\n" % e)
            return [obj if r == self.rank else None for r in range(self.world)]
        return out

    def close(self):
        if self.dist is not None:
            self.dist.barrier()
            self.dist.destroy_process_group()
            self.dist = None


class GlobalGainStepper:
    """One global-gain call of an `Rx` instance whose channels are one rank's shard.

    Phase 1 (fused chain with its own AGC off + envelope fold), the MAX all-reduce of the per-block envelopes and
    phase 2 (gain law + scale) all run on ONE explicit torch side stream: the library is put on that stream
    (a non-zero handle -- a NULL handle would select the library's own stream, which neither torch's null stream nor
    RCCL's stream ever waits on) and the collective is enqueued inside `torch.cuda.stream(...)`, so ProcessGroupNCCL
    orders its internal stream against it with events.  No host synchronisation inside a step.
    """

    def __init__(self, rx, env, blocks_per_call, device_index):
        torch = env.torch
        if torch is None:
            import torch
        self.torch, self.rx, self.env = torch, rx, env
        self.stream = torch.cuda.Stream(device=device_index)
        assert self.stream.cuda_stream != 0
        with torch.cuda.stream(self.stream):
            self.env_t = torch.zeros(blocks_per_call, dtype=torch.float32, device="cuda:%d" % device_index)
        self.stream.synchronize()
        rx.set_stream(self.stream.cuda_stream)

    def step(self, d_in, d_out, block_size):
        torch, dist = self.torch, self.env.dist
        with torch.cuda.stream(self.stream):
            self.rx.global_phase1(d_in, d_out, self.env_t.data_ptr(), block_size)
            if dist is not None:
                dist.all_reduce(self.env_t, op=dist.ReduceOp.MAX)          # RCCL over xGMI: 4 B per DSP block
            self.rx.global_phase2(d_out, self.env_t.data_ptr(), block_size)

    def synchronize(self):
        self.stream.synchronize()
