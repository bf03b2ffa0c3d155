"""CPU, world_size 2 over gloo: the N>1 path of bench.py.

Checks by construction what the 8-GPU run relies on: (1) the channel partition covers every
channel once; (2) a rank that processes only its shard (with the synthetic generator offset by its
first channel) produces exactly the slice of the unsharded result; (3) the global-gain exchange
(per-block envelope all-reduced with MAX) makes sharded == unsharded bit for bit.  The compute in
this CPU test is the oracle standing in for the HIP library -- the protocol is what is under test.
"""
import os
import socket

import numpy as np
import pytest

import rxcommon as rc
from selenite_rx.shard import channel_range, global_gain_call


def test_channel_range_partitions():
    for total in (1, 7, 8, 65536, 1 << 20, 1000003):
        for world in (1, 2, 3, 4, 8):
            got = [channel_range(total, r, world) for r in range(world)]
            assert got[0][0] == 0 and sum(n for _, n in got) == total
            for (f0, n0), (f1, _) in zip(got, got[1:]):
                assert f0 + n0 == f1
            assert max(n for _, n in got) - min(n for _, n in got) <= 1


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, total, global_gain, q):
    import torch
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        first, n = channel_range(total, rank, world)
        kw = dict(block=64, decim=2, nd_taps=21, nh_taps=15, mode=rc.MODE_USB, nco=True,
                  nco_step_all=0x02000000, agc=True, agc_global=global_gain)
        full = rc.CpuChain(rc.ChainSpec(total, **kw), "orc")
        mine = rc.CpuChain(rc.ChainSpec(n, **kw), "orc")
        shadow = rc.CpuChain(rc.ChainSpec(n, **kw), "orc")          # phase-1 stand-in (see module doc)
        ok = True
        for call in range(3):
            bs = 256
            iq_full = rc.synth_iq(0, total, call * bs, bs)
            iq = rc.synth_iq(first, n, call * bs, bs)
            ok &= rc.bits_equal(iq, iq_full[first:first + n])
            want = full.process(iq_full)[first:first + n]
            if global_gain:
                box = {}
                env_t = torch.zeros(bs // 64, dtype=torch.float32)

                def phase1():
                    _, env = shadow.process_env(iq)                 # local envelopes per DSP block
                    env_t.copy_(torch.from_numpy(env))

                def allreduce():
                    dist.all_reduce(env_t, op=dist.ReduceOp.MAX)

                def phase2():
                    box["y"], _ = mine.process_env(iq, env_override=env_t.numpy().copy())

                global_gain_call(phase1, allreduce, phase2)
                # keep the shadow's gains in step with the real instance
                shadow.L.orc_rx_set_state(shadow.h, rc.C.byref(rc.state_view(mine.state())))
                got = box["y"]
            else:
                got = mine.process(iq)
            ok &= rc.bits_equal(got, want)
        # the timing reduction bench.py does
        t = torch.tensor([float(rank + 1)], dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        ok &= float(t.item()) == float(world)
        dist.barrier()
        q.put((rank, bool(ok)))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("global_gain", [False, True])
def test_sharded_equals_unsharded_world2(global_gain):
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, 7, global_gain, q)) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    res = dict(q.get(timeout=5) for _ in range(2))
    assert res == {0: True, 1: True}
