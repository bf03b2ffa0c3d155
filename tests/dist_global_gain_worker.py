"""Worker of test_gpu_bench.py::test_global_gain_ranks_...: one rank of a gloo world whose ranks share the test box's
single GPU.  Runs the SAME GlobalGainStepper bench.py times (phase 1 -> all-reduce MAX -> phase 2 on one explicit
stream, no host synchronisation in between) on its channel shard and checks the audio against the UNSHARDED oracle
(on torch's null stream the library fell back to its own stream and the three steps were unordered)."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, os.path.join(ROOT, "selenite-lite_amd"))


def main():
    from selenite_rx import shard
    env = shard.RankEnv()
    env.local_rank = 0
    env.init_process_group("gloo")
    import rxcommon as rc
    import selenite_rx as sr
    total, bs, ncalls = 96, 1024, 6
    first, n = shard.channel_range(total, env.rank, env.world)
    sr.lib().selenite_rx_set_device(0)
    for arith, exact in ((rc.ARITH_CMSIS, True), (rc.ARITH_SPLIT16, False)):
        rx = sr.Rx(rc.baseline_spec("cfg3", n, arith, agc_global=True).config())
        ref_arith = rc.ARITH_CMSIS
        full = rc.CpuChain(rc.baseline_spec("cfg3", total, ref_arith, agc_global=True), "orc")
        nout = bs // 4
        d_in, d_out = sr.DeviceBuffer(n * bs * 8), sr.DeviceBuffer(n * nout * 4)
        gg = shard.GlobalGainStepper(rx, env, bs // 256, 0)
        for call in range(ncalls):
            iq_full = rc.synth_iq(0, total, call * bs, bs)
            # different levels per rank so the global envelope is NOT this rank's own
            iq_full[: total // 2] *= np.float32(0.05)
            want = full.process(iq_full)[first:first + n]
            d_in.upload(np.ascontiguousarray(iq_full[first:first + n]))
            gg.step(d_in.ptr, d_out.ptr, bs)             # no host sync between phase 1, the all-reduce and phase 2
            gg.synchronize()
            got = d_out.download((n, nout), np.float32)
            if exact:
                assert rc.bits_equal(got, want), "rank %d call %d: sharded != unsharded (rel %g)" % (env.rank, call, rc.rel_err(got, want))
            else:
                for b in range(nout // 64):
                    assert rc.rel_err(got[:, 64 * b:64 * b + 64], want[:, 64 * b:64 * b + 64]) <= 1e-5
        rx.check()
        rx.close()
        d_in.free()
        d_out.free()
    env.close()
    if env.rank == 0:
        print("OK")


if __name__ == "__main__":
    main()
