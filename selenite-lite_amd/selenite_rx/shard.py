"""Channel sharding across ranks (one process per GPU).

Channels are independent units: rank r of W owns a contiguous channel range and
runs the whole chain on it with NO data-path collective.  The only exchange the chain can need is
the global-gain AGC: per DSP block one float (max |audio| over the rank's channels) all-reduced
with MAX -- exact, so sharded == unsharded bit for bit.  `allreduce_max` is any callable that
all-reduces a float32 buffer in place (torch.distributed over RCCL on GPUs, gloo in the CPU tests).
"""


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
