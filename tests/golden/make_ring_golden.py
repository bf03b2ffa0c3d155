"""Generates tests/golden/ring_trace.npz FROM THE REFERENCE ITSELF: a fixed traffic pattern through the firmware's
own DSP_In_Buff_Write / DSP_In_Buff_Read / DSP_Out_Buff_Write / DSP_Out_Buff_Read / DSP_Out_Buff_Mute
(Core/Src/dsp_if.c:116-340, compiled from /root/reference by oracle/Makefile into oracle/_ref/libdsp_if_ref.so and
driven through oracle/ref_ring.c), with the reads it returns and the final state.  `replay` is shared with the tests so
the GPU library, the restatement oracle/ring_oracle.c and the reference are driven through exactly the same calls.
Build container only (needs /root/reference).   Run:  python tests/golden/make_ring_golden.py"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))

CHANNELS, FRAMES = 5, 768          # DSP_BUFF_SIZE of the firmware build (USBD_AUDIO_FREQ 96000: dsp_if.h:81-84)


def replay(ring):
    """USB side 48 kHz nominal, codec side slightly fast then slightly slow: both drift branches."""
    rng = np.random.default_rng(0x5E1E)
    C_ = ring.channels
    st = ring.state()
    st["rd_ptr"][:] = (np.arange(C_) * 37) % ring.frames
    st["wr_ptr"][:] = (np.arange(C_) * 91 + 5) % ring.frames
    ring.set_state(st)
    reads = []
    for step in range(60):
        pkt = rng.integers(-32768, 32768, (C_, 96), dtype=np.int64).astype(np.int16)
        ring.in_write(pkt)                                   # DSP_In_Buff_Write(rx, 96 words)
        nread = 47 if step < 30 else 49                      # frames per USB packet: reader slow, then fast
        reads.append(ring.in_read(4 * nread))
        if step % 7 == 3:
            ring.out_write(pkt[:, ::-1].copy())              # exercise the OUT flavour on the same rings
            reads.append(ring.out_read(2 * 50))
        if step == 41:
            ring.mute()                                      # DSP_Out_Buff_Mute (dsp_if.c:185-195)
    return reads, ring.state()


if __name__ == "__main__":
    from rxcommon import RefRing
    outs, state = replay(RefRing(CHANNELS, FRAMES))
    np.savez_compressed(os.path.join(HERE, "ring_trace.npz"), channels=CHANNELS, frames=FRAMES,
                        reads=np.concatenate([o.ravel() for o in outs]), **state)
    print("wrote ring_trace.npz:", sum(o.size for o in outs), "read words")
