"""Chain description on the Python side of the C-ABI: coefficient design, the `selenite_rx_config` a
BASELINE.json configuration maps to, and the bench workloads.

Host logic of the product's Python face (bench.py, tools/, tests).  No oracle, no CPU compute path:
the arrays built here only parameterise the HIP library.  The chain itself is [build-defined]
(DESIGN.md section 2): the reference (/root/reference, Core/Src/dsp_if.c:367-370) has an empty mode hook.
"""
import ctypes as C

import numpy as np

from . import (ARITH_AUTO, ARITH_CMSIS, ARITH_FMA, ARITH_SPLIT16, MODE_CW, MODE_USB, Config, f32p, u32p)  # noqa: F401

SEED = 0x5E1E917E            # synthetic I/Q generator seed


def fptr(a):
    return a.ctypes.data_as(f32p) if a is not None else None


def as_f32(a):
    return np.ascontiguousarray(a, dtype=np.float32)


# ---------------------------------------------------------------------------------------------
# coefficient design (numpy, float64 -> float32).  The library ships the same designs in C
# (selenite_rx_design_*); tests/test_design.py checks they agree.  Parity never depends on this: the
# chain only sees the arrays.
# ---------------------------------------------------------------------------------------------
def design_lowpass(num_taps, cutoff):
    n = np.arange(num_taps, dtype=np.float64)
    m = n - (num_taps - 1) / 2.0
    h = 2.0 * cutoff * np.sinc(2.0 * cutoff * m)
    w = 0.54 - 0.46 * np.cos(2.0 * np.pi * n / max(num_taps - 1, 1)) if num_taps > 1 else np.ones(1)
    h = h * w
    h = h / h.sum()
    return as_f32(h[::-1])          # CMSIS order {b[N-1]..b[0]}


def design_hilbert(num_taps):
    assert num_taps % 2 == 1
    c = (num_taps - 1) // 2
    n = np.arange(num_taps, dtype=np.float64)
    m = n - c
    h = np.zeros(num_taps, dtype=np.float64)
    odd = (m.astype(np.int64) % 2) != 0
    h[odd] = 2.0 / (np.pi * m[odd])
    w = 0.54 - 0.46 * np.cos(2.0 * np.pi * n / max(num_taps - 1, 1)) if num_taps > 1 else np.ones(1)
    h = h * w
    d = np.zeros(num_taps, dtype=np.float64)
    d[c] = 1.0
    return as_f32(h[::-1]), as_f32(d[::-1])


def design_bandpass(n_stages, f0, q):
    w0 = 2.0 * np.pi * f0
    alpha = np.sin(w0) / (2.0 * q)
    a0 = 1.0 + alpha
    b = np.array([alpha, 0.0, -alpha]) / a0
    a1, a2 = -2.0 * np.cos(w0) / a0, (1.0 - alpha) / a0
    one = np.array([b[0], b[1], b[2], -a1, -a2], dtype=np.float64)   # CMSIS adds the feedback
    return as_f32(np.tile(one, n_stages))


class ChainSpec:
    """Python-side description of one instance; keeps the numpy arrays alive for ctypes."""

    def __init__(self, channels, block, decim=1, nd_taps=0, nh_taps=0, n_biquad=0, mode=MODE_USB,
                 arith=ARITH_CMSIS, nco=False, nco_step_all=0, nco_steps=None, agc=True,
                 agc_global=False, dec_cutoff=None, bp_f0=500.0 / 48000.0, bp_q=4.0,
                 agc_params=None, q15_rounding=False):
        self.channels, self.block, self.decim = channels, block, decim
        self.nd_taps, self.nh_taps, self.n_biquad = nd_taps, nh_taps, n_biquad
        self.mode, self.arith = mode, arith
        self.nco, self.nco_step_all = nco, nco_step_all
        self.nco_steps = None if nco_steps is None else np.ascontiguousarray(nco_steps, dtype=np.uint32)
        self.agc, self.agc_global = agc, agc_global
        self.q15_rounding = bool(q15_rounding)   # int16 output: the ARM_MATH_ROUNDING variant of arm_float_to_q15
        if dec_cutoff is None:
            dec_cutoff = 0.4 / decim
        self.dec = design_lowpass(nd_taps, dec_cutoff) if nd_taps else None
        self.hilb, self.delay = design_hilbert(nh_taps) if nh_taps else (None, None)
        self.biq = design_bandpass(n_biquad, bp_f0, bp_q) if n_biquad else None
        self.agc_params = dict(target=0.5, attack=0.5, decay=0.05, gain_min=1e-3, gain_max=1e4,
                               env_floor=1e-6, gain_init=1.0)
        if agc_params:
            self.agc_params.update(agc_params)

    def config(self):
        g = Config()
        # (SELENITE_RX_LIB may name an older build for an A/B comparison: a version-1 library gets the version-1 struct_size -- the library
        # then reads nothing behind agc_gain_init but q15_rounding, which the round-5 builds kept in that padding)
        from . import lib, CONFIG_SIZE_V1
        g.struct_size = C.sizeof(Config) if lib().selenite_rx_abi_version() >= 2 else CONFIG_SIZE_V1
        g.channels, g.block, g.decim = self.channels, self.block, self.decim
        g.nd_taps, g.nh_taps, g.n_biquad, g.arith = self.nd_taps, self.nh_taps, self.n_biquad, self.arith
        g.mode, g.nco_enable = self.mode, int(self.nco)
        g.agc_enable, g.agc_global = int(self.agc), int(self.agc_global)
        g.nco_step_all = self.nco_step_all
        g.dec_coeffs, g.hilb_coeffs = fptr(self.dec), fptr(self.hilb)
        g.delay_coeffs, g.biquad_coeffs = fptr(self.delay), fptr(self.biq)
        g.nco_step = self.nco_steps.ctypes.data_as(u32p) if self.nco_steps is not None else None
        p = self.agc_params
        g.agc_target, g.agc_attack, g.agc_decay = p["target"], p["attack"], p["decay"]
        g.agc_gain_min, g.agc_gain_max = p["gain_min"], p["gain_max"]
        g.agc_env_floor, g.agc_gain_init = p["env_floor"], p["gain_init"]
        g.q15_rounding = int(self.q15_rounding)
        g.abi_version = 2
        g._keepalive = self          # the struct only holds raw pointers into this spec's arrays
        return g

    def state_arrays(self):
        c = self.channels
        return dict(
            dec_state=np.zeros((c, 2, max(self.nd_taps - 1, 0)), np.float32),
            fir_state=np.zeros((c, 2, max(self.nh_taps - 1, 0)), np.float32),
            biq_state=np.zeros((c, self.n_biquad, 4), np.float32),
            agc_gain=np.zeros((c,), np.float32),
            nco_phase=np.zeros((c,), np.uint32),
        )

    def out_len(self, block_size):
        return block_size // self.decim


# named BASELINE.json configurations (scaled by the caller through `channels` / block_size)
def baseline_spec(name, channels, arith=ARITH_CMSIS, **kw):
    kw.setdefault("agc", True)
    if name == "cfg1":   # single 256-sample block, USB, 63-tap Hilbert + AGC
        return ChainSpec(channels, 256, 1, 0, 63, 0, MODE_USB, arith, **kw)
    if name == "cfg2":   # 127-tap Hilbert SSB + AGC @48k
        return ChainSpec(channels, 256, 1, 0, 127, 0, MODE_USB, arith, **kw)
    if name == "cfg2_48k":   # the same chain with DSP blocks of 192 frames: BASELINE cfg2's literal 48 000 samples per second are 250 of them
        return ChainSpec(channels, 192, 1, 0, 127, 0, MODE_USB, arith, **kw)
    if name == "cfg2_48k128":   # ... and with DSP blocks of 128 frames (375 of them): passes of 256 outputs = two blocks, the last pass of a call one block
        return ChainSpec(channels, 128, 1, 0, 127, 0, MODE_USB, arith, **kw)
    if name == "cfg3":   # NCO + 256-tap /4 + 63-tap SSB + AGC  (the headline config)
        return ChainSpec(channels, 256, 4, 256, 63, 0, MODE_USB, arith, nco=True,
                         nco_step_all=0x01000000, **kw)
    if name == "cfg3_by8":   # the cfg3 chain decimating by 8 (k_ssb_split16 on the by-4 product, every second output kept)
        return ChainSpec(channels, 256, 8, 256, 63, 0, MODE_USB, arith, nco=True,
                         nco_step_all=0x01000000, **kw)
    if name == "cfg4":   # CW narrow: NCO (BFO) + 4-stage biquad @500 Hz + AGC
        return ChainSpec(channels, 256, 1, 0, 0, 4, MODE_CW, arith, nco=True,
                         nco_step_all=0x00800000, **kw)
    raise KeyError(name)


# bench.py workloads: name -> (BASELINE configuration, channels per GPU, complex samples per channel and call)
WORKLOADS = {
    "cfg3": ("cfg3", 65536, 4096),     # headline: NCO + 256-tap /4 + 63-tap SSB + AGC
    "cfg2": ("cfg2_48k128", 4096, 48000),  # BASELINE cfg2 literally: 4096 channels x 48 kS/s (one second per call), DSP block 128 = 375 blocks (187 passes of 256 outputs + one of 128)
    "cfg2_192": ("cfg2_48k", 4096, 48000),  # the same second in DSP blocks of 192 frames (250 blocks, passes of 192 outputs: three quarters of every tile)
    "cfg4": ("cfg4", 65536, 4096),
    "cfg5": ("cfg2", 131072, 1024),    # weak-scaling shape of BASELINE cfg5 (cfg2 chain)
}


def flops_per_sample(spec, nco_flops=20.0):
    """SURVEY.md 8d algorithmic flops per complex input sample (`nco_flops`: what the NCO of the timed kernel
    executes per sample -- 20 for the per-sample arm_sin/cos_f32 + complex multiply, 6 for the complex multiply
    alone when the LO comes from the shared per-call table)."""
    f = 0.0
    if spec.nd_taps:
        f += 2.0 * 2.0 * spec.nd_taps / spec.decim
    if spec.nh_taps:
        f += 2.0 * spec.nh_taps / spec.decim
    if spec.nco:
        f += nco_flops
    if spec.n_biquad:
        f += 9.0 * spec.n_biquad / spec.decim
    return f
