"""Shared test plumbing: ctypes views of include/selenite_rx.h, the oracle (oracle/librx_oracle.so)
and -- in the build container only -- the real CMSIS-DSP composition (oracle/_ref/libcmsis_ref.so).

TEST INFRASTRUCTURE.  Nothing here is imported by the product.
"""
import ctypes as C
import os
import subprocess
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORACLE_DIR = os.path.join(ROOT, "oracle")
# SELENITE_ORACLE_SAN=1: the AddressSanitizer / UBSan builds of the same libraries (make -C oracle SAN=1 -> oracle/_san/;
# python needs libasan preloaded: tools/run_cpu_tests_sanitized.sh)
SAN = os.environ.get("SELENITE_ORACLE_SAN") == "1"
LIB_DIR = os.path.join(ORACLE_DIR, "_san") if SAN else ORACLE_DIR
REF_DIR = os.path.join(ORACLE_DIR, "_san" if SAN else "_ref")
GOLDEN_DIR = os.path.join(ROOT, "tests", "golden")
PKG_DIR = os.path.join(ROOT, "selenite-lite_amd")
if PKG_DIR not in sys.path:
    sys.path.insert(0, PKG_DIR)

MODE_LSB, MODE_USB, MODE_CW, MODE_CWR, MODE_AM, MODE_FM, MODE_DIG, MODE_PKT = 0, 1, 2, 3, 4, 8, 0x0A, 0x0C
ARITH_CMSIS, ARITH_FMA, ARITH_SPLIT16, ARITH_AUTO = 0, 1, 2, 3
SUCCESS, ARGUMENT_ERROR, LENGTH_ERROR, NANINF, DEVICE_ERROR = 0, -1, -2, -4, -7
SEED = 0x5E1E917E

f32p = C.POINTER(C.c_float)
u32p = C.POINTER(C.c_uint32)
i16p = C.POINTER(C.c_int16)


from selenite_rx import Config, StateView, TxConfig, TxStateView  # noqa: E402  (ctypes structs of include/selenite_rx.h)
from selenite_rx.chain import (ChainSpec, as_f32, baseline_spec, design_bandpass, design_hilbert,  # noqa: E402,F401
                               design_lowpass, fptr)


def state_view(arrs):
    v = StateView()
    v.dec_state = fptr(arrs["dec_state"]) if arrs["dec_state"].size else None
    v.fir_state = fptr(arrs["fir_state"]) if arrs["fir_state"].size else None
    v.biq_state = fptr(arrs["biq_state"]) if arrs["biq_state"].size else None
    v.agc_gain = fptr(arrs["agc_gain"])
    v.nco_phase = arrs["nco_phase"].ctypes.data_as(u32p)
    return v


# ---------------------------------------------------------------------------------------------
# oracle / reference libraries
# ---------------------------------------------------------------------------------------------
def build_oracle():
    """(Re)build oracle/librx_oracle.so (gcc) and, where /root/reference exists, oracle/_ref."""
    subprocess.run(["make", "-s", "-C", ORACLE_DIR] + (["SAN=1"] if SAN else []), check=True,
                   stdout=subprocess.DEVNULL, stderr=subprocess.PIPE)


_oracle = None
_ref = None


def oracle_lib():
    global _oracle
    if _oracle is None:
        path = os.path.join(LIB_DIR, "librx_oracle.so")
        src_m = max(os.path.getmtime(os.path.join(ORACLE_DIR, f)) for f in ("rx_oracle.c", "rx_oracle.h"))
        if not os.path.exists(path) or os.path.getmtime(path) < src_m:
            build_oracle()
        L = C.CDLL(path)
        L.orc_sin_table.restype = f32p
        L.orc_sin_f32.restype = C.c_float
        L.orc_sin_f32.argtypes = [C.c_float, C.c_int]
        L.orc_cos_f32.restype = C.c_float
        L.orc_cos_f32.argtypes = [C.c_float, C.c_int]
        L.orc_agc_update.restype = C.c_float
        L.orc_agc_update.argtypes = [C.POINTER(Config), C.c_float, C.c_float, C.c_int]
        L.orc_fm_atan2_f32.restype = C.c_float
        L.orc_fm_atan2_f32.argtypes = [C.c_float, C.c_float]
        L.orc_rx_create.argtypes = [C.POINTER(C.c_void_p), C.POINTER(Config)]
        L.orc_rx_destroy.argtypes = [C.c_void_p]
        L.orc_rx_set_mode.argtypes = [C.c_void_p, C.c_uint8]
        L.orc_rx_process_f32.argtypes = [C.c_void_p, f32p, f32p, C.c_uint32, C.c_int]
        L.orc_rx_process_q15.argtypes = [C.c_void_p, i16p, i16p, C.c_uint32, C.c_int]
        L.orc_rx_process_f32_env.argtypes = [C.c_void_p, f32p, f32p, C.c_uint32, f32p, f32p]
        L.orc_rx_get_state.argtypes = [C.c_void_p, C.POINTER(StateView)]
        L.orc_rx_set_state.argtypes = [C.c_void_p, C.POINTER(StateView)]
        L.orc_synth_iq.argtypes = [f32p, C.c_uint32, C.c_uint32, C.c_uint64, C.c_uint32, C.c_uint64]
        u32, ci = C.c_uint32, C.c_int
        L.orc_cmplx_mult_cmplx_f32.argtypes = [f32p, f32p, f32p, u32, ci]
        L.orc_cmplx_mag_f32.argtypes = [f32p, f32p, u32, ci]
        L.orc_fir_decimate_f32.argtypes = [f32p, u32, u32, f32p, f32p, f32p, u32, ci]
        L.orc_fir_f32.argtypes = [f32p, u32, f32p, f32p, f32p, u32, ci]
        L.orc_biquad_cascade_df1_f32.argtypes = [f32p, u32, f32p, f32p, f32p, u32, ci]
        L.orc_add_f32.argtypes = [f32p, f32p, f32p, u32]
        L.orc_sub_f32.argtypes = [f32p, f32p, f32p, u32]
        L.orc_abs_f32.argtypes = [f32p, f32p, u32]
        L.orc_max_f32.argtypes = [f32p, u32, f32p, u32p]
        L.orc_scale_f32.argtypes = [f32p, C.c_float, f32p, u32]
        L.orc_q15_to_float.argtypes = [i16p, f32p, u32]
        L.orc_float_to_q15.argtypes = [f32p, i16p, u32]
        for fn in ("orc_cmplx_mult_cmplx_f32", "orc_cmplx_mag_f32", "orc_fir_decimate_f32", "orc_fir_f32",
                   "orc_biquad_cascade_df1_f32", "orc_add_f32", "orc_sub_f32", "orc_abs_f32", "orc_max_f32",
                   "orc_scale_f32", "orc_q15_to_float", "orc_float_to_q15", "orc_synth_iq",
                   "orc_rx_process_f32", "orc_rx_process_q15", "orc_rx_process_f32_env", "orc_rx_destroy"):
            getattr(L, fn).restype = None
        _oracle = L
    return _oracle


def ref_available():
    return os.path.exists(os.path.join(REF_DIR, "libcmsis_ref.so"))


def ref_lib():
    """Real CMSIS-DSP 1.5.3 (+ composition harness).  Build container only."""
    global _ref
    if _ref is None:
        L = C.CDLL(os.path.join(REF_DIR, "libcmsis_ref.so"))
        L.ref_sin_table.restype = f32p
        L.ref_rx_create.argtypes = [C.POINTER(C.c_void_p), C.POINTER(Config)]
        L.ref_rx_destroy.argtypes = [C.c_void_p]
        L.ref_rx_set_mode.argtypes = [C.c_void_p, C.c_uint8]
        L.ref_rx_process_f32.argtypes = [C.c_void_p, f32p, f32p, C.c_uint32]
        L.ref_rx_process_q15.argtypes = [C.c_void_p, i16p, i16p, C.c_uint32]
        L.ref_rx_process_f32_env.argtypes = [C.c_void_p, f32p, f32p, C.c_uint32, f32p, f32p]
        L.ref_rx_get_state.argtypes = [C.c_void_p, C.POINTER(StateView)]
        L.ref_fir_decimate.argtypes = [f32p, C.c_uint32, C.c_uint32, f32p, f32p, f32p, C.c_uint32]
        L.ref_fir.argtypes = [f32p, C.c_uint32, f32p, f32p, f32p, C.c_uint32]
        L.ref_biquad.argtypes = [f32p, C.c_uint32, f32p, f32p, f32p, C.c_uint32]
        L.ref_sin_cos.argtypes = [f32p, f32p, f32p, C.c_uint32]
        L.ref_fir_decimate_init_status.argtypes = [C.c_uint32, C.c_uint32, C.c_uint32]
        _ref = L
    return _ref


def synth_iq(first_channel, nch, first_sample, nsamp, seed=SEED):
    iq = np.empty((nch, nsamp, 2), np.float32)
    oracle_lib().orc_synth_iq(fptr(iq), first_channel, nch, first_sample, nsamp, seed)
    return iq


class CpuChain:
    """Oracle ('orc') or real-CMSIS composition ('ref') behind one Python face."""

    def __init__(self, spec, which="orc"):
        self.spec, self.which = spec, which
        self.L = oracle_lib() if which == "orc" else ref_lib()
        self.h = C.c_void_p()
        self.cfg = spec.config()
        self.rc = getattr(self.L, which + "_rx_create")(C.byref(self.h), C.byref(self.cfg))

    def ok(self):
        return self.rc == 0

    def set_mode(self, mode):
        return getattr(self.L, self.which + "_rx_set_mode")(self.h, mode)

    def process(self, iq, nthreads=1):
        iq = as_f32(iq)
        c, bs = iq.shape[0], iq.shape[1]
        out = np.empty((c, self.spec.out_len(bs)), np.float32)
        if self.which == "orc":
            self.L.orc_rx_process_f32(self.h, fptr(iq), fptr(out), bs, nthreads)
        else:
            self.L.ref_rx_process_f32(self.h, fptr(iq), fptr(out), bs)
        return out

    def process_env(self, iq, env_override=None):
        iq = as_f32(iq)
        c, bs = iq.shape[0], iq.shape[1]
        out = np.empty((c, self.spec.out_len(bs)), np.float32)
        env = np.empty((bs // self.spec.block,), np.float32)
        fn = getattr(self.L, self.which + "_rx_process_f32_env")
        fn(self.h, fptr(iq), fptr(out), bs, fptr(env_override) if env_override is not None else None, fptr(env))
        return out, env

    def process_q15(self, iq):
        iq = np.ascontiguousarray(iq, dtype=np.int16)
        c, bs = iq.shape[0], iq.shape[1]
        out = np.empty((c, self.spec.out_len(bs)), np.int16)
        if self.which == "orc":
            self.L.orc_rx_process_q15(self.h, iq.ctypes.data_as(i16p), out.ctypes.data_as(i16p), bs, 1)
        else:
            self.L.ref_rx_process_q15(self.h, iq.ctypes.data_as(i16p), out.ctypes.data_as(i16p), bs)
        return out

    def state(self):
        arrs = self.spec.state_arrays()
        v = state_view(arrs)
        getattr(self.L, self.which + "_rx_get_state")(self.h, C.byref(v))
        return arrs

    def close(self):
        if self.h:
            getattr(self.L, self.which + "_rx_destroy")(self.h)
            self.h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def bits_equal(a, b):
    """Bit-for-bit equality of float32 arrays, +0/-0 and NaN payloads included."""
    a = np.ascontiguousarray(a, np.float32)
    b = np.ascontiguousarray(b, np.float32)
    return a.shape == b.shape and np.array_equal(a.view(np.uint32), b.view(np.uint32))


def rel_err(got, ref):
    """SURVEY.md 8d parity metric: max|got-ref| / max(|ref|_inf, 1e-30)."""
    got = np.asarray(got, np.float64)
    ref = np.asarray(ref, np.float64)
    return float(np.max(np.abs(got - ref)) / max(float(np.max(np.abs(ref))), 1e-30))


# ---- DSP ring buffer oracle (oracle/ring_oracle.c, pinned against the reference's own dsp_if.c: RefRing) ----
_ring_lib = None


def ring_oracle_lib():
    global _ring_lib
    if _ring_lib is None:
        path = os.path.join(LIB_DIR, "libring_oracle.so")
        if not os.path.exists(path):
            build_oracle()
        L = C.CDLL(path)
        L.orc_ring_new.restype = C.c_void_p
        L.orc_ring_new.argtypes = [C.c_uint32, C.c_uint32]
        L.orc_ring_free.argtypes = [C.c_void_p]
        L.orc_ring_in_write.argtypes = [C.c_void_p, C.c_void_p, C.c_uint16]
        L.orc_ring_in_read.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32]
        L.orc_ring_out_write.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32]
        L.orc_ring_out_read.argtypes = [C.c_void_p, C.c_void_p, C.c_uint16]
        L.orc_ring_mute.argtypes = [C.c_void_p]
        for n, t in (("i", C.c_int16), ("q", C.c_int16), ("enable", C.c_uint8), ("rd", C.c_uint16), ("wr", C.c_uint16)):
            getattr(L, "orc_ring_" + n).restype = C.POINTER(t)
            getattr(L, "orc_ring_" + n).argtypes = [C.c_void_p]
        _ring_lib = L
    return _ring_lib


def ring_ref_available():
    return os.path.exists(os.path.join(REF_DIR, "libdsp_if_ref.so"))


class RefRing:
    """The reference's OWN ring-buffer code (Core/Src/dsp_if.c:116-340 compiled from /root/reference into
    oracle/_ref/libdsp_if_ref.so by oracle/Makefile) behind the Python face of selenite_rx.Ring.  The firmware has
    one global IN ring and one OUT ring; a batch of independent rings is run one ring at a time by moving its state
    into the global the called function uses (oracle/ref_ring.c).  DSP_BUFF_SIZE is the firmware's: 768 frames."""

    def __init__(self, channels, frames=None):
        # RTLD_LAZY: dsp_if.c / main.c also hold functions that call the HAL; the harness never reaches them
        self.L = C.CDLL(os.path.join(REF_DIR, "libdsp_if_ref.so"), mode=os.RTLD_LAZY)
        vp, L = C.c_void_p, self.L
        L.ref_ring_set.argtypes = [C.c_int, vp, vp, C.c_uint8, C.c_uint16, C.c_uint16]
        L.ref_ring_get.argtypes = [C.c_int, vp, vp, C.POINTER(C.c_uint8), C.POINTER(C.c_uint16), C.POINTER(C.c_uint16)]
        L.ref_ring_in_write.argtypes = [vp, C.c_uint16]
        L.ref_ring_in_read.argtypes = [vp, C.c_uint32]
        L.ref_ring_out_write.argtypes = [vp, C.c_uint32]
        L.ref_ring_out_read.argtypes = [vp, C.c_uint16]
        for fn in ("ref_ring_set", "ref_ring_get", "ref_ring_in_write", "ref_ring_in_read", "ref_ring_out_write",
                   "ref_ring_out_read", "ref_ring_out_mute"):
            getattr(L, fn).restype = None
        self.frames = int(self.L.ref_ring_frames())
        assert frames in (None, self.frames), "the firmware's DSP_BUFF_SIZE is %d" % self.frames
        self.channels = channels
        self.s = {"i": np.zeros((channels, self.frames), np.int16), "q": np.zeros((channels, self.frames), np.int16),
                  "buff_enable": np.zeros(channels, np.uint8), "rd_ptr": np.zeros(channels, np.uint16),
                  "wr_ptr": np.zeros(channels, np.uint16)}

    def _each(self, out, fn):
        s = self.s
        en, rd, wr = C.c_uint8(), C.c_uint16(), C.c_uint16()
        for c in range(self.channels):
            self.L.ref_ring_set(out, s["i"][c].ctypes.data, s["q"][c].ctypes.data, int(s["buff_enable"][c]),
                                int(s["rd_ptr"][c]), int(s["wr_ptr"][c]))
            fn(c)
            self.L.ref_ring_get(out, s["i"][c].ctypes.data, s["q"][c].ctypes.data, C.byref(en), C.byref(rd), C.byref(wr))
            s["buff_enable"][c], s["rd_ptr"][c], s["wr_ptr"][c] = en.value, rd.value, wr.value

    def in_write(self, pkt):
        pkt = np.ascontiguousarray(pkt, np.int16)
        self._each(0, lambda c: self.L.ref_ring_in_write(pkt[c].ctypes.data, pkt.shape[1]))

    def out_write(self, pkt):
        pkt = np.ascontiguousarray(pkt, np.int16)
        self._each(1, lambda c: self.L.ref_ring_out_write(pkt[c].ctypes.data, 2 * pkt.shape[1]))

    def in_read(self, size_bytes):
        out = np.zeros((self.channels, size_bytes // 2), np.int16)
        self._each(0, lambda c: self.L.ref_ring_in_read(out[c].ctypes.data, size_bytes))
        return out

    def out_read(self, size_words):
        out = np.zeros((self.channels, size_words), np.int16)
        self._each(1, lambda c: self.L.ref_ring_out_read(out[c].ctypes.data, size_words))
        return out

    def mute(self):
        self._each(1, lambda c: self.L.ref_ring_out_mute())

    def state(self):
        return {k: v.copy() for k, v in self.s.items()}

    def set_state(self, a):
        for k in self.s:
            self.s[k][...] = a[k]


class OracleRing:
    """oracle/ring_oracle.c behind the same Python face as selenite_rx.Ring."""

    def __init__(self, channels, frames=384):
        self.L = ring_oracle_lib()
        self.channels, self.frames = channels, frames
        self.h = self.L.orc_ring_new(channels, frames)

    def in_write(self, pkt):
        pkt = np.ascontiguousarray(pkt, np.int16)
        self.L.orc_ring_in_write(self.h, pkt.ctypes.data, pkt.shape[1])

    def out_write(self, pkt):
        pkt = np.ascontiguousarray(pkt, np.int16)
        self.L.orc_ring_out_write(self.h, pkt.ctypes.data, 2 * pkt.shape[1])

    def in_read(self, size_bytes):
        out = np.empty((self.channels, size_bytes // 2), np.int16)
        self.L.orc_ring_in_read(self.h, out.ctypes.data, size_bytes)
        return out

    def out_read(self, size_words):
        out = np.empty((self.channels, size_words), np.int16)
        self.L.orc_ring_out_read(self.h, out.ctypes.data, size_words)
        return out

    def mute(self):
        self.L.orc_ring_mute(self.h)

    def _np(self, name, shape, dtype):
        ptr = getattr(self.L, "orc_ring_" + name)(self.h)
        return np.ctypeslib.as_array(ptr, shape=shape).view(dtype)

    def state(self):
        C_, N = self.channels, self.frames
        return {"i": self._np("i", (C_, N), np.int16).copy(), "q": self._np("q", (C_, N), np.int16).copy(),
                "buff_enable": self._np("enable", (C_,), np.uint8).copy(),
                "rd_ptr": self._np("rd", (C_,), np.uint16).copy(), "wr_ptr": self._np("wr", (C_,), np.uint16).copy()}

    def set_state(self, a):
        C_, N = self.channels, self.frames
        self._np("i", (C_, N), np.int16)[...] = a["i"]
        self._np("q", (C_, N), np.int16)[...] = a["q"]
        self._np("enable", (C_,), np.uint8)[...] = a["buff_enable"]
        self._np("rd", (C_,), np.uint16)[...] = a["rd_ptr"]
        self._np("wr", (C_,), np.uint16)[...] = a["wr_ptr"]

    def __del__(self):
        try:
            self.L.orc_ring_free(self.h)
        except Exception:
            pass


# ---- TX chain (include/selenite_tx.h): spec, oracle (tx_oracle.c) and real-CMSIS harness (ref_tx.c) ----
class TxSpec:
    """Python-side description of one TX instance; keeps the numpy arrays alive for ctypes."""

    def __init__(self, channels, block=64, interp=4, ni_taps=256, nh_taps=63, mode=MODE_USB, arith=ARITH_CMSIS,
                 nco=True, nco_step_all=0x01000000, nco_steps=None, alc=True, interp_cutoff=None, alc_params=None, q15_rounding=False):
        self.channels, self.block, self.interp = channels, block, interp
        self.q15_rounding = bool(q15_rounding)       # int16 I/Q output: the ARM_MATH_ROUNDING build of arm_float_to_q15
        self.ni_taps, self.nh_taps, self.mode, self.arith = ni_taps, nh_taps, mode, arith
        self.nco, self.nco_step_all, self.alc = nco, nco_step_all, alc
        self.nco_steps = None if nco_steps is None else np.ascontiguousarray(nco_steps, dtype=np.uint32)
        if interp_cutoff is None:
            interp_cutoff = 0.4 / interp
        # interpolation low-pass with gain L so that the pass-band level survives zero stuffing
        self.ic = (design_lowpass(ni_taps, interp_cutoff) * np.float32(interp)).astype(np.float32) if ni_taps else None
        self.hilb, self.delay = design_hilbert(nh_taps) if nh_taps else (None, None)
        self.alc_params = dict(target=0.5, attack=0.5, decay=0.05, gain_min=1e-3, gain_max=1e2,
                               env_floor=1e-6, gain_init=1.0)
        if alc_params:
            self.alc_params.update(alc_params)

    def config(self):
        g = TxConfig()
        g.struct_size = C.sizeof(TxConfig)
        g.channels, g.block, g.interp = self.channels, self.block, self.interp
        g.ni_taps, g.nh_taps, g.arith, g.mode = self.ni_taps, self.nh_taps, self.arith, self.mode
        g.nco_enable, g.alc_enable, g.nco_step_all = int(self.nco), int(self.alc), self.nco_step_all
        g.interp_coeffs, g.hilb_coeffs, g.delay_coeffs = fptr(self.ic), fptr(self.hilb), fptr(self.delay)
        g.nco_step = self.nco_steps.ctypes.data_as(u32p) if self.nco_steps is not None else None
        p = self.alc_params
        g.alc_target, g.alc_attack, g.alc_decay = p["target"], p["attack"], p["decay"]
        g.alc_gain_min, g.alc_gain_max = p["gain_min"], p["gain_max"]
        g.alc_env_floor, g.alc_gain_init = p["env_floor"], p["gain_init"]
        g.q15_rounding = int(self.q15_rounding)
        g.abi_version = 2
        g._keepalive = self
        return g

    def state_arrays(self):
        c = self.channels
        return {"fir_state": np.zeros((c, 2, max(self.nh_taps - 1, 0)), np.float32),
                "interp_state": np.zeros((c, 2, max(self.ni_taps // self.interp - 1, 0) if self.ni_taps else 0), np.float32),
                "alc_gain": np.zeros(c, np.float32), "nco_phase": np.zeros(c, np.uint32)}


def synth_audio(first_channel, nch, first_sample, nsamp):
    """Test audio: the I rail of the synthetic RX input (three tones + noise, |x| < 1)."""
    return np.ascontiguousarray(synth_iq(first_channel, nch, first_sample, nsamp)[:, :, 0])


class TxCpuChain:
    """tx_oracle.c ('orc') or the real-CMSIS composition ref_tx.c ('ref') behind one face."""

    def __init__(self, spec, which="orc"):
        self.spec, self.which = spec, which
        self.L = oracle_lib() if which == "orc" else ref_lib()
        self.cfg = spec.config()
        self.h = C.c_void_p()
        pre = which + "_tx_"
        getattr(self.L, pre + "create").argtypes = [C.POINTER(C.c_void_p), C.POINTER(TxConfig)]
        for n in ("process_f32", "process_q15"):
            getattr(self.L, pre + n).argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint32]
            getattr(self.L, pre + n).restype = None
        getattr(self.L, pre + "get_state").argtypes = [C.c_void_p, C.POINTER(TxStateView)]
        getattr(self.L, pre + "set_mode").argtypes = [C.c_void_p, C.c_uint8]
        getattr(self.L, pre + "destroy").argtypes = [C.c_void_p]
        self.rc = getattr(self.L, pre + "create")(C.byref(self.h), C.byref(self.cfg))

    def ok(self):
        return self.rc == 0 and bool(self.h)

    def set_mode(self, mode):
        return getattr(self.L, self.which + "_tx_set_mode")(self.h, mode)

    def process(self, audio):
        audio = np.ascontiguousarray(audio, np.float32)
        c, bs = audio.shape
        out = np.empty((c, bs * self.spec.interp, 2), np.float32)
        getattr(self.L, self.which + "_tx_process_f32")(self.h, audio.ctypes.data, out.ctypes.data, bs)
        return out

    def process_q15(self, audio):
        audio = np.ascontiguousarray(audio, np.int16)
        c, bs = audio.shape
        out = np.empty((c, bs * self.spec.interp, 2), np.int16)
        getattr(self.L, self.which + "_tx_process_q15")(self.h, audio.ctypes.data, out.ctypes.data, bs)
        return out

    def state(self):
        a = self.spec.state_arrays()
        v = TxStateView()
        v.fir_state = fptr(a["fir_state"]) if a["fir_state"].size else None
        v.interp_state = fptr(a["interp_state"]) if a["interp_state"].size else None
        v.alc_gain = fptr(a["alc_gain"])
        v.nco_phase = a["nco_phase"].ctypes.data_as(u32p)
        getattr(self.L, self.which + "_tx_get_state")(self.h, C.byref(v))
        return a

    def __del__(self):
        try:
            if self.h:
                getattr(self.L, self.which + "_tx_destroy")(self.h)
        except Exception:
            pass
