"""ctypes face of libselenite_rx.so (include/selenite_rx.h).

Thin plumbing for tests and bench.py: every call goes straight through the C-ABI into the HIP
library.  There is no Python or CPU implementation of the chain here -- if the shared library is
missing or no HIP device is usable, construction fails loudly.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
PKG_ROOT = os.path.dirname(_HERE)
LIB_PATH = os.path.join(PKG_ROOT, "libselenite_rx.so")
if os.environ.get("SELENITE_RX_LIB"):            # A/B experiments: another build of the same library
    LIB_PATH = os.environ["SELENITE_RX_LIB"]

MODE_LSB, MODE_USB, MODE_CW, MODE_CWR, MODE_AM, MODE_FM, MODE_DIG, MODE_PKT = 0, 1, 2, 3, 4, 8, 0x0A, 0x0C
ARITH_CMSIS, ARITH_FMA, ARITH_SPLIT16, ARITH_AUTO = 0, 1, 2, 3
ABI_VERSION, CONFIG_SIZE_V1, TX_CONFIG_SIZE_V1 = 2, 112, 96      # include/selenite_rx.h, selenite_tx.h
OPT_FORCE_GENERIC, OPT_NO_SHARED_LO, OPT_NO_PERIODIC_LO, OPT_RERUN_GRID, OPT_TX_FORCE_GENERIC, OPT_CW_GRID = 0, 1, 2, 3, 4, 5      # selenite_rx_set_plan_option
SUCCESS, ARGUMENT_ERROR, LENGTH_ERROR, NANINF, DEVICE_ERROR = 0, -1, -2, -4, -7

f32p = C.POINTER(C.c_float)
u32p = C.POINTER(C.c_uint32)
i16p = C.POINTER(C.c_int16)


class Config(C.Structure):
    """struct selenite_rx_config."""
    _fields_ = [
        ("struct_size", C.c_uint32), ("channels", C.c_uint32), ("block", C.c_uint32),
        ("decim", C.c_uint32), ("nd_taps", C.c_uint32), ("nh_taps", C.c_uint32),
        ("n_biquad", C.c_uint32), ("arith", C.c_uint32),
        ("mode", C.c_uint8), ("nco_enable", C.c_uint8), ("agc_enable", C.c_uint8), ("agc_global", C.c_uint8),
        ("nco_step_all", C.c_uint32),
        ("dec_coeffs", f32p), ("hilb_coeffs", f32p), ("delay_coeffs", f32p), ("biquad_coeffs", f32p),
        ("nco_step", u32p),
        ("agc_target", C.c_float), ("agc_attack", C.c_float), ("agc_decay", C.c_float),
        ("agc_gain_min", C.c_float), ("agc_gain_max", C.c_float), ("agc_env_floor", C.c_float),
        ("agc_gain_init", C.c_float),
        ("q15_rounding", C.c_uint32), ("abi_version", C.c_uint32), ("reserved", C.c_uint32),     # ABI version 2 (version 1 ends with agc_gain_init)
    ]


class StateView(C.Structure):
    """struct selenite_rx_state_view."""
    _fields_ = [("dec_state", f32p), ("fir_state", f32p), ("biq_state", f32p),
                ("agc_gain", f32p), ("nco_phase", u32p)]


# every symbol include/selenite_rx.h declares (tests check the library exports all of them)
ABI_SYMBOLS = [
    "selenite_rx_init", "selenite_rx_free", "selenite_rx_set_mode", "selenite_rx_status",
    "selenite_rx_error_string", "selenite_rx_process_f32", "selenite_rx_process_f32_device",
    "selenite_rx_process_q15", "selenite_rx_process_q15_device",
    "selenite_rx_global_phase1_device", "selenite_rx_global_phase2_device",
    "selenite_rx_set_stream", "selenite_rx_sync", "selenite_rx_get_state", "selenite_rx_set_state",
    "selenite_rx_reset", "selenite_rx_device_alloc", "selenite_rx_device_free",
    "selenite_rx_memcpy_h2d", "selenite_rx_memcpy_d2h", "selenite_rx_device_count",
    "selenite_rx_set_device", "selenite_rx_synth_iq_host", "selenite_rx_synth_iq_device",
    "selenite_rx_time_process_device", "selenite_rx_time_process_q15_device", "selenite_rx_kernel_name", "selenite_rx_nco_path", "selenite_rx_algorithmic_bytes",
    "selenite_rx_design_lowpass", "selenite_rx_design_hilbert", "selenite_rx_design_bandpass",
    "selenite_rx_abi_version",
    "selenite_rx_global_process_f32_device",
    "selenite_rx_host_alloc", "selenite_rx_host_free", "selenite_rx_host_register", "selenite_rx_host_unregister",
    "selenite_rx_time_process_each_device", "selenite_rx_time_streaming_roof_device", "selenite_rx_time_pattern_roof_device", "selenite_rx_device_pci_bus_id", "selenite_rx_set_plan_option", "selenite_rx_get_plan_option",
    "selenite_rx_set_guard_ratio", "selenite_rx_guard_stats", "selenite_rx_guard_channels", "selenite_rx_auto_words", "selenite_rx_guard_handover", "selenite_rx_set_handover_repair", "selenite_rx_set_auto_launches", "selenite_rx_auto_launches_last", "selenite_rx_guard_clear",
]

class TxConfig(C.Structure):
    """struct selenite_tx_config (include/selenite_tx.h)."""
    _fields_ = [
        ("struct_size", C.c_uint32), ("channels", C.c_uint32), ("block", C.c_uint32),
        ("interp", C.c_uint32), ("ni_taps", C.c_uint32), ("nh_taps", C.c_uint32),
        ("arith", C.c_uint8), ("mode", C.c_uint8), ("nco_enable", C.c_uint8), ("alc_enable", C.c_uint8),
        ("nco_step_all", C.c_uint32),
        ("interp_coeffs", f32p), ("hilb_coeffs", f32p), ("delay_coeffs", f32p), ("nco_step", u32p),
        ("alc_target", C.c_float), ("alc_attack", C.c_float), ("alc_decay", C.c_float),
        ("alc_gain_min", C.c_float), ("alc_gain_max", C.c_float), ("alc_env_floor", C.c_float),
        ("alc_gain_init", C.c_float),
        ("q15_rounding", C.c_uint32), ("abi_version", C.c_uint32), ("reserved", C.c_uint32),     # ABI version 2
    ]


class TxStateView(C.Structure):
    """struct selenite_tx_state_view."""
    _fields_ = [("fir_state", f32p), ("interp_state", f32p), ("alc_gain", f32p), ("nco_phase", u32p)]


# every symbol include/selenite_tx.h declares
TX_ABI_SYMBOLS = [
    "selenite_tx_init", "selenite_tx_free", "selenite_tx_set_mode", "selenite_tx_status", "selenite_tx_error_string",
    "selenite_tx_kernel_name",
    "selenite_tx_process_f32", "selenite_tx_process_q15", "selenite_tx_process_f32_device",
    "selenite_tx_process_q15_device", "selenite_tx_set_stream", "selenite_tx_sync", "selenite_tx_get_state",
    "selenite_tx_set_state", "selenite_tx_reset", "selenite_tx_time_process_device",
]

# every symbol include/selenite_ring.h declares
RING_ABI_SYMBOLS = [
    "selenite_ring_init", "selenite_ring_free", "selenite_ring_status", "selenite_ring_error_string",
    "selenite_ring_set_stream", "selenite_ring_sync",
    "selenite_ring_in_write_device", "selenite_ring_in_read_device",
    "selenite_ring_out_write_device", "selenite_ring_out_read_device", "selenite_ring_mute",
    "selenite_ring_in_write", "selenite_ring_in_read", "selenite_ring_out_write", "selenite_ring_out_read",
    "selenite_ring_get_state", "selenite_ring_set_state", "selenite_ring_time_device",
]

_lib = None


def build(force=False):
    """Compile libselenite_rx.so in-tree with hipcc for gfx950 (selenite-lite_amd/Makefile)."""
    if force:
        subprocess.run(["make", "-s", "-C", PKG_ROOT, "clean"], check=True)
    subprocess.run(["make", "-s", "-j4", "-C", PKG_ROOT], check=True)


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(
                "libselenite_rx.so is not built (run __graft_entry__.build() or make -C selenite-lite_amd); "
                "there is no CPU fallback for the RX chain")
        L = C.CDLL(LIB_PATH)
        vp = C.c_void_p
        L.selenite_rx_init.argtypes = [C.POINTER(vp), C.POINTER(Config)]
        L.selenite_rx_free.argtypes = [vp]
        L.selenite_rx_set_mode.argtypes = [vp, C.c_uint8]
        L.selenite_rx_status.argtypes = [vp]
        L.selenite_rx_error_string.argtypes = [vp]
        L.selenite_rx_error_string.restype = C.c_char_p
        L.selenite_rx_process_f32.argtypes = [vp, f32p, f32p, C.c_uint32]
        L.selenite_rx_process_f32_device.argtypes = [vp, vp, vp, C.c_uint32]
        L.selenite_rx_process_q15.argtypes = [vp, i16p, i16p, C.c_uint32]
        L.selenite_rx_process_q15_device.argtypes = [vp, vp, vp, C.c_uint32]
        L.selenite_rx_global_phase1_device.argtypes = [vp, vp, vp, vp, C.c_uint32]
        L.selenite_rx_global_phase2_device.argtypes = [vp, vp, vp, C.c_uint32]
        L.selenite_rx_set_stream.argtypes = [vp, vp]
        L.selenite_rx_sync.argtypes = [vp]
        L.selenite_rx_get_state.argtypes = [vp, C.POINTER(StateView)]
        L.selenite_rx_set_state.argtypes = [vp, C.POINTER(StateView)]
        L.selenite_rx_reset.argtypes = [vp]
        L.selenite_rx_device_alloc.argtypes = [C.c_size_t]
        L.selenite_rx_device_alloc.restype = vp
        L.selenite_rx_device_free.argtypes = [vp]
        L.selenite_rx_host_alloc.argtypes = [C.c_size_t]
        L.selenite_rx_host_alloc.restype = vp
        L.selenite_rx_host_free.argtypes = [vp]
        L.selenite_rx_host_free.restype = None
        L.selenite_rx_host_register.argtypes = [vp, C.c_size_t]
        L.selenite_rx_host_unregister.argtypes = [vp]
        L.selenite_rx_memcpy_h2d.argtypes = [vp, vp, C.c_size_t]
        L.selenite_rx_memcpy_d2h.argtypes = [vp, vp, C.c_size_t]
        L.selenite_rx_set_device.argtypes = [C.c_int]
        L.selenite_rx_synth_iq_host.argtypes = [f32p, C.c_uint32, C.c_uint32, C.c_uint64, C.c_uint32, C.c_uint64]
        L.selenite_rx_synth_iq_device.argtypes = [vp, vp, C.c_uint32, C.c_uint32, C.c_uint64, C.c_uint32, C.c_uint64]
        L.selenite_rx_time_process_device.argtypes = [vp, vp, vp, C.c_uint32, C.c_uint32, f32p]
        L.selenite_rx_kernel_name.argtypes = [vp]
        L.selenite_rx_kernel_name.restype = C.c_char_p
        L.selenite_rx_nco_path.argtypes = [vp]
        L.selenite_rx_nco_path.restype = C.c_char_p
        L.selenite_rx_algorithmic_bytes.argtypes = [C.POINTER(Config), C.c_uint32, C.POINTER(C.c_uint64)]
        L.selenite_rx_algorithmic_bytes.restype = C.c_uint64
        u64p = C.POINTER(C.c_uint64)
        L.selenite_rx_time_process_each_device.argtypes = [vp, vp, vp, C.c_uint32, C.c_uint32, f32p, C.c_int]
        L.selenite_rx_time_streaming_roof_device.argtypes = [vp, vp, vp, C.c_uint32, C.c_uint32, f32p, C.c_int]
        if hasattr(L, "selenite_rx_set_plan_option"):      # (round-6 entry points: an older build named by SELENITE_RX_LIB for an A/B run lacks them)
            L.selenite_rx_set_plan_option.argtypes = [C.c_int, C.c_uint32]
            L.selenite_rx_get_plan_option.argtypes = [C.c_int]
            L.selenite_rx_get_plan_option.restype = C.c_uint32
            L.selenite_rx_time_pattern_roof_device.argtypes = [vp, vp, vp, C.c_uint32, C.c_uint32, f32p, C.c_int, C.c_uint32]
        L.selenite_rx_device_pci_bus_id.argtypes = [C.c_int, C.c_char_p, C.c_size_t]
        L.selenite_rx_set_guard_ratio.argtypes = [vp, C.c_float]
        L.selenite_rx_guard_stats.argtypes = [vp, u64p, u64p, u64p]
        L.selenite_rx_guard_channels.argtypes = [vp, u32p]
        L.selenite_rx_auto_words.argtypes = [vp, u32p]
        L.selenite_rx_guard_clear.argtypes = [vp]
        L.selenite_rx_guard_handover.argtypes = [vp, u64p]
        L.selenite_rx_set_handover_repair.argtypes = [vp, C.c_int]
        L.selenite_rx_set_auto_launches.argtypes = [vp, C.c_int]
        L.selenite_rx_auto_launches_last.argtypes = [vp]
        L.selenite_rx_design_lowpass.argtypes = [f32p, C.c_uint32, C.c_double]
        L.selenite_rx_design_hilbert.argtypes = [f32p, f32p, C.c_uint32]
        L.selenite_rx_design_bandpass.argtypes = [f32p, C.c_uint32, C.c_double, C.c_double]
        _lib = L
    return _lib


def _fp(a):
    return a.ctypes.data_as(f32p)


def device_pci_bus_id(ordinal):
    buf = C.create_string_buffer(32)
    rc = lib().selenite_rx_device_pci_bus_id(ordinal, buf, 32)
    return buf.value.decode() if rc == 0 else "unknown"


def design_lowpass(num_taps, cutoff):
    h = np.empty(num_taps, np.float32)
    rc = lib().selenite_rx_design_lowpass(_fp(h), num_taps, cutoff)
    if rc:
        raise ValueError("selenite_rx_design_lowpass: %d" % rc)
    return h


def design_hilbert(num_taps):
    h, d = np.empty(num_taps, np.float32), np.empty(num_taps, np.float32)
    rc = lib().selenite_rx_design_hilbert(_fp(h), _fp(d), num_taps)
    if rc:
        raise ValueError("selenite_rx_design_hilbert: %d" % rc)
    return h, d


def design_bandpass(n_stages, f0, q):
    c = np.empty(5 * n_stages, np.float32)
    rc = lib().selenite_rx_design_bandpass(_fp(c), n_stages, f0, q)
    if rc:
        raise ValueError("selenite_rx_design_bandpass: %d" % rc)
    return c


def synth_iq_host(first_channel, nch, first_sample, nsamp, seed):
    iq = np.empty((nch, nsamp, 2), np.float32)
    lib().selenite_rx_synth_iq_host(_fp(iq), first_channel, nch, first_sample, nsamp, seed)
    return iq


class _PinnedOwner:
    def __init__(self, nbytes):
        self.ptr = lib().selenite_rx_host_alloc(nbytes)
        if not self.ptr:
            raise MemoryError("selenite_rx_host_alloc(%d) failed" % nbytes)

    def __del__(self):
        try:
            lib().selenite_rx_host_free(self.ptr)
        except Exception:
            pass


def pinned_array(shape, dtype):
    """numpy array over page-locked host memory (selenite_rx_host_alloc): the host-pointer process calls DMA
    straight from / into it."""
    dtype = np.dtype(dtype)
    n = int(np.prod(shape)) * dtype.itemsize
    owner = _PinnedOwner(max(n, 1))
    buf = (C.c_char * max(n, 1)).from_address(owner.ptr)
    buf._owner = owner                       # the array keeps `buf` alive (arr.base), `buf` keeps the allocation
    return np.frombuffer(buf, dtype=dtype, count=int(np.prod(shape))).reshape(shape)


class DeviceBuffer:
    """hipMalloc'ed bytes owned through the C-ABI helpers."""

    def __init__(self, nbytes):
        self.nbytes = int(nbytes)
        self.ptr = lib().selenite_rx_device_alloc(self.nbytes)
        if not self.ptr:
            raise MemoryError("selenite_rx_device_alloc(%d) failed" % nbytes)

    def upload(self, arr):
        arr = np.ascontiguousarray(arr)
        assert arr.nbytes <= self.nbytes
        rc = lib().selenite_rx_memcpy_h2d(self.ptr, arr.ctypes.data, arr.nbytes)
        if rc:
            raise RuntimeError("h2d failed")

    def download(self, shape, dtype):
        out = np.empty(shape, dtype)
        assert out.nbytes <= self.nbytes
        rc = lib().selenite_rx_memcpy_d2h(out.ctypes.data, self.ptr, out.nbytes)
        if rc:
            raise RuntimeError("d2h failed")
        return out

    def free(self):
        if self.ptr:
            lib().selenite_rx_device_free(self.ptr)
            self.ptr = None

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass


class RxError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__("selenite_rx error %d: %s" % (code, msg))
        self.code = code


class plan_option:
    """`with plan_option(OPT_FORCE_GENERIC):` -- a kernel-selection override of the library (selenite_rx_set_plan_option) for the instances created
    inside the block; restored afterwards.  Results do not depend on it: the tests use it to reach every product path."""

    def __init__(self, option, value=1):
        self.option, self.value = option, value

    def __enter__(self):
        self.old = lib().selenite_rx_get_plan_option(self.option)
        if lib().selenite_rx_set_plan_option(self.option, self.value) != SUCCESS:
            raise ValueError("selenite_rx_set_plan_option(%d, %d)" % (self.option, self.value))
        return self

    def __exit__(self, *exc):
        lib().selenite_rx_set_plan_option(self.option, self.old)
        return False


class Rx:
    """One selenite_rx_instance.  `cfg` is a filled Config whose coefficient arrays the caller keeps
    alive until construction returns (the library copies them)."""

    def __init__(self, cfg):
        self.L = lib()
        self.cfg = cfg
        self.h = C.c_void_p()
        rc = self.L.selenite_rx_init(C.byref(self.h), C.byref(cfg))
        if rc != SUCCESS:
            raise RxError(rc, self.L.selenite_rx_error_string(None).decode())

    # -- CMSIS-style calls ------------------------------------------------------------------
    def out_len(self, block_size):
        return block_size // self.cfg.decim

    def process(self, iq, out=None):
        """selenite_rx_process_f32 on host arrays (pageable numpy memory, or page-locked `pinned_array`s)."""
        iq = np.ascontiguousarray(iq, np.float32)
        c, bs = iq.shape[0], iq.shape[1]
        if out is None:
            out = np.empty((c, self.out_len(bs)), np.float32)
        self.L.selenite_rx_process_f32(self.h, _fp(iq), _fp(out), bs)
        self.check()
        return out

    def process_q15(self, iq, out=None):
        iq = np.ascontiguousarray(iq, np.int16)
        c, bs = iq.shape[0], iq.shape[1]
        if out is None:
            out = np.empty((c, self.out_len(bs)), np.int16)
        self.L.selenite_rx_process_q15(self.h, iq.ctypes.data_as(i16p), out.ctypes.data_as(i16p), bs)
        self.check()
        return out

    def process_device(self, d_src, d_dst, block_size):
        self.L.selenite_rx_process_f32_device(self.h, d_src, d_dst, block_size)

    def process_q15_device(self, d_src, d_dst, block_size):
        self.L.selenite_rx_process_q15_device(self.h, d_src, d_dst, block_size)

    def global_phase1(self, d_src, d_dst, d_env, block_size):
        self.L.selenite_rx_global_phase1_device(self.h, d_src, d_dst, d_env, block_size)

    def global_phase2(self, d_dst, d_env, block_size):
        self.L.selenite_rx_global_phase2_device(self.h, d_dst, d_env, block_size)

    def set_mode(self, mode):
        return self.L.selenite_rx_set_mode(self.h, mode)

    def set_stream(self, stream_ptr):
        return self.L.selenite_rx_set_stream(self.h, stream_ptr)

    def sync(self):
        rc = self.L.selenite_rx_sync(self.h)
        if rc:
            raise RxError(rc, self.error())

    def reset(self):
        return self.L.selenite_rx_reset(self.h)

    def status(self):
        return self.L.selenite_rx_status(self.h)

    def error(self):
        return self.L.selenite_rx_error_string(self.h).decode()

    def check(self):
        rc = self.status()
        if rc:
            raise RxError(rc, self.error())

    def kernel_name(self):
        return self.L.selenite_rx_kernel_name(self.h).decode()

    # -- parity guard of the split-precision arithmetic ----------------------------------------
    def set_guard_ratio(self, ratio):
        rc = self.L.selenite_rx_set_guard_ratio(self.h, ratio)
        if rc:
            raise RxError(rc, "selenite_rx_set_guard_ratio")

    def guard_stats(self):
        """dict(blocks, channel_calls, rerun_channel_calls, handover_blocks) since init / the last guard_clear()."""
        b, c, r = C.c_uint64(), C.c_uint64(), C.c_uint64()
        rc = self.L.selenite_rx_guard_stats(self.h, C.byref(b), C.byref(c), C.byref(r))
        if rc:
            raise RxError(rc, self.error())
        h = C.c_uint64()
        rc = self.L.selenite_rx_guard_handover(self.h, C.byref(h))
        if rc:
            raise RxError(rc, self.error())
        return dict(blocks=int(b.value), channel_calls=int(c.value), rerun_channel_calls=int(r.value), handover_blocks=int(h.value))

    def set_handover_repair(self, on):
        return self.L.selenite_rx_set_handover_repair(self.h, int(bool(on)))

    def set_auto_launches(self, launches):
        """SELENITE_ARITH_AUTO: 1 (default) = the matrix kernel recomputes a channel it guarded itself where it can (k_hilb_split16: one launch
        per call), 3 = always k_hist_exact and the rerun pass behind it.  Same bits either way."""
        return self.L.selenite_rx_set_auto_launches(self.h, int(launches))

    def auto_launches_last(self):
        """1 / 3: the form the last SELENITE_ARITH_AUTO call on a matrix kernel took (0: none yet)."""
        return int(self.L.selenite_rx_auto_launches_last(self.h))

    def guard_channels(self):
        out = np.zeros(self.cfg.channels, np.uint32)
        rc = self.L.selenite_rx_guard_channels(self.h, out.ctypes.data_as(u32p))
        if rc:
            raise RxError(rc, self.error())
        return out

    def auto_words(self):
        """Diagnostic: the per-channel words of SELENITE_ARITH_AUTO (bit 0 rerun, bits 1-2 provenance, bit 5 held by the exact kernel, ...)."""
        out = np.zeros(self.cfg.channels, np.uint32)
        rc = self.L.selenite_rx_auto_words(self.h, out.ctypes.data_as(u32p))
        if rc:
            raise RxError(rc, self.error())
        return out

    def guard_clear(self):
        return self.L.selenite_rx_guard_clear(self.h)

    def nco_path(self):
        return self.L.selenite_rx_nco_path(self.h).decode()

    def time_process(self, d_src, d_dst, block_size, iters):
        ms = C.c_float()
        rc = self.L.selenite_rx_time_process_device(self.h, d_src, d_dst, block_size, iters, C.byref(ms))
        if rc:
            raise RxError(rc, self.error())
        return ms.value

    def time_process_q15(self, d_src, d_dst, block_size, iters):
        ms = C.c_float()
        self.L.selenite_rx_time_process_q15_device.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint32,
                                                               C.c_uint32, C.POINTER(C.c_float)]
        rc = self.L.selenite_rx_time_process_q15_device(self.h, d_src, d_dst, block_size, iters, C.byref(ms))
        if rc:
            raise RxError(rc, self.error())
        return ms.value

    def time_process_each(self, d_src, d_dst, block_size, iters, q15=False):
        """per-call durations (ms) of `iters` back-to-back device calls, one HIP event between calls"""
        ms = np.zeros(iters, np.float32)
        rc = self.L.selenite_rx_time_process_each_device(self.h, d_src, d_dst, block_size, iters, _fp(ms), int(q15))
        if rc:
            raise RxError(rc, self.error())
        return ms

    def time_streaming_roof(self, d_src, d_dst, block_size, iters, q15=False):
        """per-launch durations (ms) of the no-arithmetic kernel that moves the algorithmic bytes of one call (d_dst is overwritten)"""
        ms = np.zeros(iters, np.float32)
        rc = self.L.selenite_rx_time_streaming_roof_device(self.h, d_src, d_dst, block_size, iters, _fp(ms), int(q15))
        if rc:
            raise RxError(rc, self.error())
        return ms

    def time_pattern_roof(self, d_src, d_dst, block_size, iters, q15=False, work=0):
        """per-launch durations (ms) of the no-DSP kernel with the fetch pattern and launch shape of the systolic CW kernel, `work` dependent
        vector instructions per chunk where its biquad steps are (d_dst is overwritten; CW shapes only)"""
        ms = np.zeros(iters, np.float32)
        rc = self.L.selenite_rx_time_pattern_roof_device(self.h, d_src, d_dst, block_size, iters, _fp(ms), int(q15), int(work))
        if rc:
            raise RxError(rc, self.error())
        return ms

    def synth_device(self, d_iq, first_channel, nch, first_sample, nsamp, seed):
        rc = self.L.selenite_rx_synth_iq_device(self.h, d_iq, first_channel, nch, first_sample, nsamp, seed)
        if rc:
            raise RxError(rc, self.error())

    def algorithmic_bytes(self, block_size):
        rd = C.c_uint64()
        tot = self.L.selenite_rx_algorithmic_bytes(C.byref(self.cfg), block_size, C.byref(rd))
        return int(tot), int(rd.value)

    # -- state ------------------------------------------------------------------------------
    def _state_arrays(self):
        g = self.cfg
        c = g.channels
        return dict(
            dec_state=np.zeros((c, 2, max(g.nd_taps - 1, 0)), np.float32),
            fir_state=np.zeros((c, 2, max(g.nh_taps - 1, 0)), np.float32),
            biq_state=np.zeros((c, g.n_biquad, 4), np.float32),
            agc_gain=np.zeros((c,), np.float32),
            nco_phase=np.zeros((c,), np.uint32),
        )

    @staticmethod
    def _view(arrs):
        v = StateView()
        v.dec_state = _fp(arrs["dec_state"]) if arrs["dec_state"].size else None
        v.fir_state = _fp(arrs["fir_state"]) if arrs["fir_state"].size else None
        v.biq_state = _fp(arrs["biq_state"]) if arrs["biq_state"].size else None
        v.agc_gain = _fp(arrs["agc_gain"])
        v.nco_phase = arrs["nco_phase"].ctypes.data_as(u32p)
        return v

    def state(self):
        arrs = self._state_arrays()
        v = self._view(arrs)
        rc = self.L.selenite_rx_get_state(self.h, C.byref(v))
        if rc:
            raise RxError(rc, self.error())
        return arrs

    def set_state(self, arrs):
        arrs = {k: np.ascontiguousarray(a) for k, a in arrs.items()}
        v = self._view(arrs)
        rc = self.L.selenite_rx_set_state(self.h, C.byref(v))
        if rc:
            raise RxError(rc, self.error())

    def close(self):
        if self.h:
            self.L.selenite_rx_free(self.h)
            self.h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class RingStateView(C.Structure):
    """struct selenite_ring_state_view."""
    _fields_ = [("i", C.POINTER(C.c_int16)), ("q", C.POINTER(C.c_int16)), ("buff_enable", C.POINTER(C.c_uint8)),
                ("rd_ptr", C.POINTER(C.c_uint16)), ("wr_ptr", C.POINTER(C.c_uint16))]


class Ring:
    """`channels` reference DSP_Buff_TypeDef rings in HBM (include/selenite_ring.h; dsp_if.c:83-340).
    Method names and size units follow the reference functions they replace."""

    def __init__(self, channels, frames=384):
        self.L = lib()
        L = self.L
        vp = C.c_void_p
        L.selenite_ring_init.argtypes = [C.POINTER(vp), C.c_uint32, C.c_uint32]
        L.selenite_ring_free.argtypes = [vp]
        L.selenite_ring_status.argtypes = [vp]
        L.selenite_ring_error_string.argtypes = [vp]
        L.selenite_ring_error_string.restype = C.c_char_p
        L.selenite_ring_set_stream.argtypes = [vp, vp]
        L.selenite_ring_sync.argtypes = [vp]
        for n, a in (("in_write", C.c_uint16), ("in_read", C.c_uint32), ("out_write", C.c_uint32), ("out_read", C.c_uint16)):
            getattr(L, "selenite_ring_%s" % n).argtypes = [vp, vp, a]
            getattr(L, "selenite_ring_%s" % n).restype = None
            getattr(L, "selenite_ring_%s_device" % n).argtypes = [vp, vp, a]
            getattr(L, "selenite_ring_%s_device" % n).restype = None
        L.selenite_ring_mute.argtypes = [vp]
        L.selenite_ring_mute.restype = None
        L.selenite_ring_get_state.argtypes = [vp, C.POINTER(RingStateView)]
        L.selenite_ring_set_state.argtypes = [vp, C.POINTER(RingStateView)]
        L.selenite_ring_time_device.argtypes = [vp, vp, vp, C.c_uint16, C.c_uint32, C.POINTER(C.c_float)]
        self.channels, self.frames = int(channels), int(frames)
        self.h = vp()
        rc = L.selenite_ring_init(C.byref(self.h), self.channels, self.frames)
        if rc:
            raise RxError(rc, "selenite_ring_init failed (no GPU, or bad channels/frames)")

    def check(self):
        rc = self.L.selenite_ring_status(self.h)
        if rc:
            raise RxError(rc, self.L.selenite_ring_error_string(self.h).decode())

    def _packet(self, arr, words):
        arr = np.ascontiguousarray(arr, np.int16)
        assert arr.shape == (self.channels, words), arr.shape
        return arr

    def in_write(self, pkt):                       # DSP_In_Buff_Write (words)
        pkt = np.ascontiguousarray(pkt, np.int16)
        self.L.selenite_ring_in_write(self.h, self._packet(pkt, pkt.shape[1]).ctypes.data, pkt.shape[1])
        self.check()

    def out_write(self, pkt):                      # DSP_Out_Buff_Write (bytes)
        pkt = np.ascontiguousarray(pkt, np.int16)
        self.L.selenite_ring_out_write(self.h, self._packet(pkt, pkt.shape[1]).ctypes.data, 2 * pkt.shape[1])
        self.check()

    def in_read(self, size_bytes):                 # DSP_In_Buff_Read (bytes)
        out = np.empty((self.channels, size_bytes // 2), np.int16)
        self.L.selenite_ring_in_read(self.h, out.ctypes.data, size_bytes)
        self.check()
        return out

    def out_read(self, size_words):                # DSP_Out_Buff_Read (words)
        out = np.empty((self.channels, size_words), np.int16)
        self.L.selenite_ring_out_read(self.h, out.ctypes.data, size_words)
        self.check()
        return out

    def mute(self):
        self.L.selenite_ring_mute(self.h)
        self.L.selenite_ring_sync(self.h)
        self.check()

    def _arrays(self):
        return {"i": np.zeros((self.channels, self.frames), np.int16),
                "q": np.zeros((self.channels, self.frames), np.int16),
                "buff_enable": np.zeros(self.channels, np.uint8),
                "rd_ptr": np.zeros(self.channels, np.uint16),
                "wr_ptr": np.zeros(self.channels, np.uint16)}

    @staticmethod
    def _view(a):
        return RingStateView(a["i"].ctypes.data_as(C.POINTER(C.c_int16)), a["q"].ctypes.data_as(C.POINTER(C.c_int16)),
                             a["buff_enable"].ctypes.data_as(C.POINTER(C.c_uint8)),
                             a["rd_ptr"].ctypes.data_as(C.POINTER(C.c_uint16)),
                             a["wr_ptr"].ctypes.data_as(C.POINTER(C.c_uint16)))

    def state(self):
        a = self._arrays()
        v = self._view(a)
        rc = self.L.selenite_ring_get_state(self.h, C.byref(v))
        if rc:
            raise RxError(rc, self.L.selenite_ring_error_string(self.h).decode())
        return a

    def set_state(self, a):
        a = {k: np.ascontiguousarray(v) for k, v in a.items()}
        v = self._view(a)
        rc = self.L.selenite_ring_set_state(self.h, C.byref(v))
        if rc:
            raise RxError(rc, self.L.selenite_ring_error_string(self.h).decode())

    def time_pair(self, d_src, d_dst, size_words, iters):
        ms = C.c_float()
        rc = self.L.selenite_ring_time_device(self.h, d_src, d_dst, size_words, iters, C.byref(ms))
        if rc:
            raise RxError(rc, self.L.selenite_ring_error_string(self.h).decode())
        return ms.value

    def close(self):
        if self.h:
            self.L.selenite_ring_free(self.h)
            self.h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class Tx:
    """One batched TX chain instance (include/selenite_tx.h); mirror of Rx."""

    def __init__(self, cfg):
        self.L = lib()
        L = self.L
        vp = C.c_void_p
        L.selenite_tx_init.argtypes = [C.POINTER(vp), C.POINTER(TxConfig)]
        L.selenite_tx_free.argtypes = [vp]
        L.selenite_tx_set_mode.argtypes = [vp, C.c_uint8]
        L.selenite_tx_status.argtypes = [vp]
        L.selenite_tx_error_string.argtypes = [vp]
        L.selenite_tx_error_string.restype = C.c_char_p
        for n in ("process_f32", "process_q15", "process_f32_device", "process_q15_device"):
            getattr(L, "selenite_tx_" + n).argtypes = [vp, vp, vp, C.c_uint32]
            getattr(L, "selenite_tx_" + n).restype = None
        L.selenite_tx_set_stream.argtypes = [vp, vp]
        L.selenite_tx_sync.argtypes = [vp]
        L.selenite_tx_get_state.argtypes = [vp, C.POINTER(TxStateView)]
        L.selenite_tx_set_state.argtypes = [vp, C.POINTER(TxStateView)]
        L.selenite_tx_reset.argtypes = [vp]
        L.selenite_tx_time_process_device.argtypes = [vp, vp, vp, C.c_uint32, C.c_uint32, C.POINTER(C.c_float)]
        self.cfg = cfg
        self.h = vp()
        rc = L.selenite_tx_init(C.byref(self.h), C.byref(cfg))
        if rc:
            raise RxError(rc, "selenite_tx_init failed")

    def check(self):
        rc = self.L.selenite_tx_status(self.h)
        if rc:
            raise RxError(rc, self.L.selenite_tx_error_string(self.h).decode())

    def process(self, audio):
        audio = np.ascontiguousarray(audio, np.float32)
        c, bs = audio.shape
        out = np.empty((c, bs * self.cfg.interp, 2), np.float32)
        self.L.selenite_tx_process_f32(self.h, audio.ctypes.data, out.ctypes.data, bs)
        self.check()
        return out

    def process_q15(self, audio):
        audio = np.ascontiguousarray(audio, np.int16)
        c, bs = audio.shape
        out = np.empty((c, bs * self.cfg.interp, 2), np.int16)
        self.L.selenite_tx_process_q15(self.h, audio.ctypes.data, out.ctypes.data, bs)
        self.check()
        return out

    def process_device(self, d_src, d_dst, block_size):
        self.L.selenite_tx_process_f32_device(self.h, d_src, d_dst, block_size)

    def kernel_name(self):
        self.L.selenite_tx_kernel_name.argtypes = [C.c_void_p]
        self.L.selenite_tx_kernel_name.restype = C.c_char_p
        return self.L.selenite_tx_kernel_name(self.h).decode()

    def set_mode(self, mode):
        return self.L.selenite_tx_set_mode(self.h, mode)

    def sync(self):
        return self.L.selenite_tx_sync(self.h)

    def reset(self):
        return self.L.selenite_tx_reset(self.h)

    def _arrays(self):
        g = self.cfg
        return {"fir_state": np.zeros((g.channels, 2, max(g.nh_taps - 1, 0)), np.float32),
                "interp_state": np.zeros((g.channels, 2, max(g.ni_taps // g.interp - 1, 0) if g.ni_taps else 0), np.float32),
                "alc_gain": np.zeros(g.channels, np.float32), "nco_phase": np.zeros(g.channels, np.uint32)}

    @staticmethod
    def _view(a):
        v = TxStateView()
        v.fir_state = a["fir_state"].ctypes.data_as(f32p) if a["fir_state"].size else None
        v.interp_state = a["interp_state"].ctypes.data_as(f32p) if a["interp_state"].size else None
        v.alc_gain = a["alc_gain"].ctypes.data_as(f32p)
        v.nco_phase = a["nco_phase"].ctypes.data_as(u32p)
        return v

    def state(self):
        a = self._arrays()
        v = self._view(a)
        rc = self.L.selenite_tx_get_state(self.h, C.byref(v))
        if rc:
            raise RxError(rc, self.L.selenite_tx_error_string(self.h).decode())
        return a

    def set_state(self, a):
        a = {k: np.ascontiguousarray(x) for k, x in a.items()}
        v = self._view(a)
        rc = self.L.selenite_tx_set_state(self.h, C.byref(v))
        if rc:
            raise RxError(rc, self.L.selenite_tx_error_string(self.h).decode())

    def time_process(self, d_src, d_dst, block_size, iters):
        ms = C.c_float()
        rc = self.L.selenite_tx_time_process_device(self.h, d_src, d_dst, block_size, iters, C.byref(ms))
        if rc:
            raise RxError(rc, self.L.selenite_tx_error_string(self.h).decode())
        return ms.value

    def close(self):
        if self.h:
            self.L.selenite_tx_free(self.h)
            self.h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
