// rx_api.hip -- the C-ABI of include/selenite_rx.h over the HIP kernels.
//
// Host side of the drop-in boundary: mirrors the CMSIS-DSP init/process convention
// (arm_fir_decimate_init_f32.c:63-101 validation and state clearing; process calls return void)
// and the firmware's DSP_Set_Mode hook (Core/Src/dsp_if.c:367-370).  No CPU compute path exists
// here: without a usable HIP device init fails with SELENITE_RX_DEVICE_ERROR.
#include "rx_internal.h"

#include <cmath>
#include <cstddef>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <algorithm>
#include <dlfcn.h>
#include <mutex>
#include <thread>

using namespace srx;

static thread_local std::string g_last_error = "";

static int fail(selenite_rx_instance *S, int code, const std::string &msg)
{
    g_last_error = msg;
    if (S) {
        if (S->status == SELENITE_RX_SUCCESS) S->status = code;
        S->err = msg;
    }
    return code;
}
#define HIPCHK(S, call)                                                                     \
    do {                                                                                    \
        hipError_t e_ = (call);                                                             \
        if (e_ != hipSuccess)                                                               \
            return fail((S), SELENITE_RX_DEVICE_ERROR,                                      \
                        std::string(#call) + ": " + hipGetErrorString(e_));                 \
    } while (0)

// sinTable_f32[513] (CommonTables/arm_common_tables.c:21895) regenerated from the generator the
// reference documents (:21881-21891): the source holds every entry as an 8-decimal literal, so
// entry n = float("%.8f" % sin(2*pi*n/512)), sign of the (tiny negative) last entry included.
const float *srx::host_sin_table()
{
    static float table[513];
    static std::once_flag once;
    std::call_once(once, [] {
        for (int n = 0; n <= 512; ++n) {
            const double s = std::sin(2.0 * 3.14159265358979323846 * (double)n / 512.0);
            char buf[32];
            std::snprintf(buf, sizeof buf, "%.8f", s);
            table[n] = std::strtof(buf, nullptr);
        }
    });
    return table;
}

static bool mode_valid(uint8_t m, uint32_t nh_taps)
{
    if (m == SELENITE_MODE_FM) return nh_taps >= 2;        // the discriminator's one-sample memory lives in the FIR pair's delay lines
    return m == SELENITE_MODE_LSB || m == SELENITE_MODE_USB || m == SELENITE_MODE_CW ||
           m == SELENITE_MODE_CWR || m == SELENITE_MODE_AM || m == SELENITE_MODE_DIG ||
           m == SELENITE_MODE_PKT;
}

template <typename T>
static hipError_t dev_upload(T **d, const T *h, size_t n)
{
    *d = nullptr;
    if (n == 0) return hipSuccess;
    hipError_t e = hipMalloc((void **)d, n * sizeof(T));
    if (e != hipSuccess) return e;
    return hipMemcpy(*d, h, n * sizeof(T), hipMemcpyHostToDevice);
}
// allocation only: every state buffer is initialised by reset_state() on the instance's own stream
// (a null-stream hipMemset here could land AFTER reset_state's writes: the streams do not order)
template <typename T>
static hipError_t dev_alloc(T **d, size_t n)
{
    *d = nullptr;
    if (n == 0) return hipSuccess;
    return hipMalloc((void **)d, n * sizeof(T));
}

static void classify_coeffs(selenite_rx_instance *S)
{
    const uint32_t nh = S->cfg.nh_taps;
    S->delay_is_impulse = false;
    S->hilb_odd_only = false;
    if (!nh) return;
    // delay FIR that is exactly a unit impulse: arm_fir_f32 then returns x + 0.0f
    int ones = 0, idx = -1;
    bool rest_zero = true;
    for (uint32_t k = 0; k < nh; ++k) {
        const float v = S->h_delay[k];
        if (v == 1.0f) { ++ones; idx = (int)k; }
        else if (!(v == 0.0f && !std::signbit(v))) rest_zero = false;
    }
    if (ones == 1 && rest_zero) { S->delay_is_impulse = true; S->delay_index = idx; }
    // type-III Hilbert: taps at even distance from the centre are exactly +0.0f
    if (nh % 2 == 1) {
        const int c = (int)(nh - 1) / 2;
        bool ok = true;
        for (uint32_t k = 0; k < nh && ok; ++k)
            if ((((int)k - c) & 1) == 0) {
                const float v = S->h_hilb[k];
                if (!(v == 0.0f && !std::signbit(v))) ok = false;
            }
        S->hilb_odd_only = ok;
    }
}

static void free_device(selenite_rx_instance *S)
{
    void *ptrs[] = { S->d_flags, S->d_guard_ch, S->d_rerun_flag, S->d_rerun_list, S->d_hist_ext, S->d_conv_in, S->d_dec_c, S->d_hilb_c, S->d_delay_c, S->d_biq_c, S->d_sintab, S->d_step, S->d_phase,
                     S->d_dec_state, S->d_fir_state, S->d_biq_state, S->d_gain, S->d_scratch, S->d_env, S->d_env_part,
                     S->d_io_in, S->d_io_out, S->d_lo, S->pipe.d_in[0], S->pipe.d_in[1], S->pipe.d_out[0], S->pipe.d_out[1] };
    for (void *p : ptrs)
        if (p) (void)hipFree(p);
    if (S->h_rerun_seen) (void)hipHostFree(S->h_rerun_seen);
    S->h_rerun_seen = nullptr;
    for (int i = 0; i < 2; ++i) {
        if (S->pipe.h_in[i]) (void)hipHostFree(S->pipe.h_in[i]);
        if (S->pipe.h_out[i]) (void)hipHostFree(S->pipe.h_out[i]);
        if (S->pipe.ev_in[i]) (void)hipEventDestroy(S->pipe.ev_in[i]);
        if (S->pipe.ev_done[i]) (void)hipEventDestroy(S->pipe.ev_done[i]);
        if (S->pipe.ev_out[i]) (void)hipEventDestroy(S->pipe.ev_out[i]);
    }
    if (S->pipe.h2d) (void)hipStreamDestroy(S->pipe.h2d);
    if (S->pipe.d2h) (void)hipStreamDestroy(S->pipe.d2h);
    free_fused(S->plan);
    if (S->own_stream) (void)hipStreamDestroy(S->own_stream);
}

static int reset_state(selenite_rx_instance *S)
{
    const selenite_rx_config &g = S->cfg;
    const size_t C = g.channels;
    if (g.nd_taps > 1) HIPCHK(S, hipMemsetAsync(S->d_dec_state, 0, C * 2 * (g.nd_taps - 1) * sizeof(float), S->stream));
    if (g.nh_taps > 1) HIPCHK(S, hipMemsetAsync(S->d_fir_state, 0, C * 2 * (g.nh_taps - 1) * sizeof(float), S->stream));
    if (g.n_biquad) HIPCHK(S, hipMemsetAsync(S->d_biq_state, 0, C * 4 * g.n_biquad * sizeof(float), S->stream));
    HIPCHK(S, hipMemsetAsync(S->d_phase, 0, C * sizeof(uint32_t), S->stream));
    HIPCHK(S, hipMemsetAsync(S->d_flags, 0, kFlagWords * sizeof(uint32_t), S->stream));
    HIPCHK(S, hipMemsetAsync(S->d_guard_ch, 0, 3 * C * sizeof(uint32_t), S->stream));
    // (0 = no rerun pending, the channel's state is in exact arithmetic (kProvExact): what a cleared state is)
    if (S->d_rerun_flag) HIPCHK(S, hipMemsetAsync(S->d_rerun_flag, 0, C * sizeof(uint32_t), S->stream));
    if (S->d_rerun_list) HIPCHK(S, hipMemsetAsync(S->d_rerun_list, 0, 2 * sizeof(uint32_t), S->stream));      // both counters
    S->rerun_par = 0;
    std::vector<float> gi(C, g.agc_gain_init);
    HIPCHK(S, hipMemcpyAsync(S->d_gain, gi.data(), C * sizeof(float), hipMemcpyHostToDevice, S->stream));
    HIPCHK(S, hipStreamSynchronize(S->stream));
    S->phase_uniform = true;
    S->phase_host = 0;
    return SELENITE_RX_SUCCESS;
}

// the kernels' flag word (non-finite audio: ARM_MATH_NANINF), read after the stream has drained; latches the status
static int check_device_flags(selenite_rx_instance *S)
{
    uint32_t f = 0;
    HIPCHK(S, hipMemcpy(&f, S->d_flags, sizeof f, hipMemcpyDeviceToHost));
    if (f & 1u) fail(S, SELENITE_RX_NANINF, "a process call produced NaN / Inf audio (non-finite input samples?)");
    return S->status;
}

extern "C" int selenite_rx_abi_version(void) { return SELENITE_RX_ABI_VERSION; }

// ---- plan options (rx_diag.h) ----
namespace { uint32_t g_plan_opt[SELENITE_RX_OPT_COUNT] = { 0u, 0u, 0u, 0u, 0u, 0u }; }
namespace srx {
uint32_t plan_option(int option)
{
    return option >= 0 && option < SELENITE_RX_OPT_COUNT ? __atomic_load_n(&g_plan_opt[option], __ATOMIC_RELAXED) : 0u;
}
}  // namespace srx
extern "C" int selenite_rx_set_plan_option(int option, uint32_t value)
{
    if (option < 0 || option >= SELENITE_RX_OPT_COUNT) return SELENITE_RX_ARGUMENT_ERROR;
    if (option == SELENITE_RX_OPT_RERUN_GRID || option == SELENITE_RX_OPT_CW_GRID ? value > (1u << 20) : value > 1u) return SELENITE_RX_ARGUMENT_ERROR;
    __atomic_store_n(&g_plan_opt[option], value, __ATOMIC_RELAXED);
    return SELENITE_RX_SUCCESS;
}
extern "C" uint32_t selenite_rx_get_plan_option(int option) { return srx::plan_option(option); }

extern "C" int selenite_rx_device_count(void)
{
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

extern "C" int selenite_rx_set_device(int ordinal)
{
    HIPCHK(nullptr, hipSetDevice(ordinal));
    return SELENITE_RX_SUCCESS;
}

extern "C" int selenite_rx_init(selenite_rx_instance **out, const selenite_rx_config *caller_cfg)
{
    if (!out) return fail(nullptr, SELENITE_RX_ARGUMENT_ERROR, "selenite_rx_init: S is NULL");
    *out = nullptr;
    if (!caller_cfg) return fail(nullptr, SELENITE_RX_ARGUMENT_ERROR, "selenite_rx_init: cfg is NULL");
    // The caller owns the struct and says, through struct_size, which header it was built against (include/selenite_rx.h: ABI versions).
    // Version 1 ends with agc_gain_init: nothing behind it is read (it is padding of the caller's), int16 output truncates.
    static_assert(offsetof(selenite_rx_config, q15_rounding) == 108 && sizeof(selenite_rx_config) == 120, "selenite_rx_config layout (LP64)");
    selenite_rx_config own{};
    if (caller_cfg->struct_size == SELENITE_RX_CONFIG_SIZE_V1) {
        std::memcpy(&own, caller_cfg, offsetof(selenite_rx_config, q15_rounding));
        own.abi_version = 1;
    } else if (caller_cfg->struct_size == sizeof(selenite_rx_config)) {
        own = *caller_cfg;
        if (own.abi_version != SELENITE_RX_ABI_VERSION || own.reserved != 0)
            return fail(nullptr, SELENITE_RX_ARGUMENT_ERROR, "selenite_rx_init: abi_version must be SELENITE_RX_ABI_VERSION (2) and reserved 0 with this struct_size");
        if (own.q15_rounding > 1u)
            return fail(nullptr, SELENITE_RX_ARGUMENT_ERROR, "selenite_rx_init: q15_rounding is 0 or 1");
    } else {
        return fail(nullptr, SELENITE_RX_ARGUMENT_ERROR, "selenite_rx_init: struct_size matches neither this header's selenite_rx_config nor version 1's");
    }
    own.struct_size = (uint32_t)sizeof(selenite_rx_config);
    const selenite_rx_config *cfg = &own;
    if (cfg->channels == 0 || cfg->block == 0 || cfg->decim == 0)
        return fail(nullptr, SELENITE_RX_ARGUMENT_ERROR, "selenite_rx_init: channels, block, decim must be non-zero");
    if (!mode_valid(cfg->mode, cfg->nh_taps))
        return fail(nullptr, SELENITE_RX_ARGUMENT_ERROR, "selenite_rx_init: unsupported mode (FM needs the FIR pair's delay lines: nh_taps >= 2)");
    if (cfg->arith != SELENITE_ARITH_CMSIS && cfg->arith != SELENITE_ARITH_FMA && cfg->arith != SELENITE_ARITH_SPLIT16 &&
        cfg->arith != SELENITE_ARITH_AUTO)
        return fail(nullptr, SELENITE_RX_ARGUMENT_ERROR, "selenite_rx_init: bad arith");
    if (cfg->nd_taps == 0 && cfg->decim != 1)
        return fail(nullptr, SELENITE_RX_ARGUMENT_ERROR, "selenite_rx_init: decim > 1 needs a decimator (nd_taps)");
    if (cfg->nd_taps && !cfg->dec_coeffs)
        return fail(nullptr, SELENITE_RX_ARGUMENT_ERROR, "selenite_rx_init: dec_coeffs is NULL");
    if (cfg->nh_taps && (!cfg->hilb_coeffs || !cfg->delay_coeffs))
        return fail(nullptr, SELENITE_RX_ARGUMENT_ERROR, "selenite_rx_init: hilb_coeffs / delay_coeffs is NULL");
    if (cfg->n_biquad && !cfg->biquad_coeffs)
        return fail(nullptr, SELENITE_RX_ARGUMENT_ERROR, "selenite_rx_init: biquad_coeffs is NULL");
    if (cfg->nd_taps > 65535 || cfg->nh_taps > 65535 || cfg->decim > 255)   // CMSIS field widths
        return fail(nullptr, SELENITE_RX_ARGUMENT_ERROR, "selenite_rx_init: taps > 65535 or decim > 255");
    // arm_fir_decimate_init_f32.c:74-97
    if (cfg->block % cfg->decim != 0)
        return fail(nullptr, SELENITE_RX_LENGTH_ERROR, "selenite_rx_init: block is not a multiple of decim");

    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0)
        return fail(nullptr, SELENITE_RX_DEVICE_ERROR,
                    "selenite_rx_init: no HIP device (this library has no CPU fallback)");

    selenite_rx_instance *S = new selenite_rx_instance();
    S->cfg = *cfg;
    const size_t C = cfg->channels;
    if (cfg->nd_taps) S->h_dec.assign(cfg->dec_coeffs, cfg->dec_coeffs + cfg->nd_taps);
    if (cfg->nh_taps) {
        S->h_hilb.assign(cfg->hilb_coeffs, cfg->hilb_coeffs + cfg->nh_taps);
        S->h_delay.assign(cfg->delay_coeffs, cfg->delay_coeffs + cfg->nh_taps);
    }
    if (cfg->n_biquad) S->h_biq.assign(cfg->biquad_coeffs, cfg->biquad_coeffs + 5 * cfg->n_biquad);
    S->h_step.resize(C);
    for (size_t c = 0; c < C; ++c) S->h_step[c] = cfg->nco_step ? cfg->nco_step[c] : cfg->nco_step_all;
    S->cfg.dec_coeffs = S->h_dec.data(); S->cfg.hilb_coeffs = S->h_hilb.data();
    S->cfg.delay_coeffs = S->h_delay.data(); S->cfg.biquad_coeffs = S->h_biq.data();
    S->cfg.nco_step = S->h_step.data();
    S->steps_uniform = true;
    for (size_t c = 1; c < C; ++c) S->steps_uniform = S->steps_uniform && S->h_step[c] == S->h_step[0];
    S->steps_grid256 = true;
    for (size_t c = 0; c < C; ++c) S->steps_grid256 = S->steps_grid256 && (S->h_step[c] & 0x00FFFFFFu) == 0;
    // kernel-selection overrides of the tests (selenite_rx_set_plan_option): taken over at init, results do not depend on them
    S->force_generic = plan_option(SELENITE_RX_OPT_FORCE_GENERIC) ? 1 : 0;
    S->no_shared_lo = plan_option(SELENITE_RX_OPT_NO_SHARED_LO) ? 1 : 0;
    S->no_periodic_lo = plan_option(SELENITE_RX_OPT_NO_PERIODIC_LO) ? 1 : 0;

#define INITCHK(call)                                                                        \
    do {                                                                                     \
        hipError_t e_ = (call);                                                              \
        if (e_ != hipSuccess) {                                                              \
            int rc_ = fail(nullptr, SELENITE_RX_DEVICE_ERROR,                                \
                           std::string("selenite_rx_init: " #call ": ") + hipGetErrorString(e_)); \
            free_device(S);                                                                  \
            delete S;                                                                        \
            return rc_;                                                                      \
        }                                                                                    \
    } while (0)

    INITCHK(hipGetDevice(&S->device));
    INITCHK(hipStreamCreateWithFlags(&S->own_stream, hipStreamNonBlocking));
    S->stream = S->own_stream;
    INITCHK(dev_upload(&S->d_dec_c, S->h_dec.data(), S->h_dec.size()));
    INITCHK(dev_upload(&S->d_hilb_c, S->h_hilb.data(), S->h_hilb.size()));
    INITCHK(dev_upload(&S->d_delay_c, S->h_delay.data(), S->h_delay.size()));
    INITCHK(dev_upload(&S->d_biq_c, S->h_biq.data(), S->h_biq.size()));
    INITCHK(dev_upload(&S->d_sintab, host_sin_table(), (size_t)513));
    INITCHK(dev_upload(&S->d_step, S->h_step.data(), C));
    INITCHK(dev_alloc(&S->d_phase, C));
    INITCHK(dev_alloc(&S->d_dec_state, cfg->nd_taps > 1 ? C * 2 * (cfg->nd_taps - 1) : 0));
    INITCHK(dev_alloc(&S->d_fir_state, cfg->nh_taps > 1 ? C * 2 * (cfg->nh_taps - 1) : 0));
    INITCHK(dev_alloc(&S->d_biq_state, C * 4 * cfg->n_biquad));
    INITCHK(dev_alloc(&S->d_gain, C));
    INITCHK(dev_alloc(&S->d_flags, (size_t)kFlagWords));
    INITCHK(dev_alloc(&S->d_guard_ch, 3 * C));
    INITCHK(dev_alloc(&S->d_rerun_flag, cfg->arith == SELENITE_ARITH_AUTO ? C : 0));
    INITCHK(dev_alloc(&S->d_rerun_list, cfg->arith == SELENITE_ARITH_AUTO ? C + 2 : 0));
    if (cfg->arith == SELENITE_ARITH_AUTO) {
        INITCHK(hipHostMalloc(reinterpret_cast<void **>(&S->h_rerun_seen), sizeof(uint32_t), hipHostMallocMapped));
        *S->h_rerun_seen = 0u;
    }
    if (cfg->arith == SELENITE_ARITH_AUTO && cfg->nd_taps >= 2 && cfg->nh_taps >= 2 &&
        split16_template_nd((int)cfg->nd_taps, (int)cfg->decim, (int)cfg->nh_taps) > 0 && !diag_env("SELENITE_RX_NO_HIST_EXT")) {
        // k_ssb_split16 leaves the mixed samples in front of the decimator state here (two buffers: the one a channel's state points
        // at stays intact while the next call fills the other), for k_hist_exact
        // (round 4: allocated by the first call that needs it -- 2 x channels x ext_len x 8 bytes, 4 KB per channel for the cfg3 chain --
        // and released when the repair is switched off: ensure_hist_ext / selenite_rx_set_handover_repair)
        S->ext_len = cfg->decim * ((cfg->nh_taps - 1u + 3u) & ~3u);
    }
#undef INITCHK
    classify_coeffs(S);
    if (plan_fused(S->cfg, S->delay_is_impulse, S->delay_index, S->hilb_odd_only, S->plan) != hipSuccess) {
        int rc_ = fail(nullptr, SELENITE_RX_DEVICE_ERROR, "selenite_rx_init: building fused-kernel tables failed");
        free_device(S);
        delete S;
        return rc_;
    }
    int rc = reset_state(S);
    if (rc != SELENITE_RX_SUCCESS) { free_device(S); delete S; return rc; }
    *out = S;
    return SELENITE_RX_SUCCESS;
}

extern "C" void selenite_rx_free(selenite_rx_instance *S)
{
    if (!S) return;
    (void)hipSetDevice(S->device);
    if (S->stream) (void)hipStreamSynchronize(S->stream);
    free_device(S);
    delete S;
}

extern "C" int selenite_rx_set_mode(selenite_rx_instance *S, uint8_t mode)
{
    if (!S) return fail(nullptr, SELENITE_RX_ARGUMENT_ERROR, "selenite_rx_set_mode: S is NULL");
    if (!mode_valid(mode, S->cfg.nh_taps)) {
        g_last_error = "selenite_rx_set_mode: unsupported mode";
        return SELENITE_RX_ARGUMENT_ERROR;          // instance stays usable in its old mode
    }
    S->cfg.mode = mode;
    HIPCHK(S, plan_fused(S->cfg, S->delay_is_impulse, S->delay_index, S->hilb_odd_only, S->plan));
    return SELENITE_RX_SUCCESS;
}

extern "C" int selenite_rx_status(const selenite_rx_instance *S) { return S ? S->status : SELENITE_RX_ARGUMENT_ERROR; }
extern "C" const char *selenite_rx_error_string(const selenite_rx_instance *S)
{
    return S ? S->err.c_str() : g_last_error.c_str();
}
extern "C" const char *selenite_rx_kernel_name(const selenite_rx_instance *S)
{
    if (!S) return "";
    if (S->force_generic) return "generic";
    if (S->plan.kind != 0) return S->plan.name;
    if (cw_fused_ok(S->cfg, S->cfg.block)) {
        static thread_local std::string name;
        name = "k_cw_fused<" + std::to_string(S->cfg.n_biquad) + "," + std::to_string(S->cfg.block) + ">";
        return name.c_str();
    }
    return "generic";
}

// the shared LO repeats every 256 samples and the kernel of this instance can keep it in registers
static bool periodic_lo(const selenite_rx_instance *S)
{
    const selenite_rx_config &g = S->cfg;
    if (!(g.nco_enable && S->steps_uniform && (S->h_step[0] & 0x00FFFFFFu) == 0 && !S->no_periodic_lo)) return false;
    if ((g.arith == SELENITE_ARITH_SPLIT16 || g.arith == SELENITE_ARITH_AUTO) && S->plan.d_btab16 && g.nd_taps)
        return 256u % (g.block / g.decim) == 0 && ssb_split16_periodic_lo(split16_template_nd((int)g.nd_taps, (int)g.decim, (int)g.nh_taps), (int)g.decim, (int)g.nh_taps);
    if (g.arith == SELENITE_ARITH_AUTO) return false;      // (runs the bit-exact k_ssb_fused)
    // k_ssb_mfma (fma arithmetic, and split16 shapes without a matrix kernel of their own): decimation by 4, 1024-sample passes
    return g.arith != SELENITE_ARITH_CMSIS && S->plan.use_mfma && g.nd_taps && g.decim == 4;
}

// every channel has its own LO, each of them periodic in 256 samples (all steps multiples of 2^24), and the kernel that serves
// this instance's whole-pass calls computes one period per channel and call and keeps it in registers (NCO == 4 flavour of
// k_ssb_split16 and of k_ssb_fused; k_ssb_mfma and k_hilb_split16 have none: per-sample NCO there)
static bool periodic_lo_per_channel(const selenite_rx_instance *S)
{
    const selenite_rx_config &g = S->cfg;
    if (!(g.nco_enable && S->steps_grid256 && !S->no_periodic_lo && S->plan.kind != 0)) return false;
    if (256u % (g.block / g.decim) != 0) return false;                       // passes of 256 outputs only
    const bool split = (g.arith == SELENITE_ARITH_SPLIT16 || g.arith == SELENITE_ARITH_AUTO) && S->plan.d_btab16;
    if (split && g.nd_taps) return ssb_split16_periodic_lo(split16_template_nd((int)g.nd_taps, (int)g.decim, (int)g.nh_taps), (int)g.decim, (int)g.nh_taps);
    if (split) return false;                                                 // k_hilb_split16: per-sample NCO (and its AUTO rerun with it)
    const bool exact = g.arith == SELENITE_ARITH_CMSIS || g.arith == SELENITE_ARITH_AUTO;
    return exact || !(S->plan.use_mfma && g.decim == 4);                     // k_ssb_fused; the fma arithmetic by 4 runs k_ssb_mfma
}

extern "C" const char *selenite_rx_nco_path(const selenite_rx_instance *S)
{
    if (!S) return "";
    if (!S->cfg.nco_enable) return "off";
    const bool fused = !S->force_generic && (S->plan.kind != 0 || cw_fused_ok(S->cfg, S->cfg.block));
    if (!fused || !S->steps_uniform || !S->phase_uniform || S->no_shared_lo) {
        if (fused && S->plan.kind != 0 && periodic_lo_per_channel(S))
            return "per-channel LO, period 256 samples (arm_sin/cos_f32 once per channel and call), held in registers";
        return "per-channel arm_sin/cos_f32 in the kernel";
    }
    return periodic_lo(S) ? "shared LO, period 256 samples, held in registers" : "shared LO table per call";
}

extern "C" int selenite_rx_set_stream(selenite_rx_instance *S, void *hip_stream)
{
    if (!S) return SELENITE_RX_ARGUMENT_ERROR;
    hipStream_t next = hip_stream ? (hipStream_t)hip_stream : S->own_stream;
    if (next != S->stream) {
        // calls already queued on the old stream and calls on the new one share the streaming state (filter
        // histories, gains, phases, the LO table, scratch): drain the old stream before switching
        HIPCHK(S, hipSetDevice(S->device));
        HIPCHK(S, hipStreamSynchronize(S->stream));
        S->stream = next;
    }
    return SELENITE_RX_SUCCESS;
}

extern "C" int selenite_rx_sync(selenite_rx_instance *S)
{
    if (!S) return SELENITE_RX_ARGUMENT_ERROR;
    HIPCHK(S, hipStreamSynchronize(S->stream));
    return check_device_flags(S);
}

extern "C" int selenite_rx_set_guard_ratio(selenite_rx_instance *S, float ratio)
{
    if (!S || !(ratio >= 0.0f)) return SELENITE_RX_ARGUMENT_ERROR;      // (NaN rejected)
    S->guard_ratio = ratio;
    return SELENITE_RX_SUCCESS;
}

extern "C" int selenite_rx_guard_stats(selenite_rx_instance *S, uint64_t *guard_blocks, uint64_t *guard_channel_calls,
                                       uint64_t *rerun_channel_calls)
{
    if (!S) return SELENITE_RX_ARGUMENT_ERROR;
    HIPCHK(S, hipSetDevice(S->device));
    HIPCHK(S, hipStreamSynchronize(S->stream));
    // the kernels keep two words per channel (no atomics on shared counters); summed here, on the host
    const size_t C = S->cfg.channels;
    std::vector<uint32_t> w(2 * C);
    HIPCHK(S, hipMemcpy(w.data(), S->d_guard_ch, 2 * C * sizeof(uint32_t), hipMemcpyDeviceToHost));
    uint64_t blocks = 0, calls = 0;
    for (size_t c = 0; c < C; ++c) { blocks += w[c]; calls += w[C + c]; }
    if (guard_blocks) *guard_blocks = blocks;
    if (guard_channel_calls) *guard_channel_calls = calls;
    // SELENITE_ARITH_AUTO recomputes every guarded channel-call (the counts only ever come from the split-precision kernels)
    if (rerun_channel_calls) *rerun_channel_calls = S->cfg.arith == SELENITE_ARITH_AUTO ? calls : 0;
    return SELENITE_RX_SUCCESS;
}

extern "C" int selenite_rx_guard_channels(selenite_rx_instance *S, uint32_t *per_channel)
{
    if (!S || !per_channel) return SELENITE_RX_ARGUMENT_ERROR;
    HIPCHK(S, hipSetDevice(S->device));
    HIPCHK(S, hipStreamSynchronize(S->stream));
    HIPCHK(S, hipMemcpy(per_channel, S->d_guard_ch, (size_t)S->cfg.channels * sizeof(uint32_t), hipMemcpyDeviceToHost));
    return SELENITE_RX_SUCCESS;
}

// diagnostic (tests, tools): the per-channel words of SELENITE_ARITH_AUTO (rx_internal.h: rerun / provenance / hold bits, level)
extern "C" int selenite_rx_auto_words(selenite_rx_instance *S, uint32_t *per_channel)
{
    if (!S || !per_channel) return SELENITE_RX_ARGUMENT_ERROR;
    HIPCHK(S, hipSetDevice(S->device));
    HIPCHK(S, hipStreamSynchronize(S->stream));
    if (!S->d_rerun_flag) { std::memset(per_channel, 0, (size_t)S->cfg.channels * sizeof(uint32_t)); return SELENITE_RX_SUCCESS; }
    HIPCHK(S, hipMemcpy(per_channel, S->d_rerun_flag, (size_t)S->cfg.channels * sizeof(uint32_t), hipMemcpyDeviceToHost));
    return SELENITE_RX_SUCCESS;
}

extern "C" int selenite_rx_guard_clear(selenite_rx_instance *S)
{
    if (!S) return SELENITE_RX_ARGUMENT_ERROR;
    HIPCHK(S, hipSetDevice(S->device));
    HIPCHK(S, hipMemsetAsync(S->d_guard_ch, 0, 3 * (size_t)S->cfg.channels * sizeof(uint32_t), S->stream));
    return SELENITE_RX_SUCCESS;
}

extern "C" int selenite_rx_set_auto_launches(selenite_rx_instance *S, int launches)
{
    if (!S || !(launches == 1 || launches == 3)) return SELENITE_RX_ARGUMENT_ERROR;
    S->auto_launches = launches;
    return SELENITE_RX_SUCCESS;
}

extern "C" int selenite_rx_auto_launches_last(const selenite_rx_instance *S)
{
    return S ? (int)S->auto_form_last : 0;
}

extern "C" int selenite_rx_set_handover_repair(selenite_rx_instance *S, int on)
{
    if (!S) return SELENITE_RX_ARGUMENT_ERROR;
    S->handover_repair = on != 0;
    if (!S->handover_repair && S->d_hist_ext) {
        // the rows go (0.27 GB at 65 536 channels of the cfg3 chain); a state that pointed at them is "matrix kernel, no samples" from now on
        HIPCHK(S, hipSetDevice(S->device));
        HIPCHK(S, hipStreamSynchronize(S->stream));
        const size_t C = S->cfg.channels;
        std::vector<uint32_t> w(C);
        HIPCHK(S, hipMemcpy(w.data(), S->d_rerun_flag, C * sizeof(uint32_t), hipMemcpyDeviceToHost));
        for (size_t c = 0; c < C; ++c)
            if (((w[c] >> kProvShift) & kProvMask) == kProvSplitExt)
                w[c] = (w[c] & ~((kProvMask << kProvShift) | kExtQ15)) | (kProvSplit << kProvShift);
        HIPCHK(S, hipMemcpy(S->d_rerun_flag, w.data(), C * sizeof(uint32_t), hipMemcpyHostToDevice));
        HIPCHK(S, hipFree(S->d_hist_ext));
        S->d_hist_ext = nullptr;
    }
    return SELENITE_RX_SUCCESS;
}

// SELENITE_ARITH_AUTO on a shape with a split-precision decimator: the rows k_ssb_split16 leaves for k_hist_exact, allocated by the
// first call that can use them
static int ensure_hist_ext(selenite_rx_instance *S)
{
    if (S->d_hist_ext || !S->ext_len || !S->handover_repair || !S->d_rerun_flag) return SELENITE_RX_SUCCESS;
    const size_t n = 2 * (size_t)S->cfg.channels * S->ext_len;
    HIPCHK(S, hipMalloc((void **)&S->d_hist_ext, n * sizeof(float2)));
    HIPCHK(S, hipMemsetAsync(S->d_hist_ext, 0, n * sizeof(float2), S->stream));
    return SELENITE_RX_SUCCESS;
}

extern "C" int selenite_rx_guard_handover(selenite_rx_instance *S, uint64_t *handover_blocks)
{
    if (!S || !handover_blocks) return SELENITE_RX_ARGUMENT_ERROR;
    HIPCHK(S, hipSetDevice(S->device));
    HIPCHK(S, hipStreamSynchronize(S->stream));
    const size_t C = S->cfg.channels;
    std::vector<uint32_t> w(C);
    HIPCHK(S, hipMemcpy(w.data(), S->d_guard_ch + 2 * C, C * sizeof(uint32_t), hipMemcpyDeviceToHost));
    uint64_t n = 0;
    for (size_t c = 0; c < C; ++c) n += w[c];
    *handover_blocks = n;
    return SELENITE_RX_SUCCESS;
}

extern "C" int selenite_rx_reset(selenite_rx_instance *S)
{
    if (!S) return SELENITE_RX_ARGUMENT_ERROR;
    return reset_state(S);
}

// ------------------------------------------------------------------------------------------
static RxParams make_params(selenite_rx_instance *S, uint32_t block_size)
{
    const selenite_rx_config &g = S->cfg;
    RxParams p{};
    p.channels = g.channels; p.block = g.block; p.decim = g.decim;
    p.nd = g.nd_taps; p.nh = g.nh_taps; p.nbiq = g.n_biquad; p.mode = g.mode; p.q15_round = g.q15_rounding ? 1u : 0u;
    p.nco = g.nco_enable ? 1 : 0; p.agc = g.agc_enable ? 1 : 0;
    p.block_size = block_size; p.nout = block_size / g.decim;
    p.in_stride = p.block_size; p.out_stride = p.nout;
    p.dec_c = S->d_dec_c; p.hilb_c = S->d_hilb_c; p.delay_c = S->d_delay_c; p.biq_c = S->d_biq_c;
    p.sintab = S->d_sintab; p.step = S->d_step; p.phase = S->d_phase;
    p.dec_state = S->d_dec_state; p.fir_state = S->d_fir_state; p.biq_state = S->d_biq_state;
    p.gain = S->d_gain;
    p.flags = S->d_flags;
    p.guard_ratio = S->guard_ratio;
    p.guard_ch = S->d_guard_ch;
    p.guard_calls = S->d_guard_ch + g.channels;
    p.guard_hand = S->d_guard_ch + 2 * (size_t)g.channels;
    p.hist_ext = S->handover_repair ? S->d_hist_ext : nullptr; p.ext_len = S->ext_len; p.ext_buf_stride = (size_t)g.channels * S->ext_len;
    if (S->sub_count) {                                     // a contiguous channel range of the instance: every per-channel array moves with it
        const size_t c0 = S->sub_first;
        p.channels = S->sub_count;
        p.step += c0; p.phase += c0; p.gain += c0; p.guard_ch += c0; p.guard_calls += c0; p.guard_hand += c0;
        if (p.hist_ext) p.hist_ext += c0 * p.ext_len;
        if (p.dec_state) p.dec_state += c0 * 2 * (g.nd_taps - 1);
        if (p.fir_state) p.fir_state += c0 * 2 * (g.nh_taps - 1);
        if (p.biq_state) p.biq_state += c0 * 4 * g.n_biquad;
    }
    p.agcp = AgcParams{ g.agc_target, g.agc_attack, g.agc_decay, g.agc_gain_min, g.agc_gain_max, g.agc_env_floor };
    // generic front kernel: largest pass (<= 256 outputs) whose LDS image fits 64 KiB
    uint32_t P = 256;
    for (;;) {
        p.pass_out = P;
        if (front_generic_lds_bytes(p) <= 64 * 1024 || P == 1) break;
        P >>= 1;
    }
    return p;
}

static int ensure(selenite_rx_instance *S, void **buf, size_t *cap, size_t need)
{
    if (*cap >= need) return SELENITE_RX_SUCCESS;
    if (*buf) { HIPCHK(S, hipStreamSynchronize(S->stream)); HIPCHK(S, hipFree(*buf)); *buf = nullptr; *cap = 0; }
    HIPCHK(S, hipMalloc(buf, need));
    *cap = need;
    return SELENITE_RX_SUCCESS;
}

static bool block_size_ok(selenite_rx_instance *S, uint32_t block_size, const char *who)
{
    if (block_size == 0 || block_size % S->cfg.block != 0) {
        fail(S, SELENITE_RX_LENGTH_ERROR, std::string(who) + ": blockSize is not a non-zero multiple of cfg.block");
        return false;
    }
    return true;
}

enum Phase { kAll, kPhase1, kPhase2 };

// The one dispatcher behind every process entry point.
static int run_part(selenite_rx_instance *S, const void *src, bool src_q15, void *dst, bool dst_q15,
                    uint32_t block_size, Phase phase, float *ext_env, uint32_t in_stride, uint32_t out_stride)
{
    const selenite_rx_config &g = S->cfg;
    HIPCHK(S, hipSetDevice(S->device));
    if (phase != kPhase2 && S->plan.kind != 0 && S->plan.d_btab16 && !S->force_generic)
        if (int rc = ensure_hist_ext(S)) return rc;
    RxParams p = make_params(S, block_size);
    p.in_stride = in_stride; p.out_stride = out_stride;
    if (front_generic_lds_bytes(p) > 64 * 1024 && (S->force_generic || S->plan.kind == 0))
        return fail(S, SELENITE_RX_LENGTH_ERROR, "filter lengths exceed the LDS budget of the generic kernel");
    const int arith = (int)g.arith;
    const int garith = arith == SELENITE_ARITH_AUTO ? SELENITE_ARITH_CMSIS : arith;      // the generic kernels: AUTO is bit-exact there
    const bool global = g.agc_enable && g.agc_global;
    const bool cw = mode_is_cw(g.mode) && g.n_biquad;
    hipStream_t st = S->stream;

    // host copy of the common NCO phase (valid while every channel shares step and phase)
    // (advanced by commit_phase() once the kernels of the call are enqueued: a call that fails before that leaves
    // the host copy in step with d_phase)
    const uint32_t phase_now = S->phase_host;
    auto commit_phase = [&]() { if (phase != kPhase2 && g.nco_enable) S->phase_host = phase_now + block_size * S->h_step[0]; };

    // Fused kernels serve the global-gain variant too: they run with their own AGC off (un-scaled
    // audio out), then the envelope reduction and the gain pass below finish the call.
    // (the fused kernels convert in and out symmetrically, and a global gain needs f32 audio between its two phases: int16 slots with
    // a global gain get their input converted once, up front -- arm_q15_to_float over the whole buffer, the very operation the fused
    // int16 load performs -- and run as an f32-input call whose gain pass stores int16; round 2 left them to the generic kernels)
    const bool fusable = phase != kPhase2 && !S->force_generic;
    const bool ssb_fused = fusable && S->plan.kind != 0;
    const bool cw_fused = fusable && cw_fused_ok(g, block_size) && cw_strides_ok(p.in_stride, p.out_stride);    // (wider strides: the generic kernels)
    if (global && src_q15 && (ssb_fused || cw_fused)) {
        const size_t nval = (size_t)p.channels * p.in_stride * 2;            // int16 values of the call (block_size % 4 == 0 for every fused shape)
        if (nval % 8 == 0) {
            int rc = ensure(S, (void **)&S->d_conv_in, &S->conv_in_bytes, nval * sizeof(float));
            if (rc) return rc;
            HIPCHK(S, launch_q15_to_f32(static_cast<const int16_t *>(src), S->d_conv_in, nval, S->stream));
            src = S->d_conv_in;
            src_q15 = false;
        }
    }
    float *audio = (float *)dst;      // un-scaled audio: dst itself when dst is f32, else scratch
    if (dst_q15 && (global || !(ssb_fused || cw_fused))) {
        const size_t need = (size_t)g.channels * p.out_stride * sizeof(float);
        int rc = ensure(S, (void **)&S->d_scratch, &S->scratch_bytes, need);
        if (rc) return rc;
        audio = S->d_scratch;
    }
    // (SELENITE_ARITH_AUTO outside the SSB fused kernels -- CW, generic: every channel's state stays in exact arithmetic, and the
    // provenance words k_ssb_split16 reads at its next call say so)
    // (a channel the matrix kernel left with its samples gets its Hilbert-pair history recomputed in exact arithmetic first: the
    // generic / CW kernels read it -- advisor finding, round 3)
    if (phase != kPhase2 && S->d_rerun_flag && !ssb_fused) {
        uint32_t *words = S->d_rerun_flag + (S->sub_count ? S->sub_first : 0u);
        if (p.hist_ext) {
            RxParams ph = p;
            ph.chan_flags = words;
            HIPCHK(S, launch_hist_exact(ph, true, st));
        }
        HIPCHK(S, hipMemsetAsync(words, 0, p.channels * sizeof(uint32_t), st));
    }
    bool env_emitted = false;      // global gain: the fused kernel wrote the per-channel block maxima
    if (ssb_fused || cw_fused) {
        RxParams pf = p;
        if (g.nco_enable && S->steps_uniform && S->phase_uniform && !S->no_shared_lo) {
            // one LO for all channels: computed once per call, read from L2 by every wavefront
            // the table is a pure function of (start phase, step, length): a call that starts where the table in d_lo
            // starts reuses it -- every chunk of a pipelined host call, and EVERY call when the phase advance of a call
            // is a multiple of 2^32 (an LO on the fs / 256 grid with calls of whole DSP blocks)
            // (at least one whole period: the register-resident flavour reads LO[0 .. 255] whatever the call length)
            const uint32_t lo_n = block_size < 256u ? 256u : block_size;
            if (!(S->lo_valid && S->lo_phase == phase_now && S->lo_step == S->h_step[0] && S->lo_n >= lo_n)) {
                S->lo_valid = false;
                int rc = ensure(S, (void **)&S->d_lo, &S->lo_bytes, (size_t)lo_n * sizeof(float2));
                if (rc) return rc;
                HIPCHK(S, launch_lo_table(S->d_lo, S->d_sintab, phase_now, S->h_step[0], lo_n, st));
                S->lo_valid = true; S->lo_phase = phase_now; S->lo_step = S->h_step[0]; S->lo_n = lo_n;
            }
            pf.nco = 2;
            pf.lo = S->d_lo;
            // a step that is a multiple of 2^24 repeats the LO every 256 samples (channelised receivers: LO
            // frequencies on a grid of fs / 256): k_ssb_split16 then keeps it in registers (its NCO == 3 flavour)
            pf.lo_period = periodic_lo(S) ? 256u : 0u;
        } else if (ssb_fused && g.nco_enable && periodic_lo_per_channel(S)) {
            pf.lo_period = 256u;                          // pf.nco stays 1: every channel computes its own period once
        }
        if (arith == SELENITE_ARITH_AUTO && ssb_fused) {
            // the split16 kernel raises the rerun flag of the channels it guards and leaves their state alone; the bit-exact
            // kernel then recomputes the flagged channels (launch_fused)
            pf.rerun_flag = S->d_rerun_flag + (S->sub_count ? S->sub_first : 0u);
            pf.chan_list = S->d_rerun_list + 2;
            pf.chan_count = S->d_rerun_list;              // the two counters; launch_shape picks by *rerun_par_host where it launches the prepare kernel
            pf.rerun_par_host = &S->rerun_par;
            pf.rerun_seen = S->h_rerun_seen;
            pf.auto_inline = S->auto_launches == 1 ? 1u : 0u;
            pf.form_host = &S->auto_form_last;
        }
        void *fdst = dst;
        bool fq15 = dst_q15;
        if (global) {
            pf.agc = 0; fdst = audio; fq15 = false;
            pf.out_cached = 1;                            // phase 2 (and, without block maxima from the kernel, the envelope fold) reads this audio back
            // k_ssb_split16 (16-lane DSP blocks, whole passes) leaves the block maxima of every channel behind: the
            // envelope reduction below then folds channels x blocks floats instead of reading the audio again
            if (ssb_fused && (arith == SELENITE_ARITH_SPLIT16 || arith == SELENITE_ARITH_AUTO) && S->plan.d_btab16 && g.nd_taps && g.decim == 4 && (g.block / g.decim) / 4 == 16 &&
                (block_size / g.decim) % 256 == 0 && g.nco_enable && g.mode != SELENITE_MODE_AM && g.mode != SELENITE_MODE_FM) {   // the launches with the DPP block reductions (decimation by 4, 64-sample audio blocks)
                const size_t need = sizeof(float) * env_fold_scratch_floats(p.channels, block_size / g.block);
                int rc = ensure(S, (void **)&S->d_env_part, &S->env_part_cap, need);
                if (rc) return rc;
                pf.env_part = S->d_env_part;
                env_emitted = true;
            }
        }
        if (ssb_fused) HIPCHK(S, launch_fused(S->plan, pf, arith, src, src_q15, fdst, fq15, S->delay_index, st));
        else HIPCHK(S, launch_cw_fused(pf, src, src_q15, fdst, fq15, st));
        commit_phase();
        if (!global) return SELENITE_RX_SUCCESS;
    }

    // generic path: front -> [biquad] -> AGC / convert
    if (phase != kPhase2 && !(ssb_fused || cw_fused)) {
        HIPCHK(S, launch_front_generic(p, garith, src, src_q15, audio, st));
        commit_phase();
        if (cw) HIPCHK(S, launch_biquad_generic(p, garith, audio, st));
    }
    if (global) {
        float *env = ext_env;
        if (!env) {
            const size_t need = sizeof(float) * (block_size / g.block);
            int rc = ensure(S, (void **)&S->d_env, &S->env_cap, need);
            if (rc) return rc;
            env = S->d_env;
        }
        if (phase != kPhase2) {
            if (env_emitted) {
                HIPCHK(S, launch_env_fold(S->d_env_part, env, p.channels, block_size / g.block, st));
            } else {
                const size_t need = sizeof(float) * env_global_rows(p) * (block_size / g.block);
                int rc = ensure(S, (void **)&S->d_env_part, &S->env_part_cap, need);
                if (rc) return rc;
                HIPCHK(S, launch_env_global(p, audio, S->d_env_part, env, st));
            }
        }
        if (phase != kPhase1) HIPCHK(S, launch_agc_apply_global(p, garith, audio, env, dst, dst_q15, st));
    } else if (g.agc_enable || dst_q15) {
        HIPCHK(S, launch_agc_generic(p, garith, audio, dst, dst_q15, st));
    }
    return SELENITE_RX_SUCCESS;
}

// Entry of every process call.  Any call length (a whole number of DSP blocks) runs on the fused kernels; a
// split-precision call that ends in a partial pass too short for the matrix kernel is cut in two launches on the same
// streaming state (fused_tail_split; both parts address the caller's buffers with the full per-channel stride; in
// SELENITE_ARITH_AUTO the tail runs in the bit-exact arithmetic, rx_fused.hip launch_shape).
static int run_chain(selenite_rx_instance *S, const void *src, bool src_q15, void *dst, bool dst_q15,
                     uint32_t block_size, Phase phase, float *ext_env)
{
    const selenite_rx_config &g = S->cfg;
    const uint32_t nout = block_size / g.decim;
    const bool global = g.agc_enable && g.agc_global;
    if (phase == kAll && !global && !S->force_generic && S->plan.kind != 0 && fused_tail_split(S->plan, g, block_size)) {
        const uint32_t unit = split16_pass_out(g.block, g.decim) * g.decim, bs1 = block_size / unit * unit;
        {
            int rc = run_part(S, src, src_q15, dst, dst_q15, bs1, kAll, nullptr, block_size, nout);
            if (rc) return rc;
            const size_t ein = src_q15 ? sizeof(int16_t) : sizeof(float), eout = dst_q15 ? sizeof(int16_t) : sizeof(float);
            const char *src2 = static_cast<const char *>(src) + (size_t)bs1 * 2 * ein;
            char *dst2 = static_cast<char *>(dst) + (size_t)(bs1 / g.decim) * eout;
            return run_part(S, src2, src_q15, dst2, dst_q15, block_size - bs1, kAll, nullptr, block_size, nout);
        }
    }
    return run_part(S, src, src_q15, dst, dst_q15, block_size, phase, ext_env, block_size, nout);
}

extern "C" void selenite_rx_process_f32_device(selenite_rx_instance *S, const float *dSrcIQ,
                                               float *dDstAudio, uint32_t blockSize)
{
    if (!S || !block_size_ok(S, blockSize, "selenite_rx_process_f32_device")) return;
    run_chain(S, dSrcIQ, false, dDstAudio, false, blockSize, kAll, nullptr);
}

extern "C" void selenite_rx_process_q15_device(selenite_rx_instance *S, const int16_t *dSrcIQ,
                                               int16_t *dDstAudio, uint32_t blockSize)
{
    if (!S || !block_size_ok(S, blockSize, "selenite_rx_process_q15_device")) return;
    run_chain(S, dSrcIQ, true, dDstAudio, true, blockSize, kAll, nullptr);
}

extern "C" void selenite_rx_global_phase1_device(selenite_rx_instance *S, const float *dSrcIQ,
                                                 float *dDstAudio, float *dEnv, uint32_t blockSize)
{
    if (!S || !block_size_ok(S, blockSize, "selenite_rx_global_phase1_device")) return;
    if (!(S->cfg.agc_enable && S->cfg.agc_global)) {
        fail(S, SELENITE_RX_ARGUMENT_ERROR, "selenite_rx_global_phase1_device: instance is not agc_global");
        return;
    }
    run_chain(S, dSrcIQ, false, dDstAudio, false, blockSize, kPhase1, dEnv);
}

extern "C" void selenite_rx_global_phase2_device(selenite_rx_instance *S, float *dDstAudio,
                                                 const float *dEnv, uint32_t blockSize)
{
    if (!S || !block_size_ok(S, blockSize, "selenite_rx_global_phase2_device")) return;
    if (!(S->cfg.agc_enable && S->cfg.agc_global)) {
        fail(S, SELENITE_RX_ARGUMENT_ERROR, "selenite_rx_global_phase2_device: instance is not agc_global");
        return;
    }
    run_chain(S, nullptr, false, dDstAudio, false, blockSize, kPhase2, const_cast<float *>(dEnv));
}

// ---- global-gain call with the exchange done HERE, for a plain C host: phase 1, ncclAllReduce(MAX) of the
// per-block envelopes over RCCL / xGMI, phase 2 -- all on the instance's stream.  RCCL is bound at run time
// (the process's already loaded librccl -- e.g. the one torch ships -- or librccl.so.1), so the library carries no
// link-time dependency on it and a host that never uses global gain never loads it.
typedef int (*nccl_allreduce_fn)(const void *, void *, size_t, int, int, void *, hipStream_t);
static nccl_allreduce_fn rccl_allreduce()
{
    static nccl_allreduce_fn fn = [] {
        void *sym = dlsym(RTLD_DEFAULT, "ncclAllReduce");
        if (!sym) {
            void *h = dlopen("librccl.so.1", RTLD_NOW | RTLD_GLOBAL);
            if (!h) h = dlopen("librccl.so", RTLD_NOW | RTLD_GLOBAL);
            if (h) sym = dlsym(h, "ncclAllReduce");
        }
        return reinterpret_cast<nccl_allreduce_fn>(sym);
    }();
    return fn;
}

extern "C" int selenite_rx_global_process_f32_device(selenite_rx_instance *S, const float *dSrcIQ, float *dDstAudio,
                                                     uint32_t blockSize, void *rccl_comm)
{
    if (!S || !block_size_ok(S, blockSize, "selenite_rx_global_process_f32_device")) return S ? S->status : SELENITE_RX_ARGUMENT_ERROR;
    if (!(S->cfg.agc_enable && S->cfg.agc_global))
        return fail(S, SELENITE_RX_ARGUMENT_ERROR, "selenite_rx_global_process_f32_device: instance is not agc_global");
    const size_t nblk = blockSize / S->cfg.block;
    int rc = ensure(S, (void **)&S->d_env, &S->env_cap, sizeof(float) * nblk);
    if (rc) return rc;
    rc = run_chain(S, dSrcIQ, false, dDstAudio, false, blockSize, kPhase1, S->d_env);
    if (rc) return rc;
    if (rccl_comm) {                                        // NULL: single rank, nothing to exchange
        nccl_allreduce_fn ar = rccl_allreduce();
        if (!ar) return fail(S, SELENITE_RX_DEVICE_ERROR, "selenite_rx_global_process_f32_device: RCCL (ncclAllReduce) is not available");
        const int nccl_float = 7, nccl_max = 2;             // ncclFloat32, ncclMax (rccl.h)
        const int e = ar(S->d_env, S->d_env, nblk, nccl_float, nccl_max, rccl_comm, S->stream);
        if (e != 0) return fail(S, SELENITE_RX_DEVICE_ERROR, "selenite_rx_global_process_f32_device: ncclAllReduce failed (" + std::to_string(e) + ")");
    }
    return run_chain(S, nullptr, false, dDstAudio, false, blockSize, kPhase2, S->d_env);
}

// ---- host-pointer entry points: the literal drop-in signature (float* / int16_t* I/Q in, audio out) ----
//
// Channels are independent, so a call over host buffers is cut into channel chunks and pipelined: chunk k+1 crosses
// PCIe (H2D stream) while chunk k computes (the instance's stream) and chunk k-1 returns (D2H stream), two device
// buffers each way, ordered by events only.  Caller memory that is page-locked (selenite_rx_host_alloc /
// selenite_rx_host_register, or any hipHostMalloc / hipHostRegister memory) is the DMA source and target itself;
// pageable caller memory goes through the library's own pinned staging buffers, filled and drained by a few host
// threads (a pageable hipMemcpy is bounced by the driver at ~10 GB/s on this stack).  No allocation per call once
// the buffers have grown to the call's chunk size.  The global-gain variant needs every channel's envelope before
// any gain and stays one chunk.
static bool host_ptr_is_pinned(const void *p)
{
    hipPointerAttribute_t a;
    if (hipPointerGetAttributes(&a, p) != hipSuccess) { (void)hipGetLastError(); return false; }   // plain malloc memory: "invalid value"
    return a.type == hipMemoryTypeHost;
}

static void parallel_memcpy(void *dst, const void *src, size_t bytes)
{
    const unsigned hw = std::thread::hardware_concurrency();
    const size_t nt = bytes < (8u << 20) ? 1 : std::min<size_t>(8, hw ? hw : 1);
    if (nt <= 1) { std::memcpy(dst, src, bytes); return; }
    std::vector<std::thread> th;
    const size_t per = ((bytes + nt - 1) / nt + 4095) & ~(size_t)4095;
    for (size_t i = 0; i < nt; ++i) {
        const size_t off = i * per;
        if (off >= bytes) break;
        const size_t n = std::min(per, bytes - off);
        th.emplace_back([=] { std::memcpy(static_cast<char *>(dst) + off, static_cast<const char *>(src) + off, n); });
    }
    for (auto &t : th) t.join();
}

static int pipe_setup(selenite_rx_instance *S, size_t in_bytes, size_t out_bytes, bool stage_in, bool stage_out)
{
    auto &P = S->pipe;
    if (!P.h2d) {
        HIPCHK(S, hipStreamCreateWithFlags(&P.h2d, hipStreamNonBlocking));
        HIPCHK(S, hipStreamCreateWithFlags(&P.d2h, hipStreamNonBlocking));
        for (int i = 0; i < 2; ++i) {
            HIPCHK(S, hipEventCreateWithFlags(&P.ev_in[i], hipEventDisableTiming));
            HIPCHK(S, hipEventCreateWithFlags(&P.ev_done[i], hipEventDisableTiming));
            HIPCHK(S, hipEventCreateWithFlags(&P.ev_out[i], hipEventDisableTiming));
        }
    }
    auto grow_dev = [&](void *(&b)[2], size_t &cap, size_t need) -> int {
        if (cap >= need) return 0;
        HIPCHK(S, hipDeviceSynchronize());
        for (int i = 0; i < 2; ++i) { if (b[i]) HIPCHK(S, hipFree(b[i])); b[i] = nullptr; HIPCHK(S, hipMalloc(&b[i], need)); }
        cap = need;
        return 0;
    };
    auto grow_host = [&](void *(&b)[2], size_t &cap, size_t need) -> int {
        if (cap >= need) return 0;
        HIPCHK(S, hipDeviceSynchronize());
        for (int i = 0; i < 2; ++i) { if (b[i]) HIPCHK(S, hipHostFree(b[i])); b[i] = nullptr; HIPCHK(S, hipHostMalloc(&b[i], need, hipHostMallocDefault)); }
        cap = need;
        return 0;
    };
    if (grow_dev(P.d_in, P.d_in_bytes, in_bytes) || grow_dev(P.d_out, P.d_out_bytes, out_bytes)) return S->status;
    if (stage_in && grow_host(P.h_in, P.h_in_bytes, in_bytes)) return S->status;
    if (stage_out && grow_host(P.h_out, P.h_out_bytes, out_bytes)) return S->status;
    return SELENITE_RX_SUCCESS;
}

static void process_host(selenite_rx_instance *S, const void *src, void *dst, uint32_t block_size, bool q15,
                         const char *who)
{
    if (!S || !block_size_ok(S, block_size, who)) return;
    const selenite_rx_config &g = S->cfg;
    const size_t esz = q15 ? sizeof(int16_t) : sizeof(float);
    const size_t in_ch = (size_t)block_size * 2 * esz, out_ch = (size_t)(block_size / g.decim) * esz;   // bytes per channel
    if (hipSetDevice(S->device) != hipSuccess) { fail(S, SELENITE_RX_DEVICE_ERROR, "hipSetDevice"); return; }

    if (g.agc_enable && g.agc_global) {                     // one chunk: every envelope before any gain
        const size_t nin = g.channels * in_ch, nout = g.channels * out_ch;
        if (ensure(S, &S->d_io_in, &S->io_in_bytes, nin)) return;
        if (ensure(S, &S->d_io_out, &S->io_out_bytes, nout)) return;
        if (hipMemcpyAsync(S->d_io_in, src, nin, hipMemcpyHostToDevice, S->stream) != hipSuccess) { fail(S, SELENITE_RX_DEVICE_ERROR, "H2D copy failed"); return; }
        if (run_chain(S, S->d_io_in, q15, S->d_io_out, q15, block_size, kAll, nullptr)) return;
        if (hipMemcpyAsync(dst, S->d_io_out, nout, hipMemcpyDeviceToHost, S->stream) != hipSuccess) { fail(S, SELENITE_RX_DEVICE_ERROR, "D2H copy failed"); return; }
        if (hipStreamSynchronize(S->stream) != hipSuccess) fail(S, SELENITE_RX_DEVICE_ERROR, "stream sync failed");
        return;
    }

    // chunk: about 32 MiB of input (16 MiB measured slower with pageable callers: the staging copies are threads spawned per chunk), at least 64 channels (a few waves per CU would starve the kernels), at most all
    static const size_t chunk_bytes = [] { const char *e = std::getenv("SELENITE_RX_HOST_CHUNK_MB"); return (size_t)(e && std::atoi(e) > 0 ? std::atoi(e) : 32) << 20; }();
    uint32_t cch = (uint32_t)std::max<size_t>(64, chunk_bytes / in_ch);
    cch = std::min<uint32_t>(cch, g.channels);
    const uint32_t nchunk = (g.channels + cch - 1) / cch;
    const bool stage_in = !host_ptr_is_pinned(src), stage_out = !host_ptr_is_pinned(dst);
    if (pipe_setup(S, cch * in_ch, cch * out_ch, stage_in, stage_out)) return;
    auto &P = S->pipe;
    const uint32_t phase0 = S->phase_host;                  // the shared LO of a call is one table: every chunk starts from the same phase
    uint32_t phase_end = phase0;
    const char *hs = static_cast<const char *>(src);
    char *hd = static_cast<char *>(dst);
    bool ok = true;
    auto chk = [&](hipError_t e, const char *what) { if (ok && e != hipSuccess) { fail(S, SELENITE_RX_DEVICE_ERROR, std::string(who) + ": " + what + ": " + hipGetErrorString(e)); ok = false; } };
    auto drain_out = [&](uint32_t k) {                      // pageable destination: chunk k from pinned staging to the caller
        const uint32_t c0 = k * cch, n = std::min(cch, g.channels - c0);
        chk(hipEventSynchronize(P.ev_out[k & 1]), "D2H wait");
        if (ok) parallel_memcpy(hd + (size_t)c0 * out_ch, P.h_out[k & 1], (size_t)n * out_ch);
    };
    for (uint32_t k = 0; k < nchunk && ok; ++k) {
        const int s = (int)(k & 1);
        const uint32_t c0 = k * cch, n = std::min(cch, g.channels - c0);
        const void *from = hs + (size_t)c0 * in_ch;
        if (stage_in) {
            if (k >= 2) chk(hipEventSynchronize(P.ev_in[s]), "staging wait");             // H2D of chunk k-2 left this staging buffer
            if (ok) parallel_memcpy(P.h_in[s], from, (size_t)n * in_ch);
            from = P.h_in[s];
        }
        if (k >= 2) chk(hipStreamWaitEvent(P.h2d, P.ev_done[s], 0), "H2D order");            // kernels of chunk k-2 have read d_in[s]
        chk(hipMemcpyAsync(P.d_in[s], from, (size_t)n * in_ch, hipMemcpyHostToDevice, P.h2d), "H2D copy");
        chk(hipEventRecord(P.ev_in[s], P.h2d), "event");
        chk(hipStreamWaitEvent(S->stream, P.ev_in[s], 0), "compute order");
        if (k >= 2) chk(hipStreamWaitEvent(S->stream, P.ev_out[s], 0), "compute order");     // D2H of chunk k-2 has read d_out[s]
        if (!ok) break;
        S->sub_first = c0; S->sub_count = n;
        S->phase_host = phase0;
        const int rc = run_chain(S, P.d_in[s], q15, P.d_out[s], q15, block_size, kAll, nullptr);
        phase_end = S->phase_host;
        S->sub_first = 0; S->sub_count = 0;
        if (rc) { ok = false; break; }
        chk(hipEventRecord(P.ev_done[s], S->stream), "event");
        chk(hipStreamWaitEvent(P.d2h, P.ev_done[s], 0), "D2H order");
        if (stage_out && k >= 2) drain_out(k - 2);                                           // frees h_out[s] before it is the D2H target again
        chk(hipMemcpyAsync(stage_out ? P.h_out[s] : (void *)(hd + (size_t)c0 * out_ch), P.d_out[s], (size_t)n * out_ch,
                           hipMemcpyDeviceToHost, P.d2h), "D2H copy");
        chk(hipEventRecord(P.ev_out[s], P.d2h), "event");
    }
    S->phase_host = ok ? phase_end : phase0;
    if (stage_out && ok) {
        if (nchunk >= 2) drain_out(nchunk - 2);
        drain_out(nchunk - 1);
    }
    chk(hipStreamSynchronize(P.h2d), "sync");
    chk(hipStreamSynchronize(S->stream), "sync");
    chk(hipStreamSynchronize(P.d2h), "sync");
    if (ok) (void)check_device_flags(S);
}

extern "C" void *selenite_rx_host_alloc(size_t bytes)
{
    void *p = nullptr;
    if (hipHostMalloc(&p, bytes ? bytes : 1, hipHostMallocDefault) != hipSuccess) { g_last_error = "hipHostMalloc failed"; return nullptr; }
    return p;
}
extern "C" void selenite_rx_host_free(void *hptr) { if (hptr) (void)hipHostFree(hptr); }
extern "C" int selenite_rx_host_register(void *hptr, size_t bytes)
{
    HIPCHK(nullptr, hipHostRegister(hptr, bytes, hipHostRegisterDefault));
    return SELENITE_RX_SUCCESS;
}
extern "C" int selenite_rx_host_unregister(void *hptr)
{
    HIPCHK(nullptr, hipHostUnregister(hptr));
    return SELENITE_RX_SUCCESS;
}

extern "C" void selenite_rx_process_f32(selenite_rx_instance *S, const float *pSrcIQ, float *pDstAudio,
                                        uint32_t blockSize)
{
    process_host(S, pSrcIQ, pDstAudio, blockSize, false, "selenite_rx_process_f32");
}
extern "C" void selenite_rx_process_q15(selenite_rx_instance *S, const int16_t *pSrcIQ, int16_t *pDstAudio,
                                        uint32_t blockSize)
{
    process_host(S, pSrcIQ, pDstAudio, blockSize, true, "selenite_rx_process_q15");
}

// ------------------------------------------------------------------------------------------
extern "C" int selenite_rx_get_state(selenite_rx_instance *S, const selenite_rx_state_view *v)
{
    if (!S || !v) return SELENITE_RX_ARGUMENT_ERROR;
    const selenite_rx_config &g = S->cfg;
    const size_t C = g.channels;
    HIPCHK(S, hipSetDevice(S->device));
    HIPCHK(S, hipStreamSynchronize(S->stream));
    if (v->dec_state && g.nd_taps > 1)
        HIPCHK(S, hipMemcpy(v->dec_state, S->d_dec_state, C * 2 * (g.nd_taps - 1) * sizeof(float), hipMemcpyDeviceToHost));
    if (v->fir_state && g.nh_taps > 1)
        HIPCHK(S, hipMemcpy(v->fir_state, S->d_fir_state, C * 2 * (g.nh_taps - 1) * sizeof(float), hipMemcpyDeviceToHost));
    if (v->biq_state && g.n_biquad)
        HIPCHK(S, hipMemcpy(v->biq_state, S->d_biq_state, C * 4 * g.n_biquad * sizeof(float), hipMemcpyDeviceToHost));
    if (v->agc_gain) HIPCHK(S, hipMemcpy(v->agc_gain, S->d_gain, C * sizeof(float), hipMemcpyDeviceToHost));
    if (v->nco_phase) HIPCHK(S, hipMemcpy(v->nco_phase, S->d_phase, C * sizeof(uint32_t), hipMemcpyDeviceToHost));
    return SELENITE_RX_SUCCESS;
}

extern "C" int selenite_rx_set_state(selenite_rx_instance *S, const selenite_rx_state_view *v)
{
    if (!S || !v) return SELENITE_RX_ARGUMENT_ERROR;
    const selenite_rx_config &g = S->cfg;
    const size_t C = g.channels;
    HIPCHK(S, hipSetDevice(S->device));
    HIPCHK(S, hipStreamSynchronize(S->stream));
    if (v->dec_state && g.nd_taps > 1)
        HIPCHK(S, hipMemcpy(S->d_dec_state, v->dec_state, C * 2 * (g.nd_taps - 1) * sizeof(float), hipMemcpyHostToDevice));
    if (v->fir_state && g.nh_taps > 1)
        HIPCHK(S, hipMemcpy(S->d_fir_state, v->fir_state, C * 2 * (g.nh_taps - 1) * sizeof(float), hipMemcpyHostToDevice));
    if (v->biq_state && g.n_biquad)
        HIPCHK(S, hipMemcpy(S->d_biq_state, v->biq_state, C * 4 * g.n_biquad * sizeof(float), hipMemcpyHostToDevice));
    if (v->agc_gain) HIPCHK(S, hipMemcpy(S->d_gain, v->agc_gain, C * sizeof(float), hipMemcpyHostToDevice));
    if (v->nco_phase) {
        HIPCHK(S, hipMemcpy(S->d_phase, v->nco_phase, C * sizeof(uint32_t), hipMemcpyHostToDevice));
        S->phase_uniform = true;
        for (size_t c = 1; c < C; ++c) S->phase_uniform = S->phase_uniform && v->nco_phase[c] == v->nco_phase[0];
        S->phase_host = v->nco_phase[0];
    }
    if (S->d_rerun_flag) HIPCHK(S, hipMemsetAsync(S->d_rerun_flag, 0, C * sizeof(uint32_t), S->stream));      // a given state counts as exact
    return SELENITE_RX_SUCCESS;
}

extern "C" void *selenite_rx_device_alloc(size_t bytes)
{
    void *p = nullptr;
    if (hipMalloc(&p, bytes ? bytes : 1) != hipSuccess) { g_last_error = "hipMalloc failed"; return nullptr; }
    return p;
}
extern "C" void selenite_rx_device_free(void *dptr) { if (dptr) (void)hipFree(dptr); }
extern "C" int selenite_rx_memcpy_h2d(void *dptr, const void *hptr, size_t bytes)
{
    HIPCHK(nullptr, hipMemcpy(dptr, hptr, bytes, hipMemcpyHostToDevice));
    return SELENITE_RX_SUCCESS;
}
extern "C" int selenite_rx_memcpy_d2h(void *hptr, const void *dptr, size_t bytes)
{
    HIPCHK(nullptr, hipMemcpy(hptr, dptr, bytes, hipMemcpyDeviceToHost));
    return SELENITE_RX_SUCCESS;
}

// ------------------------------------------------------------------------------------------
extern "C" void selenite_rx_synth_iq_host(float *iq, uint32_t first_channel, uint32_t nch,
                                          uint64_t first_sample, uint32_t nsamp, uint64_t seed)
{
    synth_host(iq, host_sin_table(), first_channel, nch, first_sample, nsamp, seed);
}

extern "C" int selenite_rx_synth_iq_device(selenite_rx_instance *S, float *dIQ, uint32_t first_channel,
                                           uint32_t nch, uint64_t first_sample, uint32_t nsamp, uint64_t seed)
{
    if (!S) return SELENITE_RX_ARGUMENT_ERROR;
    HIPCHK(S, hipSetDevice(S->device));
    HIPCHK(S, launch_synth(dIQ, S->d_sintab, first_channel, nch, first_sample, nsamp, seed, S->stream));
    return SELENITE_RX_SUCCESS;
}

static int time_process(selenite_rx_instance *S, const void *src, void *dst, bool q15, uint32_t blockSize,
                        uint32_t iters, float *ms_per_call, const char *who)
{
    if (!S || !ms_per_call || iters == 0) return SELENITE_RX_ARGUMENT_ERROR;
    if (!block_size_ok(S, blockSize, who)) return S->status;
    HIPCHK(S, hipSetDevice(S->device));
    struct EventPair {                                     // destroyed on every exit path
        hipEvent_t e0 = nullptr, e1 = nullptr;
        ~EventPair() { if (e0) (void)hipEventDestroy(e0); if (e1) (void)hipEventDestroy(e1); }
    } ev;
    HIPCHK(S, hipEventCreate(&ev.e0));
    HIPCHK(S, hipEventCreate(&ev.e1));
    HIPCHK(S, hipEventRecord(ev.e0, S->stream));
    for (uint32_t i = 0; i < iters; ++i) {
        int rc = run_chain(S, src, q15, dst, q15, blockSize, kAll, nullptr);
        if (rc) return rc;
    }
    HIPCHK(S, hipEventRecord(ev.e1, S->stream));
    HIPCHK(S, hipEventSynchronize(ev.e1));
    float ms = 0.0f;
    HIPCHK(S, hipEventElapsedTime(&ms, ev.e0, ev.e1));
    *ms_per_call = ms / (float)iters;
    return SELENITE_RX_SUCCESS;
}

extern "C" int selenite_rx_time_process_device(selenite_rx_instance *S, const float *dSrcIQ, float *dDstAudio,
                                               uint32_t blockSize, uint32_t iters, float *ms_per_call)
{
    return time_process(S, dSrcIQ, dDstAudio, false, blockSize, iters, ms_per_call, "selenite_rx_time_process_device");
}

extern "C" int selenite_rx_time_process_q15_device(selenite_rx_instance *S, const int16_t *dSrcIQ, int16_t *dDstAudio,
                                                   uint32_t blockSize, uint32_t iters, float *ms_per_call)
{
    return time_process(S, dSrcIQ, dDstAudio, true, blockSize, iters, ms_per_call, "selenite_rx_time_process_q15_device");
}

extern "C" int selenite_rx_time_process_each_device(selenite_rx_instance *S, const void *dSrcIQ, void *dDstAudio, uint32_t blockSize,
                                                    uint32_t iters, float *ms_each, int q15)
{
    if (!S || !ms_each || iters == 0) return SELENITE_RX_ARGUMENT_ERROR;
    if (!block_size_ok(S, blockSize, "selenite_rx_time_process_each_device")) return S->status;
    HIPCHK(S, hipSetDevice(S->device));
    struct Events {                                        // destroyed on every exit path
        std::vector<hipEvent_t> e;
        ~Events() { for (hipEvent_t x : e) if (x) (void)hipEventDestroy(x); }
    } ev;
    ev.e.assign((size_t)iters + 1, nullptr);
    for (auto &x : ev.e) HIPCHK(S, hipEventCreate(&x));
    HIPCHK(S, hipEventRecord(ev.e[0], S->stream));
    for (uint32_t i = 0; i < iters; ++i) {
        int rc = run_chain(S, dSrcIQ, q15 != 0, dDstAudio, q15 != 0, blockSize, kAll, nullptr);
        if (rc) return rc;
        HIPCHK(S, hipEventRecord(ev.e[i + 1], S->stream));
    }
    HIPCHK(S, hipEventSynchronize(ev.e[iters]));
    for (uint32_t i = 0; i < iters; ++i) HIPCHK(S, hipEventElapsedTime(&ms_each[i], ev.e[i], ev.e[i + 1]));
    return SELENITE_RX_SUCCESS;
}

extern "C" int selenite_rx_time_streaming_roof_device(selenite_rx_instance *S, const void *dSrcIQ, void *dDstAudio, uint32_t blockSize,
                                                      uint32_t iters, float *ms_each, int q15)
{
    if (!S || !ms_each || iters == 0 || !dSrcIQ || !dDstAudio) return SELENITE_RX_ARGUMENT_ERROR;
    if (!block_size_ok(S, blockSize, "selenite_rx_time_streaming_roof_device")) return S->status;
    const selenite_rx_config &g = S->cfg;
    HIPCHK(S, hipSetDevice(S->device));
    // the per-channel state of SURVEY.md 8d (what selenite_rx_algorithmic_bytes counts), in a scratch buffer: the instance's own stays untouched
    uint32_t words = 0;
    if (g.nd_taps > 1) words += 2 * (g.nd_taps - 1);
    if (g.nh_taps > 1) words += 2 * (g.nh_taps - 1);
    words += 4 * g.n_biquad + (g.agc_enable ? 1 : 0) + (g.nco_enable ? 1 : 0);
    if (words > 1024) return fail(S, SELENITE_RX_LENGTH_ERROR, "selenite_rx_time_streaming_roof_device: state larger than the roof kernel handles");
    struct Scratch {
        float *p = nullptr; std::vector<hipEvent_t> e;
        ~Scratch() { if (p) (void)hipFree(p); for (hipEvent_t x : e) if (x) (void)hipEventDestroy(x); }
    } sc;
    const size_t nst = (size_t)g.channels * (words ? words : 1);
    HIPCHK(S, hipMalloc((void **)&sc.p, nst * sizeof(float)));
    HIPCHK(S, hipMemsetAsync(sc.p, 0, nst * sizeof(float), S->stream));
    sc.e.assign((size_t)iters + 1, nullptr);
    for (auto &x : sc.e) HIPCHK(S, hipEventCreate(&x));
    const uint32_t in_bytes = blockSize * (q15 ? 4u : 8u), out_bytes = (blockSize / g.decim) * (q15 ? 2u : 4u);
    for (int w = 0; w < 3; ++w) HIPCHK(S, launch_stream_roof(dSrcIQ, dDstAudio, sc.p, g.channels, in_bytes, out_bytes, words, S->stream));
    HIPCHK(S, hipEventRecord(sc.e[0], S->stream));
    for (uint32_t i = 0; i < iters; ++i) {
        HIPCHK(S, launch_stream_roof(dSrcIQ, dDstAudio, sc.p, g.channels, in_bytes, out_bytes, words, S->stream));
        HIPCHK(S, hipEventRecord(sc.e[i + 1], S->stream));
    }
    HIPCHK(S, hipEventSynchronize(sc.e[iters]));
    for (uint32_t i = 0; i < iters; ++i) HIPCHK(S, hipEventElapsedTime(&ms_each[i], sc.e[i], sc.e[i + 1]));
    return SELENITE_RX_SUCCESS;
}

extern "C" int selenite_rx_time_pattern_roof_device(selenite_rx_instance *S, const void *dSrcIQ, void *dDstAudio, uint32_t blockSize,
                                                    uint32_t iters, float *ms_each, int q15, uint32_t work)
{
    if (!S || !ms_each || iters == 0 || !dSrcIQ || !dDstAudio) return SELENITE_RX_ARGUMENT_ERROR;
    if (!block_size_ok(S, blockSize, "selenite_rx_time_pattern_roof_device")) return S->status;
    const selenite_rx_config &g = S->cfg;
    if (!cw_fused_ok(g, blockSize) || (g.block != 128 && g.block != 256 && g.block != 512) || (g.block == 512 && g.n_biquad == 2))
        return fail(S, SELENITE_RX_ARGUMENT_ERROR, "selenite_rx_time_pattern_roof_device: only the shapes of the systolic CW kernel (DSP blocks of 128 / 256 / 512) have a pattern of their own");
    HIPCHK(S, hipSetDevice(S->device));
    struct Scratch {
        float4 *p = nullptr; std::vector<hipEvent_t> e;
        ~Scratch() { if (p) (void)hipFree(p); for (hipEvent_t x : e) if (x) (void)hipEventDestroy(x); }
    } sc;
    const uint32_t ch_per_wg = 64u / g.n_biquad;
    const size_t nst = (size_t)((g.channels + ch_per_wg - 1) / ch_per_wg) * 64u;
    HIPCHK(S, hipMalloc((void **)&sc.p, nst * sizeof(float4)));
    HIPCHK(S, hipMemsetAsync(sc.p, 0, nst * sizeof(float4), S->stream));
    sc.e.assign((size_t)iters + 1, nullptr);
    for (auto &x : sc.e) HIPCHK(S, hipEventCreate(&x));
    const RxParams p = make_params(S, blockSize);
    for (int w = 0; w < 3; ++w) HIPCHK(S, launch_cw_roof(p, dSrcIQ, q15 != 0, dDstAudio, sc.p, work, S->stream));
    HIPCHK(S, hipEventRecord(sc.e[0], S->stream));
    for (uint32_t i = 0; i < iters; ++i) {
        HIPCHK(S, launch_cw_roof(p, dSrcIQ, q15 != 0, dDstAudio, sc.p, work, S->stream));
        HIPCHK(S, hipEventRecord(sc.e[i + 1], S->stream));
    }
    HIPCHK(S, hipEventSynchronize(sc.e[iters]));
    for (uint32_t i = 0; i < iters; ++i) HIPCHK(S, hipEventElapsedTime(&ms_each[i], sc.e[i], sc.e[i + 1]));
    return SELENITE_RX_SUCCESS;
}

extern "C" int selenite_rx_device_pci_bus_id(int ordinal, char *buf, size_t len)
{
    if (!buf || len < 13) return SELENITE_RX_ARGUMENT_ERROR;
    HIPCHK(nullptr, hipDeviceGetPCIBusId(buf, (int)len, ordinal));
    return SELENITE_RX_SUCCESS;
}

extern "C" uint64_t selenite_rx_algorithmic_bytes(const selenite_rx_config *g, uint32_t blockSize, uint64_t *read_bytes)
{
    if (!g || g->decim == 0) return 0;
    // state = FIR histories (both rails) + 16 B per biquad stage + AGC/NCO scalars,
    // read once and written once per channel-block.
    uint64_t state = 0;
    if (g->nd_taps > 1) state += 4ull * 2 * (g->nd_taps - 1);
    if (g->nh_taps > 1) state += 4ull * 2 * (g->nh_taps - 1);
    state += 16ull * g->n_biquad;
    state += 4ull * ((g->agc_enable ? 1 : 0) + (g->nco_enable ? 1 : 0));
    const uint64_t rd = 8ull * blockSize + state;
    const uint64_t wr = 4ull * (blockSize / g->decim) + state;
    if (read_bytes) *read_bytes = rd * g->channels;
    return (rd + wr) * g->channels;
}
