// tx.hip -- batched TX chain (include/selenite_tx.h): ALC -> Hilbert pair -> sideband select ->
// arm_fir_interpolate_f32 by L -> NCO up-mix, one wavefront per channel, ALC block by ALC block.
// Arithmetic contract as in rx_device.h: every step keeps the CMSIS-DSP 1.5.3 operation order
// (oracle/tx_oracle.c restates it, oracle/ref_tx.c composes the real functions); SELENITE_ARITH_FMA
// fuses the FIR tap loops only.  This is the "any configuration" kernel of the TX direction (the
// counterpart of rx_generic.hip): plain LDS arrays, runtime tap loops, coalesced 8-byte I/Q stores.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <cmath>
#include <cstddef>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "../../include/selenite_tx.h"
#include "rx_device.h"
#include "rx_internal.h"
#include "tx_internal.h"

struct selenite_tx_instance {
    selenite_tx_config cfg;
    int device = 0;
    float *d_ic = nullptr, *d_hc = nullptr, *d_dc = nullptr, *d_sintab = nullptr;
    uint32_t *d_step = nullptr, *d_phase = nullptr;
    float *d_fir_state = nullptr, *d_int_state = nullptr, *d_gain = nullptr;
    void *d_io_in = nullptr, *d_io_out = nullptr;
    size_t io_in_bytes = 0, io_out_bytes = 0;
    std::vector<uint32_t> h_step;
    // fused-kernel planning (tx_fused.hip) and the shared LO of a call
    bool delay_is_impulse = false, hilb_odd_only = false, phase_uniform = true, steps_same = true, force_generic = false, no_periodic_lo = false;
    uint32_t delay_index = 0, phase_host = 0;
    float2 *d_lo = nullptr;
    bool lo_valid = false; uint32_t lo_phase = 0, lo_step = 0, lo_n = 0;   // what d_lo holds
    size_t lo_bytes = 0;
    void *d_ttab16 = nullptr;     // k_tx_split16: Toeplitz fragments of the interpolator phases
    int tpost = 0;                // tap scale exponent of that table
    hipStream_t stream = nullptr, own_stream = nullptr;
    int status = 0;
    std::string err;
};

namespace {

using namespace srx;

__device__ __forceinline__ float load_audio(const float *p, size_t i) { return p[i]; }
__device__ __forceinline__ float load_audio(const int16_t *p, size_t i) { return q15_to_float(p[i]); }
__device__ __forceinline__ void store_iq(float *p, size_t i, float re, float im, uint32_t = 0u)
{
    reinterpret_cast<float2 *>(p)[i] = make_float2(re, im);
}
__device__ __forceinline__ void store_iq(int16_t *p, size_t i, float re, float im, uint32_t round = 0u)
{
    short2 v;
    v.x = float_to_q15(re, round);
    v.y = float_to_q15(im, round);
    reinterpret_cast<short2 *>(p)[i] = v;
}

typedef float v2f __attribute__((ext_vector_type(2)));

static __host__ __device__ inline uint32_t up4(uint32_t v) { return (v + 3u) & ~3u; }

// LDS of k_tx_generic, in floats: sine table | interpolator taps | (delay, Hilbert) tap pairs |
// (delay-instance, Hilbert-instance) state pairs | (I, Q) interpolator state pairs.  Pairs: one
// ds_read_b64 and one packed MAC serve both members.
struct TxLds { uint32_t tab, ci, chd, H, Z, total; };
static __host__ __device__ inline TxLds tx_layout(uint32_t ni, uint32_t nh, uint32_t P, uint32_t nb)
{
    TxLds L;
    const uint32_t nh1 = nh ? nh - 1 : 0;
    uint32_t o = 0;
    L.tab = o; o += 516;
    L.ci = o;  o += up4(ni);
    L.chd = o; o += up4(2 * nh);
    L.H = o;   o += up4(2 * (nh1 + nb));
    L.Z = o;   o += up4(2 * ((P - 1) + nb));
    L.total = o;
    return L;
}

size_t tx_lds_bytes(const TxParams &p)
{
    return sizeof(float) * tx_layout(p.ni, p.nh, p.P, p.block).total;
}

template <int ARITH>
__device__ __forceinline__ v2f tx_mac2(v2f acc, v2f w, v2f c2)
{
    if constexpr (ARITH == 1) return __builtin_elementwise_fma(w, c2, acc);
    else { const v2f pr = w * c2; return acc + pr; }
}

template <int ARITH, typename TIn, typename TOut>
__global__ __launch_bounds__(64) void k_tx_generic(TxParams p, const TIn *__restrict__ src, TOut *__restrict__ dst)
{
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int lane = threadIdx.x;
    const uint32_t c = blockIdx.x;
    const uint32_t nb = p.block, L = p.L, P = p.P, nh = p.nh, nh1 = nh ? nh - 1 : 0, p1 = P - 1;
    const TxLds Y = tx_layout(p.ni, nh, P, nb);
    float *tab = lds + Y.tab, *ci = lds + Y.ci;
    v2f *chd = reinterpret_cast<v2f *>(lds + Y.chd);             // (delay tap, Hilbert tap)
    v2f *H = reinterpret_cast<v2f *>(lds + Y.H);                 // arm_fir_f32 pState of the (delay, Hilbert) instances
    v2f *Z = reinterpret_cast<v2f *>(lds + Y.Z);                 // arm_fir_interpolate_f32 pState of the (I, Q) rails
    if (p.nco)
        for (int i = lane; i < 513; i += kWave) tab[i] = p.sintab[i];
    for (uint32_t i = lane; i < p.ni; i += kWave) ci[i] = p.ic[i];
    for (uint32_t i = lane; i < nh; i += kWave) chd[i] = v2f{ p.dc[i], p.hc[i] };
    for (uint32_t i = lane; i < nh1; i += kWave)
        H[i] = v2f{ p.fir_state[(size_t)c * 2 * nh1 + i], p.fir_state[(size_t)c * 2 * nh1 + nh1 + i] };
    for (uint32_t i = lane; i < p1; i += kWave)
        Z[i] = v2f{ p.int_state[(size_t)c * 2 * p1 + i], p.int_state[(size_t)c * 2 * p1 + p1 + i] };
    float gain = p.alc ? p.gain[c] : 1.0f;
    const uint32_t ph0 = p.nco ? p.phase[c] : 0u, step = p.nco ? p.step[c] : 0u;
    const bool am = p.mode == SELENITE_MODE_AM, up = mode_is_upper(p.mode);
    const uint32_t nblk = p.block_size / nb;
    __syncthreads();

    for (uint32_t b = 0; b < nblk; ++b) {
        // 1. ALC: arm_abs_f32 + arm_max_f32 -> gain law -> arm_scale_f32
        const size_t ib = (size_t)c * p.block_size + (size_t)b * nb;
        float m = 0.0f;
        for (uint32_t i = lane; i < nb; i += kWave) m = fmaxf(m, fabsf(load_audio(src, ib + i)));
        if (p.alc) {
            m = wave_max(m);
            gain = agc_update<0>(p.alcp, gain, m);
        }
        for (uint32_t i = lane; i < nb; i += kWave) {
            float a = load_audio(src, ib + i);
            if (p.alc) a = a * gain;
            if (nh) H[nh1 + i] = v2f{ a, a };
            else Z[p1 + i] = v2f{ a, 0.0f };
        }
        __syncthreads();
        // 2.-3. Hilbert pair (arm_fir_f32 x2 as one packed tap loop) and sideband select
        if (nh) {
            for (uint32_t i = lane; i < nb; i += kWave) {
                v2f r = { 0.0f, 0.0f };
#pragma unroll 4
                for (uint32_t k = 0; k < nh; ++k) r = tx_mac2<ARITH>(r, H[i + k], chd[k]);
                Z[p1 + i] = r;
            }
            __syncthreads();
            for (uint32_t i0 = 0; i0 < nh1; i0 += kWave) {     // history tails, 64-wide slices
                const uint32_t i = i0 + lane;
                const v2f t = (i < nh1) ? H[nb + i] : v2f{ 0.0f, 0.0f };
                __syncthreads();
                if (i < nh1) H[i] = t;
                __syncthreads();
            }
        }
        for (uint32_t i = lane; i < nb; i += kWave) {
            v2f z = Z[p1 + i];
            if (am) {                                          // arm_scale_f32(0.5) then arm_offset_f32(0.5); Q = 0
                const float t = z.x * 0.5f;
                z = v2f{ t + 0.5f, 0.0f };
            } else if (!up) {
                z.y = -z.y;                                    // arm_negate_f32
            }
            Z[p1 + i] = z;
        }
        __syncthreads();
        // 4.-5. interpolator on both rails (one packed MAC per tap), NCO up-mix, store
        const size_t ob = ((size_t)c * p.block_size + (size_t)b * nb) * L;
        for (uint32_t o = lane; o < nb * L; o += kWave) {
            const uint32_t n = o / L, phs = o % L;
            v2f u;
            if (p.ni) {
                u = v2f{ 0.0f, 0.0f };
                const float *cf = ci + (L - 1 - phs);
#pragma unroll 4
                for (uint32_t t = 0; t < P; ++t) {
                    const float cc = cf[t * L];
                    u = tx_mac2<ARITH>(u, Z[n + t], v2f{ cc, cc });
                }
            } else {
                u = Z[n];
            }
            float re = u.x, im = u.y;
            if (p.nco) {
                const uint32_t phase = ph0 + (uint32_t)(b * nb * L + o) * step;
                const float x = (float)(phase >> 8) * kNcoK;
                const float lc = cos_f32<0>(tab, x), ls = sin_f32<0>(tab, x);
                const float2 r = cmul<0>(make_float2(u.x, u.y), make_float2(lc, ls));
                re = r.x; im = r.y;
            }
            store_iq(dst, ob + o, re, im, p.q15_round);
        }
        __syncthreads();
        if (p.ni) {
            for (uint32_t i0 = 0; i0 < p1; i0 += kWave) {
                const uint32_t i = i0 + lane;
                const v2f t = (i < p1) ? Z[nb + i] : v2f{ 0.0f, 0.0f };
                __syncthreads();
                if (i < p1) Z[i] = t;
                __syncthreads();
            }
        }
    }
    for (uint32_t i = lane; i < nh1; i += kWave) {
        p.fir_state[(size_t)c * 2 * nh1 + i] = H[i].x;
        p.fir_state[(size_t)c * 2 * nh1 + nh1 + i] = H[i].y;
    }
    for (uint32_t i = lane; i < p1; i += kWave) {
        p.int_state[(size_t)c * 2 * p1 + i] = Z[i].x;
        p.int_state[(size_t)c * 2 * p1 + p1 + i] = Z[i].y;
    }
    if (lane == 0) {
        if (p.alc) p.gain[c] = gain;
        if (p.nco) p.phase[c] = ph0 + p.block_size * L * step;
    }
}

int fail(selenite_tx_instance *S, int code, const std::string &msg)
{
    if (S && S->status == 0) { S->status = code; S->err = msg; }
    return code;
}

#define TCHK(S, call)                                                                          \
    do {                                                                                       \
        hipError_t e_ = (call);                                                                \
        if (e_ != hipSuccess)                                                                  \
            return fail((S), SELENITE_RX_DEVICE_ERROR, std::string(#call ": ") + hipGetErrorString(e_)); \
    } while (0)

bool tx_mode_ok(uint8_t m)
{
    return m == SELENITE_MODE_LSB || m == SELENITE_MODE_USB || m == SELENITE_MODE_CW || m == SELENITE_MODE_CWR ||
           m == SELENITE_MODE_AM || m == SELENITE_MODE_DIG || m == SELENITE_MODE_PKT;
}

TxParams make_params(const selenite_tx_instance *S, uint32_t block_size)
{
    const selenite_tx_config &g = S->cfg;
    TxParams p{};
    p.channels = g.channels; p.block = g.block; p.L = g.interp; p.ni = g.ni_taps;
    p.P = g.ni_taps ? g.ni_taps / g.interp : 1; p.nh = g.nh_taps; p.mode = g.mode;
    p.nco = g.nco_enable ? 1 : 0; p.alc = g.alc_enable ? 1 : 0; p.block_size = block_size; p.q15_round = g.q15_rounding ? 1u : 0u;
    p.ic = S->d_ic; p.hc = S->d_hc; p.dc = S->d_dc; p.sintab = S->d_sintab;
    p.step = S->d_step; p.phase = S->d_phase;
    p.fir_state = S->d_fir_state; p.int_state = S->d_int_state; p.gain = S->d_gain;
    p.alcp = AgcParams{ g.alc_target, g.alc_attack, g.alc_decay, g.alc_gain_min, g.alc_gain_max, g.alc_env_floor };
    return p;
}

template <int ARITH, typename TIn, typename TOut>
hipError_t launch(const TxParams &p, const void *src, void *dst, hipStream_t st)
{
    const size_t lds = tx_lds_bytes(p);
    auto k = k_tx_generic<ARITH, TIn, TOut>;
    if (lds > 48 * 1024) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(k), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return e;
    }
    hipLaunchKernelGGL(k, dim3(p.channels), dim3(64), lds, st, p, static_cast<const TIn *>(src), static_cast<TOut *>(dst));
    return hipGetLastError();
}

bool block_size_ok(selenite_tx_instance *S, uint32_t bs, const char *who)
{
    if (!S) return false;
    if (bs == 0 || bs % S->cfg.block != 0) {
        fail(S, SELENITE_RX_LENGTH_ERROR, std::string(who) + ": blockSize is not a non-zero multiple of cfg.block");
        return false;
    }
    return true;
}

int ensure(selenite_tx_instance *S, void **buf, size_t *cap, size_t need);

int run(selenite_tx_instance *S, const void *src, void *dst, bool q15, uint32_t bs)
{
    TCHK(S, hipSetDevice(S->device));
    const selenite_tx_config &g = S->cfg;
    const TxParams p = make_params(S, bs);
    const uint32_t phase_now = S->phase_host;
    if (g.nco_enable) S->phase_host += bs * g.interp * S->h_step[0];
    if (!S->force_generic && tx_fused_ok(g, S->delay_is_impulse, S->hilb_odd_only, bs)) {
        const float2 *lo = nullptr;
        if (g.nco_enable && S->phase_uniform) {
            // every channel shares step and phase: one LO per call, read from L2 by every wavefront
            // the table is a pure function of (start phase, step, length): reused when the call starts where it starts
            // (every call for an LO on the fs / 256 grid: the phase advance of a call is then a multiple of 2^32)
            const uint32_t nlo = bs * g.interp;
            if (!(S->lo_valid && S->lo_phase == phase_now && S->lo_step == S->h_step[0] && S->lo_n >= nlo)) {
                S->lo_valid = false;
                if (ensure(S, (void **)&S->d_lo, &S->lo_bytes, (size_t)nlo * sizeof(float2))) return S->status;
                TCHK(S, launch_lo_table(S->d_lo, S->d_sintab, phase_now, S->h_step[0], nlo, S->stream));
                S->lo_valid = true; S->lo_phase = phase_now; S->lo_step = S->h_step[0]; S->lo_n = nlo;
            }
            lo = S->d_lo;
        }
        if (g.arith == SELENITE_ARITH_SPLIT16 && S->d_ttab16) {
            TxParams ps = p;
            ps.lo_period = (lo && (S->h_step[0] & 0x00FFFFFFu) == 0 && !S->no_periodic_lo) ? 256u : 0u;   // k_tx_split16 keeps it in registers
            TCHK(S, launch_tx_split16(ps, S->delay_index, lo, S->d_ttab16, S->tpost, src, q15, dst, S->stream));
        }
        else
            TCHK(S, launch_tx_fused(p, (int)g.arith, S->delay_index, lo, src, q15, dst, S->stream));
        return 0;
    }
    if (tx_lds_bytes(p) > 64 * 1024) return fail(S, SELENITE_RX_LENGTH_ERROR, "filter lengths exceed the LDS budget of the TX kernel");
    const bool fma = g.arith != SELENITE_ARITH_CMSIS;
    hipError_t e;
    if (q15) e = fma ? launch<1, int16_t, int16_t>(p, src, dst, S->stream) : launch<0, int16_t, int16_t>(p, src, dst, S->stream);
    else e = fma ? launch<1, float, float>(p, src, dst, S->stream) : launch<0, float, float>(p, src, dst, S->stream);
    TCHK(S, e);
    return 0;
}

template <typename T>
int upload(selenite_tx_instance *S, T **dp, const T *h, size_t n)
{
    *dp = nullptr;
    if (!n) return 0;
    TCHK(S, hipMalloc((void **)dp, n * sizeof(T)));
    if (h) TCHK(S, hipMemcpy(*dp, h, n * sizeof(T), hipMemcpyHostToDevice));
    return 0;
}

int reset_state(selenite_tx_instance *S)
{
    const selenite_tx_config &g = S->cfg;
    const size_t C = g.channels, nh1 = g.nh_taps ? g.nh_taps - 1 : 0, p1 = g.ni_taps ? g.ni_taps / g.interp - 1 : 0;
    if (nh1) TCHK(S, hipMemsetAsync(S->d_fir_state, 0, C * 2 * nh1 * sizeof(float), S->stream));
    if (p1) TCHK(S, hipMemsetAsync(S->d_int_state, 0, C * 2 * p1 * sizeof(float), S->stream));
    TCHK(S, hipMemsetAsync(S->d_phase, 0, C * sizeof(uint32_t), S->stream));
    std::vector<float> gi(C, g.alc_gain_init);
    TCHK(S, hipMemcpyAsync(S->d_gain, gi.data(), C * sizeof(float), hipMemcpyHostToDevice, S->stream));
    TCHK(S, hipStreamSynchronize(S->stream));
    S->phase_host = 0;
    S->phase_uniform = S->steps_same;
    return 0;
}

int ensure(selenite_tx_instance *S, void **buf, size_t *cap, size_t need)
{
    if (*cap >= need) return 0;
    TCHK(S, hipStreamSynchronize(S->stream));
    if (*buf) (void)hipFree(*buf);
    *buf = nullptr; *cap = 0;
    TCHK(S, hipMalloc(buf, need));
    *cap = need;
    return 0;
}

void process_host(selenite_tx_instance *S, const void *src, void *dst, uint32_t bs, bool q15, const char *who)
{
    if (!block_size_ok(S, bs, who)) return;
    const selenite_tx_config &g = S->cfg;
    const size_t esz = q15 ? sizeof(int16_t) : sizeof(float);
    const size_t nin = (size_t)g.channels * bs * esz, nout = (size_t)g.channels * bs * g.interp * 2 * esz;
    if (hipSetDevice(S->device) != hipSuccess) { fail(S, SELENITE_RX_DEVICE_ERROR, "hipSetDevice"); return; }
    if (ensure(S, &S->d_io_in, &S->io_in_bytes, nin) || ensure(S, &S->d_io_out, &S->io_out_bytes, nout)) return;
    if (hipMemcpyAsync(S->d_io_in, src, nin, hipMemcpyHostToDevice, S->stream) != hipSuccess) { fail(S, SELENITE_RX_DEVICE_ERROR, "H2D copy failed"); return; }
    if (run(S, S->d_io_in, S->d_io_out, q15, bs)) return;
    if (hipMemcpyAsync(dst, S->d_io_out, nout, hipMemcpyDeviceToHost, S->stream) != hipSuccess ||
        hipStreamSynchronize(S->stream) != hipSuccess)
        fail(S, SELENITE_RX_DEVICE_ERROR, "D2H copy failed");
}

}  // namespace

extern "C" int selenite_tx_init(selenite_tx_instance **out, const selenite_tx_config *caller_cfg)
{
    if (!out) return SELENITE_RX_ARGUMENT_ERROR;
    *out = nullptr;
    if (!caller_cfg) return SELENITE_RX_ARGUMENT_ERROR;
    // struct_size says which header the caller was built against (include/selenite_rx.h: ABI versions); version 1 ends with alc_gain_init
    static_assert(offsetof(selenite_tx_config, q15_rounding) == 92 && sizeof(selenite_tx_config) == 104, "selenite_tx_config layout (LP64)");
    selenite_tx_config own{};
    if (caller_cfg->struct_size == SELENITE_TX_CONFIG_SIZE_V1) {
        std::memcpy(&own, caller_cfg, offsetof(selenite_tx_config, q15_rounding));
        own.abi_version = 1;
    } else if (caller_cfg->struct_size == sizeof(selenite_tx_config)) {
        own = *caller_cfg;
        if (own.abi_version != SELENITE_RX_ABI_VERSION || own.reserved != 0) return SELENITE_RX_ARGUMENT_ERROR;
    } else {
        return SELENITE_RX_ARGUMENT_ERROR;
    }
    own.struct_size = (uint32_t)sizeof(selenite_tx_config);
    const selenite_tx_config *g = &own;
    if (g->q15_rounding > 1u || !g->channels || !g->block || !g->interp || !tx_mode_ok(g->mode) ||
        g->arith > SELENITE_ARITH_SPLIT16)
        return SELENITE_RX_ARGUMENT_ERROR;
    if ((g->interp > 1) != (g->ni_taps > 0)) return SELENITE_RX_ARGUMENT_ERROR;
    if (g->ni_taps && !g->interp_coeffs) return SELENITE_RX_ARGUMENT_ERROR;
    if (g->nh_taps && (!g->hilb_coeffs || !g->delay_coeffs)) return SELENITE_RX_ARGUMENT_ERROR;
    if (g->ni_taps % g->interp) return SELENITE_RX_LENGTH_ERROR;        // arm_fir_interpolate_init_f32.c:91-96
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0) return SELENITE_RX_DEVICE_ERROR;   // no CPU fallback
    selenite_tx_instance *S = new selenite_tx_instance;
    S->cfg = *g;
    S->cfg.interp_coeffs = S->cfg.hilb_coeffs = S->cfg.delay_coeffs = nullptr;
    S->cfg.nco_step = nullptr;
    auto bail = [&](int code) { selenite_tx_free(S); return code; };
    if (hipGetDevice(&S->device) != hipSuccess) return bail(SELENITE_RX_DEVICE_ERROR);
    if (hipStreamCreateWithFlags(&S->own_stream, hipStreamNonBlocking) != hipSuccess) return bail(SELENITE_RX_DEVICE_ERROR);
    S->stream = S->own_stream;
    const size_t C = g->channels, nh1 = g->nh_taps ? g->nh_taps - 1 : 0, p1 = g->ni_taps ? g->ni_taps / g->interp - 1 : 0;
    S->h_step.resize(C);
    for (size_t c = 0; c < C; ++c) S->h_step[c] = g->nco_step ? g->nco_step[c] : g->nco_step_all;
    S->steps_same = true;
    for (size_t c = 1; c < C; ++c) S->steps_same = S->steps_same && S->h_step[c] == S->h_step[0];
    {   // what tx_fused.hip relies on: a unit-impulse delay FIR and a type-III Hilbert (taps at even
        // distance from the centre exactly +0.0f) -- same classification as the RX side (rx_api.hip)
        const uint32_t nh = g->nh_taps;
        int ones = 0, idx = -1;
        bool rest_zero = true;
        for (uint32_t k = 0; k < nh; ++k) {
            const float v = g->delay_coeffs[k];
            if (v == 1.0f) { ++ones; idx = (int)k; }
            else if (!(v == 0.0f && !std::signbit(v))) rest_zero = false;
        }
        if (nh && ones == 1 && rest_zero) { S->delay_is_impulse = true; S->delay_index = (uint32_t)idx; }
        if (nh % 2 == 1) {
            const int cc = (int)(nh - 1) / 2;
            bool ok = true;
            for (uint32_t k = 0; k < nh && ok; ++k)
                if ((((int)k - cc) & 1) == 0) {
                    const float v = g->hilb_coeffs[k];
                    if (!(v == 0.0f && !std::signbit(v))) ok = false;
                }
            S->hilb_odd_only = ok;
        }
        // kernel-selection overrides of the tests (selenite_rx_set_plan_option)
        S->force_generic = srx::plan_option(SELENITE_RX_OPT_TX_FORCE_GENERIC) != 0;
        S->no_periodic_lo = srx::plan_option(SELENITE_RX_OPT_NO_PERIODIC_LO) != 0;
    }
    if (g->arith == SELENITE_ARITH_SPLIT16 && tx_fused_ok(*g, S->delay_is_impulse, S->hilb_odd_only, 256) &&
        build_tx_split16_table(g->interp_coeffs, &S->d_ttab16, &S->tpost) != hipSuccess)
        return bail(SELENITE_RX_DEVICE_ERROR);
    if (upload(S, &S->d_ic, g->interp_coeffs, (size_t)g->ni_taps) || upload(S, &S->d_hc, g->hilb_coeffs, (size_t)g->nh_taps) ||
        upload(S, &S->d_dc, g->delay_coeffs, (size_t)g->nh_taps) || upload(S, &S->d_sintab, srx::host_sin_table(), (size_t)513) ||
        upload(S, &S->d_step, S->h_step.data(), C) || upload<uint32_t>(S, &S->d_phase, nullptr, C) ||
        upload<float>(S, &S->d_fir_state, nullptr, C * 2 * nh1) || upload<float>(S, &S->d_int_state, nullptr, C * 2 * p1) ||
        upload<float>(S, &S->d_gain, nullptr, C))
        return bail(SELENITE_RX_DEVICE_ERROR);
    if (reset_state(S)) return bail(SELENITE_RX_DEVICE_ERROR);
    *out = S;
    return SELENITE_RX_SUCCESS;
}

extern "C" void selenite_tx_free(selenite_tx_instance *S)
{
    if (!S) return;
    void *ptrs[] = { S->d_ic, S->d_hc, S->d_dc, S->d_sintab, S->d_step, S->d_phase, S->d_fir_state, S->d_int_state,
                     S->d_gain, S->d_io_in, S->d_io_out, S->d_lo, S->d_ttab16 };
    for (void *p : ptrs)
        if (p) (void)hipFree(p);
    if (S->own_stream) (void)hipStreamDestroy(S->own_stream);
    delete S;
}

extern "C" int selenite_tx_set_mode(selenite_tx_instance *S, uint8_t mode)
{
    if (!S || !tx_mode_ok(mode)) return SELENITE_RX_ARGUMENT_ERROR;     // instance stays usable in its old mode
    S->cfg.mode = mode;
    return SELENITE_RX_SUCCESS;
}

extern "C" const char *selenite_tx_kernel_name(const selenite_tx_instance *S)
{
    if (!S) return "";
    if (S->force_generic || !tx_fused_ok(S->cfg, S->delay_is_impulse, S->hilb_odd_only, 256)) return "k_tx_generic";
    return (S->cfg.arith == SELENITE_ARITH_SPLIT16 && S->d_ttab16) ? "k_tx_split16<4,256,63>" : "k_tx_fused<4,256,63>";
}

extern "C" int selenite_tx_status(const selenite_tx_instance *S) { return S ? S->status : SELENITE_RX_ARGUMENT_ERROR; }
extern "C" const char *selenite_tx_error_string(const selenite_tx_instance *S) { return S ? S->err.c_str() : "null instance"; }

extern "C" void selenite_tx_process_f32(selenite_tx_instance *S, const float *src, float *dst, uint32_t bs)
{
    process_host(S, src, dst, bs, false, "selenite_tx_process_f32");
}
extern "C" void selenite_tx_process_q15(selenite_tx_instance *S, const int16_t *src, int16_t *dst, uint32_t bs)
{
    process_host(S, src, dst, bs, true, "selenite_tx_process_q15");
}
extern "C" void selenite_tx_process_f32_device(selenite_tx_instance *S, const float *src, float *dst, uint32_t bs)
{
    if (block_size_ok(S, bs, "selenite_tx_process_f32_device")) run(S, src, dst, false, bs);
}
extern "C" void selenite_tx_process_q15_device(selenite_tx_instance *S, const int16_t *src, int16_t *dst, uint32_t bs)
{
    if (block_size_ok(S, bs, "selenite_tx_process_q15_device")) run(S, src, dst, true, bs);
}

extern "C" int selenite_tx_set_stream(selenite_tx_instance *S, void *hip_stream)
{
    if (!S) return SELENITE_RX_ARGUMENT_ERROR;
    hipStream_t next = hip_stream ? (hipStream_t)hip_stream : S->own_stream;
    if (next != S->stream) {                                // the streaming state is shared: drain the old stream first
        TCHK(S, hipStreamSynchronize(S->stream));
        S->stream = next;
    }
    return SELENITE_RX_SUCCESS;
}

extern "C" int selenite_tx_sync(selenite_tx_instance *S)
{
    if (!S) return SELENITE_RX_ARGUMENT_ERROR;
    TCHK(S, hipStreamSynchronize(S->stream));
    return S->status;
}

extern "C" int selenite_tx_get_state(selenite_tx_instance *S, const selenite_tx_state_view *v)
{
    if (!S || !v) return SELENITE_RX_ARGUMENT_ERROR;
    const selenite_tx_config &g = S->cfg;
    const size_t C = g.channels, nh1 = g.nh_taps ? g.nh_taps - 1 : 0, p1 = g.ni_taps ? g.ni_taps / g.interp - 1 : 0;
    TCHK(S, hipStreamSynchronize(S->stream));
    if (v->fir_state && nh1) TCHK(S, hipMemcpy(v->fir_state, S->d_fir_state, C * 2 * nh1 * sizeof(float), hipMemcpyDeviceToHost));
    if (v->interp_state && p1) TCHK(S, hipMemcpy(v->interp_state, S->d_int_state, C * 2 * p1 * sizeof(float), hipMemcpyDeviceToHost));
    if (v->alc_gain) TCHK(S, hipMemcpy(v->alc_gain, S->d_gain, C * sizeof(float), hipMemcpyDeviceToHost));
    if (v->nco_phase) TCHK(S, hipMemcpy(v->nco_phase, S->d_phase, C * sizeof(uint32_t), hipMemcpyDeviceToHost));
    return 0;
}

extern "C" int selenite_tx_set_state(selenite_tx_instance *S, const selenite_tx_state_view *v)
{
    if (!S || !v) return SELENITE_RX_ARGUMENT_ERROR;
    const selenite_tx_config &g = S->cfg;
    const size_t C = g.channels, nh1 = g.nh_taps ? g.nh_taps - 1 : 0, p1 = g.ni_taps ? g.ni_taps / g.interp - 1 : 0;
    TCHK(S, hipStreamSynchronize(S->stream));
    if (v->fir_state && nh1) TCHK(S, hipMemcpy(S->d_fir_state, v->fir_state, C * 2 * nh1 * sizeof(float), hipMemcpyHostToDevice));
    if (v->interp_state && p1) TCHK(S, hipMemcpy(S->d_int_state, v->interp_state, C * 2 * p1 * sizeof(float), hipMemcpyHostToDevice));
    if (v->alc_gain) TCHK(S, hipMemcpy(S->d_gain, v->alc_gain, C * sizeof(float), hipMemcpyHostToDevice));
    if (v->nco_phase) {
        TCHK(S, hipMemcpy(S->d_phase, v->nco_phase, C * sizeof(uint32_t), hipMemcpyHostToDevice));
        bool same = S->steps_same;
        for (size_t c = 1; c < C; ++c) same = same && v->nco_phase[c] == v->nco_phase[0];
        S->phase_uniform = same;
        S->phase_host = v->nco_phase[0];
    }
    return 0;
}

extern "C" int selenite_tx_reset(selenite_tx_instance *S)
{
    if (!S) return SELENITE_RX_ARGUMENT_ERROR;
    return reset_state(S);
}

extern "C" int selenite_tx_time_process_device(selenite_tx_instance *S, const float *src, float *dst, uint32_t bs,
                                               uint32_t iters, float *ms_per_call)
{
    if (!S || !ms_per_call || iters == 0) return SELENITE_RX_ARGUMENT_ERROR;
    if (!block_size_ok(S, bs, "selenite_tx_time_process_device")) return S->status;
    struct EventPair {                                      // destroyed on every exit path
        hipEvent_t e0 = nullptr, e1 = nullptr;
        ~EventPair() { if (e0) (void)hipEventDestroy(e0); if (e1) (void)hipEventDestroy(e1); }
    } ev;
    TCHK(S, hipEventCreate(&ev.e0));
    TCHK(S, hipEventCreate(&ev.e1));
    TCHK(S, hipEventRecord(ev.e0, S->stream));
    for (uint32_t i = 0; i < iters; ++i)
        if (run(S, src, dst, false, bs)) return S->status;
    TCHK(S, hipEventRecord(ev.e1, S->stream));
    TCHK(S, hipEventSynchronize(ev.e1));
    float ms = 0.0f;
    TCHK(S, hipEventElapsedTime(&ms, ev.e0, ev.e1));
    *ms_per_call = ms / (float)iters;
    return 0;
}
