// rx_generic.hip -- generic (any configuration) HIP path of the Selenite RX block chain.
//
// Three simple kernels that cover every configuration the C-ABI accepts; the fused kernels in
// rx_fused.hip replace them for the BASELINE.json shapes.  Same arithmetic contract as the
// fused path (bit-exact vs the oracle in both ARITH modes), so tests cross-check the two.
//
//   k_front_generic   one wavefront per channel; per pass: coalesced I/Q load -> NCO mix
//                     (arm_sin/cos + arm_cmplx_mult_cmplx) -> LDS -> arm_fir_decimate on both
//                     rails -> LDS -> arm_fir pair / arm_cmplx_mag -> un-scaled audio
//   k_biquad_generic  one lane per channel, arm_biquad_cascade_df1, stage-outer like the reference
//   k_agc_generic     one wavefront per channel; arm_abs+arm_max by shuffle reduction, gain law,
//                     arm_scale, optional arm_float_to_q15 on the store
#include "rx_internal.h"

#pragma clang fp contract(off)

namespace srx {

static __host__ __device__ inline uint32_t up4(uint32_t v) { return (v + 3u) & ~3u; }

struct FrontLds {
    uint32_t tab, cd, ch, cdl, sI, sQ, dI, dQ, total;   // offsets in floats
};
static __host__ __device__ inline FrontLds front_layout(uint32_t nd, uint32_t nh, uint32_t M, uint32_t P, bool nco)
{
    FrontLds L;
    const uint32_t Hd = nd ? nd - 1 : 0, Hh = nh ? nh - 1 : 0;
    uint32_t o = 0;
    L.tab = o; o += nco ? 516u : 0u;
    L.cd = o;  o += up4(nd);
    L.ch = o;  o += up4(nh);
    L.cdl = o; o += up4(nh);
    L.sI = o;  o += up4(Hd + P * M);
    L.sQ = o;  o += up4(Hd + P * M);
    L.dI = o;  o += up4(Hh + P);
    L.dQ = o;  o += up4(Hh + P);
    L.total = o;
    return L;
}

size_t front_generic_lds_bytes(const RxParams &p)
{
    return (size_t)front_layout(p.nd, p.nh, p.decim, p.pass_out, p.nco != 0).total * sizeof(float);
}

// move s[adv .. adv+H) down to s[0 .. H) (history copy-back, arm_fir_decimate_f32.c:396-426)
__device__ __forceinline__ void shift_down(float *s, uint32_t H, uint32_t adv, int lane)
{
    for (uint32_t base = 0; base < H; base += kWave) {
        uint32_t i = base + lane;
        float t = (i < H) ? s[adv + i] : 0.0f;
        __syncthreads();
        if (i < H) s[i] = t;
        __syncthreads();
    }
}

template <int ARITH, typename TIn>
__global__ __launch_bounds__(64) void k_front_generic(RxParams p, const TIn *__restrict__ src,
                                                      float *__restrict__ audio)
{
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int lane = threadIdx.x;
    const uint32_t c = blockIdx.x;
    const uint32_t M = p.decim, nd = p.nd, nh = p.nh, P = p.pass_out;
    const uint32_t Hd = nd ? nd - 1 : 0, Hh = nh ? nh - 1 : 0;
    const bool am = (p.mode == SELENITE_MODE_AM);
    const bool use_fir = nh && !am;
    const bool upper = mode_is_upper(p.mode);
    const FrontLds L = front_layout(nd, nh, M, P, p.nco != 0);
    float *tab = lds + L.tab, *cd = lds + L.cd, *ch = lds + L.ch, *cdl = lds + L.cdl;
    float *sI = lds + L.sI, *sQ = lds + L.sQ, *dI = lds + L.dI, *dQ = lds + L.dQ;

    if (p.nco)
        for (uint32_t i = lane; i < 513; i += kWave) tab[i] = p.sintab[i];
    for (uint32_t i = lane; i < nd; i += kWave) cd[i] = p.dec_c[i];
    for (uint32_t i = lane; i < nh; i += kWave) { ch[i] = p.hilb_c[i]; cdl[i] = p.delay_c[i]; }
    for (uint32_t i = lane; i < Hd; i += kWave) {
        sI[i] = p.dec_state[((size_t)c * 2 + 0) * Hd + i];
        sQ[i] = p.dec_state[((size_t)c * 2 + 1) * Hd + i];
    }
    if (use_fir)
        for (uint32_t i = lane; i < Hh; i += kWave) {
            dI[i] = p.fir_state[((size_t)c * 2 + 0) * Hh + i];
            dQ[i] = p.fir_state[((size_t)c * 2 + 1) * Hh + i];
        }
    const uint32_t ph0 = p.nco ? p.phase[c] : 0u;
    const uint32_t step = p.nco ? p.step[c] : 0u;
    const uint32_t fo = use_fir ? Hh : 0u;     // where new decimated samples start in dI/dQ
    __syncthreads();

    const size_t in_base = (size_t)c * p.block_size, out_base = (size_t)c * p.nout;
    for (uint32_t o0 = 0; o0 < p.nout; o0 += P) {
        const uint32_t cnt = (p.nout - o0 < P) ? (p.nout - o0) : P;
        const uint32_t tin = cnt * M, n0 = o0 * M;
        // 1. load + NCO mix
        for (uint32_t i = lane; i < tin; i += kWave) {
            float2 v = load_iq(src, in_base + n0 + i);
            if (p.nco) v = cmul<ARITH>(v, nco_lo<ARITH>(tab, ph0 + (n0 + i) * step));
            sI[Hd + i] = v.x;
            sQ[Hd + i] = v.y;
        }
        __syncthreads();
        // 2. decimating FIR, both rails (arm_fir_decimate_f32: y[j] = sum_k c[k] s[jM+k])
        for (uint32_t j = lane; j < cnt; j += kWave) {
            float aI, aQ;
            if (nd) {
                aI = 0.0f; aQ = 0.0f;
                const uint32_t b = j * M;
                for (uint32_t k = 0; k < nd; ++k) {
                    const float ck = cd[k];
                    aI = mac<ARITH>(aI, sI[b + k], ck);
                    aQ = mac<ARITH>(aQ, sQ[b + k], ck);
                }
            } else {
                aI = sI[j]; aQ = sQ[j];
            }
            dI[fo + j] = aI;
            dQ[fo + j] = aQ;
        }
        __syncthreads();
        // 3. demodulator
        for (uint32_t j = lane; j < cnt; j += kWave) {
            float a;
            if (am) {
                a = cmag<ARITH>(dI[fo + j], dQ[fo + j]);
            } else if (use_fir) {
                float i2 = 0.0f, q2 = 0.0f;            // arm_fir_f32: y[n] = sum_k c[k] s[n+k]
                for (uint32_t k = 0; k < nh; ++k) {
                    i2 = mac<ARITH>(i2, dI[j + k], cdl[k]);
                    q2 = mac<ARITH>(q2, dQ[j + k], ch[k]);
                }
                a = upper ? (i2 - q2) : (i2 + q2);     // arm_sub_f32 / arm_add_f32
            } else {
                a = dI[j];
            }
            audio[out_base + o0 + j] = a;
        }
        __syncthreads();
        // 4. history copy-back
        if (Hd) { shift_down(sI, Hd, tin, lane); shift_down(sQ, Hd, tin, lane); }
        if (use_fir && Hh) { shift_down(dI, Hh, cnt, lane); shift_down(dQ, Hh, cnt, lane); }
    }
    for (uint32_t i = lane; i < Hd; i += kWave) {
        p.dec_state[((size_t)c * 2 + 0) * Hd + i] = sI[i];
        p.dec_state[((size_t)c * 2 + 1) * Hd + i] = sQ[i];
    }
    if (use_fir)
        for (uint32_t i = lane; i < Hh; i += kWave) {
            p.fir_state[((size_t)c * 2 + 0) * Hh + i] = dI[i];
            p.fir_state[((size_t)c * 2 + 1) * Hh + i] = dQ[i];
        }
    if (p.nco && lane == 0) p.phase[c] = ph0 + p.block_size * step;
}

template <int ARITH>
__global__ __launch_bounds__(64) void k_biquad_generic(RxParams p, float *__restrict__ audio)
{
    const uint32_t c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= p.channels) return;
    float *a = audio + (size_t)c * p.nout;
    for (uint32_t s = 0; s < p.nbiq; ++s) {
        const float b0 = p.biq_c[5 * s], b1 = p.biq_c[5 * s + 1], b2 = p.biq_c[5 * s + 2];
        const float a1 = p.biq_c[5 * s + 3], a2 = p.biq_c[5 * s + 4];
        float *st = p.biq_state + ((size_t)c * p.nbiq + s) * 4;
        float x1 = st[0], x2 = st[1], y1 = st[2], y2 = st[3];
        for (uint32_t n = 0; n < p.nout; ++n)
            a[n] = biquad_step<ARITH>(b0, b1, b2, a1, a2, a[n], x1, x2, y1, y2);
        st[0] = x1; st[1] = x2; st[2] = y1; st[3] = y2;
    }
}

template <int ARITH, typename TOut>
__global__ __launch_bounds__(64) void k_agc_generic(RxParams p, const float *audio, TOut *dst)
{
    const int lane = threadIdx.x;
    const uint32_t c = blockIdx.x;
    const uint32_t na = p.block / p.decim, nblk = p.block_size / p.block;
    float g = p.agc ? p.gain[c] : 1.0f;
    for (uint32_t b = 0; b < nblk; ++b) {
        const size_t base = (size_t)c * p.nout + (size_t)b * na;
        if (p.agc) {
            float m = 0.0f;
            for (uint32_t i = lane; i < na; i += kWave) m = fmaxf(m, fabsf(audio[base + i]));
            const float env = wave_max(m);                 // arm_abs_f32 + arm_max_f32
            g = agc_update<ARITH>(p.agcp, g, env);
            for (uint32_t i = lane; i < na; i += kWave) store_audio(dst, base + i, audio[base + i] * g);
        } else {
            for (uint32_t i = lane; i < na; i += kWave) store_audio(dst, base + i, audio[base + i]);
        }
    }
    if (p.agc && lane == 0) p.gain[c] = g;
}

// env[b] = max over channels; non-negative floats order like their bit patterns
__global__ __launch_bounds__(64) void k_env_global(RxParams p, const float *audio, float *env)
{
    const int lane = threadIdx.x;
    const uint32_t c = blockIdx.x;
    const uint32_t na = p.block / p.decim, nblk = p.block_size / p.block;
    for (uint32_t b = 0; b < nblk; ++b) {
        const size_t base = (size_t)c * p.nout + (size_t)b * na;
        float m = 0.0f;
        for (uint32_t i = lane; i < na; i += kWave) m = fmaxf(m, fabsf(audio[base + i]));
        m = wave_max(m);
        if (lane == 0) atomicMax(reinterpret_cast<unsigned int *>(env) + b, __float_as_uint(m));
    }
}

template <int ARITH, typename TOut>
__global__ __launch_bounds__(64) void k_agc_apply_global(RxParams p, const float *audio,
                                                         const float *env, TOut *dst)
{
    const int lane = threadIdx.x;
    const uint32_t c = blockIdx.x;
    const uint32_t na = p.block / p.decim, nblk = p.block_size / p.block;
    float g = p.gain[c];
    for (uint32_t b = 0; b < nblk; ++b) {
        const size_t base = (size_t)c * p.nout + (size_t)b * na;
        g = agc_update<ARITH>(p.agcp, g, env[b]);
        for (uint32_t i = lane; i < na; i += kWave) store_audio(dst, base + i, audio[base + i] * g);
    }
    if (lane == 0) p.gain[c] = g;
}

// ------------------------------------------------------------------------------------------
template <int ARITH, typename TIn>
static hipError_t front_launch(const RxParams &p, const void *src, float *audio, hipStream_t st)
{
    const size_t lds = front_generic_lds_bytes(p);
    auto k = k_front_generic<ARITH, TIn>;
    if (lds > 48 * 1024) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(k),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return e;
    }
    hipLaunchKernelGGL(k, dim3(p.channels), dim3(64), lds, st, p, static_cast<const TIn *>(src), audio);
    return hipGetLastError();
}

hipError_t launch_front_generic(const RxParams &p, int arith, const void *src, bool src_q15,
                                float *audio, hipStream_t st)
{
    if (arith != SELENITE_ARITH_CMSIS)
        return src_q15 ? front_launch<1, int16_t>(p, src, audio, st) : front_launch<1, float>(p, src, audio, st);
    return src_q15 ? front_launch<0, int16_t>(p, src, audio, st) : front_launch<0, float>(p, src, audio, st);
}

hipError_t launch_biquad_generic(const RxParams &p, int arith, float *audio, hipStream_t st)
{
    const dim3 grid((p.channels + 63) / 64), blk(64);
    if (arith != SELENITE_ARITH_CMSIS) hipLaunchKernelGGL(k_biquad_generic<1>, grid, blk, 0, st, p, audio);
    else hipLaunchKernelGGL(k_biquad_generic<0>, grid, blk, 0, st, p, audio);
    return hipGetLastError();
}

hipError_t launch_agc_generic(const RxParams &p, int arith, const float *audio, void *dst,
                              bool dst_q15, hipStream_t st)
{
    const dim3 grid(p.channels), blk(64);
    if (arith != SELENITE_ARITH_CMSIS) {
        if (dst_q15) hipLaunchKernelGGL((k_agc_generic<1, int16_t>), grid, blk, 0, st, p, audio, (int16_t *)dst);
        else hipLaunchKernelGGL((k_agc_generic<1, float>), grid, blk, 0, st, p, audio, (float *)dst);
    } else {
        if (dst_q15) hipLaunchKernelGGL((k_agc_generic<0, int16_t>), grid, blk, 0, st, p, audio, (int16_t *)dst);
        else hipLaunchKernelGGL((k_agc_generic<0, float>), grid, blk, 0, st, p, audio, (float *)dst);
    }
    return hipGetLastError();
}

hipError_t launch_env_global(const RxParams &p, const float *audio, float *env, hipStream_t st)
{
    hipError_t e = hipMemsetAsync(env, 0, sizeof(float) * (p.block_size / p.block), st);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(k_env_global, dim3(p.channels), dim3(64), 0, st, p, audio, env);
    return hipGetLastError();
}

hipError_t launch_agc_apply_global(const RxParams &p, int arith, const float *audio, const float *env,
                                   void *dst, bool dst_q15, hipStream_t st)
{
    const dim3 grid(p.channels), blk(64);
    if (arith != SELENITE_ARITH_CMSIS) {
        if (dst_q15) hipLaunchKernelGGL((k_agc_apply_global<1, int16_t>), grid, blk, 0, st, p, audio, env, (int16_t *)dst);
        else hipLaunchKernelGGL((k_agc_apply_global<1, float>), grid, blk, 0, st, p, audio, env, (float *)dst);
    } else {
        if (dst_q15) hipLaunchKernelGGL((k_agc_apply_global<0, int16_t>), grid, blk, 0, st, p, audio, env, (int16_t *)dst);
        else hipLaunchKernelGGL((k_agc_apply_global<0, float>), grid, blk, 0, st, p, audio, env, (float *)dst);
    }
    return hipGetLastError();
}

}  // namespace srx
