// tx_fused.hip -- fused TX kernel for the BASELINE-like shape: ALC block 64, L = 4, 256-tap
// interpolator (P = 64 taps per phase), 63-tap Hilbert pair (unit-impulse delay, type-III Hilbert).
// One wavefront per channel, passes of 256 audio samples -> 1024 complex output samples.
//
// The interpolator is the RX decimator's twin (128 MAC per complex output sample).  Both rails use
// the same tap, so the interpolator state lives in LDS as (I, Q) pairs and every MAC is one
// v_pk_fma_f32 (or v_pk_mul + v_pk_add in the CMSIS arithmetic) with the tap in an SGPR fetched by
// v_readlane from four lane-distributed VGPRs.  Lane l owns input samples l, l+64, l+128, l+192 of
// the pass: its four outputs per sample (the L phases) are 32 contiguous bytes and the lanes of a
// wavefront store 2 KB contiguous -- coalesced without a transpose.  Per output the taps are visited
// in the reference's order (t ascending, arm_fir_interpolate_f32.c:389-440), single accumulator.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <cmath>
#include <vector>

#include "rx_internal.h"
#include "tx_internal.h"
#include <cstdlib>

namespace srx {

namespace {

typedef float v2f __attribute__((ext_vector_type(2)));

constexpr int kL = 4, kNI = 256, kP = kNI / kL, kNH = 63, kBlk = 64, kPass = 256;
constexpr int kHH = kNH - 1, kHH4 = (kHH + 3) & ~3, kFH = kHH4 - kHH;        // Hilbert history 62 -> 64, lead pad 2
constexpr int kHLen = kHH4 + kPass + 4;
constexpr int kZH = 64;                                                       // interpolator history slots (63 used)
constexpr int oTab = 0, oHI = 516, oHQ = oHI + kHLen, oZ = oHQ + kHLen, kTotal = oZ + 2 * (kZH + kPass);

__device__ __forceinline__ void wave_lds_sync()      // single-wave workgroup: program order is enough (rx_fused.hip)
{
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

template <int ARITH>
__device__ __forceinline__ v2f mac2(v2f acc, v2f w, float c)
{
    const v2f c2 = { c, c };
    if constexpr (ARITH == 1) return __builtin_elementwise_fma(w, c2, acc);
    else { const v2f pr = w * c2; return acc + pr; }
}

__device__ __forceinline__ float lane_bcast(float v, int l)
{
    return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), l));
}

typedef float tx_v4f __attribute__((ext_vector_type(4)));
template <typename T> struct AudioIO;
template <> struct AudioIO<float> {
    typedef float4 raw;
    static __device__ __forceinline__ raw load(const float *p, size_t i)      // non-temporal: streamed once
    {
        const tx_v4f v = __builtin_nontemporal_load(reinterpret_cast<const tx_v4f *>(p + i));
        return make_float4(v.x, v.y, v.z, v.w);
    }
    static __device__ __forceinline__ void unpack(const raw &r, float (&a)[4]) { a[0] = r.x; a[1] = r.y; a[2] = r.z; a[3] = r.w; }
    static __device__ __forceinline__ void store4(float *p, size_t cplx, const v2f (&o)[4], uint32_t = 0u)
    {
        tx_v4f *d = reinterpret_cast<tx_v4f *>(p + 2 * cplx);                    // non-temporal: written once, never read back
        __builtin_nontemporal_store(tx_v4f{ o[0].x, o[0].y, o[1].x, o[1].y }, d);
        __builtin_nontemporal_store(tx_v4f{ o[2].x, o[2].y, o[3].x, o[3].y }, d + 1);
    }
};
template <> struct AudioIO<int16_t> {
    typedef short4 raw;
    static __device__ __forceinline__ raw load(const int16_t *p, size_t i) { return *reinterpret_cast<const short4 *>(p + i); }
    static __device__ __forceinline__ void unpack(const raw &r, float (&a)[4])
    {
        a[0] = q15_to_float(r.x); a[1] = q15_to_float(r.y); a[2] = q15_to_float(r.z); a[3] = q15_to_float(r.w);
    }
    static __device__ __forceinline__ void store4(int16_t *p, size_t cplx, const v2f (&o)[4], uint32_t round = 0u)
    {
        uint2 a, b;
        float4_to_q15(o[0].x, o[0].y, o[1].x, o[1].y, round, a.x, a.y);
        float4_to_q15(o[2].x, o[2].y, o[3].x, o[3].y, round, b.x, b.y);
        uint2 *d = reinterpret_cast<uint2 *>(p + 2 * cplx);
        d[0] = a; d[1] = b;
    }
};

// NCO: 0 off, 1 per-channel LO computed here, 2 shared LO table (cos, -sin) of the call in `lo`
template <int ARITH, int NCO, typename TIn, typename TOut>
__global__ __launch_bounds__(64, 2) void k_tx_fused(TxParams p, uint32_t delay_idx, const float2 *__restrict__ lo,
                                                    const TIn *__restrict__ src, TOut *__restrict__ dst)
{
    using IO = AudioIO<TIn>;
    using OO = AudioIO<TOut>;
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int lane = threadIdx.x;
    const uint32_t c = blockIdx.x;
    float *tab = lds + oTab, *HI = lds + oHI, *HQ = lds + oHQ;
    v2f *Z = reinterpret_cast<v2f *>(lds + oZ);                   // Z[kZH + n]: n-th new (I,Q) of the pass; state[j] = Z[1 + j]
    const size_t in_base = (size_t)c * p.block_size, out_base = (size_t)c * p.block_size * kL;
    const uint32_t npass = p.block_size / kPass;
    typename IO::raw raw = IO::load(src, in_base + 4u * lane);

    // taps: interpolator lane-distributed (4 VGPRs), Hilbert (1 VGPR), fetched by v_readlane
    float creg[kNI / 64];
#pragma unroll
    for (int v = 0; v < kNI / 64; ++v) creg[v] = p.ic[64 * v + lane];
    const float hreg = (lane < kNH) ? p.hc[lane] : 0.0f;
    if constexpr (NCO == 1)
        for (int i = lane; i < 513; i += kWave) tab[i] = p.sintab[i];
    // state: branch-free loads (clamped index, masked value), all issued before the first LDS store
    {
        const float *stF = p.fir_state + (size_t)c * 2 * kHH, *stZ = p.int_state + (size_t)c * 2 * (kP - 1);
        float f[2], z[2];
#pragma unroll
        for (int j = 0; j < 2; ++j) {                            // 2 x 64 slots: rail j, slot lane
            const int sidx = lane - kFH;
            const float x = stF[j * kHH + (sidx < 0 ? 0 : sidx)];
            f[j] = sidx < 0 ? 0.0f : x;
            const int zi = lane - 1;
            const float y = stZ[j * (kP - 1) + (zi < 0 ? 0 : zi)];
            z[j] = zi < 0 ? 0.0f : y;
        }
        HI[lane] = f[0]; HQ[lane] = f[1];
        Z[lane] = v2f{ z[0], z[1] };
    }
    float gain = p.alc ? p.gain[c] : 1.0f;
    const uint32_t ph0 = NCO ? p.phase[c] : 0u, step = NCO ? p.step[c] : 0u;
    const bool am = p.mode == SELENITE_MODE_AM, up = mode_is_upper(p.mode);
    wave_lds_sync();

    for (uint32_t pass = 0; pass < npass; ++pass) {
        // ---- 1. ALC on the 4 blocks of the pass (16 lanes each): abs/max, gain law, scale ----
        float a[4];
        IO::unpack(raw, a);
        if (pass + 1 < npass) raw = IO::load(src, in_base + (size_t)(pass + 1) * kPass + 4u * lane);
        if (p.alc) {
            float m = fmaxf(fmaxf(fabsf(a[0]), fabsf(a[1])), fmaxf(fabsf(a[2]), fabsf(a[3])));
#pragma unroll
            for (int off = 1; off < 16; off <<= 1) m = fmaxf(m, __shfl_xor(m, off, 64));
            const float dmine = agc_desired(p.alcp, m);
            float g = gain, mine = gain;
#pragma unroll
            for (int b = 0; b < 4; ++b) {
                g = agc_step(p.alcp, g, lane_bcast(dmine, 16 * b));
                mine = (b == (lane >> 4)) ? g : mine;
            }
            gain = g;
#pragma unroll
            for (int r = 0; r < 4; ++r) a[r] = a[r] * mine;
        }
        *reinterpret_cast<float4 *>(HI + kHH4 + 4 * lane) = make_float4(a[0], a[1], a[2], a[3]);
        *reinterpret_cast<float4 *>(HQ + kHH4 + 4 * lane) = make_float4(a[0], a[1], a[2], a[3]);
        wave_lds_sync();
        // ---- 2.-3. Hilbert pair on the lane's 4 samples, sideband select, (I,Q) pairs into Z ----
        {
            float q2[4] = { 0.0f, 0.0f, 0.0f, 0.0f };
            constexpr int C = (kNH - 1) / 2;
#pragma unroll
            for (int t = 0; t <= (kHH4 + 3) / 4; ++t) {
                const float4 W = lds_ld4f(HQ + 4 * lane + 4 * t);
                const float w[4] = { W.x, W.y, W.z, W.w };
#pragma unroll
                for (int e = 0; e < 4; ++e)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const int k = 4 * t + e - r - kFH;
                        if (k < 0 || k >= kNH || (((k - C) & 1) == 0)) continue;      // structural zeros (exact for finite data)
                        q2[r] = mac<ARITH>(q2[r], w[e], lane_bcast(hreg, k));
                    }
            }
            const float *di = HI + kFH + delay_idx + 4 * lane;       // unit-impulse delay FIR
            v2f z[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                float ri = di[r] + 0.0f, rq = q2[r];
                if (am) { const float t = ri * 0.5f; ri = t + 0.5f; rq = 0.0f; }
                else if (!up) rq = -rq;
                z[r] = v2f{ ri, rq };
            }
            float4 *zp = reinterpret_cast<float4 *>(Z + kZH + 4 * lane);
            zp[0] = make_float4(z[0].x, z[0].y, z[1].x, z[1].y);
            zp[1] = make_float4(z[2].x, z[2].y, z[3].x, z[3].y);
        }
        wave_lds_sync();
        {   // Hilbert-pair history: last 64 slots to the front
            const float ti = HI[kPass + lane], tq = HQ[kPass + lane];
            wave_lds_sync();
            HI[lane] = ti; HQ[lane] = tq;
        }
        // ---- 4. interpolator: samples lane + 64 s, phases 0..3, taps t ascending ----
        v2f acc[4][kL];
#pragma unroll
        for (int s = 0; s < 4; ++s)
#pragma unroll
            for (int ph = 0; ph < kL; ++ph) acc[s][ph] = v2f{ 0.0f, 0.0f };
        const v2f *zb = Z + 1 + lane;                               // state[n + t] = Z[1 + n + t]
#pragma unroll
        for (int t = 0; t < kP; ++t) {
            float cf[kL];
#pragma unroll
            for (int ph = 0; ph < kL; ++ph) {
                const int k = (kL - 1 - ph) + kL * t;               // pCoeffs[(L - j) + t L], j = ph + 1
                cf[ph] = lane_bcast(creg[k >> 6], k & 63);
            }
#pragma unroll
            for (int s = 0; s < 4; ++s) {
                const v2f w = zb[64 * s + t];
#pragma unroll
                for (int ph = 0; ph < kL; ++ph) acc[s][ph] = mac2<ARITH>(acc[s][ph], w, cf[ph]);
            }
        }
        // ---- 5. NCO up-mix (LO = (cos, +sin)) and store: 4 complex samples per (lane, s) ----
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            const uint32_t o = kL * (lane + 64 * s);                // first output sample of this quad in the pass
            v2f out[kL];
            if constexpr (NCO == 2) {
                const float4 *lp = reinterpret_cast<const float4 *>(lo + (size_t)pass * kPass * kL + o);
                const float4 l01 = lp[0], l23 = lp[1];
                const float2 l[4] = { make_float2(l01.x, -l01.y), make_float2(l01.z, -l01.w),
                                      make_float2(l23.x, -l23.y), make_float2(l23.z, -l23.w) };   // conj of the RX LO: exact
#pragma unroll
                for (int ph = 0; ph < kL; ++ph) {
                    const float2 r = cmul<0>(make_float2(acc[s][ph].x, acc[s][ph].y), l[ph]);
                    out[ph] = v2f{ r.x, r.y };
                }
            } else if constexpr (NCO == 1) {
#pragma unroll
                for (int ph = 0; ph < kL; ++ph) {
                    const uint32_t phase = ph0 + (pass * kPass * kL + o + ph) * step;
                    const float x = (float)(phase >> 8) * kNcoK;
                    const float2 r = cmul<0>(make_float2(acc[s][ph].x, acc[s][ph].y),
                                             make_float2(cos_f32<0>(tab, x), sin_f32<0>(tab, x)));
                    out[ph] = v2f{ r.x, r.y };
                }
            } else {
#pragma unroll
                for (int ph = 0; ph < kL; ++ph) out[ph] = acc[s][ph];
            }
            OO::store4(dst, out_base + (size_t)pass * kPass * kL + o, out, p.q15_round);
        }
        wave_lds_sync();
        {   // interpolator history: last 64 pairs to the front
            const v2f tz = Z[kPass + lane];
            wave_lds_sync();
            Z[lane] = tz;
        }
        wave_lds_sync();
    }

    if (lane >= kFH) {
        p.fir_state[(size_t)c * 2 * kHH + (lane - kFH)] = HI[lane];
        p.fir_state[(size_t)c * 2 * kHH + kHH + (lane - kFH)] = HQ[lane];
    }
    if (lane >= 1) {
        const v2f z = Z[lane];
        p.int_state[(size_t)c * 2 * (kP - 1) + (lane - 1)] = z.x;
        p.int_state[(size_t)c * 2 * (kP - 1) + (kP - 1) + (lane - 1)] = z.y;
    }
    if (lane == 0) {
        if (p.alc) p.gain[c] = gain;
        if constexpr (NCO != 0) p.phase[c] = ph0 + p.block_size * kL * step;
    }
}

// ------------------------------------------------------------------------------------------
// k_tx_split16<NCO, TIn, TOut> -- SELENITE_ARITH_SPLIT16: the interpolator on the 16-bit matrix pipe.
// Phase ph of the polyphase interpolator is a plain 64-tap FIR over the state (taps c[(3-ph) + 4t]),
// i.e. the banded-Toeplitz product of rx_fused.hip's k_hilb_split16:  D_ph[i][m] = sum_k st'[16 i + k]
// B_ph[k][m],  B_ph[k][m] = c'_ph[k - m],  where st' = [0 | 63 history | 256 new] carries one extra
// oldest slot under a zero tap so that the new samples start 16-byte aligned (c'_ph[0] = 0,
// c'_ph[t + 1] = c[(3-ph) + 4t]; K = 65 + 15 -> 3 k-steps of 32).  Samples (I', Q' x 2^8) and taps
// (x 2^SC) are split into f16 hi + lo; xh*ch + xh*cl + xl*ch with f32 accumulation: 72
// v_mfma_f32_16x16x32_f16 per 256-sample pass instead of 1024 v_pk_fma + 256 v_readlane.  The four
// A fragments of a k-step serve all four phases.  The MFMA result layout gives every lane the four
// phases of four input samples = 32 contiguous output bytes each.  ALC, Hilbert pair and NCO stay f32
// VALU code as in k_tx_fused; the interpolator state is written from the f32 samples in registers and
// stays bit-exact.  Output is tolerance-based (<= 1e-5 relative per ALC block against the CMSIS chain).
// ------------------------------------------------------------------------------------------
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef _Float16 h2 __attribute__((ext_vector_type(2)));
typedef float v4f __attribute__((ext_vector_type(4)));

constexpr int kZS = 64;                                            // image slots in front of the new samples
constexpr int kKS = 3;                                             // k-steps of 32 (K = 80 <= 96)
constexpr int kZN = 240 + 32 * kKS;                                // highest image index read + 1
__host__ __device__ constexpr int zphys(int u) { return u; }        // linear: the A-fragment ds_read_b128 (16-byte slot 2 (l&15) + (l>>4) + 4 kk of lane l) is conflict free
                                                                    // under the instruction's real lane groups (GeoH::phys, rx_fused_common.h); the pad per 128 samples of rounds 3-4 made it 2-way
constexpr int kZIMG = (kZN + 8 + 7) & ~7;                          // halfs per image
constexpr int oZ16 = oHQ + kHLen, oZF = oZ16 + 2 * kZIMG;          // 4 images of kZIMG halfs = 2 kZIMG floats
constexpr int kTotal16 = oZF + 2 * kPass;                          // + the pass's interpolator input in f32, both rails

template <int NCO, typename TIn, typename TOut>
__global__ __launch_bounds__(64, 2) void k_tx_split16(TxParams p, uint32_t delay_idx, const float2 *__restrict__ lo,
                                                      const void *__restrict__ ttab16, int tap_sc,
                                                      const TIn *__restrict__ src, TOut *__restrict__ dst)
{
    using IO = AudioIO<TIn>;
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int lane = threadIdx.x;
    float *tab = lds + oTab, *HI = lds + oHI, *HQ = lds + oHQ;
    _Float16 *ZI = reinterpret_cast<_Float16 *>(lds + oZ16);       // [I hi | I lo | Q hi | Q lo]
    float *ZF = lds + oZF;                                         // [rail][256] f32: the last 64 are the next pass's history slots
    const uint32_t npass = p.block_size / kPass;
    // persistent grid (as k_ssb_split16): this workgroup runs channels blockIdx.x, + gridDim.x, ...; the 96 VGPRs of Toeplitz
    // fragments (24 KB per workgroup), the Hilbert taps and the periodic LO are loaded once, and the first audio samples of
    // the next channel are prefetched under the last pass of the current one
    typename IO::raw raw = IO::load(src, (size_t)blockIdx.x * p.block_size + 4u * lane);

    h8 Bh[kL][kKS], Bl[kL][kKS];                                   // 96 VGPRs of Toeplitz fragments
    {
        const h8 *bt = static_cast<const h8 *>(ttab16);
#pragma unroll
        for (int ph = 0; ph < kL; ++ph)
#pragma unroll
            for (int kk = 0; kk < kKS; ++kk) {
                Bh[ph][kk] = bt[((ph * kKS + kk) * 2 + 0) * 64 + lane];
                Bl[ph][kk] = bt[((ph * kKS + kk) * 2 + 1) * 64 + lane];
            }
    }
    const float hreg = (lane < kNH) ? p.hc[lane] : 0.0f;
    if constexpr (NCO == 1)
        for (int i = lane; i < 513; i += kWave) tab[i] = p.sintab[i];
    // periodic shared LO (NCO == 3): the table repeats every 256 output samples, a pass is four periods: step j of any
    // pass multiplies by LO[(128 j + 2 lane, + 1) mod 256] -- two register quads for the whole kernel
    float4 lo_per[2] = { make_float4(0, 0, 0, 0), make_float4(0, 0, 0, 0) };
    if constexpr (NCO == 3) {
        lo_per[0] = *reinterpret_cast<const float4 *>(lo + 2 * lane);
        lo_per[1] = *reinterpret_cast<const float4 *>(lo + 128 + 2 * lane);
    }
    // block floating point (rx_split16.hip): the scale 2^s of a pass puts the largest |component| of the image
    // ([64 history slots | 256 new samples]) into [2^14, 2^15); `pre` = 2^s
    auto putz = [&](int u, float i0, float q0, float i1, float q1, float pre) {     // image slots u (even), u + 1, both rails
        const float a[4] = { i0 * pre, i1 * pre, q0 * pre, q1 * pre };
        _Float16 h[4], l[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) { h[j] = (_Float16)a[j]; l[j] = (_Float16)(a[j] - (float)h[j]); }
        const int ph = zphys(u);
        *reinterpret_cast<h2 *>(ZI + 0 * kZIMG + ph) = h2{ h[0], h[1] };
        *reinterpret_cast<h2 *>(ZI + 1 * kZIMG + ph) = h2{ l[0], l[1] };
        *reinterpret_cast<h2 *>(ZI + 2 * kZIMG + ph) = h2{ h[2], h[3] };
        *reinterpret_cast<h2 *>(ZI + 3 * kZIMG + ph) = h2{ l[2], l[3] };
    };
    for (int u = kZS + kPass + 2 * lane; u < kZN; u += 2 * kWave) putz(u, 0.0f, 0.0f, 0.0f, 0.0f, 1.0f);   // finite slack under zero taps
    const bool am = p.mode == SELENITE_MODE_AM, up = mode_is_upper(p.mode);
    const int mcol = lane & 15, rg = lane >> 4;
    // streaming state of a channel, branch-free (Hilbert-pair histories, interpolator history, ALC gain, NCO phase / step): loaded
    // into registers -- for the next channel of this workgroup right after the current one is installed, so that its memory
    // round trip hides under the current channel's passes -- and installed into LDS when the channel starts
    float st_f[2], st_zi, st_zq, st_gain;
    uint32_t st_ph0, st_step;
    auto load_state = [&](uint32_t ch) {
        const float *stF = p.fir_state + (size_t)ch * 2 * kHH, *stZ = p.int_state + (size_t)ch * 2 * (kP - 1);
        const int sidx = lane - kFH, s0 = lane - 1;
#pragma unroll
        for (int j = 0; j < 2; ++j) st_f[j] = stF[j * kHH + (sidx < 0 ? 0 : sidx)];
        st_zi = stZ[s0 < 0 ? 0 : s0];
        st_zq = stZ[(kP - 1) + (s0 < 0 ? 0 : s0)];
        st_gain = p.alc ? p.gain[ch] : 1.0f;
        st_ph0 = NCO ? p.phase[ch] : 0u;
        st_step = NCO ? p.step[ch] : 0u;
    };
    load_state(blockIdx.x);
    for (uint32_t c = blockIdx.x; c < p.channels; c += gridDim.x) {
    const size_t in_base = (size_t)c * p.block_size, out_base = (size_t)c * p.block_size * kL;
    const uint32_t cn = c + gridDim.x < p.channels ? c + gridDim.x : c;     // next channel of this workgroup (or a harmless re-read)
    uint32_t e_hist;
    {
        const int sidx = lane - kFH, s0 = lane - 1;
        HI[lane] = sidx < 0 ? 0.0f : st_f[0];
        HQ[lane] = sidx < 0 ? 0.0f : st_f[1];
        // history slot u = 1 + state index (slot 0 = the extra zero); lane owns slot `lane` of both rails
        const float hi = s0 < 0 ? 0.0f : st_zi, hq = s0 < 0 ? 0.0f : st_zq;
        ZF[kPass - kZS + lane] = hi;
        ZF[kPass + kPass - kZS + lane] = hq;
        e_hist = wave_umax_bits(fmaxf(fabsf(hi), fabsf(hq))) >> 23;
    }
    int s_cur = 0x7fff;
    float gain = st_gain;
    const uint32_t ph0 = st_ph0, step = st_step;
    load_state(cn);                                                 // the next channel's state: in flight during this channel's passes
    wave_lds_sync();

    for (uint32_t pass = 0; pass < npass; ++pass) {
        // ---- 1. ALC ----
        float a[4];
        IO::unpack(raw, a);
        raw = pass + 1 < npass ? IO::load(src, in_base + (size_t)(pass + 1) * kPass + 4u * lane)
                               : IO::load(src, (size_t)cn * p.block_size + 4u * lane);
        if (p.alc) {
            float m = fmaxf(fmaxf(fabsf(a[0]), fabsf(a[1])), fmaxf(fabsf(a[2]), fabsf(a[3])));
            m = row16_fmax(m);                                     // ALC block = 64 audio samples = one 16-lane row (DPP)
            const float dmine = agc_desired(p.alcp, m);
            float g = gain, mine = gain;
#pragma unroll
            for (int b = 0; b < 4; ++b) {
                g = agc_step(p.alcp, g, lane_bcast(dmine, 16 * b));
                mine = (b == (lane >> 4)) ? g : mine;
            }
            gain = g;
#pragma unroll
            for (int r = 0; r < 4; ++r) a[r] = a[r] * mine;
        }
        *reinterpret_cast<float4 *>(HI + kHH4 + 4 * lane) = make_float4(a[0], a[1], a[2], a[3]);
        *reinterpret_cast<float4 *>(HQ + kHH4 + 4 * lane) = make_float4(a[0], a[1], a[2], a[3]);
        wave_lds_sync();
        // ---- 2.-3. Hilbert pair (f32 VALU), sideband select, split into the four images ----
        {
            float q2[4] = { 0.0f, 0.0f, 0.0f, 0.0f };
            constexpr int C = (kNH - 1) / 2;
#pragma unroll
            for (int t = 0; t <= (kHH4 + 3) / 4; ++t) {
                const float4 W = lds_ld4f(HQ + 4 * lane + 4 * t);
                const float w[4] = { W.x, W.y, W.z, W.w };
#pragma unroll
                for (int e = 0; e < 4; ++e)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const int k = 4 * t + e - r - kFH;
                        if (k < 0 || k >= kNH || (((k - C) & 1) == 0)) continue;
                        q2[r] = mac<1>(q2[r], w[e], lane_bcast(hreg, k));
                    }
            }
            const float *di = HI + kFH + delay_idx + 4 * lane;
            float zi[4], zq[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                float ri = di[r] + 0.0f, rq = q2[r];
                if (am) { const float t = ri * 0.5f; ri = t + 0.5f; rq = 0.0f; }
                else if (!up) rq = -rq;
                zi[r] = ri; zq[r] = rq;
            }
            // block exponent over the new samples and the history; the samples that become the next history
            // (the last 64 of the pass) on the side
            float mq = 0.0f, mt = 0.0f;
#pragma unroll
            for (int r = 0; r < 4; ++r) mq = fmaxf(mq, fmaxf(fabsf(zi[r]), fabsf(zq[r])));
            mt = 4 * lane >= kPass - kZS ? mq : 0.0f;
            const uint32_t e_tail = wave_umax_bits(mt) >> 23;
            const uint32_t e_need = max(max(wave_umax_bits(mq) >> 23, e_tail), e_hist);
            int s_new = 141 - (int)e_need;
            s_new = s_new > 127 ? 127 : (s_new < -126 ? -126 : s_new);
            if (s_new != s_cur) {                                  // wave-uniform; always in the first pass
                const int u = 2 * (lane & 31);                     // both half-waves write the same words
                const float2 hi = *reinterpret_cast<const float2 *>(ZF + kPass - kZS + u);
                const float2 hq = *reinterpret_cast<const float2 *>(ZF + kPass + kPass - kZS + u);
                putz(u, hi.x, hq.x, hi.y, hq.y, __uint_as_float((uint32_t)(s_new + 127) << 23));
                s_cur = s_new;
            }
            e_hist = e_tail;
            wave_lds_sync();                                       // history reads above, f32 rail writes below
            const float pre = __uint_as_float((uint32_t)(s_cur + 127) << 23);
            putz(kZS + 4 * lane, zi[0], zq[0], zi[1], zq[1], pre);
            putz(kZS + 4 * lane + 2, zi[2], zq[2], zi[3], zq[3], pre);
            *reinterpret_cast<float4 *>(ZF + 4 * lane) = make_float4(zi[0], zi[1], zi[2], zi[3]);
            *reinterpret_cast<float4 *>(ZF + kPass + 4 * lane) = make_float4(zq[0], zq[1], zq[2], zq[3]);
        }
        wave_lds_sync();
        {   // Hilbert-pair history
            const float ti = HI[kPass + lane], tq = HQ[kPass + lane];
            wave_lds_sync();
            HI[lane] = ti; HQ[lane] = tq;
        }
        // ---- 4. interpolator: 4 phases x 2 rails x 3 k-steps x 3 MFMAs ----
        v4f aI[kL], aQ[kL];
#pragma unroll
        for (int ph = 0; ph < kL; ++ph) { aI[ph] = v4f{ 0.0f, 0.0f, 0.0f, 0.0f }; aQ[ph] = v4f{ 0.0f, 0.0f, 0.0f, 0.0f }; }
#pragma unroll
        for (int kk = 0; kk < kKS; ++kk) {
            const int u = 16 * mcol + 8 * rg + 32 * kk;
            const int pz = zphys(u);
            const h8 xIh = *reinterpret_cast<const h8 *>(ZI + 0 * kZIMG + pz), xIl = *reinterpret_cast<const h8 *>(ZI + 1 * kZIMG + pz);
            const h8 xQh = *reinterpret_cast<const h8 *>(ZI + 2 * kZIMG + pz), xQl = *reinterpret_cast<const h8 *>(ZI + 3 * kZIMG + pz);
#pragma unroll
            for (int ph = 0; ph < kL; ++ph) {
                aI[ph] = __builtin_amdgcn_mfma_f32_16x16x32_f16(xIh, Bh[ph][kk], aI[ph], 0, 0, 0);
                aQ[ph] = __builtin_amdgcn_mfma_f32_16x16x32_f16(xQh, Bh[ph][kk], aQ[ph], 0, 0, 0);
                aI[ph] = __builtin_amdgcn_mfma_f32_16x16x32_f16(xIh, Bl[ph][kk], aI[ph], 0, 0, 0);
                aQ[ph] = __builtin_amdgcn_mfma_f32_16x16x32_f16(xQh, Bl[ph][kk], aQ[ph], 0, 0, 0);
                aI[ph] = __builtin_amdgcn_mfma_f32_16x16x32_f16(xIl, Bh[ph][kk], aI[ph], 0, 0, 0);
                aQ[ph] = __builtin_amdgcn_mfma_f32_16x16x32_f16(xQl, Bh[ph][kk], aQ[ph], 0, 0, 0);
            }
        }
        // ---- 5. transpose through LDS (lane holds the 4 phases of input samples 64 rg + 16 r + mcol), then
        //         NCO up-mix and store with every wave instruction covering 1 KB of contiguous output ----
        {
            float *T = lds + kTotal16;                                  // [1024][2] un-mixed output tile of the pass
            const int pex = -(s_cur + tap_sc);                          // exact power-of-two rescale
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                float4 *tp = reinterpret_cast<float4 *>(T + 2 * kL * (64 * rg + 16 * r + mcol));
                tp[0] = make_float4(__builtin_ldexpf(aI[0][r], pex), __builtin_ldexpf(aQ[0][r], pex), __builtin_ldexpf(aI[1][r], pex), __builtin_ldexpf(aQ[1][r], pex));
                tp[1] = make_float4(__builtin_ldexpf(aI[2][r], pex), __builtin_ldexpf(aQ[2][r], pex), __builtin_ldexpf(aI[3][r], pex), __builtin_ldexpf(aQ[3][r], pex));
            }
            wave_lds_sync();
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const uint32_t o = 2u * (64u * j + lane);                 // two complex outputs per lane and step
                const float4 t = lds_ld4f(T + 2 * o);
                float2 y0 = make_float2(t.x, t.y), y1 = make_float2(t.z, t.w);
                if constexpr (NCO == 2) {
                    const float4 l = *reinterpret_cast<const float4 *>(lo + (size_t)pass * kPass * kL + o);
                    y0 = cmul<0>(y0, make_float2(l.x, -l.y));
                    y1 = cmul<0>(y1, make_float2(l.z, -l.w));
                } else if constexpr (NCO == 3) {                          // periodic shared LO: o mod 256 = 128 (j & 1) + 2 lane
                    const float4 l = lo_per[j & 1];
                    y0 = cmul<0>(y0, make_float2(l.x, -l.y));
                    y1 = cmul<0>(y1, make_float2(l.z, -l.w));
                } else if constexpr (NCO == 1) {
                    const uint32_t phase = ph0 + (pass * kPass * kL + o) * step;
                    lo_v2f la, lb;                                        // (cos, -sin) pairs, arm_sin/cos_f32 restated (rx_device.h): same bits
                    nco_lo_pair(tab, phase, phase + step, la, lb);
                    y0 = cmul<0>(y0, make_float2(la.x, -la.y));
                    y1 = cmul<0>(y1, make_float2(lb.x, -lb.y));
                }
                const size_t at = out_base + (size_t)pass * kPass * kL + o;
                if constexpr (sizeof(TOut) == 4) {
                    __builtin_nontemporal_store(v4f{ y0.x, y0.y, y1.x, y1.y }, reinterpret_cast<v4f *>(reinterpret_cast<float *>(dst) + 2 * at));   // written once, never read back
                } else {
                    uint2 q;
                    float4_to_q15(y0.x, y0.y, y1.x, y1.y, p.q15_round, q.x, q.y);
                    *reinterpret_cast<uint2 *>(reinterpret_cast<int16_t *>(dst) + 2 * at) = q;
                }
            }
        }
        wave_lds_sync();
        {   // image history: slots [256, 320) -> [0, 64) of all four images (32-bit moves)
            uint32_t t4[4] = { 0, 0, 0, 0 };
            const int u = 2 * (lane & 31);
            if (lane < 32) {
#pragma unroll
                for (int j = 0; j < 4; ++j) t4[j] = *reinterpret_cast<const uint32_t *>(ZI + j * kZIMG + zphys(kPass + u));
            }
            wave_lds_sync();
            if (lane < 32) {
#pragma unroll
                for (int j = 0; j < 4; ++j) *reinterpret_cast<uint32_t *>(ZI + j * kZIMG + zphys(u)) = t4[j];
            }
        }
        wave_lds_sync();
    }

    if (lane >= kFH) {
        p.fir_state[(size_t)c * 2 * kHH + (lane - kFH)] = HI[lane];
        p.fir_state[(size_t)c * 2 * kHH + kHH + (lane - kFH)] = HQ[lane];
    }
    if (lane >= 1) {                                               // arm_fir_interpolate_f32 pState tails, exact f32
        float *stI = p.int_state + (size_t)c * 2 * (kP - 1), *stQ = stI + (kP - 1);
        stI[lane - 1] = ZF[kPass - kZS + lane];
        stQ[lane - 1] = ZF[kPass + kPass - kZS + lane];
    }
    if (lane == 0) {
        if (p.alc) p.gain[c] = gain;
        if constexpr (NCO != 0) p.phase[c] = ph0 + p.block_size * kL * step;
    }
    wave_lds_sync();                                               // the state reads above before the next channel's installs
    }
}

template <typename TIn, typename TOut>
hipError_t launch_s16(const TxParams &p, uint32_t delay_idx, const float2 *lo, const void *ttab16, int tap_sc, const void *src,
                      void *dst, hipStream_t st)
{
    constexpr size_t lds = (size_t)(kTotal16 + 2 * kPass * kL) * sizeof(float);      // + the 8 KB output tile
    // persistent grid: as many single-wave workgroups as the device keeps resident (SELENITE_TX_SPLIT16_GRID=0: one per channel)
    static int resident = 0;
    if (resident == 0) {
        int per_cu = 0, dev = 0;
        hipDeviceProp_t prop;
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, k_tx_split16<2, TIn, TOut>, 64, lds) != hipSuccess ||
            hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&prop, dev) != hipSuccess || per_cu <= 0)
            resident = -1;
        else
            resident = per_cu * prop.multiProcessorCount;
        if (const char *e = diag_env("SELENITE_TX_SPLIT16_GRID")) resident = std::atoi(e) > 0 ? std::atoi(e) : -1;
    }
    const dim3 grid(resident > 0 && (uint32_t)resident < p.channels ? (uint32_t)resident : p.channels), blk(64);
    const TIn *s = static_cast<const TIn *>(src);
    TOut *d = static_cast<TOut *>(dst);
    if (!p.nco) hipLaunchKernelGGL((k_tx_split16<0, TIn, TOut>), grid, blk, lds, st, p, delay_idx, lo, ttab16, tap_sc, s, d);
    else if (lo && p.lo_period == 256) hipLaunchKernelGGL((k_tx_split16<3, TIn, TOut>), grid, blk, lds, st, p, delay_idx, lo, ttab16, tap_sc, s, d);
    else if (lo) hipLaunchKernelGGL((k_tx_split16<2, TIn, TOut>), grid, blk, lds, st, p, delay_idx, lo, ttab16, tap_sc, s, d);
    else hipLaunchKernelGGL((k_tx_split16<1, TIn, TOut>), grid, blk, lds, st, p, delay_idx, lo, ttab16, tap_sc, s, d);
    return hipGetLastError();
}

template <int ARITH, typename TIn, typename TOut>
hipError_t launch_a(const TxParams &p, uint32_t delay_idx, const float2 *lo, const void *src, void *dst, hipStream_t st)
{
    constexpr size_t lds = (size_t)kTotal * sizeof(float);
    const dim3 grid(p.channels), blk(64);
    const TIn *s = static_cast<const TIn *>(src);
    TOut *d = static_cast<TOut *>(dst);
    if (!p.nco) hipLaunchKernelGGL((k_tx_fused<ARITH, 0, TIn, TOut>), grid, blk, lds, st, p, delay_idx, lo, s, d);
    else if (lo) hipLaunchKernelGGL((k_tx_fused<ARITH, 2, TIn, TOut>), grid, blk, lds, st, p, delay_idx, lo, s, d);
    else hipLaunchKernelGGL((k_tx_fused<ARITH, 1, TIn, TOut>), grid, blk, lds, st, p, delay_idx, lo, s, d);
    return hipGetLastError();
}

}  // namespace

bool tx_fused_ok(const selenite_tx_config &g, bool delay_is_impulse, bool hilb_odd_only, uint32_t block_size)
{
    return g.interp == kL && g.ni_taps == kNI && g.nh_taps == kNH && g.block == kBlk && delay_is_impulse &&
           hilb_odd_only && block_size % kPass == 0;
}

// Toeplitz fragments of the four interpolator phases for k_tx_split16 (f16 hi / lo, taps x 2^SC);
// *tap_sc = SC.  Layout: [phase][k-step][hi, lo][lane][8 halfs].
hipError_t build_tx_split16_table(const float *interp_coeffs, void **d_table, int *tap_sc)
{
    float cmax = 0.0f;
    for (int k = 0; k < kNI; ++k) cmax = std::fmax(cmax, std::fabs(interp_coeffs[k]));
    int ex = 0;
    if (cmax > 0.0f) std::frexp(cmax, &ex);
    const int SC = 15 - ex;                                        // largest |tap| * 2^SC in [2^14, 2^15)
    std::vector<_Float16> b16((size_t)kL * kKS * 2 * 64 * 8, (_Float16)0.0f);
    for (int ph = 0; ph < kL; ++ph)
        for (int kk = 0; kk < kKS; ++kk)
            for (int l = 0; l < 64; ++l)
                for (int j = 0; j < 8; ++j) {
                    const int tp = 32 * kk + 8 * (l >> 4) + j - (l & 15);       // c'_ph index: 0 = the extra zero tap
                    float cv = 0.0f;
                    if (tp >= 1 && tp <= kP) cv = std::ldexp(interp_coeffs[(kL - 1 - ph) + kL * (tp - 1)], SC);
                    const _Float16 hi = (_Float16)cv;
                    const _Float16 lo = (_Float16)(cv - (float)hi);
                    b16[((((size_t)ph * kKS + kk) * 2 + 0) * 64 + l) * 8 + j] = hi;
                    b16[((((size_t)ph * kKS + kk) * 2 + 1) * 64 + l) * 8 + j] = lo;
                }
    hipError_t e = hipMalloc(d_table, b16.size() * sizeof(_Float16));
    if (e != hipSuccess) return e;
    e = hipMemcpy(*d_table, b16.data(), b16.size() * sizeof(_Float16), hipMemcpyHostToDevice);
    *tap_sc = SC;
    return e;
}

hipError_t launch_tx_split16(const TxParams &p, uint32_t delay_idx, const float2 *lo, const void *ttab16, int tap_sc,
                             const void *src, bool q15, void *dst, hipStream_t st)
{
    return q15 ? launch_s16<int16_t, int16_t>(p, delay_idx, lo, ttab16, tap_sc, src, dst, st)
               : launch_s16<float, float>(p, delay_idx, lo, ttab16, tap_sc, src, dst, st);
}

hipError_t launch_tx_fused(const TxParams &p, int arith, uint32_t delay_idx, const float2 *lo, const void *src, bool q15,
                           void *dst, hipStream_t st)
{
    if (arith != SELENITE_ARITH_CMSIS)
        return q15 ? launch_a<1, int16_t, int16_t>(p, delay_idx, lo, src, dst, st) : launch_a<1, float, float>(p, delay_idx, lo, src, dst, st);
    return q15 ? launch_a<0, int16_t, int16_t>(p, delay_idx, lo, src, dst, st) : launch_a<0, float, float>(p, delay_idx, lo, src, dst, st);
}

}  // namespace srx
