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
#include "rx_hist_exact.h"

#include <type_traits>

#pragma clang fp contract(off)

namespace srx {

static __host__ __device__ inline uint32_t up4(uint32_t v) { return (v + 3u) & ~3u; }

typedef float v2f __attribute__((ext_vector_type(2)));

// LDS layout of k_front_generic (offsets in floats).  Both rails travel together as (I, Q) pairs: one
// ds_read_b64 and one packed MAC serve both.  The decimator input is kept POLYPHASE -- phase array
// pp holds the samples u' = m*M + pp of the padded state [F zeros | nd-1 history | new] -- so that the
// lanes of a wavefront (consecutive outputs j) read consecutive pairs S[pp][j + q] for tap k' = q*M + pp
// instead of pairs M apart (bank conflicts M-fold in the flat layout).
struct FrontLds {
    uint32_t tab, cd, chd, S, D, Hq, F, PL, total;
};
static __host__ __device__ inline FrontLds front_layout(uint32_t nd, uint32_t nh, uint32_t M, uint32_t P, bool nco)
{
    FrontLds L;
    const uint32_t Hh = nh ? nh - 1 : 0;
    L.Hq = nd ? (nd - 1 + M - 1) / M : 0;                 // history length per phase
    L.F = nd ? L.Hq * M - (nd - 1) : 0;                   // leading pad of the padded state
    L.PL = L.Hq + P;                                      // pairs per phase array
    uint32_t o = 0;
    L.tab = o; o += nco ? 516u : 0u;
    L.cd = o;  o += up4(nd);
    L.chd = o; o += up4(2 * nh);                          // (delay tap, Hilbert tap) pairs
    L.S = o;   o += nd ? up4(2 * M * L.PL) : 0u;
    L.D = o;   o += up4(2 * (Hh + P));
    L.total = o;
    return L;
}

size_t front_generic_lds_bytes(const RxParams &p)
{
    return (size_t)front_layout(p.nd, p.nh, p.decim, p.pass_out, p.nco != 0).total * sizeof(float);
}

// move s[adv .. adv+H) down to s[0 .. H) (history copy-back, arm_fir_decimate_f32.c:396-426)
__device__ __forceinline__ void shift_down2(v2f *s, uint32_t H, uint32_t adv, int lane)
{
    for (uint32_t base = 0; base < H; base += kWave) {
        const uint32_t i = base + lane;
        const v2f t = (i < H) ? s[adv + i] : v2f{ 0.0f, 0.0f };
        __syncthreads();
        if (i < H) s[i] = t;
        __syncthreads();
    }
}

template <int ARITH>
__device__ __forceinline__ v2f mac2s(v2f acc, v2f w, float c)       // both rails, one tap
{
    const v2f c2 = { c, c };
    if constexpr (ARITH == 1) return __builtin_elementwise_fma(w, c2, acc);
    else { const v2f pr = w * c2; return acc + pr; }
}
template <int ARITH>
__device__ __forceinline__ v2f mac2v(v2f acc, v2f w, v2f c2)        // (I, Q) with a tap of its own per rail
{
    if constexpr (ARITH == 1) return __builtin_elementwise_fma(w, c2, acc);
    else { const v2f pr = w * c2; return acc + pr; }
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
    const bool fm = (p.mode == SELENITE_MODE_FM);          // the delay lines of the FIR pair run, the taps are not evaluated
    const bool use_fir = nh && !am;
    const bool upper = mode_is_upper(p.mode);
    const FrontLds L = front_layout(nd, nh, M, P, p.nco != 0);
    float *tab = lds + L.tab, *cd = lds + L.cd;
    v2f *chd = reinterpret_cast<v2f *>(lds + L.chd), *S = reinterpret_cast<v2f *>(lds + L.S), *D = reinterpret_cast<v2f *>(lds + L.D);
    const uint32_t Hq = L.Hq, F = L.F, PL = L.PL;

    if (p.nco)
        for (uint32_t i = lane; i < 513; i += kWave) tab[i] = p.sintab[i];
    for (uint32_t i = lane; i < nd; i += kWave) cd[i] = p.dec_c[i];
    for (uint32_t i = lane; i < nh; i += kWave) chd[i] = v2f{ p.delay_c[i], p.hilb_c[i] };
    for (uint32_t i = lane; i < Hd; i += kWave) {            // CMSIS state sample i sits at padded index i + F
        const uint32_t u = i + F;
        S[(u % M) * PL + u / M] = v2f{ p.dec_state[((size_t)c * 2 + 0) * Hd + i], p.dec_state[((size_t)c * 2 + 1) * Hd + i] };
    }
    if (use_fir)
        for (uint32_t i = lane; i < Hh; i += kWave)
            D[i] = v2f{ p.fir_state[((size_t)c * 2 + 0) * Hh + i], p.fir_state[((size_t)c * 2 + 1) * Hh + i] };
    const uint32_t ph0 = p.nco ? p.phase[c] : 0u;
    const uint32_t step = p.nco ? p.step[c] : 0u;
    const uint32_t fo = use_fir ? Hh : 0u;     // where new decimated samples start in D
    __syncthreads();

    const size_t in_base = (size_t)c * p.in_stride, out_base = (size_t)c * p.out_stride;
    for (uint32_t o0 = 0; o0 < p.nout; o0 += P) {
        const uint32_t cnt = (p.nout - o0 < P) ? (p.nout - o0) : P;
        const uint32_t tin = cnt * M, n0 = o0 * M;
        // 1. load + NCO mix; new sample i of the pass is padded-state index Hq*M + i
        for (uint32_t i = lane; i < tin; i += kWave) {
            float2 v = load_iq(src, in_base + n0 + i);
            if (p.nco) v = cmul<ARITH>(v, nco_lo<ARITH>(tab, ph0 + (n0 + i) * step));
            if (nd) S[(i % M) * PL + Hq + i / M] = v2f{ v.x, v.y };
            else D[fo + i] = v2f{ v.x, v.y };
        }
        __syncthreads();
        // 2. decimating FIR on both rails (arm_fir_decimate_f32: y[j] = sum_k c[k] s[jM+k]), four outputs per
        //    lane share every tap fetch; taps ascending for every output
        if (nd) {
            auto run = [&](auto rc, uint32_t j0) {            // R outputs per lane: j0 + lane + 64 r
                constexpr int R = decltype(rc)::value;
                v2f acc[R];
#pragma unroll
                for (int r = 0; r < R; ++r) acc[r] = v2f{ 0.0f, 0.0f };
                const uint32_t jb = j0 + lane;
                uint32_t pp = F % M;
                const v2f *row = S + pp * PL + F / M + jb;
                for (uint32_t k = 0; k < nd; ++k) {
                    const float ck = cd[k];
#pragma unroll
                    for (int r = 0; r < R; ++r) acc[r] = mac2s<ARITH>(acc[r], row[64 * r], ck);   // slack reads stay inside LDS
                    if (++pp == M) { pp = 0; row += 1 - (M - 1) * PL; } else row += PL;
                }
#pragma unroll
                for (int r = 0; r < R; ++r)
                    if (jb + 64 * r < cnt) D[fo + jb + 64 * r] = acc[r];
            };
            for (uint32_t j0 = 0; j0 < cnt;) {
                if (cnt - j0 > kWave) { run(std::integral_constant<int, 4>{}, j0); j0 += 4 * kWave; }
                else { run(std::integral_constant<int, 1>{}, j0); j0 += kWave; }
            }
            __syncthreads();
        }
        // 3. demodulator
        for (uint32_t j = lane; j < cnt; j += kWave) {
            float a;
            if (am) {
                const v2f d = D[fo + j];
                a = cmag<ARITH>(d.x, d.y);
            } else if (fm) {
                const v2f z = D[fo + j], zp = D[fo + j - 1];     // (nh >= 2: fo >= 1; the sample in front of the call is the newest history entry)
                a = fm_disc(z.x, z.y, zp.x, zp.y);
            } else if (use_fir) {
                v2f acc = { 0.0f, 0.0f };                   // arm_fir_f32 x2: y[n] = sum_k c[k] s[n+k]; I: delay taps, Q: Hilbert taps
#pragma unroll 4
                for (uint32_t k = 0; k < nh; ++k) acc = mac2v<ARITH>(acc, D[j + k], chd[k]);
                a = upper ? (acc.x - acc.y) : (acc.x + acc.y);     // arm_sub_f32 / arm_add_f32
            } else {
                a = D[j].x;
            }
            audio[out_base + o0 + j] = a;
        }
        __syncthreads();
        // 4. history copy-back
        if (Hq)
            for (uint32_t ppi = 0; ppi < M; ++ppi) shift_down2(S + ppi * PL, Hq, cnt, lane);
        if (use_fir && Hh) shift_down2(D, Hh, cnt, lane);
    }
    for (uint32_t i = lane; i < Hd; i += kWave) {
        const uint32_t u = i + F;
        const v2f v = S[(u % M) * PL + u / M];
        p.dec_state[((size_t)c * 2 + 0) * Hd + i] = v.x;
        p.dec_state[((size_t)c * 2 + 1) * Hd + i] = v.y;
    }
    if (use_fir)
        for (uint32_t i = lane; i < Hh; i += kWave) {
            p.fir_state[((size_t)c * 2 + 0) * Hh + i] = D[i].x;
            p.fir_state[((size_t)c * 2 + 1) * Hh + i] = D[i].y;
        }
    if (p.nco && lane == 0) p.phase[c] = ph0 + p.block_size * step;
}

// ------------------------------------------------------------------------------------------
// k_hist_exact -- SELENITE_ARITH_AUTO, what makes the rerun exact across a call boundary.
// A channel the call before left on the matrix kernel carries a Hilbert-pair history (the last nh - 1 decimated samples of
// both rails) of split16 precision.  When THIS call has to be recomputed for it (p.chan_flags: rerun bit, provenance
// kProvSplitExt), the history is first recomputed here in the reference's arithmetic -- arm_fir_decimate_f32.c:193-284: one
// accumulator from 0, taps ascending, product rounded, then sum rounded -- from the exact mixed samples k_ssb_split16 left behind:
// T = (one unused slot) ++ hist_ext[0 .. L - 1) (positions [E - H - L + 1, E - H)) ++ decimator state ([E - H, E)), H = nd - 1, L = ext_len = M * HH4.  History entry r
// (r = nh - 2 the newest) is the decimator output whose newest sample sits at E - M (nh - 1 - r): T[t0 .. t0 + nd), t0 = L - M (nh - 1 - r).
// One wavefront per flagged channel, the flag array walked 64 channels per workgroup and trip (which channels, and how many, only the
// device knows); rare by construction (a channel whose level crosses the guard ratio downwards at a call boundary).
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(64) void k_hist_exact(RxParams p, uint32_t all, uint32_t repair)     // all: every channel with that provenance (a call that runs the exact kernel on all channels), not only the flagged ones
{
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int lane = threadIdx.x;
    if (blockIdx.x == 0 && lane == 0 && p.chan_count_next) *p.chan_count_next = 0u;      // the counter the NEXT call's launch counts in
    if (p.chan_list) {
        // round 4: the channels to recompute as a dense list for the rerun pass.  A workgroup takes 1024 channels at a time: 16 words
        // per lane, a wave prefix sum of the per-lane counts, ONE atomic on the list's counter (64 atomics for 65 536 channels: one per
        // 16-channel window cost 50 us of serialised atomics when most windows had a flagged channel).  The order of the chunks in the
        // list varies from run to run, the results do not: every channel is computed on its own.
        const uint32_t nchunk = (p.channels + 1023u) / 1024u;
        for (uint32_t ch = blockIdx.x; ch < nchunk; ch += gridDim.x) {
            const uint32_t c0 = 1024u * ch + 16u * (uint32_t)lane;
            uint32_t bits = 0u;                                           // bit k: channel c0 + k has its rerun bit up
#pragma unroll
            for (int k = 0; k < 16; ++k) {
                const uint32_t ci = c0 + (uint32_t)k;
                const uint32_t w = ci < p.channels ? p.chan_flags[ci] : 0u;
                bits |= (w & kFlagRerun) << k;
            }
            const uint32_t n = (uint32_t)__builtin_popcount(bits);
            uint32_t incl = n;                                            // inclusive prefix sum over the wave
#pragma unroll
            for (int off = 1; off < 64; off <<= 1) {
                const uint32_t v = (uint32_t)__shfl_up((int)incl, off, 64);
                if (lane >= off) incl += v;
            }
            const uint32_t total = (uint32_t)__builtin_amdgcn_readlane((int)incl, 63);
            if (total == 0u) continue;                                    // wave-uniform
            uint32_t base = 0u;
            if (lane == 0) base = atomicAdd(p.chan_count, total);
            base = (uint32_t)__builtin_amdgcn_readfirstlane((int)base) + incl - n;
            for (uint32_t b2 = bits; b2 != 0u; b2 &= b2 - 1u) p.chan_list[base++] = c0 + (uint32_t)__builtin_ctz(b2);
        }
    }
    if (!repair) return;
    // (64 channels per workgroup and trip: one wave load of words -- 65 536 channels on 1024 workgroups are ONE memory round trip; the
    // 16-channel windows of round 3 were four dependent ones, 2-3 us of a launch that runs in front of every rerun pass)
    for (uint32_t base = 64u * blockIdx.x; base < p.channels; base += 64u * gridDim.x) {
        const uint32_t ci = base + (uint32_t)lane;
        const uint32_t f = ci < p.channels ? p.chan_flags[ci] : 0u;
        uint64_t todo = __builtin_amdgcn_ballot_w64(((f & kFlagRerun) != 0u || all != 0u) && ((f >> kProvShift) & kProvMask) == kProvSplitExt);
        while (todo != 0) {                                           // wave-uniform
            const uint32_t c = base + (uint32_t)__builtin_ctzll(todo);
            todo &= todo - 1;
            const uint32_t word = p.chan_flags[c];
            hist_exact_channel(p, c, word, lds, lane);
            // the channel's Hilbert-pair history is exact now: say so (what follows may be a kernel that hands the provenance on as it
            // finds it -- AM, which neither reads nor writes that history; everything else rewrites the word anyway)
            if (lane == 0) p.chan_flags[c] = word & ~(kProvMask << kProvShift);
            __syncthreads();
        }
    }
}

template <int ARITH>
__global__ __launch_bounds__(64) void k_biquad_generic(RxParams p, float *__restrict__ audio)
{
    const uint32_t c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= p.channels) return;
    float *a = audio + (size_t)c * p.out_stride;
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
        const size_t base = (size_t)c * p.out_stride + (size_t)b * na;
        if (p.agc) {
            float m = 0.0f;
            for (uint32_t i = lane; i < na; i += kWave) m = fmaxf(m, fabsf(audio[base + i]));
            const float env = wave_max(m);                 // arm_abs_f32 + arm_max_f32
            g = agc_update<ARITH>(p.agcp, g, env);
            for (uint32_t i = lane; i < na; i += kWave) store_audio(dst, base + i, audio[base + i] * g, p.q15_round);
        } else {
            for (uint32_t i = lane; i < na; i += kWave) store_audio(dst, base + i, audio[base + i], p.q15_round);
        }
    }
    if (p.agc && lane == 0) p.gain[c] = g;
}

// Lane -> (channel, float4) mapping shared by the two global-gain kernels: a DSP block of one
// channel is L = na/4 float4 (na % 4 == 0), and a wavefront covers cpw = max(1, 64 / L) channels
// at a time so short blocks (cfg3: na = 64, L = 16) still use every lane.
struct GlobalMap {
    uint32_t L, cpw, sub, coff;
    bool live;
    __device__ GlobalMap(uint32_t na, int lane)
    {
        L = na / 4;
        cpw = L >= kWave ? 1u : kWave / L;
        sub = L >= kWave ? (uint32_t)lane : (uint32_t)lane % L;
        coff = L >= kWave ? 0u : (uint32_t)lane / L;
        live = coff < cpw;
    }
};

// env[b] = max over channels, in two deterministic steps without atomics: every wavefront folds a
// strided set of channels into part[wave][b]; k_env_fold then folds the waves.  (One atomicMax per
// channel-block on nblk shared words cost ~1 ms at 65536 channels.)
constexpr int kEnvWaves = 4;
constexpr uint32_t kEnvMaxGrid = 1024;
__global__ __launch_bounds__(64 * kEnvWaves) void k_env_global(RxParams p, const float *audio, float *part)
{
    const int lane = threadIdx.x & 63;
    const uint32_t w = blockIdx.x * kEnvWaves + (threadIdx.x >> 6), nw = gridDim.x * kEnvWaves;
    const uint32_t na = p.block / p.decim, nblk = p.block_size / p.block;
    if (na % 4 == 0) {
        const GlobalMap gm(na, lane);
        for (uint32_t b = 0; b < nblk; ++b) {
            float m = 0.0f;
            for (uint32_t c = w * gm.cpw + gm.coff; c < p.channels && gm.live; c += nw * gm.cpw) {
                const float *a = audio + (size_t)c * p.out_stride + (size_t)b * na;
                for (uint32_t i = gm.sub; i < gm.L; i += kWave) {
                    const float4 v = *reinterpret_cast<const float4 *>(a + 4 * i);
                    m = fmaxf(fmaxf(m, fmaxf(fabsf(v.x), fabsf(v.y))), fmaxf(fabsf(v.z), fabsf(v.w)));
                }
            }
            m = wave_max(m);
            if (lane == 0) part[(size_t)w * nblk + b] = m;
        }
        return;
    }
    for (uint32_t b = 0; b < nblk; ++b) {
        float m = 0.0f;
        for (uint32_t c = w; c < p.channels; c += nw) {
            const size_t base = (size_t)c * p.out_stride + (size_t)b * na;
            for (uint32_t i = lane; i < na; i += kWave) m = fmaxf(m, fabsf(audio[base + i]));
        }
        m = wave_max(m);
        if (lane == 0) part[(size_t)w * nblk + b] = m;
    }
}

// env[b] = max over the n partial rows part[r][b]; one wavefront per DSP block
// (256 threads, four independent loads per thread and trip: with 64 threads and one load per trip a 4096-row fold took 13 us)
constexpr int kFoldThreads = 256;
__global__ __launch_bounds__(kFoldThreads) void k_env_fold(const float *part, float *env, uint32_t n, uint32_t nblk)
{
    __shared__ float wmax[kFoldThreads / kWave];
    const uint32_t b = blockIdx.x;
    float m0 = 0.0f, m1 = 0.0f, m2 = 0.0f, m3 = 0.0f;
    uint32_t r = threadIdx.x;
    for (; r + 3 * kFoldThreads < n; r += 4 * kFoldThreads) {
        const float a0 = part[(size_t)r * nblk + b], a1 = part[(size_t)(r + kFoldThreads) * nblk + b];
        const float a2 = part[(size_t)(r + 2 * kFoldThreads) * nblk + b], a3 = part[(size_t)(r + 3 * kFoldThreads) * nblk + b];
        m0 = fmaxf(m0, a0); m1 = fmaxf(m1, a1); m2 = fmaxf(m2, a2); m3 = fmaxf(m3, a3);
    }
    for (; r < n; r += kFoldThreads) m0 = fmaxf(m0, part[(size_t)r * nblk + b]);
    const float m = wave_max(fmaxf(fmaxf(m0, m1), fmaxf(m2, m3)));
    if ((threadIdx.x & (kWave - 1)) == 0) wmax[threadIdx.x / kWave] = m;
    __syncthreads();
    if (threadIdx.x == 0) {
        float t = wmax[0];
        for (int w = 1; w < kFoldThreads / kWave; ++w) t = fmaxf(t, wmax[w]);
        env[b] = t;
    }
}

// The same for the common geometry -- f32 audio, DSP blocks of na audio samples with na | 256 and na >= 4, calls of whole
// 256-sample rows -- with ONE wavefront streaming ONE channel in rows of 1 KiB (64 lanes x 16 bytes contiguous per
// wave-instruction; the general kernel below gives a wavefront 64 / (na / 4) channels with 4 na bytes contiguous each: 256-byte
// segments for na = 64, 4.0 TB/s).  The gains of the row's 256 / na blocks are wave-uniform: every lane runs the recurrence
// (same operations, same order as k_agc_apply_global) and keeps the gain of the block its four samples belong to.
template <int ARITH>
__global__ __launch_bounds__(64) void k_agc_apply_global_rows(RxParams p, const float *audio, const float *env, float *dst)
{
    const int lane = threadIdx.x;
    const uint32_t c = blockIdx.x;
    const uint32_t na = p.block / p.decim, bpr = 256u / na, nrow = p.nout / 256u;      // blocks per row, rows per channel
    const uint32_t myb = (4u * (uint32_t)lane) / na;                                  // this lane's block within a row
    const float *a = audio + (size_t)c * p.out_stride + 4 * lane;
    float *d = dst + (size_t)c * p.out_stride + 4 * lane;
    float g = p.gain[c];
    constexpr uint32_t RB = 4;                                                        // rows in flight (loads before stores: audio may alias dst)
    for (uint32_t r0 = 0; r0 < nrow; r0 += RB) {
        v4f v[RB];
#pragma unroll
        for (uint32_t k = 0; k < RB; ++k)
            if (r0 + k < nrow) v[k] = __builtin_nontemporal_load(reinterpret_cast<const v4f *>(a + (size_t)(r0 + k) * 256));
#pragma unroll
        for (uint32_t k = 0; k < RB; ++k)
            if (r0 + k < nrow) {
                float mine = g;
                for (uint32_t b = 0; b < bpr; ++b) {
                    g = agc_update<ARITH>(p.agcp, g, env[(r0 + k) * bpr + b]);
                    mine = (b == myb) ? g : mine;
                }
                __builtin_nontemporal_store(v[k] * mine, reinterpret_cast<v4f *>(d + (size_t)(r0 + k) * 256));
            }
    }
    if (lane == 0) p.gain[c] = g;
}

// gain recurrence on the shared envelope + arm_scale_f32 of every channel.  grid: one wavefront
// per GlobalMap::cpw channels when na % 4 == 0 (launch_agc_apply_global sizes it), else per channel.
template <int ARITH, typename TOut>
__global__ __launch_bounds__(64) void k_agc_apply_global(RxParams p, const float *audio,
                                                         const float *env, TOut *dst)
{
    const int lane = threadIdx.x;
    const uint32_t na = p.block / p.decim, nblk = p.block_size / p.block;
    if (na % 4 == 0) {
        const GlobalMap gm(na, lane);
        const uint32_t c = blockIdx.x * gm.cpw + gm.coff;
        if (!gm.live || c >= p.channels) return;
        float g = p.gain[c];
        auto put = [&](size_t at, const float4 &v, float gg) {
            const float4 r = make_float4(v.x * gg, v.y * gg, v.z * gg, v.w * gg);
            if constexpr (sizeof(TOut) == 4) {
                *reinterpret_cast<float4 *>(reinterpret_cast<float *>(dst) + at) = r;
            } else {
                uint2 w;
                float4_to_q15(r.x, r.y, r.z, r.w, p.q15_round, w.x, w.y);
                *reinterpret_cast<uint2 *>(reinterpret_cast<int16_t *>(dst) + at) = w;
            }
        };
        if (gm.L <= kWave) {
            // audio may alias dst (in place), so the loads of a chunk of blocks are issued before
            // the first store of the chunk to keep them in flight together
            constexpr uint32_t CH = 8;
            for (uint32_t b0 = 0; b0 < nblk; b0 += CH) {
                float4 v[CH];
#pragma unroll
                for (uint32_t k = 0; k < CH; ++k)
                    if (b0 + k < nblk && gm.sub < gm.L)
                        v[k] = *reinterpret_cast<const float4 *>(audio + (size_t)c * p.out_stride + (size_t)(b0 + k) * na + 4 * gm.sub);
#pragma unroll
                for (uint32_t k = 0; k < CH; ++k)
                    if (b0 + k < nblk) {
                        g = agc_update<ARITH>(p.agcp, g, env[b0 + k]);
                        if (gm.sub < gm.L) put((size_t)c * p.out_stride + (size_t)(b0 + k) * na + 4 * gm.sub, v[k], g);
                    }
            }
        } else {
            for (uint32_t b = 0; b < nblk; ++b) {
                const size_t base = (size_t)c * p.out_stride + (size_t)b * na;
                g = agc_update<ARITH>(p.agcp, g, env[b]);
                for (uint32_t i = gm.sub; i < gm.L; i += kWave)
                    put(base + 4 * i, *reinterpret_cast<const float4 *>(audio + base + 4 * i), g);
            }
        }
        if (gm.sub == 0) p.gain[c] = g;
        return;
    }
    const uint32_t c = blockIdx.x;
    float g = p.gain[c];
    for (uint32_t b = 0; b < nblk; ++b) {
        const size_t base = (size_t)c * p.out_stride + (size_t)b * na;
        g = agc_update<ARITH>(p.agcp, g, env[b]);
        for (uint32_t i = lane; i < na; i += kWave) store_audio(dst, base + i, audio[base + i] * g, p.q15_round);
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

// First step of a fold over MANY rows (one per channel): thread j of the launch takes the elements j, j + n, j + 2n, ...
// of part[rows][nblk] with n = threads of the launch, a multiple of nblk, so all of them belong to block j % nblk;
// out[j] is again a [n / nblk][nblk] array for k_env_fold.  Coalesced; exact (max is associative).
constexpr uint32_t kEnvColsGrid = 256, kEnvColsThreads = 256;
__global__ __launch_bounds__(kEnvColsThreads) void k_env_cols(const float *part, float *out, size_t total)
{
    const size_t j = (size_t)blockIdx.x * kEnvColsThreads + threadIdx.x, n = (size_t)gridDim.x * kEnvColsThreads;
    float m = 0.0f;
    for (size_t i = j; i < total; i += n) m = fmaxf(m, part[i]);
    out[j] = m;
}

// `part` needs room for kEnvColsGrid * kEnvColsThreads more floats behind its rows * nblk when rows is large
size_t env_fold_scratch_floats(uint32_t rows, uint32_t nblk)
{
    return (size_t)rows * nblk + (size_t)kEnvColsGrid * kEnvColsThreads;
}

hipError_t launch_env_fold(float *part, float *env, uint32_t rows, uint32_t nblk, hipStream_t st)
{
    const uint32_t n = kEnvColsGrid * kEnvColsThreads;
    if (rows > 4096 && n % nblk == 0) {
        float *cols = part + (size_t)rows * nblk;
        hipLaunchKernelGGL(k_env_cols, dim3(kEnvColsGrid), dim3(kEnvColsThreads), 0, st, part, cols, (size_t)rows * nblk);
        hipLaunchKernelGGL(k_env_fold, dim3(nblk), dim3(kFoldThreads), 0, st, cols, env, n / nblk, nblk);
    } else {
        hipLaunchKernelGGL(k_env_fold, dim3(nblk), dim3(kFoldThreads), 0, st, part, env, rows, nblk);
    }
    return hipGetLastError();
}

uint32_t env_global_rows(const RxParams &p)
{
    const uint32_t grid = (p.channels + kEnvWaves - 1) / kEnvWaves;
    return (grid < kEnvMaxGrid ? grid : kEnvMaxGrid) * kEnvWaves;
}

// part: env_global_rows(p) * (block_size / block) floats of scratch
hipError_t launch_env_global(const RxParams &p, const float *audio, float *part, float *env, hipStream_t st)
{
    const uint32_t rows = env_global_rows(p), nblk = p.block_size / p.block;
    hipLaunchKernelGGL(k_env_global, dim3(rows / kEnvWaves), dim3(64 * kEnvWaves), 0, st, p, audio, part);
    hipLaunchKernelGGL(k_env_fold, dim3(nblk), dim3(kFoldThreads), 0, st, part, env, rows, nblk);
    return hipGetLastError();
}

// arm_q15_to_float (SupportFunctions/arm_q15_to_float.c:87: (float)x / 32768.0f) over a whole buffer: eight values per thread and step,
// streamed (non-temporal both ways).  Used for int16 slots with a global gain: the fused kernels convert in and out symmetrically,
// and the global gain needs f32 audio between its two phases -- so the input is converted once, up front, and the call runs as an
// f32-input call whose gain pass stores int16 (the same operations on the same values as the fused int16 load: bit-identical)
__global__ __launch_bounds__(256) void k_q15_to_f32(const int16_t *__restrict__ src, float *__restrict__ dst, size_t n8)
{
    typedef short s8v __attribute__((ext_vector_type(8)));
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n8; i += (size_t)gridDim.x * blockDim.x) {
        const s8v v = __builtin_nontemporal_load(reinterpret_cast<const s8v *>(src) + i);
        v4f a = { q15_to_float(v[0]), q15_to_float(v[1]), q15_to_float(v[2]), q15_to_float(v[3]) };
        v4f b = { q15_to_float(v[4]), q15_to_float(v[5]), q15_to_float(v[6]), q15_to_float(v[7]) };
        __builtin_nontemporal_store(a, reinterpret_cast<v4f *>(dst) + 2 * i);
        __builtin_nontemporal_store(b, reinterpret_cast<v4f *>(dst) + 2 * i + 1);
    }
}

hipError_t launch_hist_exact(const RxParams &p, bool all, hipStream_t st)
{
    if (!p.chan_flags) return hipSuccess;
    // (in front of an AM call too, round 4: AM neither reads nor moves the Hilbert-pair history, but the decimator state moves on under it --
    // the last moment the samples kept in front of that state still belong to the history is now)
    static const bool off = diag_env("SELENITE_RX_NO_HIST_EXACT") != nullptr;      // diagnostic: what the rerun does without it
    const bool repair = p.hist_ext && p.nd >= 2 && p.nh >= 2 && !off;
    if (!repair && !p.chan_list) return hipSuccess;
    const uint32_t nwin = (p.channels + 63u) / 64u;
    const size_t lds = repair ? 2 * (size_t)(p.ext_len + p.nd - 1u) * sizeof(float) : 0;
    hipLaunchKernelGGL(k_hist_exact, dim3(nwin < 1024u ? nwin : 1024u), dim3(64), lds, st, p, all ? 1u : 0u, repair ? 1u : 0u);
    return hipGetLastError();
}

hipError_t launch_q15_to_f32(const int16_t *src, float *dst, size_t n, hipStream_t st)
{
    if (n % 8 != 0) return hipErrorInvalidValue;
    const size_t n8 = n / 8;
    const unsigned grid = (unsigned)((n8 + 255) / 256 < 16384 ? (n8 + 255) / 256 : 16384);
    hipLaunchKernelGGL(k_q15_to_f32, dim3(grid ? grid : 1), dim3(256), 0, st, src, dst, n8);
    return hipGetLastError();
}

hipError_t launch_agc_apply_global(const RxParams &p, int arith, const float *audio, const float *env,
                                   void *dst, bool dst_q15, hipStream_t st)
{
    const uint32_t na = p.block / p.decim;
    if (!dst_q15 && na >= 4 && 256u % na == 0 && p.nout % 256u == 0 && p.out_stride % 4u == 0) {     // 1 KiB rows, one wavefront per channel
        if (arith != SELENITE_ARITH_CMSIS) hipLaunchKernelGGL(k_agc_apply_global_rows<1>, dim3(p.channels), dim3(64), 0, st, p, audio, env, (float *)dst);
        else hipLaunchKernelGGL(k_agc_apply_global_rows<0>, dim3(p.channels), dim3(64), 0, st, p, audio, env, (float *)dst);
        return hipGetLastError();
    }
    const uint32_t cpw = (na % 4 == 0 && na / 4 < 64) ? 64 / (na / 4) : 1;
    const dim3 grid((p.channels + cpw - 1) / cpw), blk(64);
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
