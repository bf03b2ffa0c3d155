// rx_fused.hip -- fused single-launch kernels for the SSB receive chain (gfx950).
//
//   k_ssb_fused<ARITH, ND, M, NH, TIn, TOut>
//
// One wavefront per channel (64-thread workgroups, grid = channels >> 256 CUs).  Per pass the
// wavefront turns 256*M complex input samples into 256 audio samples entirely on chip:
//
//   HBM --dwordx4, 1 KiB/wave-instr--> VGPR --NCO mix (arm_sin/cos table in LDS, cmplx_mult)-->
//   LDS polyphase image (M arrays of (I,Q) pairs S_p[m] = s[m*M+p], history of HQ4 phase-samples
//   in front, 48-byte lane groups) --ds_read_b128, conflict free--> arm_fir_decimate taps as
//   v_pk_{mul,add,fma}_f32 on (I,Q) with the coefficient from v_readlane (4 adjacent outputs per
//   lane, so one b128 read feeds 8 packed MACs) --> LDS (decimated rails, NH-1
//   history) --> Hilbert FIR on Q (structurally-zero taps skipped), delay on I (unit impulse =
//   one LDS read), arm_sub/arm_add --> AGC: |.| and max by 16-lane xor-shuffle, gain law,
//   arm_scale --> one dwordx4 store per lane.
//
// Arithmetic: identical per-output operation order to the reference (single accumulator from
// 0.0f, taps ascending).  Exact-zero taps are skipped: acc + 0*x == acc for finite x and an
// accumulator that is never -0 (it starts at +0 and x + y = -0 only for -0 + -0), so results are
// bit-identical to the dense loop (Inf/NaN inputs excepted; documented in DESIGN.md).
//
// Covered: decimator (ND>0, M==4) or none (ND==0, M==1); Hilbert pair with unit-impulse delay
// and type-III (odd-only) Hilbert taps; USB/LSB/DIG/PKT (and CW/CWR without biquads); per-channel
// AGC with block/M in {4..256, power of two}.  Everything else runs on rx_generic.hip.
#include "rx_internal.h"

#include <cmath>
#include <cstdio>
#include <cstdlib>

#pragma clang fp contract(off)

namespace srx {

typedef float v2f __attribute__((ext_vector_type(2)));

template <int ND, int M, int NH>
struct Geo {
    static constexpr int P = 256;                                   // decimated outputs per pass
    static constexpr int T = P * M;                                 // complex inputs per pass
    static constexpr int HQ = ND ? (ND - 1 + M - 1) / M : 0;        // decimator history, phase-samples
    static constexpr int HQ4 = (HQ + 3) & ~3;
    static constexpr int F = ND ? (HQ4 * M + 1 - ND) : 0;           // leading zero-pad taps
    static constexpr int NCQ = HQ4 * M + 1;                         // padded taps cq[0 .. HQ4*M]
    static constexpr int NCR = (NCQ + 63) / 64;                     // coefficient VGPRs per lane
    static constexpr int PLEN = HQ4 + P;                            // complex elements per phase
    // Polyphase image: per phase p an array of (I,Q) float2 elements; each group of 4 elements
    // (the 4 outputs one lane owns) occupies THREE 16-byte slots (48 B, last slot unused), so the
    // ds_read_b128 of lane l at compile-time offset o is at 48*l + imm: lane stride 3 slots is
    // conflict-free for every b128 lane group and needs no per-read address arithmetic.
    static constexpr int PSF = 12 * (PLEN / 4);                     // floats per phase array
    static constexpr int HH = NH ? NH - 1 : 0;                      // Hilbert history
    static constexpr int HH4 = (HH + 3) & ~3;
    static constexpr int FH = HH4 - HH;                             // leading pad of the FIR window
    static constexpr int DLEN = HH4 + P + 4;
    static constexpr int oTab = 0;
    static constexpr int oS = 516;                                  // [M][PSF]          (ND > 0)
    static constexpr int oD = oS + (ND ? M * PSF : 0);              // [2 rails][DLEN]
    static constexpr int total = oD + 2 * DLEN;
    __host__ __device__ static constexpr int elem(int idx) { return 12 * (idx >> 2) + 2 * (idx & 3); }
};

struct FusedArgs {
    const float *cq;        // padded decimator taps: cq[k'] = dec[k' - F] (k' >= F), else 0; 64*NCR floats
    uint32_t delay_idx;     // index of the unit tap in delay_coeffs
    uint32_t upper;         // 1: audio = I' - Q'   0: audio = I' + Q'
    uint32_t am;            // 1: audio = |I + jQ| (arm_cmplx_mag_f32); the Hilbert pair and its state are untouched
    uint32_t group;         // lanes per DSP block = (block / M) / 4
    uint32_t grp_shift;     // k_ssb_mfma: phase group of a wave = (wave >> grp_shift) & 1
    const void *btab16;     // k_ssb_split16: Toeplitz operand, f16 hi/lo fragments
    float split_post;       // k_ssb_split16: exact power-of-two rescale of the MFMA result
    unsigned long long *dbg; // diagnostics (SELENITE_RX_DEBUG_TIMING): s_memtime stamps of workgroup 0, else NULL
};

// Workgroups of these kernels are ONE wavefront: LDS instructions of a wave execute in issue order,
// so a store is visible to any lane's later load without s_barrier.  What is needed is only that
// the compiler keeps the program order of LDS accesses: a wavefront-scope fence (emits nothing)
// plus the wave_barrier scheduling fence.  __syncthreads() would add "s_waitcnt vmcnt(0)", which
// drains the next pass's HBM prefetch and stalls the wave for a full memory latency per pass.
__device__ __forceinline__ void wave_lds_sync()
{
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// Prologue fill of N LDS words from HBM state.  `index(i)` is the element of `base` that slot i takes,
// negative for slots in front of the state (they are zero).  Loads are unconditional from a clamped
// index and masked afterwards: a conditional load becomes an exec-masked branch with its own
// s_waitcnt vmcnt(0), and the N/64 round trips of a wavefront then queue behind one another
// (measured: 8 serialized trips = 25 % of a workgroup's lifetime).  All loads are issued before the
// first store.
template <int N, typename IndexFn, typename StoreFn>
__device__ __forceinline__ void batched_fill(int lane, const float *__restrict__ base, IndexFn index, StoreFn store)
{
    constexpr int NI = (N + 63) / 64;
    float v[NI];
#pragma unroll
    for (int j = 0; j < NI; ++j) {
        const int i = (N % 64 == 0) ? j * 64 + lane : min(j * 64 + lane, N - 1);
        const int e = index(i);
        const float x = base[e < 0 ? 0 : e];
        v[j] = e < 0 ? 0.0f : x;
    }
#pragma unroll
    for (int j = 0; j < NI; ++j) {
        const int i = j * 64 + lane;
        if (N % 64 == 0 || i < N) store(i, v[j]);
    }
}

__device__ __forceinline__ float f4get(const float4 &v, int e)
{
    return e == 0 ? v.x : (e == 1 ? v.y : (e == 2 ? v.z : v.w));
}

// (I,Q) += (I,Q) * c  -- one v_pk_fma_f32, or v_pk_mul_f32 + v_pk_add_f32 in the CMSIS arithmetic
template <int ARITH>
__device__ __forceinline__ v2f mac2(v2f acc, v2f w, float c)
{
    const v2f c2 = { c, c };
    if constexpr (ARITH == 1) {
        return __builtin_elementwise_fma(w, c2, acc);
    } else {
        const v2f p = w * c2;
        return acc + p;
    }
}

// arm_cmplx_mult_cmplx_f32 on one (re, im) register pair: (a*c - b*d, a*d + b*c) with the four
// products and the two sums rounded separately (ComplexMathFunctions/arm_cmplx_mult_cmplx_f32.c:186-187).
// Three packed instructions: v_pk_mul_f32 x2 (operand halves picked by op_sel) + v_pk_add_f32.
__device__ __forceinline__ v2f cmul_pk(v2f A, v2f L)
{
    // The compiler does not fold the half swaps into op_sel (it emits v_mov/v_xor pairs), hence asm:
    //   t1 = (a*c, a*d)   t2 = (b*d, b*c)   r = (t1.lo - t2.lo, t1.hi + t2.hi)
    // s_nop: packed-f32 results need one wait state before a non-packed consumer on gfx950 (the
    // compiler inserts the same s_nop in its own code; it cannot see into the asm block).
    v2f t1, t2, r;
    asm("v_pk_mul_f32 %0, %3, %4 op_sel_hi:[0,1]\n\t"
        "v_pk_mul_f32 %1, %3, %4 op_sel:[1,1] op_sel_hi:[1,0]\n\t"
        "s_nop 0\n\t"
        "v_pk_add_f32 %2, %0, %1 neg_lo:[0,1]\n\t"
        "s_nop 0"
        : "=&v"(t1), "=&v"(t2), "=v"(r)
        : "v"(A), "v"(L));
    return r;
}

// raw global loads: two complex samples per lane per instruction
template <typename TIn> struct Raw;
template <> struct Raw<float> {
    typedef float4 type;
    static __device__ __forceinline__ type load(const float *src, size_t cplx_index)
    {
        // streamed once: non-temporal, so the shared LO / coefficient tables keep their cache lines
        typedef float f4v __attribute__((ext_vector_type(4)));
        const f4v v = __builtin_nontemporal_load(reinterpret_cast<const f4v *>(src + 2 * cplx_index));
        return make_float4(v.x, v.y, v.z, v.w);
    }
    static __device__ __forceinline__ void unpack(const type &r, float2 &a, float2 &b)
    {
        a = make_float2(r.x, r.y); b = make_float2(r.z, r.w);
    }
};
template <> struct Raw<int16_t> {
    typedef short4 type;
    static __device__ __forceinline__ type load(const int16_t *src, size_t cplx_index)
    {
        return *reinterpret_cast<const short4 *>(src + 2 * cplx_index);
    }
    static __device__ __forceinline__ void unpack(const type &r, float2 &a, float2 &b)
    {
        a = make_float2(q15_to_float(r.x), q15_to_float(r.y));
        b = make_float2(q15_to_float(r.z), q15_to_float(r.w));
    }
};

// arm_fir_decimate_f32 on BOTH rails for the 4 adjacent outputs j = 4*lane + r.
// Output j needs s[(j - HQ4 + q)*M + p] * cq[q*M + p], q = 0..HQ4 ascending, p ascending: the
// loops below visit (q, p) in exactly that order for every r, so each accumulator sees the
// reference's tap order.  Coefficients live lane-distributed in creg and reach the SGPR file by
// v_readlane (no scalar-memory latency, one fetch serves both rails and up to 4 outputs).
template <int ARITH, int ND, int M, int NH>
__device__ __forceinline__ void decim_quad(const float *S, int lane, const float (&creg)[Geo<ND, M, NH>::NCR],
                                           v2f (&acc)[4])
{
    using G = Geo<ND, M, NH>;
    const float *base = S + 12 * lane;
#pragma unroll
    for (int o = 0; o < (G::HQ4 + 4) / 2; ++o) {
        float4 W[M];
#pragma unroll
        for (int p = 0; p < M; ++p)
            W[p] = *reinterpret_cast<const float4 *>(base + p * G::PSF + 4 * (3 * (o >> 1) + (o & 1)));
#pragma unroll
        for (int e = 0; e < 2; ++e) {
#pragma unroll
            for (int p = 0; p < M; ++p) {
                const v2f w = e ? v2f{ W[p].z, W[p].w } : v2f{ W[p].x, W[p].y };
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int q = 2 * o + e - r;
                    const int kk = q * M + p;
                    if (q < 0 || q > G::HQ4 || (q == G::HQ4 && p > 0) || kk < G::F) continue;
                    const float c = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(creg[kk >> 6]), kk & 63));
                    acc[r] = mac2<ARITH>(acc[r], w, c);
                }
            }
        }
    }
}

// arm_fir_f32 with type-III Hilbert taps for 4 adjacent outputs n = 4*lane + r.
// dq: decimated Q rail, new samples start at HH4.  y[n] = sum_k h[k] * dq[n + k + FH].
// The taps live lane-distributed in hreg (lane k of hreg[k>>6] = h[k]), loaded ONCE per kernel and
// fetched by v_readlane: reading them from memory inside the pass loop costs an L2 round trip per
// pass (the compiler cannot hoist the loads above the audio stores it must assume may alias).
template <int ARITH, int ND, int M, int NH>
__device__ __forceinline__ void hilbert_quad(const float *dq, int lane, const float (&hreg)[(NH + 63) / 64 ? (NH + 63) / 64 : 1],
                                             float (&acc)[4])
{
    using G = Geo<ND, M, NH>;
    constexpr int C = (NH - 1) / 2;
#pragma unroll
    for (int t = 0; t <= (G::HH4 + 3) / 4; ++t) {
        const float4 W = *reinterpret_cast<const float4 *>(dq + 4 * lane + 4 * t);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const float w = f4get(W, e);
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int k = 4 * t + e - r - G::FH;
                if (k < 0 || k >= NH || (((k - C) & 1) == 0)) continue;   // structural zeros
                const float hk = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(hreg[k >> 6]), k & 63));
                acc[r] = mac<ARITH>(acc[r], w, hk);
            }
        }
    }
}

// Steps 3-5 of a pass, shared by the VALU and the MFMA kernels: Hilbert FIR on Q (structural zeros
// skipped) and unit-impulse delay on I from the decimated rails in LDS, sideband combine, AGC per
// DSP block (group lanes), one 4-sample store per lane.
// GROUP = lanes per DSP block as a compile-time constant (16: 64-sample blocks, 64: 256-sample blocks;
// 0 = runtime `group`): constant lane indices turn the block-envelope broadcasts into v_readlane and
// the lane reductions into DPP instead of ds_bpermute round trips.
template <int ARITH, int GROUP, int ND, int M, int NH, typename TOut, int AM = 0>
__device__ __forceinline__ void demod_agc_store(const RxParams &p, const FusedArgs &fa, const float *dI,
                                                const float *dQ, int lane, int group,
                                                const float (&hreg)[(NH + 63) / 64 ? (NH + 63) / 64 : 1], float &gain,
                                                TOut *__restrict__ dst, size_t out_index)
{
    using G = Geo<ND, M, NH>;
    float au[4];
    if constexpr (NH > 0 && AM != 0) {
        // AM: envelope of the decimated rails; new sample n of a pass sits at HH4 + n
        const float4 vi = *reinterpret_cast<const float4 *>(dI + G::HH4 + 4 * lane);
        const float4 vq = *reinterpret_cast<const float4 *>(dQ + G::HH4 + 4 * lane);
        au[0] = cmag<0>(vi.x, vq.x); au[1] = cmag<0>(vi.y, vq.y);
        au[2] = cmag<0>(vi.z, vq.z); au[3] = cmag<0>(vi.w, vq.w);
    } else if constexpr (NH > 0) {
        float q2[4] = { 0.0f, 0.0f, 0.0f, 0.0f };
        hilbert_quad<ARITH, ND, M, NH>(dQ, lane, hreg, q2);
        const float *di = dI + G::FH + fa.delay_idx + 4 * lane;   // unit-impulse delay FIR
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const float i2 = di[r] + 0.0f;                        // 0.0f + 1.0f*x of the dense loop
            au[r] = fa.upper ? (i2 - q2[r]) : (i2 + q2[r]);       // arm_sub_f32 / arm_add_f32
        }
    } else {
        const float4 v = *reinterpret_cast<const float4 *>(dI + 4 * lane);
        au[0] = v.x; au[1] = v.y; au[2] = v.z; au[3] = v.w;
    }
    // AGC: arm_abs + arm_max per DSP block (group lanes), gain law, arm_scale
    if (p.agc) {
        float m = fmaxf(fmaxf(fabsf(au[0]), fabsf(au[1])), fmaxf(fabsf(au[2]), fabsf(au[3])));
        float g = gain, mine = gain;
        if constexpr (GROUP > 0) {
#pragma unroll
            for (int off = 1; off < GROUP; off <<= 1) m = fmaxf(m, __shfl_xor(m, off, 64));
            const int myblk = lane / GROUP;
            // the divisions target/env of the blocks of this pass are independent of the gain
            // recurrence: issue them together, then run the (cheap) recurrence
            float dsr[64 / GROUP];
            // every lane divides for its own block's envelope (one division sequence for the whole
            // wavefront instead of one per block), then the per-block results are broadcast
            const float mine_d = agc_desired(p.agcp, m);
#pragma unroll
            for (int b = 0; b < 64 / GROUP; ++b)
                dsr[b] = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(mine_d), b * GROUP));
#pragma unroll
            for (int b = 0; b < 64 / GROUP; ++b) {
                g = agc_step(p.agcp, g, dsr[b]);
                mine = (b == myblk) ? g : mine;
            }
        } else {
#pragma unroll
            for (int off = 1; off < 64; off <<= 1)
                if (off < group) m = fmaxf(m, __shfl_xor(m, off, 64));
            const int nblk = 64 / group, myblk = lane / group;
            for (int b = 0; b < nblk; ++b) {
                const float env = __shfl(m, b * group, 64);
                g = agc_update<0>(p.agcp, g, env);
                if (b == myblk) mine = g;
            }
        }
        gain = g;
#pragma unroll
        for (int r = 0; r < 4; ++r) au[r] = au[r] * mine;
    }
    const size_t o = out_index + 4 * lane;
    if constexpr (sizeof(TOut) == 4) {
        *reinterpret_cast<float4 *>(reinterpret_cast<float *>(dst) + o) = make_float4(au[0], au[1], au[2], au[3]);
    } else {
        short4 s4;
        s4.x = float_to_q15(au[0]); s4.y = float_to_q15(au[1]);
        s4.z = float_to_q15(au[2]); s4.w = float_to_q15(au[3]);
        *reinterpret_cast<short4 *>(reinterpret_cast<int16_t *>(dst) + o) = s4;
    }
}

template <int ARITH, int NCO, int ND, int M, int NH, typename TIn, typename TOut, int AM = 0>
__global__ __launch_bounds__(64, 2) void k_ssb_fused(RxParams p, FusedArgs fa, const TIn *__restrict__ src,
                                                     TOut *__restrict__ dst)
{
    using G = Geo<ND, M, NH>;
    using R = Raw<TIn>;
    static_assert(ND == 0 ? M == 1 : M == 4, "fused kernel: no decimator, or decimate by 4");
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int lane = threadIdx.x;
    const uint32_t c = blockIdx.x;
    float *tab = lds + G::oTab;
    float *S = lds + G::oS;
    float *D = lds + G::oD;
    float *dI = D, *dQ = D + G::DLEN;
    constexpr int NLD = G::T / 128;                      // raw loads per lane per pass

    const size_t in_base = (size_t)c * p.in_stride, out_base = (size_t)c * p.out_stride;
    const uint32_t npass = p.nout / G::P;
    typename R::type raw[NLD];
#pragma unroll
    for (int i = 0; i < NLD; ++i) raw[i] = R::load(src, in_base + 128u * i + 2u * lane);

    // ---- prologue: tables and streaming state into LDS / registers ----
    if constexpr (NCO == 1)
        for (int i = lane; i < 513; i += kWave) tab[i] = p.sintab[i];
    float creg[G::NCR > 0 ? G::NCR : 1];
    if constexpr (ND > 0) {
#pragma unroll
        for (int v = 0; v < G::NCR; ++v) creg[v] = fa.cq[64 * v + lane];
        // history element (phase pp, index m) holds sample s = (m*M + pp) - F of the CMSIS state
        // (oldest first); slots before the state (s < 0) only ever meet zero-padded taps
        batched_fill<2 * M * G::HQ4>(lane, p.dec_state + (size_t)c * 2 * (ND - 1),
            [&](int i) {
                const int rail = i / (M * G::HQ4), sidx = i % (M * G::HQ4) - G::F;
                return sidx >= 0 ? rail * (ND - 1) + sidx : -1;
            },
            [&](int i, float v) {
                const int rail = i / (M * G::HQ4), rem = i % (M * G::HQ4);
                S[(rem % M) * G::PSF + G::elem(rem / M) + rail] = v;
            });
    }
    if constexpr (NH > 0) {
        batched_fill<2 * G::HH4>(lane, p.fir_state + (size_t)c * 2 * G::HH,
            [&](int i) {
                const int rail = i / G::HH4, sidx = i % G::HH4 - G::FH;
                return sidx >= 0 ? rail * G::HH + sidx : -1;
            },
            [&](int i, float v) { D[(i / G::HH4) * G::DLEN + i % G::HH4] = v; });
    }
    float hreg[(NH + 63) / 64 ? (NH + 63) / 64 : 1];
#pragma unroll
    for (int v = 0; v < (NH + 63) / 64; ++v) hreg[v] = (64 * v + lane < NH) ? p.hilb_c[64 * v + lane] : 0.0f;
    const uint32_t ph0 = NCO ? p.phase[c] : 0u;
    const uint32_t step = NCO ? p.step[c] : 0u;
    float gain = p.agc ? p.gain[c] : 1.0f;
    const int group = (int)fa.group;
    wave_lds_sync();

    for (uint32_t pass = 0; pass < npass; ++pass) {
        const uint32_t n0 = pass * G::T;
        // ---- 1. NCO mix of the prefetched samples, scatter into the LDS image ----
        float4 lo4[NLD];
        if constexpr (NCO == 2) {                                     // shared LO table (L2 resident):
#pragma unroll
            for (int i = 0; i < NLD; ++i)                             // all loads of the pass in flight at once
                lo4[i] = *reinterpret_cast<const float4 *>(p.lo + n0 + 128u * i + 2u * lane);
            __builtin_amdgcn_sched_barrier(0);
        }
#pragma unroll
        for (int i = 0; i < NLD; ++i) {
            const uint32_t n = 128u * i + 2u * lane;                  // even sample index in the pass
            float2 a, b;
            R::unpack(raw[i], a, b);
            if constexpr (NCO == 2) {
                const float4 l2 = lo4[i];
                a = cmul<0>(a, make_float2(l2.x, l2.y));
                b = cmul<0>(b, make_float2(l2.z, l2.w));
            } else if constexpr (NCO == 1) {
                a = cmul<0>(a, nco_lo<0>(tab, ph0 + (n0 + n) * step));
                b = cmul<0>(b, nco_lo<0>(tab, ph0 + (n0 + n + 1) * step));
            }
            if constexpr (ND > 0) {
                const int m = G::HQ4 + (int)(n / M), pp = (int)(n % M);   // n even: pp in {0,2}
                float *d = S + pp * G::PSF + G::elem(m);
                *reinterpret_cast<float2 *>(d) = a;
                *reinterpret_cast<float2 *>(d + G::PSF) = b;
            } else {
                *reinterpret_cast<float2 *>(dI + G::HH4 + n) = make_float2(a.x, b.x);
                *reinterpret_cast<float2 *>(dQ + G::HH4 + n) = make_float2(a.y, b.y);
            }
        }
        wave_lds_sync();
        // ---- prefetch the next pass while this one computes ----
        if (pass + 1 < npass) {
#pragma unroll
            for (int i = 0; i < NLD; ++i) raw[i] = R::load(src, in_base + n0 + G::T + 128u * i + 2u * lane);
        }
        // ---- 2. arm_fir_decimate_f32 on both rails, 4 adjacent outputs per lane ----
        if constexpr (ND > 0) {
            v2f acc[4] = { { 0.0f, 0.0f }, { 0.0f, 0.0f }, { 0.0f, 0.0f }, { 0.0f, 0.0f } };
            decim_quad<ARITH, ND, M, NH>(S, lane, creg, acc);
            *reinterpret_cast<float4 *>(dI + G::HH4 + 4 * lane) = make_float4(acc[0].x, acc[1].x, acc[2].x, acc[3].x);
            *reinterpret_cast<float4 *>(dQ + G::HH4 + 4 * lane) = make_float4(acc[0].y, acc[1].y, acc[2].y, acc[3].y);
            wave_lds_sync();
        }
        // ---- 3-5. Hilbert pair + sideband, AGC, store ----
        if (group == 16)
            demod_agc_store<ARITH, 16, ND, M, NH, TOut, AM>(p, fa, dI, dQ, lane, group, hreg, gain, dst, out_base + (size_t)pass * G::P);
        else if (group == 64)
            demod_agc_store<ARITH, 64, ND, M, NH, TOut, AM>(p, fa, dI, dQ, lane, group, hreg, gain, dst, out_base + (size_t)pass * G::P);
        else
            demod_agc_store<ARITH, 0, ND, M, NH, TOut, AM>(p, fa, dI, dQ, lane, group, hreg, gain, dst, out_base + (size_t)pass * G::P);
        wave_lds_sync();
        // ---- 6. history copy-back (arm_fir_decimate_f32.c:396-426, arm_fir_f32.c:947-978) ----
        if constexpr (ND > 0) {
            constexpr int NG = M * (G::HQ4 / 4);                      // 48-byte groups to move
            constexpr int NK = (NG + 63) / 64;
            float4 t0[NK], t1[NK];
#pragma unroll
            for (int k = 0; k < NK; ++k) {
                const int i = k * 64 + lane;
                if (i < NG) {
                    const float *sp = S + (i / (G::HQ4 / 4)) * G::PSF + 12 * (G::P / 4 + i % (G::HQ4 / 4));
                    t0[k] = *reinterpret_cast<const float4 *>(sp);
                    t1[k] = *reinterpret_cast<const float4 *>(sp + 4);
                }
            }
            wave_lds_sync();
#pragma unroll
            for (int k = 0; k < NK; ++k) {
                const int i = k * 64 + lane;
                if (i < NG) {
                    float *dp = S + (i / (G::HQ4 / 4)) * G::PSF + 12 * (i % (G::HQ4 / 4));
                    *reinterpret_cast<float4 *>(dp) = t0[k];
                    *reinterpret_cast<float4 *>(dp + 4) = t1[k];
                }
            }
        }
        if constexpr (NH > 0) {
            constexpr int NV = 2 * (G::HH4 / 4);
            static_assert(NV <= 64, "Hilbert history move assumes <= 64 float4");
            float4 tmp;
            const int rail = lane / (G::HH4 / 4), v = lane % (G::HH4 / 4);
            if (lane < NV) tmp = *reinterpret_cast<const float4 *>(D + rail * G::DLEN + G::P + 4 * v);
            wave_lds_sync();
            if (lane < NV) *reinterpret_cast<float4 *>(D + rail * G::DLEN + 4 * v) = tmp;
        }
        wave_lds_sync();
    }

    // ---- epilogue: streaming state back to HBM ----
    if constexpr (ND > 0) {
        for (int i = lane; i < 2 * M * G::HQ4; i += kWave) {
            const int rail = i / (M * G::HQ4), rem = i % (M * G::HQ4);
            const int s = rem - G::F, pp = rem % M, m = rem / M;
            if (s >= 0) p.dec_state[((size_t)c * 2 + rail) * (ND - 1) + s] = S[pp * G::PSF + G::elem(m) + rail];
        }
    }
    if constexpr (NH > 0) {
        if constexpr (AM == 0) {                                      // AM never ran the Hilbert pair: its state stays
            for (int i = lane; i < 2 * G::HH4; i += kWave) {
                const int rail = i / G::HH4, m = i % G::HH4, s = m - G::FH;
                if (s >= 0) p.fir_state[((size_t)c * 2 + rail) * G::HH + s] = D[rail * G::DLEN + m];
            }
        }
    }
    if (lane == 0) {
        if constexpr (NCO != 0) p.phase[c] = ph0 + p.block_size * step;
        if (p.agc) p.gain[c] = gain;
    }
}

// ------------------------------------------------------------------------------------------
// k_ssb_mfma<ND, 4, NH, TIn, TOut> -- FMA-arithmetic variant with the decimator on the matrix
// cores.  v_mfma_f32_16x16x4_f32 is bit-for-bit a k-ordered fmaf chain (cdna_hip_programming.md
// section 3), i.e. exactly SELENITE_ARITH_FMA's contract, so this kernel is bit-exact against the
// oracle's fmaf restatement.  The FIR is cast as a banded-Toeplitz product
//     D[i][n] = sum_k A[i][k] * B[k][n],   A[i][k] = x[64*i + k],   B[k][n] = cq[k - 4*n]
// rows i = 16 blocks of 16 consecutive outputs of one rail, columns n = the 16 outputs of a block,
// k = 0 .. ND+60 (80 MFMA steps of 4; 256/320 = 80 % of the multiplies are on real taps; a zero
// B entry adds fma(x, 0, acc) = acc exactly).  B (80 VGPRs) is loaded once per kernel; A is one
// ds_read_b32 per MFMA from a flat per-rail LDS image whose 64-sample rows are padded by 2 dwords
// (address = 66*(lane&15) + (lane>>4) + imm: conflict-free for both 32-lane groups).
// The VALU (NCO, Hilbert, AGC, address math) runs beside the MFMA pipe instead of in front of it.
// ------------------------------------------------------------------------------------------
typedef float v4f __attribute__((ext_vector_type(4)));

template <int ND, int M, int NH>
struct GeoM {
    using G = Geo<ND, M, NH>;
    static constexpr int HS = G::HQ4 * M;                 // history samples kept in front (>= ND-1)
    static constexpr int XN = HS + G::T;                  // samples per rail in LDS
    static constexpr int XROWS = XN / 64;
    static constexpr int XLEN = 66 * XROWS;               // padded floats per rail
    static constexpr int KTOT = ND + 4 * 15 + 1;          // padded taps 0..ND  +  shift of 15 outputs
    static constexpr int KS = (KTOT + 3) / 4;             // MFMA k-steps
    static constexpr int oTab = 0;
    static constexpr int oX = 516;
    static constexpr int oD = oX + 2 * XLEN;
    static constexpr int total = oD + 2 * G::DLEN;
    __host__ __device__ static constexpr int phys(int f) { return f + 2 * (f >> 6); }
};

constexpr int kMfmaWaves = 1;      // waves (= channels) per workgroup of k_ssb_mfma (f32 MFMA shares the FP32 ALUs with the VALU, so anti-phased multi-wave groups bring nothing)

template <int NCO, int ND, int M, int NH, typename TIn, typename TOut, int AM = 0>
__global__ __launch_bounds__(64 * kMfmaWaves, 2) void k_ssb_mfma(RxParams p, FusedArgs fa,
                                                                 const float *__restrict__ btab,
                                                                 const TIn *__restrict__ src, TOut *__restrict__ dst)
{
    using G = Geo<ND, M, NH>;
    using GM = GeoM<ND, M, NH>;
    using R = Raw<TIn>;
    static_assert(ND > 0 && M == 4 && G::T % 64 == 0 && GM::HS % 64 == 0, "MFMA decimator: /4, 64-sample rows");
    extern __shared__ __attribute__((aligned(16))) float lds_all[];
    // Eight independent wavefronts (one channel each) per workgroup.  A workgroup's waves are placed
    // on the CU's SIMDs cyclically, so waves w and w+4 share a SIMD: the two halves of the workgroup
    // run in ANTI-PHASE (s_barrier between half-steps): while waves 0-3 issue their MFMA chain, waves
    // 4-7 do their VALU work (Hilbert/AGC/store of the finished pass, NCO staging of the next), then
    // the roles swap.  No data is shared between waves -- the barrier only aligns the phases so the
    // matrix pipe and the VALU of every SIMD are busy at the same time.
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    const int grp = (wave >> fa.grp_shift) & 1;                       // 0: MFMA on even half-steps, 1: on odd
    float *lds = lds_all + wave * GM::total;
    const uint32_t c = blockIdx.x * kMfmaWaves + wave;
    float *tab = lds + GM::oTab;
    float *XI = lds + GM::oX, *XQ = XI + GM::XLEN;
    float *D = lds + GM::oD;
    float *dI = D, *dQ = D + G::DLEN;
    constexpr int NLD = G::T / 128;

    const size_t in_base = (size_t)c * p.in_stride, out_base = (size_t)c * p.out_stride;
    const uint32_t npass = p.nout / G::P;
    typename R::type raw[NLD];
#pragma unroll
    for (int i = 0; i < NLD; ++i) raw[i] = R::load(src, in_base + 128u * i + 2u * lane);

    // Toeplitz B operand: lane l holds B[k = 4*ks + (l>>4)][n = l&15] = cq[k - 4n]
    float B[GM::KS];
#pragma unroll
    for (int ks = 0; ks < GM::KS; ++ks) B[ks] = btab[64 * ks + lane];

    if constexpr (NCO == 1)
        for (int i = lane; i < 513; i += kWave) tab[i] = p.sintab[i];
    // history: flat sample f in [0, HS) is CMSIS state sample s = f - F (older slots meet zero taps)
    batched_fill<2 * GM::HS>(lane, p.dec_state + (size_t)c * 2 * (ND - 1),
        [&](int i) {
            const int rail = i / GM::HS, sidx = i % GM::HS - G::F;
            return sidx >= 0 ? rail * (ND - 1) + sidx : -1;
        },
        [&](int i, float v) { (i / GM::HS ? XQ : XI)[GM::phys(i % GM::HS)] = v; });
    if constexpr (NH > 0) {
        batched_fill<2 * G::HH4>(lane, p.fir_state + (size_t)c * 2 * G::HH,
            [&](int i) {
                const int rail = i / G::HH4, sidx = i % G::HH4 - G::FH;
                return sidx >= 0 ? rail * G::HH + sidx : -1;
            },
            [&](int i, float v) { D[(i / G::HH4) * G::DLEN + i % G::HH4] = v; });
    }
    float hreg[(NH + 63) / 64 ? (NH + 63) / 64 : 1];
#pragma unroll
    for (int v = 0; v < (NH + 63) / 64; ++v) hreg[v] = (64 * v + lane < NH) ? p.hilb_c[64 * v + lane] : 0.0f;
    const uint32_t ph0 = NCO ? p.phase[c] : 0u;
    const uint32_t step = NCO ? p.step[c] : 0u;
    float gain = p.agc ? p.gain[c] : 1.0f;
    const int group = (int)fa.group;
    const int abase = 66 * (lane & 15) + (lane >> 4);                 // A-operand lane base (dwords)

    // ---- V phase, part 2: NCO mix of the prefetched pass into the X image, then prefetch the next ----
    auto stage = [&](uint32_t pass) {
        const uint32_t n0 = pass * G::T;
        float4 lo4[NLD];
        if constexpr (NCO == 2) {
#pragma unroll
            for (int i = 0; i < NLD; ++i)
                lo4[i] = *reinterpret_cast<const float4 *>(p.lo + n0 + 128u * i + 2u * lane);
            __builtin_amdgcn_sched_barrier(0);
        }
#pragma unroll
        for (int i = 0; i < NLD; ++i) {
            const uint32_t n = 128u * i + 2u * lane;
            float2 a, b;
            R::unpack(raw[i], a, b);
            if constexpr (NCO == 2) {
                const float4 l2 = lo4[i];
                a = cmul<0>(a, make_float2(l2.x, l2.y));
                b = cmul<0>(b, make_float2(l2.z, l2.w));
            } else if constexpr (NCO == 1) {
                a = cmul<0>(a, nco_lo<0>(tab, ph0 + (n0 + n) * step));
                b = cmul<0>(b, nco_lo<0>(tab, ph0 + (n0 + n + 1) * step));
            }
            const int f = GM::HS + (int)n;                            // even: (f, f+1) share a row
            const int ph = f + 2 * (f >> 6);
            *reinterpret_cast<float2 *>(XI + ph) = make_float2(a.x, b.x);
            *reinterpret_cast<float2 *>(XQ + ph) = make_float2(a.y, b.y);
        }
        wave_lds_sync();
        if (pass + 1 < npass) {
#pragma unroll
            for (int i = 0; i < NLD; ++i) raw[i] = R::load(src, in_base + n0 + G::T + 128u * i + 2u * lane);
        }
    };
    // ---- M phase: decimator on the matrix cores, 2 accumulator tiles (I, Q), KS steps each ----
    auto mfma_phase = [&]() {
        v4f accI = { 0.0f, 0.0f, 0.0f, 0.0f }, accQ = { 0.0f, 0.0f, 0.0f, 0.0f };
        const float *aI = XI + abase, *aQ = XQ + abase;
        // A operands are fetched one group (GK k-steps, both rails) ahead of the MFMAs that use them
        constexpr int GK = 4, NG = (GM::KS + GK - 1) / GK;
        float bufI[2][GK], bufQ[2][GK];
#pragma unroll
        for (int j = 0; j < GK; ++j) {
            const int off = 4 * j + 2 * ((4 * j) >> 6);
            bufI[0][j] = aI[off];
            bufQ[0][j] = aQ[off];
        }
#pragma unroll
        for (int g = 0; g < NG; ++g) {
            if (g + 1 < NG) {
#pragma unroll
                for (int j = 0; j < GK; ++j) {
                    const int ks = (g + 1) * GK + j;
                    if (ks < GM::KS) {
                        const int off = 4 * ks + 2 * ((4 * ks) >> 6);     // phys(k0), k0 = 4*ks
                        bufI[(g + 1) & 1][j] = aI[off];
                        bufQ[(g + 1) & 1][j] = aQ[off];
                    }
                }
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int j = 0; j < GK; ++j) {
                const int ks = g * GK + j;
                if (ks < GM::KS) {
                    accI = __builtin_amdgcn_mfma_f32_16x16x4f32(bufI[g & 1][j], B[ks], accI, 0, 0, 0);
                    accQ = __builtin_amdgcn_mfma_f32_16x16x4f32(bufQ[g & 1][j], B[ks], accQ, 0, 0, 0);
                }
            }
        }
        // D layout: lane holds rows (lane>>4)*4 + r, column lane&15 -> output 16*row + col
        const int o0 = G::HH4 + 64 * (lane >> 4) + (lane & 15);
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            dI[o0 + 16 * r] = accI[r];
            dQ[o0 + 16 * r] = accQ[r];
        }
        wave_lds_sync();
    };
    // ---- V phase, part 1: Hilbert pair + sideband, AGC, store; history copy-backs ----
    auto finish = [&](uint32_t pass) {
        if (group == 16)
            demod_agc_store<1, 16, ND, M, NH, TOut, AM>(p, fa, dI, dQ, lane, group, hreg, gain, dst, out_base + (size_t)pass * G::P);
        else
            demod_agc_store<1, 0, ND, M, NH, TOut, AM>(p, fa, dI, dQ, lane, group, hreg, gain, dst, out_base + (size_t)pass * G::P);
        wave_lds_sync();
        {
            constexpr int NV = 2 * GM::HS / 2;                        // float2 moves
            constexpr int NK = (NV + 63) / 64;
            float2 t[NK];
#pragma unroll
            for (int k = 0; k < NK; ++k) {
                const int i = k * 64 + lane;
                if (i < NV) {
                    const int rail = i / (GM::HS / 2), f = 2 * (i % (GM::HS / 2));
                    t[k] = *reinterpret_cast<const float2 *>((rail ? XQ : XI) + GM::phys(G::T + f));
                }
            }
            wave_lds_sync();
#pragma unroll
            for (int k = 0; k < NK; ++k) {
                const int i = k * 64 + lane;
                if (i < NV) {
                    const int rail = i / (GM::HS / 2), f = 2 * (i % (GM::HS / 2));
                    *reinterpret_cast<float2 *>((rail ? XQ : XI) + GM::phys(f)) = t[k];
                }
            }
        }
        if constexpr (NH > 0) {
            constexpr int NV = 2 * (G::HH4 / 4);
            float4 tmp;
            const int rail = lane / (G::HH4 / 4), v = lane % (G::HH4 / 4);
            if (lane < NV) tmp = *reinterpret_cast<const float4 *>(D + rail * G::DLEN + G::P + 4 * v);
            wave_lds_sync();
            if (lane < NV) *reinterpret_cast<float4 *>(D + rail * G::DLEN + 4 * v) = tmp;
        }
        wave_lds_sync();
    };

    wave_lds_sync();
    stage(0);
    // half-steps: group 0 runs M(k) at h = 2k and V(k) at h = 2k+1; group 1 one half-step later
    const uint32_t nhalf = 2 * npass + 1;
    // diagnostics: stamp slot s of half-step h for wave w at dbg[w*128 + h*8 + s] (workgroup 0 only)
    uint32_t hcur = 0;
    auto stamp = [&](int slot) {
        if (fa.dbg && blockIdx.x == 0 && lane == 0) fa.dbg[wave * 128 + hcur * 8 + slot] = __builtin_amdgcn_s_memtime();
    };
    for (uint32_t h = 0; h < nhalf; ++h) {
        hcur = h;
        stamp(0);
        if (h >= (uint32_t)grp) {
            const uint32_t hh = h - grp, k = hh >> 1;
            if (k < npass) {
                if ((hh & 1) == 0) {
                    mfma_phase();
                } else {
                    finish(k);
                    stamp(3);
                    if (k + 1 < npass) stage(k + 1);
                }
            }
        }
        stamp(7);
        __builtin_amdgcn_s_barrier();                                 // timing alignment only: no shared data
    }

    for (int i = lane; i < 2 * GM::HS; i += kWave) {
        const int rail = i / GM::HS, f = i % GM::HS, s = f - G::F;
        if (s >= 0) p.dec_state[((size_t)c * 2 + rail) * (ND - 1) + s] = (rail ? XQ : XI)[GM::phys(f)];
    }
    if constexpr (NH > 0) {
        if constexpr (AM == 0) {                                      // AM never ran the Hilbert pair: its state stays
            for (int i = lane; i < 2 * G::HH4; i += kWave) {
                const int rail = i / G::HH4, m = i % G::HH4, s = m - G::FH;
                if (s >= 0) p.fir_state[((size_t)c * 2 + rail) * G::HH + s] = D[rail * G::DLEN + m];
            }
        }
    }
    if (lane == 0) {
        if constexpr (NCO != 0) p.phase[c] = ph0 + p.block_size * step;
        if (p.agc) p.gain[c] = gain;
    }
}

// ------------------------------------------------------------------------------------------
// k_ssb_split16<NCO, ND, 4, NH, TIn, TOut> -- SELENITE_ARITH_SPLIT16: the decimator as a
// split-precision matrix product on the 16-bit matrix cores (v_mfma_f32_16x16x32_f16).
//
// f32 MFMA executes on the FP32 vector ALUs (it ADDS to the VALU time, measured: DESIGN.md 5.1), so
// the only extra throughput on the chip is the separate 16-bit matrix pipe.  Each mixed sample x
// (scaled by 2^8) and each tap c (scaled by 2^SC) is split exactly into f16 hi + lo,
//     x' = xh + xl + O(2^-22 x'),    c' = ch + cl + O(2^-22 c'),
// and the banded-Toeplitz product of k_ssb_mfma is evaluated as  xh*ch + (xh*cl + xl*ch)  with f32
// accumulation inside the MFMA (f16 x f16 products are exact in f32); the dropped xl*cl term is
// 2^-22 relative.  The big and the two small terms use separate accumulators.  The result is
// rescaled by an exact power of two.  Error vs the CMSIS arithmetic: <2e-6 of the block maximum
// (tests), against the north star's 1e-5.  NOT bit-reproducible by a CPU loop -- parity for this
// mode is tolerance-based by construction.
// LDS: four f16 images (I/Q x hi/lo), 64-sample rows padded to 80 halfs: the 16-byte A-fragment read
// of lane l is at 160*(l&15) + 16*(l>>4) + imm, conflict-free for every b128 lane group.
// The streaming state stays exact f32: the last ND-1 mixed samples of a call go to HBM as f32.
// ------------------------------------------------------------------------------------------
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef _Float16 h2 __attribute__((ext_vector_type(2)));

template <int ND, int M, int NH>
struct GeoS {
    using G = Geo<ND, M, NH>;
    static constexpr int HS = G::HQ4 * M;                 // history samples in front
    static constexpr int XN = HS + G::T;
    static constexpr int XROWS = XN / 64;
    static constexpr int IMG = 80 * XROWS;                // halfs per image
    static constexpr int KTOT = ND + 4 * 15 + 1;
    static constexpr int KS = (KTOT + 31) / 32;           // MFMA k-steps of 32
    static constexpr int oTab = 0;                        // floats
    static constexpr int oX = 516;                        // 4 images of IMG halfs = 2*IMG floats
    static constexpr int oD = oX + 2 * IMG;
    static constexpr int total = oD + 2 * G::DLEN;
    static constexpr int XSCALE = 8;                      // samples scaled by 2^8 before the split
    __host__ __device__ static constexpr int phys(int f) { return 80 * (f >> 6) + (f & 63); }
};

template <int NCO, int ND, int M, int NH, typename TIn, typename TOut, int AM = 0>
__global__ __launch_bounds__(64, 2) void k_ssb_split16(RxParams p, FusedArgs fa, const TIn *__restrict__ src,
                                                       TOut *__restrict__ dst)
{
    using G = Geo<ND, M, NH>;
    using GS = GeoS<ND, M, NH>;
    using R = Raw<TIn>;
    static_assert(ND > 0 && M == 4 && NH > 0 && G::T % 64 == 0 && GS::HS % 64 == 0, "split16 decimator: /4 + Hilbert");
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int lane = threadIdx.x;
    const uint32_t c = blockIdx.x;
    float *tab = lds + GS::oTab;
    _Float16 *X = reinterpret_cast<_Float16 *>(lds + GS::oX);     // [rail][hi/lo][IMG]
    float *D = lds + GS::oD;
    float *dI = D, *dQ = D + G::DLEN;
    constexpr int NLD = G::T / 128;

    const size_t in_base = (size_t)c * p.in_stride, out_base = (size_t)c * p.out_stride;
    const uint32_t npass = p.nout / G::P;
    typename R::type raw[NLD];
#pragma unroll
    for (int i = 0; i < NLD; ++i) raw[i] = R::load(src, in_base + 128u * i + 2u * lane);

    // Toeplitz B fragments (8 halfs per lane): [kk][hi/lo]
    h8 Bh[GS::KS], Bl[GS::KS];
    {
        const h8 *bt = static_cast<const h8 *>(fa.btab16);
#pragma unroll
        for (int kk = 0; kk < GS::KS; ++kk) {
            Bh[kk] = bt[(2 * kk + 0) * 64 + lane];
            Bl[kk] = bt[(2 * kk + 1) * 64 + lane];
        }
    }
    float hreg[(NH + 63) / 64];
#pragma unroll
    for (int v = 0; v < (NH + 63) / 64; ++v) hreg[v] = (64 * v + lane < NH) ? p.hilb_c[64 * v + lane] : 0.0f;

    if constexpr (NCO == 1)
        for (int i = lane; i < 513; i += kWave) tab[i] = p.sintab[i];
    const float xs = (float)(1 << GS::XSCALE);
    auto put = [&](int rail, int f, float x0, float x1) {           // samples f (even), f+1 of one rail
        const float a0 = x0 * xs, a1 = x1 * xs;
        const _Float16 h0 = (_Float16)a0, h1 = (_Float16)a1;
        const _Float16 l0 = (_Float16)(a0 - (float)h0), l1 = (_Float16)(a1 - (float)h1);
        const int ph = GS::phys(f);
        *reinterpret_cast<h2 *>(X + (2 * rail + 0) * GS::IMG + ph) = h2{ h0, h1 };
        *reinterpret_cast<h2 *>(X + (2 * rail + 1) * GS::IMG + ph) = h2{ l0, l1 };
    };
    // two mixed samples (I, Q) -> four words, one per image.  Working on (I, Q) pairs keeps the
    // complex multiply, the scaling and both conversions in packed instructions.
    const v2f xs2 = { xs, xs };
    auto put_iq = [&](int f, v2f ma, v2f mb) {                        // samples f (even) and f + 1
        const v2f sa = ma * xs2, sb = mb * xs2;
        const h2 ha = __builtin_convertvector(sa, h2), hb = __builtin_convertvector(sb, h2);
        const h2 la = __builtin_convertvector(sa - __builtin_convertvector(ha, v2f), h2);
        const h2 lb = __builtin_convertvector(sb - __builtin_convertvector(hb, v2f), h2);
        const int ph = GS::phys(f);
        *reinterpret_cast<h2 *>(X + 0 * GS::IMG + ph) = h2{ ha.x, hb.x };
        *reinterpret_cast<h2 *>(X + 1 * GS::IMG + ph) = h2{ la.x, lb.x };
        *reinterpret_cast<h2 *>(X + 2 * GS::IMG + ph) = h2{ ha.y, hb.y };
        *reinterpret_cast<h2 *>(X + 3 * GS::IMG + ph) = h2{ la.y, lb.y };
    };
    // history: flat sample f in [0, HS) is CMSIS state sample s = f - F (older slots meet zero taps).
    // All state loads of the prologue are issued before the first use, so the workgroup pays one
    // memory round trip for them instead of one per loop iteration.
    {
        constexpr int NHI = 2 * (GS::HS / 2) / kWave;                    // pairs of history samples per lane
        constexpr int NFI = 2 * G::HH4 / kWave;
        static_assert(2 * (GS::HS / 2) % kWave == 0 && 2 * G::HH4 % kWave == 0, "prologue fills are whole wave-loads");
        // branch-free (clamped index, masked value): see batched_fill
        const float *stD = p.dec_state + (size_t)c * 2 * (ND - 1);
        const float *stF = p.fir_state + (size_t)c * 2 * G::HH;
        float h0[NHI], h1[NHI], fv[NFI];
#pragma unroll
        for (int j = 0; j < NHI; ++j) {
            const int i = j * kWave + lane;
            const int rail = i / (GS::HS / 2), f = 2 * (i % (GS::HS / 2));
            const int s0 = f - G::F, s1 = f + 1 - G::F;
            const float x0 = stD[rail * (ND - 1) + (s0 < 0 ? 0 : s0)], x1 = stD[rail * (ND - 1) + (s1 < 0 ? 0 : s1)];
            h0[j] = s0 < 0 ? 0.0f : x0;
            h1[j] = s1 < 0 ? 0.0f : x1;
        }
#pragma unroll
        for (int j = 0; j < NFI; ++j) {
            const int i = j * kWave + lane;
            const int rail = i / G::HH4, sidx = i % G::HH4 - G::FH;
            const float x = stF[rail * G::HH + (sidx < 0 ? 0 : sidx)];
            fv[j] = sidx < 0 ? 0.0f : x;
        }
#pragma unroll
        for (int j = 0; j < NHI; ++j) {
            const int i = j * kWave + lane;
            put(i / (GS::HS / 2), 2 * (i % (GS::HS / 2)), h0[j], h1[j]);
        }
#pragma unroll
        for (int j = 0; j < NFI; ++j) {
            const int i = j * kWave + lane;
            D[(i / G::HH4) * G::DLEN + i % G::HH4] = fv[j];
        }
    }
    const uint32_t ph0 = NCO ? p.phase[c] : 0u;
    const uint32_t step = NCO ? p.step[c] : 0u;
    float gain = p.agc ? p.gain[c] : 1.0f;
    const int group = (int)fa.group;
    const int abase = 80 * (lane & 15) + 8 * (lane >> 4);             // A-fragment lane base (halfs)
    wave_lds_sync();

    // shared LO of a pass (NCO == 2) is fetched from L2 one stage early -- pass 0 under the
    // prologue, pass p+1 right behind the matrix stage of pass p (whose fragment and accumulator
    // registers are dead by then) -- so the mix stage does not open with an exposed L2 round trip
    float4 lo4[NLD];
    if constexpr (NCO == 2) {
#pragma unroll
        for (int i = 0; i < NLD; ++i) lo4[i] = *reinterpret_cast<const float4 *>(p.lo + 128u * i + 2u * lane);
    }
    for (uint32_t pass = 0; pass < npass; ++pass) {
        const uint32_t n0 = pass * G::T;
        const bool last = (pass + 1 == npass);
        // ---- 1. NCO mix, f16 hi/lo split, four LDS images; exact f32 state from the last pass ----
#pragma unroll
        for (int i = 0; i < NLD; ++i) {
            const uint32_t n = 128u * i + 2u * lane;
            float2 a, b;
            R::unpack(raw[i], a, b);
            v2f ma, mb;                                               // mixed samples as (I, Q) pairs
            if constexpr (NCO == 2) {
                const float4 l2 = lo4[i];
                ma = cmul_pk(v2f{ a.x, a.y }, v2f{ l2.x, l2.y });
                mb = cmul_pk(v2f{ b.x, b.y }, v2f{ l2.z, l2.w });
            } else if constexpr (NCO == 1) {
                const float2 la = nco_lo<0>(tab, ph0 + (n0 + n) * step), lb = nco_lo<0>(tab, ph0 + (n0 + n + 1) * step);
                ma = cmul_pk(v2f{ a.x, a.y }, v2f{ la.x, la.y });
                mb = cmul_pk(v2f{ b.x, b.y }, v2f{ lb.x, lb.y });
            } else {
                ma = v2f{ a.x, a.y };
                mb = v2f{ b.x, b.y };
            }
            put_iq(GS::HS + (int)n, ma, mb);
            if (128 * (i + 1) > G::T - (ND - 1) && last) {            // CMSIS pState: last ND-1 mixed samples, f32
                // (the first operand is a compile-time constant of the unrolled loop: load groups
                // in front of the state window carry no store code at all)
                const int s0 = (int)n - (G::T - (ND - 1));
                float *stI = p.dec_state + ((size_t)c * 2 + 0) * (ND - 1), *stQ = stI + (ND - 1);
                if (s0 >= 0) { stI[s0] = ma.x; stQ[s0] = ma.y; }
                if (s0 + 1 >= 0) { stI[s0 + 1] = mb.x; stQ[s0 + 1] = mb.y; }
            }
        }
        wave_lds_sync();
        if (pass + 1 < npass) {
#pragma unroll
            for (int i = 0; i < NLD; ++i) raw[i] = R::load(src, in_base + n0 + G::T + 128u * i + 2u * lane);
        }
        // ---- 2. decimator: 3 f16 MFMAs per k-step and rail (hi*hi | hi*lo + lo*hi) ----
        {
            v4f bigI = { 0.0f, 0.0f, 0.0f, 0.0f }, smlI = { 0.0f, 0.0f, 0.0f, 0.0f };
            v4f bigQ = { 0.0f, 0.0f, 0.0f, 0.0f }, smlQ = { 0.0f, 0.0f, 0.0f, 0.0f };
            const _Float16 *xIh = X + 0 * GS::IMG + abase, *xIl = X + 1 * GS::IMG + abase;
            const _Float16 *xQh = X + 2 * GS::IMG + abase, *xQl = X + 3 * GS::IMG + abase;
            // A fragments are read one k-step ahead of the MFMAs that consume them (explicit software
            // pipeline + scheduling groups: left alone, the scheduler issues a fragment read right in
            // front of its MFMA and the wave eats the LDS latency twice per k-step)
            auto offA = [](int kk) { return 80 * (kk >> 1) + 32 * (kk & 1); };   // phys(32*kk): rows never straddle
            h8 aIh = *reinterpret_cast<const h8 *>(xIh + offA(0)), aIl = *reinterpret_cast<const h8 *>(xIl + offA(0));
            h8 aQh = *reinterpret_cast<const h8 *>(xQh + offA(0)), aQl = *reinterpret_cast<const h8 *>(xQl + offA(0));
            __builtin_amdgcn_sched_group_barrier(0x100, 4, 0);                            // the 4 reads of k-step 0
#pragma unroll
            for (int kk = 0; kk < GS::KS; ++kk) {
                h8 nIh = aIh, nIl = aIl, nQh = aQh, nQl = aQl;
                if (kk + 1 < GS::KS) {
                    const int off = offA(kk + 1);
                    nIh = *reinterpret_cast<const h8 *>(xIh + off); nIl = *reinterpret_cast<const h8 *>(xIl + off);
                    nQh = *reinterpret_cast<const h8 *>(xQh + off); nQl = *reinterpret_cast<const h8 *>(xQl + off);
                }
                bigI = __builtin_amdgcn_mfma_f32_16x16x32_f16(aIh, Bh[kk], bigI, 0, 0, 0);
                bigQ = __builtin_amdgcn_mfma_f32_16x16x32_f16(aQh, Bh[kk], bigQ, 0, 0, 0);
                smlI = __builtin_amdgcn_mfma_f32_16x16x32_f16(aIh, Bl[kk], smlI, 0, 0, 0);
                smlQ = __builtin_amdgcn_mfma_f32_16x16x32_f16(aQh, Bl[kk], smlQ, 0, 0, 0);
                smlI = __builtin_amdgcn_mfma_f32_16x16x32_f16(aIl, Bh[kk], smlI, 0, 0, 0);
                smlQ = __builtin_amdgcn_mfma_f32_16x16x32_f16(aQl, Bh[kk], smlQ, 0, 0, 0);
                if (kk + 1 < GS::KS) __builtin_amdgcn_sched_group_barrier(0x100, 4, 0);   // 4 DS reads (k-step kk+1)
                __builtin_amdgcn_sched_group_barrier(0x008, 6, 0);                        // 6 MFMAs (k-step kk)
                aIh = nIh; aIl = nIl; aQh = nQh; aQl = nQl;
            }
            const int o0 = G::HH4 + 64 * (lane >> 4) + (lane & 15);
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                dI[o0 + 16 * r] = (bigI[r] + smlI[r]) * fa.split_post;
                dQ[o0 + 16 * r] = (bigQ[r] + smlQ[r]) * fa.split_post;
            }
            wave_lds_sync();
        }
        if constexpr (NCO == 2) {
            if (pass + 1 < npass) {
#pragma unroll
                for (int i = 0; i < NLD; ++i)
                    lo4[i] = *reinterpret_cast<const float4 *>(p.lo + n0 + G::T + 128u * i + 2u * lane);
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        // ---- 3-5. Hilbert pair + sideband, AGC, store ----
        if (group == 16)
            demod_agc_store<1, 16, ND, M, NH, TOut, AM>(p, fa, dI, dQ, lane, group, hreg, gain, dst, out_base + (size_t)pass * G::P);
        else
            demod_agc_store<1, 0, ND, M, NH, TOut, AM>(p, fa, dI, dQ, lane, group, hreg, gain, dst, out_base + (size_t)pass * G::P);
        wave_lds_sync();
        // ---- 6. history copy-back: last HS samples of every image to its front (8-byte moves) ----
        {
            constexpr int NV = 4 * (GS::HS / 4), NK = (NV + 63) / 64;
            uint2 t[NK];
#pragma unroll
            for (int k = 0; k < NK; ++k) {
                const int i = k * 64 + lane;
                if (i < NV) {
                    const int img = i / (GS::HS / 4), f = 4 * (i % (GS::HS / 4));
                    t[k] = *reinterpret_cast<const uint2 *>(X + img * GS::IMG + GS::phys(G::T + f));
                }
            }
            wave_lds_sync();
#pragma unroll
            for (int k = 0; k < NK; ++k) {
                const int i = k * 64 + lane;
                if (i < NV) {
                    const int img = i / (GS::HS / 4), f = 4 * (i % (GS::HS / 4));
                    *reinterpret_cast<uint2 *>(X + img * GS::IMG + GS::phys(f)) = t[k];
                }
            }
        }
        {
            constexpr int NV = 2 * (G::HH4 / 4);
            float4 tmp;
            const int rail = lane / (G::HH4 / 4), v = lane % (G::HH4 / 4);
            if (lane < NV) tmp = *reinterpret_cast<const float4 *>(D + rail * G::DLEN + G::P + 4 * v);
            wave_lds_sync();
            if (lane < NV) *reinterpret_cast<float4 *>(D + rail * G::DLEN + 4 * v) = tmp;
        }
        wave_lds_sync();
    }

    if constexpr (AM == 0) {                                          // AM never ran the Hilbert pair: its state stays
        for (int i = lane; i < 2 * G::HH4; i += kWave) {
            const int rail = i / G::HH4, m = i % G::HH4, s = m - G::FH;
            if (s >= 0) p.fir_state[((size_t)c * 2 + rail) * G::HH + s] = D[rail * G::DLEN + m];
        }
    }
    if (lane == 0) {
        if constexpr (NCO != 0) p.phase[c] = ph0 + p.block_size * step;
        if (p.agc) p.gain[c] = gain;
    }
}

// ------------------------------------------------------------------------------------------
// k_hilb_split16<NCO, NH, TIn, TOut, AM> -- SELENITE_ARITH_SPLIT16 for the no-decimator shapes
// (BASELINE cfg1 / cfg2 / cfg5: M = 1, DSP block 256): the Hilbert FIR on the 16-bit matrix pipe.
// Those shapes are VALU-bound by the Hilbert tap loop (64 non-zero taps of 127 per output); as a
// banded-Toeplitz product  D[i][m] = sum_k A[i][k] B[k][m],  A[i][k] = st[16 i + k],  B[k][m] = h[k - m]
// (st = [NH-1 history | 256 new samples] of the Q rail, 16 rows of 16 outputs, K = NH + 15) it is
// 3 MFMAs per k-step of 32 with the f16 hi/lo split of k_ssb_split16 (samples x 2^8, taps x 2^SC,
// xh*ch + xh*cl + xl*ch, f32 accumulation): 15 v_mfma_f32_16x16x32_f16 per pass instead of 128
// v_pk_fma + 64 v_readlane.  The I rail is a pure delay (unit-impulse FIR) and stays f32; the
// streaming state is written from the f32 mixed samples in registers, so it stays bit-exact.
// The MFMA result layout (lane holds outputs 64(l>>4) + 16 r + (l&15)) goes through a 1 KB LDS
// transpose so the audio leaves as one coalesced float4 per lane.  Tolerance-based like k_ssb_split16.
// ------------------------------------------------------------------------------------------
template <int NH>
struct GeoH {
    static constexpr int HH = NH - 1;                              // history samples (even)
    static constexpr int KS = (NH + 15 + 31) / 32;                 // MFMA k-steps of 32
    static constexpr int XN = 240 + 32 * KS;                       // highest image index read + 1
    __host__ __device__ static constexpr int phys(int u) { return u + 8 * (u >> 7); }   // 16 B pad per 128 samples
    static constexpr int IMG = ((XN + 8 * (XN >> 7) + 8) + 7) & ~7;  // halfs per image
    static constexpr int DIL = HH + 256;                           // f32 I rail: [history | new]
    static constexpr int oTab = 0, oX = 516, oDI = oX + IMG /* 2 images of IMG halfs */, oO = oDI + DIL + 2, total = oO + 256;
    static_assert(HH % 2 == 0 && HH <= 256, "Hilbert history");
};

template <int NCO, int NH, typename TIn, typename TOut, int AM = 0>
__global__ __launch_bounds__(64, 2) void k_hilb_split16(RxParams p, FusedArgs fa, const TIn *__restrict__ src,
                                                        TOut *__restrict__ dst)
{
    using GH = GeoH<NH>;
    using R = Raw<TIn>;
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int lane = threadIdx.x;
    const uint32_t c = blockIdx.x;
    float *tab = lds + GH::oTab;
    _Float16 *Xh = reinterpret_cast<_Float16 *>(lds + GH::oX), *Xl = Xh + GH::IMG;
    float *dI = lds + GH::oDI, *O = lds + GH::oO;
    const size_t in_base = (size_t)c * p.in_stride, out_base = (size_t)c * p.out_stride;
    const uint32_t npass = p.nout / 256;
    typename R::type raw[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) raw[i] = R::load(src, in_base + 128u * i + 2u * lane);

    h8 Bh[GH::KS], Bl[GH::KS];
    {
        const h8 *bt = static_cast<const h8 *>(fa.btab16);
#pragma unroll
        for (int kk = 0; kk < GH::KS; ++kk) {
            Bh[kk] = bt[(2 * kk + 0) * 64 + lane];
            Bl[kk] = bt[(2 * kk + 1) * 64 + lane];
        }
    }
    if constexpr (NCO == 1)
        for (int i = lane; i < 513; i += kWave) tab[i] = p.sintab[i];
    const float xs = (float)(1 << 8);
    auto put = [&](int u, float x0, float x1) {                     // image slots u (even), u + 1 of the Q rail
        const float a0 = x0 * xs, a1 = x1 * xs;
        const _Float16 h0 = (_Float16)a0, h1 = (_Float16)a1;
        const _Float16 l0 = (_Float16)(a0 - (float)h0), l1 = (_Float16)(a1 - (float)h1);
        const int ph = GH::phys(u);
        *reinterpret_cast<h2 *>(Xh + ph) = h2{ h0, h1 };
        *reinterpret_cast<h2 *>(Xl + ph) = h2{ l0, l1 };
    };
    // the read-only slack behind the samples meets zero taps only, but must hold finite numbers
    for (int u = GH::HH + 256 + 2 * lane; u < GH::XN; u += 2 * kWave) put(u, 0.0f, 0.0f);
    {   // state: I history (f32), Q history (split); branch-free loads (batched_fill)
        const float *stI = p.fir_state + (size_t)c * 2 * GH::HH, *stQ = stI + GH::HH;
        const int u = 2 * lane < GH::HH ? 2 * lane : GH::HH - 2;
        const float i0 = stI[u], i1 = stI[u + 1], q0 = stQ[u], q1 = stQ[u + 1];
        if (2 * lane < GH::HH) {
            *reinterpret_cast<float2 *>(dI + u) = make_float2(i0, i1);
            put(u, q0, q1);
        }
    }
    const uint32_t ph0 = NCO ? p.phase[c] : 0u, step = NCO ? p.step[c] : 0u;
    float gain = p.agc ? p.gain[c] : 1.0f;
    const int mcol = lane & 15, rg = lane >> 4;
    wave_lds_sync();

    for (uint32_t pass = 0; pass < npass; ++pass) {
        const uint32_t n0 = pass * 256u;
        const bool last = (pass + 1 == npass);
        // ---- 1. NCO mix; I rail f32, Q rail split into the f16 images; exact f32 state from the last pass ----
        float4 lo4[2];
        if constexpr (NCO == 2) {
#pragma unroll
            for (int i = 0; i < 2; ++i) lo4[i] = *reinterpret_cast<const float4 *>(p.lo + n0 + 128u * i + 2u * lane);
        }
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const uint32_t n = 128u * i + 2u * lane;
            float2 a, b;
            R::unpack(raw[i], a, b);
            if constexpr (NCO == 2) {
                a = cmul<0>(a, make_float2(lo4[i].x, lo4[i].y));
                b = cmul<0>(b, make_float2(lo4[i].z, lo4[i].w));
            } else if constexpr (NCO == 1) {
                a = cmul<0>(a, nco_lo<0>(tab, ph0 + (n0 + n) * step));
                b = cmul<0>(b, nco_lo<0>(tab, ph0 + (n0 + n + 1) * step));
            }
            if constexpr (AM != 0) {
                *reinterpret_cast<float2 *>(O + n) = make_float2(cmag<0>(a.x, a.y), cmag<0>(b.x, b.y));
            } else {
                *reinterpret_cast<float2 *>(dI + GH::HH + n) = make_float2(a.x, b.x);
                put(GH::HH + (int)n, a.y, b.y);
                if (128 * (i + 1) > 256 - GH::HH && last) {            // arm_fir_f32 pState tails: last NH-1 samples, f32
                    const int s0 = (int)n - (256 - GH::HH);
                    float *stI = p.fir_state + (size_t)c * 2 * GH::HH, *stQ = stI + GH::HH;
                    if (s0 >= 0) { stI[s0] = a.x; stQ[s0] = a.y; stI[s0 + 1] = b.x; stQ[s0 + 1] = b.y; }
                }
            }
        }
        wave_lds_sync();
        if (pass + 1 < npass) {
#pragma unroll
            for (int i = 0; i < 2; ++i) raw[i] = R::load(src, in_base + n0 + 256u + 128u * i + 2u * lane);
        }
        if constexpr (AM == 0) {
            // ---- 2. Hilbert FIR of the Q rail: 3 f16 MFMAs per k-step ----
            v4f big = { 0.0f, 0.0f, 0.0f, 0.0f }, sml = { 0.0f, 0.0f, 0.0f, 0.0f };
#pragma unroll
            for (int kk = 0; kk < GH::KS; ++kk) {
                const int u = 16 * mcol + 8 * rg + 32 * kk;                // A[i = l&15][k = 32kk + 8(l>>4) ..+7] = st[16 i + k]
                const int ph = u + 8 * (u >> 7);
                const h8 ah = *reinterpret_cast<const h8 *>(Xh + ph), al = *reinterpret_cast<const h8 *>(Xl + ph);
                big = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, Bh[kk], big, 0, 0, 0);
                sml = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, Bl[kk], sml, 0, 0, 0);
                sml = __builtin_amdgcn_mfma_f32_16x16x32_f16(al, Bh[kk], sml, 0, 0, 0);
            }
            // ---- 3. delay on I, sideband combine; transpose through LDS ----
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int n = 64 * rg + 16 * r + mcol;                     // D[row 4 rg + r][col mcol]
                const float q2 = (big[r] + sml[r]) * fa.split_post;
                const float i2 = dI[n + fa.delay_idx] + 0.0f;
                O[n] = fa.upper ? (i2 - q2) : (i2 + q2);
            }
            wave_lds_sync();
        }
        // ---- 4.-5. AGC on the DSP block (= the pass), coalesced store ----
        const float4 o4 = *reinterpret_cast<const float4 *>(O + 4 * lane);
        float au[4] = { o4.x, o4.y, o4.z, o4.w };
        if (p.agc) {
            float m = fmaxf(fmaxf(fabsf(au[0]), fabsf(au[1])), fmaxf(fabsf(au[2]), fabsf(au[3])));
            m = wave_max(m);
            gain = agc_update<0>(p.agcp, gain, m);
#pragma unroll
            for (int r = 0; r < 4; ++r) au[r] = au[r] * gain;
        }
        const size_t o = out_base + (size_t)n0 + 4 * lane;
        if constexpr (sizeof(TOut) == 4) {
            *reinterpret_cast<float4 *>(reinterpret_cast<float *>(dst) + o) = make_float4(au[0], au[1], au[2], au[3]);
        } else {
            short4 s4;
            s4.x = float_to_q15(au[0]); s4.y = float_to_q15(au[1]);
            s4.z = float_to_q15(au[2]); s4.w = float_to_q15(au[3]);
            *reinterpret_cast<short4 *>(reinterpret_cast<int16_t *>(dst) + o) = s4;
        }
        // ---- 6. history: last NH-1 samples of the I rail and of both images to the front ----
        if constexpr (AM == 0) {
            float2 ti = make_float2(0.0f, 0.0f);
            uint32_t th = 0, tl = 0;
            const int v = 2 * lane;                                         // pair (v, v+1), v < HH
            if (v < GH::HH) {
                ti = *reinterpret_cast<const float2 *>(dI + 256 + v);
                th = *reinterpret_cast<const uint32_t *>(Xh + GH::phys(256 + v));
                tl = *reinterpret_cast<const uint32_t *>(Xl + GH::phys(256 + v));
            }
            wave_lds_sync();
            if (v < GH::HH) {
                *reinterpret_cast<float2 *>(dI + v) = ti;
                *reinterpret_cast<uint32_t *>(Xh + GH::phys(v)) = th;
                *reinterpret_cast<uint32_t *>(Xl + GH::phys(v)) = tl;
            }
        }
        wave_lds_sync();
    }
    if (lane == 0) {
        if constexpr (NCO != 0) p.phase[c] = ph0 + p.block_size * step;
        if (p.agc) p.gain[c] = gain;
    }
}

// LO[n] = (cos x, -sin x), x from the integer phase phase0 + n*step: the NCO of DESIGN.md section 2,
// evaluated once per call when every channel shares step and phase.
__global__ __launch_bounds__(256) void k_lo_table(float2 *lo, const float *sintab, uint32_t phase0, uint32_t step,
                                                  uint32_t nsamp)
{
    __shared__ float tab[516];
    for (uint32_t i = threadIdx.x; i < 513; i += blockDim.x) tab[i] = sintab[i];
    __syncthreads();
    const uint32_t n = blockIdx.x * blockDim.x + threadIdx.x;
    if (n < nsamp) lo[n] = nco_lo<0>(tab, phase0 + n * step);
}

hipError_t launch_lo_table(float2 *lo, const float *sintab, uint32_t phase0, uint32_t step, uint32_t nsamp,
                           hipStream_t st)
{
    hipLaunchKernelGGL(k_lo_table, dim3((nsamp + 255) / 256), dim3(256), 0, st, lo, sintab, phase0, step, nsamp);
    return hipGetLastError();
}

// ------------------------------------------------------------------------------------------
// host side: plan + dispatch
// ------------------------------------------------------------------------------------------
template <int ND, int M, int NH>
static bool shape_is(const selenite_rx_config &g)
{
    return (int)g.nd_taps == ND && (int)g.decim == M && (int)g.nh_taps == NH;
}

template <int ND, int M, int NH>
static hipError_t build_tables(const selenite_rx_config &g, FusedPlan &plan)
{
    using G = Geo<ND, M, NH>;
    if constexpr (ND > 0) {
        std::vector<float> cq((size_t)64 * G::NCR, 0.0f);
        for (int k = 0; k < ND; ++k) cq[(size_t)k + G::F] = g.dec_coeffs[k];
        hipError_t e = hipMalloc((void **)&plan.d_cq, cq.size() * sizeof(float));
        if (e != hipSuccess) return e;
        e = hipMemcpy(plan.d_cq, cq.data(), cq.size() * sizeof(float), hipMemcpyHostToDevice);
        if (e != hipSuccess) return e;
        if constexpr (M == 4) {
            // Toeplitz operand of k_ssb_mfma: btab[ks][lane] = cq[k - 4n], k = 4ks + (lane>>4), n = lane&15
            using GM = GeoM<ND, M, NH>;
            std::vector<float> bt((size_t)64 * GM::KS, 0.0f);
            for (int ks = 0; ks < GM::KS; ++ks)
                for (int l = 0; l < 64; ++l) {
                    const int idx = 4 * ks + (l >> 4) - 4 * (l & 15);
                    if (idx >= 0 && idx <= ND) bt[(size_t)64 * ks + l] = cq[idx];
                }
            e = hipMalloc((void **)&plan.d_btab, bt.size() * sizeof(float));
            if (e != hipSuccess) return e;
            e = hipMemcpy(plan.d_btab, bt.data(), bt.size() * sizeof(float), hipMemcpyHostToDevice);
            if (e != hipSuccess) return e;
            if constexpr (NH > 0) {
                // SELENITE_ARITH_SPLIT16: taps scaled by 2^SC (largest |tap| lands in [2^9, 2^10)), split into
                // f16 hi + lo; fragment of lane l at k-step kk: 8 halfs B[k = 32kk + 8(l>>4) + j][n = l&15]
                using GS = GeoS<ND, M, NH>;
                float cmax = 0.0f;
                for (int k = 0; k < ND; ++k) cmax = std::fmax(cmax, std::fabs(g.dec_coeffs[k]));
                int ex = 0;
                if (cmax > 0.0f) std::frexp(cmax, &ex);                       // cmax = m * 2^ex, m in [0.5, 1)
                const int SC = 10 - ex;
                std::vector<_Float16> b16((size_t)GS::KS * 2 * 64 * 8, (_Float16)0.0f);
                for (int kk = 0; kk < GS::KS; ++kk)
                    for (int l = 0; l < 64; ++l)
                        for (int j = 0; j < 8; ++j) {
                            const int idx = 32 * kk + 8 * (l >> 4) + j - 4 * (l & 15);
                            float cv = 0.0f;
                            if (idx >= 0 && idx <= ND) cv = std::ldexp(cq[idx], SC);
                            const _Float16 hi = (_Float16)cv;
                            const _Float16 lo = (_Float16)(cv - (float)hi);
                            b16[(((size_t)2 * kk + 0) * 64 + l) * 8 + j] = hi;
                            b16[(((size_t)2 * kk + 1) * 64 + l) * 8 + j] = lo;
                        }
                e = hipMalloc(&plan.d_btab16, b16.size() * sizeof(_Float16));
                if (e != hipSuccess) return e;
                e = hipMemcpy(plan.d_btab16, b16.data(), b16.size() * sizeof(_Float16), hipMemcpyHostToDevice);
                if (e != hipSuccess) return e;
                plan.split_post = std::ldexp(1.0f, -(SC + GS::XSCALE));
            }
            return hipSuccess;
        }
    }
    if constexpr (ND == 0 && M == 1 && NH > 0) {
        // k_hilb_split16: Hilbert taps scaled by 2^SC, f16 hi + lo; fragment of lane l at k-step kk:
        // 8 halfs B[k = 32kk + 8(l>>4) + j][m = l&15] = h[k - m]
        using GH = GeoH<NH>;
        float cmax = 0.0f;
        for (int k = 0; k < NH; ++k) cmax = std::fmax(cmax, std::fabs(g.hilb_coeffs[k]));
        int ex = 0;
        if (cmax > 0.0f) std::frexp(cmax, &ex);
        const int SC = 10 - ex;
        std::vector<_Float16> b16((size_t)GH::KS * 2 * 64 * 8, (_Float16)0.0f);
        for (int kk = 0; kk < GH::KS; ++kk)
            for (int l = 0; l < 64; ++l)
                for (int j = 0; j < 8; ++j) {
                    const int idx = 32 * kk + 8 * (l >> 4) + j - (l & 15);
                    float cv = 0.0f;
                    if (idx >= 0 && idx < NH) cv = std::ldexp(g.hilb_coeffs[idx], SC);
                    const _Float16 hi = (_Float16)cv;
                    const _Float16 lo = (_Float16)(cv - (float)hi);
                    b16[(((size_t)2 * kk + 0) * 64 + l) * 8 + j] = hi;
                    b16[(((size_t)2 * kk + 1) * 64 + l) * 8 + j] = lo;
                }
        hipError_t e = hipMalloc(&plan.d_btab16, b16.size() * sizeof(_Float16));
        if (e != hipSuccess) return e;
        e = hipMemcpy(plan.d_btab16, b16.data(), b16.size() * sizeof(_Float16), hipMemcpyHostToDevice);
        if (e != hipSuccess) return e;
        plan.split_post = std::ldexp(1.0f, -(SC + 8));
    }
    return hipSuccess;
}

template <int ARITH, int ND, int M, int NH, typename TIn, typename TOut>
static hipError_t launch_one(const RxParams &p, const FusedArgs &fa, const void *src, void *dst, hipStream_t st)
{
    using G = Geo<ND, M, NH>;
    constexpr size_t lds = (size_t)G::total * sizeof(float);
    auto k = fa.am ? (p.nco == 2 ? k_ssb_fused<ARITH, 2, ND, M, NH, TIn, TOut, 1>
                         : (p.nco == 1 ? k_ssb_fused<ARITH, 1, ND, M, NH, TIn, TOut, 1> : k_ssb_fused<ARITH, 0, ND, M, NH, TIn, TOut, 1>))
                   : (p.nco == 2 ? k_ssb_fused<ARITH, 2, ND, M, NH, TIn, TOut, 0>
                         : (p.nco == 1 ? k_ssb_fused<ARITH, 1, ND, M, NH, TIn, TOut, 0> : k_ssb_fused<ARITH, 0, ND, M, NH, TIn, TOut, 0>));
    if constexpr (lds > 48 * 1024) {
        static bool once = false;
        if (!once) {
            hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(k),
                                               hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
            if (e != hipSuccess) return e;
            once = true;
        }
    }
    hipLaunchKernelGGL(k, dim3(p.channels), dim3(64), lds, st, p, fa, static_cast<const TIn *>(src),
                       static_cast<TOut *>(dst));
    return hipGetLastError();
}

template <int ND, int M, int NH, typename TIn, typename TOut>
static hipError_t launch_mfma(const RxParams &p, const FusedArgs &fa, const float *btab, const void *src, void *dst,
                              hipStream_t st)
{
    using GM = GeoM<ND, M, NH>;
    constexpr size_t lds = (size_t)kMfmaWaves * GM::total * sizeof(float);
    static_assert(lds <= 160 * 1024, "k_ssb_mfma LDS image");
    auto k = fa.am ? (p.nco == 2 ? k_ssb_mfma<2, ND, M, NH, TIn, TOut, 1>
                         : (p.nco == 1 ? k_ssb_mfma<1, ND, M, NH, TIn, TOut, 1> : k_ssb_mfma<0, ND, M, NH, TIn, TOut, 1>))
                   : (p.nco == 2 ? k_ssb_mfma<2, ND, M, NH, TIn, TOut, 0>
                         : (p.nco == 1 ? k_ssb_mfma<1, ND, M, NH, TIn, TOut, 0> : k_ssb_mfma<0, ND, M, NH, TIn, TOut, 0>));
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(k), hipFuncAttributeMaxDynamicSharedMemorySize,
                                       (int)lds);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(k, dim3(p.channels / kMfmaWaves), dim3(64 * kMfmaWaves), lds, st, p, fa, btab,
                       static_cast<const TIn *>(src), static_cast<TOut *>(dst));
    return hipGetLastError();
}

template <int ND, int M, int NH, typename TIn, typename TOut>
static hipError_t launch_split16(const RxParams &p, const FusedArgs &fa, const void *src, void *dst, hipStream_t st)
{
    using GS = GeoS<ND, M, NH>;
    constexpr size_t lds = (size_t)GS::total * sizeof(float);
    static_assert(lds <= 48 * 1024, "k_ssb_split16 LDS image");
    auto k = fa.am ? (p.nco == 2 ? k_ssb_split16<2, ND, M, NH, TIn, TOut, 1>
                         : (p.nco == 1 ? k_ssb_split16<1, ND, M, NH, TIn, TOut, 1> : k_ssb_split16<0, ND, M, NH, TIn, TOut, 1>))
                   : (p.nco == 2 ? k_ssb_split16<2, ND, M, NH, TIn, TOut, 0>
                         : (p.nco == 1 ? k_ssb_split16<1, ND, M, NH, TIn, TOut, 0> : k_ssb_split16<0, ND, M, NH, TIn, TOut, 0>));
    hipLaunchKernelGGL(k, dim3(p.channels), dim3(64), lds, st, p, fa, static_cast<const TIn *>(src),
                       static_cast<TOut *>(dst));
    return hipGetLastError();
}

template <int NH, typename TIn, typename TOut>
static hipError_t launch_hilb16(const RxParams &p, const FusedArgs &fa, const void *src, void *dst, hipStream_t st)
{
    using GH = GeoH<NH>;
    constexpr size_t lds = (size_t)GH::total * sizeof(float);
    static_assert(lds <= 48 * 1024, "k_hilb_split16 LDS image");
    auto k = fa.am ? (p.nco == 2 ? k_hilb_split16<2, NH, TIn, TOut, 1>
                                 : (p.nco == 1 ? k_hilb_split16<1, NH, TIn, TOut, 1> : k_hilb_split16<0, NH, TIn, TOut, 1>))
                   : (p.nco == 2 ? k_hilb_split16<2, NH, TIn, TOut, 0>
                                 : (p.nco == 1 ? k_hilb_split16<1, NH, TIn, TOut, 0> : k_hilb_split16<0, NH, TIn, TOut, 0>));
    hipLaunchKernelGGL(k, dim3(p.channels), dim3(64), lds, st, p, fa, static_cast<const TIn *>(src),
                       static_cast<TOut *>(dst));
    return hipGetLastError();
}

template <int ND, int M, int NH>
static hipError_t launch_shape(const RxParams &p, const FusedArgs &fa, const FusedPlan &plan, int arith,
                               const void *src, bool src_q15, void *dst, bool dst_q15, hipStream_t st)
{
    if (src_q15 != dst_q15) return hipErrorNotSupported;
    if constexpr (ND > 0 && M == 4 && NH > 0) {
        if (arith == SELENITE_ARITH_SPLIT16 && plan.d_btab16) {
            if (src_q15) return launch_split16<ND, M, NH, int16_t, int16_t>(p, fa, src, dst, st);
            return launch_split16<ND, M, NH, float, float>(p, fa, src, dst, st);
        }
    }
    if constexpr (ND == 0 && M == 1 && NH > 0) {
        if (arith == SELENITE_ARITH_SPLIT16 && plan.d_btab16 && fa.group == 64) {
            if (src_q15) return launch_hilb16<NH, int16_t, int16_t>(p, fa, src, dst, st);
            return launch_hilb16<NH, float, float>(p, fa, src, dst, st);
        }
    }
    if constexpr (ND > 0 && M == 4) {
        if (arith != SELENITE_ARITH_CMSIS && plan.use_mfma && p.channels % kMfmaWaves == 0) {
            if (src_q15) return launch_mfma<ND, M, NH, int16_t, int16_t>(p, fa, plan.d_btab, src, dst, st);
            return launch_mfma<ND, M, NH, float, float>(p, fa, plan.d_btab, src, dst, st);
        }
    }
    if (arith != SELENITE_ARITH_CMSIS) {
        if (src_q15) return launch_one<1, ND, M, NH, int16_t, int16_t>(p, fa, src, dst, st);
        return launch_one<1, ND, M, NH, float, float>(p, fa, src, dst, st);
    }
    if (src_q15) return launch_one<0, ND, M, NH, int16_t, int16_t>(p, fa, src, dst, st);
    return launch_one<0, ND, M, NH, float, float>(p, fa, src, dst, st);
}

// the instantiated shapes: BASELINE.json cfg1 / cfg2 / cfg3 (+ cfg5 = cfg2 chain), and two neighbours of
// cfg3 (half the decimator taps; the long Hilbert) to show the kernels are not tied to one tap count
#define SRX_SHAPES(X) X(256, 4, 63, 1) X(0, 1, 63, 2) X(0, 1, 127, 3) X(128, 4, 63, 4) X(256, 4, 127, 5)

static bool fused_mode_ok(const selenite_rx_config &g)
{
    const uint32_t m = g.mode;
    const bool ssb = m == SELENITE_MODE_USB || m == SELENITE_MODE_LSB || m == SELENITE_MODE_DIG || m == SELENITE_MODE_PKT;
    const bool cw_plain = mode_is_cw(m) && g.n_biquad == 0;
    return ssb || cw_plain || m == SELENITE_MODE_AM;
}

hipError_t plan_fused(const selenite_rx_config &g, bool delay_is_impulse, int delay_index, bool hilb_odd_only,
                      FusedPlan &plan)
{
    (void)delay_index;
    plan.kind = 0;
    plan.name = "generic";
    if (!fused_mode_ok(g) || !g.nh_taps || !delay_is_impulse || !hilb_odd_only) return hipSuccess;
    const uint32_t na = g.block / g.decim;
    if (na < 4 || na > 256 || (na & (na - 1)) != 0) return hipSuccess;
    int kind = 0;
    const char *name = nullptr;
#define X(ND_, M_, NH_, ID_) \
    if (shape_is<ND_, M_, NH_>(g)) { kind = ID_; name = "k_ssb_fused<" #ND_ "," #M_ "," #NH_ ">"; }
    SRX_SHAPES(X)
#undef X
    if (!kind) return hipSuccess;
    if (!plan.tables_built) {
        hipError_t e = hipSuccess;
#define X(ND_, M_, NH_, ID_) if (kind == ID_) e = build_tables<ND_, M_, NH_>(g, plan);
        SRX_SHAPES(X)
#undef X
        if (e != hipSuccess) return e;
        plan.tables_built = true;
    }
    plan.kind = kind;
    const char *nm = std::getenv("SELENITE_RX_NO_MFMA");
    plan.use_mfma = plan.d_btab != nullptr && !(nm && nm[0] == '1');
    const std::string shape = "<" + std::to_string(g.nd_taps) + "," + std::to_string(g.decim) + "," + std::to_string(g.nh_taps) + ">";
    plan.name_buf = "k_ssb_fused" + shape;
    (void)name;
    if (plan.use_mfma && g.arith == SELENITE_ARITH_FMA) plan.name_buf = "k_ssb_mfma" + shape;
    if (plan.d_btab16 && g.arith == SELENITE_ARITH_SPLIT16) {
        if (g.nd_taps) plan.name_buf = "k_ssb_split16" + shape;
        else if (na == 256) plan.name_buf = "k_hilb_split16<" + std::to_string(g.nh_taps) + ">";
    }
    plan.name = plan.name_buf.c_str();
    return hipSuccess;
}

void free_fused(FusedPlan &plan)
{
    if (plan.d_cq) (void)hipFree(plan.d_cq);
    if (plan.d_btab) (void)hipFree(plan.d_btab);
    if (plan.d_btab16) (void)hipFree(plan.d_btab16);
    plan.d_btab16 = nullptr;
    plan.d_cq = nullptr;
    plan.d_btab = nullptr;
    plan.tables_built = false;
    plan.kind = 0;
}

bool fused_block_size_ok(const FusedPlan &plan, const selenite_rx_config &g, uint32_t block_size)
{
    (void)plan;
    return (block_size / g.decim) % 256 == 0;       // whole passes only; otherwise the generic path runs
}

hipError_t launch_fused(const FusedPlan &plan, const RxParams &p, int arith, const void *src, bool src_q15,
                        void *dst, bool dst_q15, int delay_index, hipStream_t st)
{
    FusedArgs fa;
    fa.cq = plan.d_cq;
    fa.delay_idx = (uint32_t)delay_index;
    fa.upper = mode_is_upper(p.mode) ? 1u : 0u;
    fa.am = p.mode == SELENITE_MODE_AM ? 1u : 0u;
    fa.group = (p.block / p.decim) / 4;
    fa.btab16 = plan.d_btab16;
    fa.split_post = plan.split_post;
    {
        const char *e = std::getenv("SELENITE_RX_GRP_SHIFT");
        fa.grp_shift = e ? (uint32_t)std::atoi(e) : 2u;
        fa.dbg = nullptr;
        static unsigned long long *dbg_buf = nullptr;
        static int dbg_calls = 0;
        if (std::getenv("SELENITE_RX_DEBUG_TIMING")) {        // phase timeline of workgroup 0 on stderr
            if (!dbg_buf) { (void)hipMalloc((void **)&dbg_buf, 1024 * 8); (void)hipMemset(dbg_buf, 0, 1024 * 8); }
            fa.dbg = dbg_buf;
            if (++dbg_calls == 8) {
                (void)hipDeviceSynchronize();
                unsigned long long h[1024];
                (void)hipMemcpy(h, dbg_buf, sizeof h, hipMemcpyDeviceToHost);
                for (int w = 0; w < 8; w += 4) {
                    fprintf(stderr, "wave %d:", w);
                    for (int i = 0; i < 12; ++i)
                        if (h[w * 128 + i * 8 + 7]) {
                            fprintf(stderr, " [h%d @%llu", i, h[w * 128 + i * 8] - h[0]);
                            if (h[w * 128 + i * 8 + 3]) fprintf(stderr, " fin %llu", h[w * 128 + i * 8 + 3] - h[w * 128 + i * 8]);
                            fprintf(stderr, " tot %llu]", h[w * 128 + i * 8 + 7] - h[w * 128 + i * 8]);
                        }
                    fprintf(stderr, "\n");
                }
            }
        }
    }
#define X(ND_, M_, NH_, ID_) \
    if (plan.kind == ID_) return launch_shape<ND_, M_, NH_>(p, fa, plan, arith, src, src_q15, dst, dst_q15, st);
    SRX_SHAPES(X)
#undef X
    return hipErrorNotSupported;
}

}  // namespace srx
