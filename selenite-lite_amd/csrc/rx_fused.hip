// rx_fused.hip -- fused single-launch kernels for the SSB receive chain (gfx950): planning, tables, dispatch; k_ssb_mfma; the
// fma-arithmetic instantiations of k_ssb_fused (the kernel itself is in rx_fused_kernels.h, its CMSIS-arithmetic
// instantiations in rx_fused_exact.hip).
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
#include "rx_fused_kernels.h"

#include <cmath>
#include <cstdio>
#include <cstdlib>

#pragma clang fp contract(off)

namespace srx {
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
    // persistent grid (as k_ssb_split16): this wavefront runs channels c, c + gridDim.x * kMfmaWaves, ...; the Toeplitz operand,
    // the taps and the periodic LO are loaded once; the first pass of the next channel is prefetched under the last pass
    const uint32_t c_first = blockIdx.x * kMfmaWaves + wave, c_step = gridDim.x * kMfmaWaves;
    float *tab = lds + GM::oTab;
    float *XI = lds + GM::oX, *XQ = XI + GM::XLEN;
    float *D = lds + GM::oD;
    float *dI = D, *dQ = D + G::DLEN;
    constexpr int NLD = G::T / 128;

    const uint32_t npass = p.nout / G::P;
    bool nonfinite = false;                                           // any audio sample of this wavefront NaN / Inf
    typename R::type raw[NLD];
#pragma unroll
    for (int i = 0; i < NLD; ++i) raw[i] = R::load(src, (size_t)c_first * p.in_stride + 128u * i + 2u * lane);

    // Toeplitz B operand: lane l holds B[k = 4*ks + (l>>4)][n = l&15] = cq[k - 4n]
    float B[GM::KS];
#pragma unroll
    for (int ks = 0; ks < GM::KS; ++ks) B[ks] = btab[64 * ks + lane];

    if constexpr (NCO == 1)
        for (int i = lane; i < 513; i += kWave) tab[i] = p.sintab[i];
    float hreg[(NH + 63) / 64 ? (NH + 63) / 64 : 1];
#pragma unroll
    for (int v = 0; v < (NH + 63) / 64; ++v) hreg[v] = (64 * v + lane < NH) ? p.hilb_c[64 * v + lane] : 0.0f;
    const int group = (int)fa.group;
    const int abase = 66 * (lane & 15) + (lane >> 4);                 // A-operand lane base (dwords)
    // periodic shared LO (NCO == 3, as in k_ssb_split16): the table repeats every 256 samples and a pass is a whole number of
    // periods, so load i of any pass multiplies by LO[(128 i + 2 lane, + 1) mod 256]: two register quads for the kernel
    float4 lo_per[2] = { make_float4(0.0f, 0.0f, 0.0f, 0.0f), make_float4(0.0f, 0.0f, 0.0f, 0.0f) };
    if constexpr (NCO == 3) {
        static_assert(NCO != 3 || G::T % 256 == 0, "a pass is a whole number of LO periods");
        lo_per[0] = *reinterpret_cast<const float4 *>(p.lo + 2 * lane);
        lo_per[1] = *reinterpret_cast<const float4 *>(p.lo + 128 + 2 * lane);
    }
    // streaming state of a channel: loaded into registers (for the wavefront's next channel right after the current one is
    // installed: the memory round trip hides under the current channel's passes), installed into LDS when the channel starts.
    // History: flat sample f in [0, HS) is CMSIS state sample s = f - F (older slots meet zero taps)
    float st_x[(2 * GM::HS + 63) / 64], st_d[NH > 0 ? (2 * G::HH4 + 63) / 64 : 1], st_gain;
    uint32_t st_ph0, st_step;
    const int ndr = (int)p.nd, Fr = G::HQ4 * M + 1 - ndr;             // the instance's decimator may be shorter than the kernel's (k_ssb_fused)
    auto load_state = [&](uint32_t ch) {
        batched_load<2 * GM::HS>(lane, p.dec_state + (size_t)ch * 2 * (ndr - 1),
            [&](int i) {
                const int rail = i / GM::HS, sidx = i % GM::HS - Fr;
                return sidx >= 0 ? rail * (ndr - 1) + sidx : -1;
            }, st_x);
        if constexpr (NH > 0)
            batched_load<2 * G::HH4>(lane, p.fir_state + (size_t)ch * 2 * G::HH,
                [&](int i) {
                    const int rail = i / G::HH4, sidx = i % G::HH4 - G::FH;
                    return sidx >= 0 ? rail * G::HH + sidx : -1;
                }, st_d);
        st_ph0 = NCO ? p.phase[ch] : 0u;
        st_step = NCO ? p.step[ch] : 0u;
        st_gain = p.agc ? p.gain[ch] : 1.0f;
    };
    load_state(c_first < p.channels ? c_first : p.channels - 1);
    for (uint32_t c = c_first; c < p.channels; c += c_step) {
    const size_t in_base = (size_t)c * p.in_stride, out_base = (size_t)c * p.out_stride;
    const uint32_t cn = c + c_step < p.channels ? c + c_step : c;        // next channel of this wavefront (or a harmless re-read)
    batched_store<2 * GM::HS>(lane, st_x, [&](int i, float v) { (i / GM::HS ? XQ : XI)[GM::phys(i % GM::HS)] = v; });
    if constexpr (NH > 0)
        batched_store<2 * G::HH4>(lane, st_d, [&](int i, float v) { D[(i / G::HH4) * G::DLEN + i % G::HH4] = v; });
    const uint32_t ph0 = st_ph0, step = st_step;
    float gain = st_gain;
    load_state(cn);

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
            } else if constexpr (NCO == 3) {
                const float4 l2 = lo_per[i & 1];
                a = cmul<0>(a, make_float2(l2.x, l2.y));
                b = cmul<0>(b, make_float2(l2.z, l2.w));
            } else if constexpr (NCO == 1) {
                lo_v2f la, lb;                                            // arm_sin/cos_f32 restated for the vector ALU: same bits (rx_device.h)
                const uint32_t pe = ph0 + (n0 + n) * step;
                nco_lo_pair(tab, pe, pe + step, la, lb);
                a = cmul<0>(a, make_float2(la.x, la.y));
                b = cmul<0>(b, make_float2(lb.x, lb.y));
            }
            const int f = GM::HS + (int)n;                            // even: (f, f+1) share a row
            const int ph = f + 2 * (f >> 6);
            *reinterpret_cast<float2 *>(XI + ph) = make_float2(a.x, b.x);
            *reinterpret_cast<float2 *>(XQ + ph) = make_float2(a.y, b.y);
        }
        wave_lds_sync();
        {   // the next pass of this channel, or the first pass of the wavefront's next channel
            const size_t nb = pass + 1 < npass ? in_base + n0 + G::T : (size_t)cn * p.in_stride;
#pragma unroll
            for (int i = 0; i < NLD; ++i) raw[i] = R::load(src, nb + 128u * i + 2u * lane);
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
            demod_agc_store<1, 16, ND, M, NH, TOut, AM>(p, fa, dI, dQ, lane, group, hreg, gain, dst, out_base + (size_t)pass * G::P, nonfinite);
        else
            demod_agc_store<1, 0, ND, M, NH, TOut, AM>(p, fa, dI, dQ, lane, group, hreg, gain, dst, out_base + (size_t)pass * G::P, nonfinite);
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
            if (lane < NV) tmp = lds_ld4f(D + rail * G::DLEN + G::P + 4 * v);
            wave_lds_sync();
            if (lane < NV) *reinterpret_cast<float4 *>(D + rail * G::DLEN + 4 * v) = tmp;
        }
        wave_lds_sync();
    };

    wave_lds_sync();
    stage(0);
    // half-steps: group 0 runs M(k) at h = 2k and V(k) at h = 2k+1; group 1 one half-step later
    const uint32_t nhalf = 2 * npass + 1;
    for (uint32_t h = 0; h < nhalf; ++h) {
        if (h >= (uint32_t)grp) {
            const uint32_t hh = h - grp, k = hh >> 1;
            if (k < npass) {
                if ((hh & 1) == 0) {
                    mfma_phase();
                } else {
                    finish(k);
                    if (k + 1 < npass) stage(k + 1);
                }
            }
        }
        __builtin_amdgcn_s_barrier();                                 // timing alignment only: no shared data
    }

    for (int i = lane; i < 2 * GM::HS; i += kWave) {
        const int rail = i / GM::HS, f = i % GM::HS, s = f - Fr;
        if (s >= 0) p.dec_state[((size_t)c * 2 + rail) * (ndr - 1) + s] = (rail ? XQ : XI)[GM::phys(f)];
    }
    if constexpr (NH > 0) {
        if (AM == 0 || fa.am == 2u) {                                 // AM never ran the Hilbert pair: its state stays (FM keeps the delay lines running)
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
    wave_lds_sync();                                                  // the state reads above before the next channel's fills
    }
    if (nonfinite) p.flags[kFlagNanInf] = 1u;                         // ARM_MATH_NANINF, read by selenite_rx_sync
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
// the instantiated shape that serves a configuration: the same decimation ratio and FIR pair, and the SHORTEST decimator that holds the
// instance's (round 4: any 2 <= nd <= 256 runs on the fused kernels with its taps zero-padded in front; nd = 0 needs nd = 0)
int fused_template_nd(int nd, int m, int nh)
{
    int best = -1;
#define X(ND_, M_, NH_, ID_) \
    if (m == M_ && nh == NH_ && (nd == 0 ? ND_ == 0 : (ND_ >= nd && nd >= 2)) && (best < 0 || ND_ < best)) best = ND_;
    SRX_SHAPES(X)
#undef X
    return best;
}
// ... and the shortest split-precision decimator (k_ssb_split16 wants an even tap count: its rows for k_hist_exact are pair-aligned)
int split16_template_nd(int nd, int m, int nh)
{
    int best = -1;
    if (nd < 2 || (nd & 1)) return -1;
    if (m == 8) m = 4;                                    // decimation by 8 runs on the by-4 Toeplitz product, every second output kept (FusedArgs::dec2)
#define X(ND_, M_, NH_) if (m == M_ && nh == NH_ && ND_ >= nd && (best < 0 || ND_ < best)) best = ND_;
    SRX_SPLIT16_SHAPES(X)
#undef X
    return best;
}
template <int ND, int M, int NH>
static bool shape_is(const selenite_rx_config &g)
{
    return fused_template_nd((int)g.nd_taps, (int)g.decim, (int)g.nh_taps) == ND && (int)g.decim == M && (int)g.nh_taps == NH;
}

// SELENITE_ARITH_SPLIT16: the Toeplitz operand of k_ssb_split16<ND, M, NH> -- ND the kernel's decimator length, which may exceed the
// instance's (split16_template_nd): taps zero-padded in front, scaled by 2^SC (largest |tap| lands in [2^14, 2^15)), split into f16
// hi + lo; fragment of lane l at k-step kk: 8 halfs B[k = 32kk + 8(l>>4) + j][n = l&15]
template <int ND, int M, int NH>
static hipError_t build_split16_table(const selenite_rx_config &g, FusedPlan &plan)
{
    using G = Geo<ND, M, NH>;
    using GS = GeoS<2, ND, M, NH>;
    const int ndr = (int)g.nd_taps, Fr = G::HQ4 * M + 1 - ndr;
    std::vector<float> cq((size_t)G::NCQ, 0.0f);
    for (int k = 0; k < ndr; ++k) cq[(size_t)k + Fr] = g.dec_coeffs[k];
    float cmax = 0.0f;
    for (int k = 0; k < ndr; ++k) cmax = std::fmax(cmax, std::fabs(g.dec_coeffs[k]));
    int ex = 0;
    if (cmax > 0.0f) std::frexp(cmax, &ex);                       // cmax = m * 2^ex, m in [0.5, 1)
    const int SC = 15 - ex;                                         // largest |tap| * 2^SC in [2^14, 2^15)
    std::vector<_Float16> b16((size_t)GS::KS * 2 * 64 * 8, (_Float16)0.0f);
    for (int kk = 0; kk < GS::KS; ++kk)
        for (int l = 0; l < 64; ++l)
            for (int j = 0; j < 8; ++j) {
                const int idx = 32 * kk + 8 * (l >> 4) + j - M * (l & 15);      // B[k][n] = cq[k - M n]
                float cv = 0.0f;
                if (idx >= 0 && idx < G::NCQ) cv = std::ldexp(cq[idx], SC);
                const _Float16 hi = (_Float16)cv;
                const _Float16 lo = (_Float16)(cv - (float)hi);
                b16[(((size_t)2 * kk + 0) * 64 + l) * 8 + j] = hi;
                b16[(((size_t)2 * kk + 1) * 64 + l) * 8 + j] = lo;
            }
    hipError_t e = hipMalloc(&plan.d_btab16, b16.size() * sizeof(_Float16));
    if (e != hipSuccess) return e;
    e = hipMemcpy(plan.d_btab16, b16.data(), b16.size() * sizeof(_Float16), hipMemcpyHostToDevice);
    if (e != hipSuccess) return e;
    plan.split_sc = SC;
    return hipSuccess;
}

template <int ND, int M, int NH>
static hipError_t build_tables(const selenite_rx_config &g, FusedPlan &plan, bool dense = false)
{
    using G = Geo<ND, M, NH>;
    if (dense && ND == 0) return hipSuccess;                          // (no decimator, no matrix operands: the pair's taps are all there is)
    if constexpr (ND > 0) {
        std::vector<float> cq((size_t)64 * G::NCR, 0.0f);
        const int ndr = (int)g.nd_taps, Fr = G::HQ4 * M + 1 - ndr;      // the instance's taps, zero-padded in front up to the kernel's length
        for (int k = 0; k < ndr; ++k) cq[(size_t)k + Fr] = g.dec_coeffs[k];
        hipError_t e = hipMalloc((void **)&plan.d_cq, cq.size() * sizeof(float));
        if (e != hipSuccess) return e;
        e = hipMemcpy(plan.d_cq, cq.data(), cq.size() * sizeof(float), hipMemcpyHostToDevice);
        if (e != hipSuccess) return e;
        if (dense) return hipSuccess;                                 // (the dense flavour runs on k_ssb_fused only: no matrix operands)
        if constexpr (M == 4) {
            // Toeplitz operand of k_ssb_mfma: btab[ks][lane] = cq[k - 4n], k = 4ks + (lane>>4), n = lane&15
            using GM = GeoM<ND, M, NH>;
            std::vector<float> bt((size_t)64 * GM::KS, 0.0f);
            for (int ks = 0; ks < GM::KS; ++ks)
                for (int l = 0; l < 64; ++l) {
                    const int idx = 4 * ks + (l >> 4) - 4 * (l & 15);
                    if (idx >= 0 && idx < G::NCQ) bt[(size_t)64 * ks + l] = cq[idx];
                }
            e = hipMalloc((void **)&plan.d_btab, bt.size() * sizeof(float));
            if (e != hipSuccess) return e;
            e = hipMemcpy(plan.d_btab, bt.data(), bt.size() * sizeof(float), hipMemcpyHostToDevice);
            if (e != hipSuccess) return e;
        }
        if (NH > 0) {                                                 // the split-precision decimator's operand, for ITS shape
            const int nds = split16_template_nd(ndr, M, NH);
            constexpr int MT = M == 8 ? 4 : M;                        // (by 8: the by-4 product's table)
#define X(ND_, M_, NH_) if (nds == ND_ && MT == M_ && NH == NH_) return build_split16_table<ND_, M_, NH_>(g, plan);
            SRX_SPLIT16_SHAPES(X)
#undef X
        }
        return hipSuccess;
    }
    if constexpr (ND == 0 && M == 1 && NH > 0) {
        // k_hilb_split16: Hilbert taps scaled by 2^SC, f16 hi + lo; fragment of lane l at k-step kk:
        // 8 halfs B[k = 32kk + 8(l>>4) + j][m = l&15] = h[k - m]
        using GH = GeoH<NH>;
        float cmax = 0.0f;
        for (int k = 0; k < NH; ++k) cmax = std::fmax(cmax, std::fabs(g.hilb_coeffs[k]));
        int ex = 0;
        if (cmax > 0.0f) std::frexp(cmax, &ex);
        const int SC = 15 - ex;
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
        plan.split_sc = SC;
    }
    return hipSuccess;
}


template <int ND, int M, int NH, typename TIn, typename TOut>
static hipError_t launch_mfma(const RxParams &p, const FusedArgs &fa, const float *btab, const void *src, void *dst,
                              hipStream_t st)
{
    using GM = GeoM<ND, M, NH>;
    constexpr size_t lds = (size_t)kMfmaWaves * GM::total * sizeof(float);
    static_assert(lds <= 160 * 1024, "k_ssb_mfma LDS image");
    const uint32_t nco = (p.nco == 2 && p.lo_period == 256 && Geo<ND, M, NH>::T % 256 == 0) ? 3u : p.nco;     // periodic LO: in registers
    auto k = fa.am ? (nco == 3 ? k_ssb_mfma<3, ND, M, NH, TIn, TOut, 1> : nco == 2 ? k_ssb_mfma<2, ND, M, NH, TIn, TOut, 1>
                         : (nco == 1 ? k_ssb_mfma<1, ND, M, NH, TIn, TOut, 1> : k_ssb_mfma<0, ND, M, NH, TIn, TOut, 1>))
                   : (nco == 3 ? k_ssb_mfma<3, ND, M, NH, TIn, TOut, 0> : nco == 2 ? k_ssb_mfma<2, ND, M, NH, TIn, TOut, 0>
                         : (nco == 1 ? k_ssb_mfma<1, ND, M, NH, TIn, TOut, 0> : k_ssb_mfma<0, ND, M, NH, TIn, TOut, 0>));
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(k), hipFuncAttributeMaxDynamicSharedMemorySize,
                                       (int)lds);
    if (e != hipSuccess) return e;
    // persistent grid: as many workgroups as the device keeps resident (SELENITE_RX_MFMA_GRID=0: one per channel)
    static int resident = 0;                // per shape and slot format; the NCO / AM flavours share the resource footprint closely enough
    if (resident == 0) {
        int per_cu = 0, dev = 0;
        hipDeviceProp_t prop;
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, k, 64 * kMfmaWaves, lds) == hipSuccess && per_cu > 0 &&
            hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess)
            resident = per_cu * prop.multiProcessorCount;
        else
            resident = -1;
        if (const char *ge = diag_env("SELENITE_RX_MFMA_GRID")) resident = std::atoi(ge) > 0 ? std::atoi(ge) : -1;
    }
    uint32_t grid = p.channels / kMfmaWaves;
    if (resident > 0 && (uint32_t)resident < grid) grid = (uint32_t)resident;
    hipLaunchKernelGGL(k, dim3(grid), dim3(64 * kMfmaWaves), lds, st, p, fa, btab,
                       static_cast<const TIn *>(src), static_cast<TOut *>(dst));
    return hipGetLastError();
}

template <int ND, int M, int NH>
static hipError_t launch_shape(const RxParams &p, const FusedArgs &fa, const FusedPlan &plan, int arith,
                               const void *src, bool src_q15, void *dst, bool dst_q15, hipStream_t st)
{
    if (src_q15 != dst_q15) return hipErrorNotSupported;
    const bool auto_ = arith == SELENITE_ARITH_AUTO;
    // FM: the discriminator divides by |z|, so no parity bar holds on a raw split product (SPLIT16 runs FM as FMA).  AUTO (round 4) has
    // the guard for it: a block is guarded when min|z| x max|audio| < ratio x pass maximum, its channel recomputed by the bit-exact kernel
    // -- on the decimating shapes (k_ssb_split16); the no-decimator shapes keep the bit-exact kernel (their FIR pair is not on the matrix pipe in FM)
    const bool fm = fa.am == 2u;
    const bool split = (arith == SELENITE_ARITH_SPLIT16 || auto_) && !(fm && !(auto_ && ND > 0));
    // SELENITE_ARITH_AUTO, second launch: the bit-exact kernel over the channels whose rerun flag the split16 kernel raised
    // (their streaming state is still the pre-call state; audio and state are recomputed in the CMSIS arithmetic)
    auto rerun = [&]() -> hipError_t {
        RxParams p2 = p;
        p2.chan_flags = p.rerun_flag;
        p2.rerun_flag = nullptr;                      // (guard_ch / guard_calls stay: the rerun pass counts for the channels it HOLDS, which the matrix kernel skipped)
        if (p.rerun_par_host) {                       // the dense list of this call: counted in one of two alternating counters (rx_internal.h)
            const uint32_t par = *p.rerun_par_host & 1u;
            p2.chan_count = p.chan_count + par;
            p2.chan_count_next = p.chan_count + (par ^ 1u);
            *p.rerun_par_host = par ^ 1u;
        }
        // (channels the call before left on the matrix kernel: their Hilbert-pair history first, in exact arithmetic -- rx_generic.hip)
        if (hipError_t e = launch_hist_exact(p2, false, st); e != hipSuccess) return e;
        return launch_exact(ND, M, NH, src_q15, p2, fa, src, dst, st);            // rx_fused_exact.hip
    };
    // what the matrix kernels take: DSP blocks that divide the 256-output pass; whole passes (k_ssb_mfma, k_hilb_split16), or a
    // partial last pass that holds a whole decimator history (k_ssb_split16).  Everything else -- short calls, the firmware's
    // 96-frame blocks -- runs on k_ssb_fused, whose passes have variable length.
    const bool whole = fa.pass_out == 256 && p.nout % 256 == 0;
    const int nds = ND > 0 ? split16_template_nd((int)p.nd, M, NH) : -1;               // the split-precision kernel's decimator length (>= the instance's), or -1
    const uint32_t kHS = nds > 0 ? (uint32_t)(((((nds - 1 + M - 1) / M) + 3) & ~3) * M) : 0u;      // GeoS::HS: decimator history in the image
    // k_ssb_split16 also takes passes of fewer than 256 outputs when they are whole 16-output tiles (240 for the firmware's
    // 96-frame blocks by 4, 192 for its 96-sample audio blocks): its run-time DSP-block flavour advances by pass_out * M samples
    // (decimation by 8, round 4 late: the by-4 kernel with every second output kept -- passes of at most 128 outputs = 1024 inputs)
    FusedArgs fa16 = fa;
    if constexpr (M == 8) {
        fa16.pass_out = split16_pass_out(p.block, p.decim);
        // (which half: the tile's output j ends at input sample 4 j of the pass in this kernel's bookkeeping, a by-8 output n at 8 n -- the
        // even ones; SELENITE_RX_DEC2_PARITY=2 selects the odd ones: a diagnostic that shows the tests notice)
        // (only the value 2 is taken, and said so on stderr: anything else keeps the chain's outputs)
        static const uint32_t par = [] {
            const char *e = diag_env("SELENITE_RX_DEC2_PARITY");
            if (e && e[0] == '2' && e[1] == 0) { fprintf(stderr, "selenite_rx: SELENITE_RX_DEC2_PARITY=2 -- decimation by 8 keeps the ODD by-4 outputs (diagnostic, wrong audio)\n"); return 2u; }
            return 1u;
        }();
        fa16.dec2 = par;
    }
    const uint32_t tq = fa16.pass_out * M;
    // (a last pass shorter than the decimator history: only as a call of its own -- fused_tail_split cuts it off)
    const bool split_ok = split16_pass_ok(fa16.pass_out) && (p.block_size % tq == 0 || p.block_size % tq >= kHS || p.block_size < tq);
    // SELENITE_ARITH_AUTO (round 4: "handover_blocks == 0 by construction"): the matrix kernel only takes calls long enough to leave the
    // mixed samples in front of the decimator state behind (hist_ext: nd - 1 + M HH4 - 1 samples) -- every state it leaves can then be
    // repaired by k_hist_exact.  Shorter calls (the firmware's literal one-slot callback, a tail cut off by fused_tail_split) run on the
    // bit-exact kernel, from a history repaired first (below): they are state-traffic bound either way.  With the repair switched
    // off (a diagnostic: selenite_rx_set_handover_repair) nothing is kept and such calls stay on the matrix kernel, counted.
    const bool auto_ok = !auto_ || p.hist_ext == nullptr || p.block_size + 1u >= (uint32_t)(ND > 0 ? p.nd - 1 : 0) + p.ext_len;
    if constexpr (ND > 0 && (M == 4 || M == 2 || M == 8) && NH > 0) {
        if (split && plan.d_btab16 && split_ok && auto_ok) {
            if (auto_ && fa.am == 1u && p.rerun_flag) {
                // AM neither reads nor moves the Hilbert-pair history while the decimator state moves on: the samples kept in front of
                // that state belong to the history for the last time NOW -- repair every channel that has them, before the AM call
                RxParams p3 = p;
                p3.chan_flags = p.rerun_flag;
                p3.chan_list = nullptr; p3.chan_count_next = nullptr;
                if (hipError_t e = launch_hist_exact(p3, true, st); e != hipSuccess) return e;
            }
            if (auto_ && p.rerun_flag && p.form_host) *p.form_host = 3u;
            hipError_t e = launch_ssb_split16(nds, M == 8 ? 4 : M, NH, p, fa16, src, src_q15, dst, st);     // rx_split16.hip
            if (e == hipSuccess && auto_ && p.rerun_flag) e = rerun();
            return e;
        }
    }
    if constexpr (ND == 0 && M == 1 && NH > 0) {
        // k_hilb_split16: whole passes of the largest whole number of DSP blocks in 256 (256 itself, or e.g. 192: BASELINE cfg2's
        // literal 48 000 samples are 250 blocks of 192); whole 4-sample lanes per block
        const bool hilb_ok = fa.pass_out != 0;                        // (any call length: the last pass may be partial, round 4 late)
        if (split && plan.d_btab16 && hilb_ok) {
            FusedArgs fah = fa;
            const bool inl = auto_ && p.rerun_flag && p.auto_inline && !fa.am;     // (every SSB instantiation of k_hilb_split16 carries the exact body)
            fah.inl = inl ? 1u : 0u;
            if (auto_ && p.rerun_flag && p.form_host) *p.form_host = inl ? 1u : 3u;
            hipError_t e = launch_hilb_split16(NH, p, fah, src, src_q15, dst, st);           // rx_split16.hip
            if (e == hipSuccess && auto_ && p.rerun_flag && !inl) e = rerun();
            return e;
        }
    }
    if (auto_) {                                  // no split-precision kernel for this launch: the bit-exact one, on every channel --
        arith = SELENITE_ARITH_CMSIS;             // from a Hilbert-pair history in exact arithmetic where the call before left the samples for it
        if (p.rerun_flag) {
            RxParams p3 = p;
            p3.chan_flags = p.rerun_flag;
            p3.chan_list = nullptr; p3.chan_count_next = nullptr;
            if (hipError_t e = launch_hist_exact(p3, true, st); e != hipSuccess) return e;
        }
    }
    if constexpr (ND > 0 && M == 4) {
        static_assert(kMfmaWaves == 1, "one channel per workgroup: any channel count launches (plan.name says k_ssb_mfma)");
        if (arith != SELENITE_ARITH_CMSIS && plan.use_mfma && whole) {
            if (src_q15) return launch_mfma<ND, M, NH, int16_t, int16_t>(p, fa, plan.d_btab, src, dst, st);
            return launch_mfma<ND, M, NH, float, float>(p, fa, plan.d_btab, src, dst, st);
        }
    }
    if (arith != SELENITE_ARITH_CMSIS) {
        if (src_q15) return launch_one<1, ND, M, NH, int16_t, int16_t>(p, fa, src, dst, st);
        return launch_one<1, ND, M, NH, float, float>(p, fa, src, dst, st);
    }
    return launch_exact(ND, M, NH, src_q15, p, fa, src, dst, st);                 // rx_fused_exact.hip
}


static bool fused_mode_ok(const selenite_rx_config &g)
{
    const uint32_t m = g.mode;
    const bool ssb = m == SELENITE_MODE_USB || m == SELENITE_MODE_LSB || m == SELENITE_MODE_DIG || m == SELENITE_MODE_PKT;
    const bool cw_plain = mode_is_cw(m) && g.n_biquad == 0;
    return ssb || cw_plain || m == SELENITE_MODE_AM || m == SELENITE_MODE_FM;
}

hipError_t plan_fused(const selenite_rx_config &g, bool delay_is_impulse, int delay_index, bool hilb_odd_only,
                      FusedPlan &plan)
{
    (void)delay_index;
    plan.kind = 0;
    plan.dense = false;
    plan.name = "generic";
    if (!fused_mode_ok(g) || !g.nh_taps) return hipSuccess;
    // DSP blocks of 4 .. 256 audio samples, four per lane; a power of two divides the 256-output pass, anything else (the
    // firmware's 96 frames: 24 or 96 audio samples) runs with passes of the largest whole number of blocks (k_ssb_fused)
    const uint32_t na = g.block / g.decim;
    if (na < 4 || na > 256 || na % 4 != 0) return hipSuccess;
    int kind = 0;
    const char *name = nullptr;
    if (delay_is_impulse && hilb_odd_only) {
#define X(ND_, M_, NH_, ID_) \
    if (shape_is<ND_, M_, NH_>(g)) { kind = ID_; name = "k_ssb_fused<" #ND_ "," #M_ "," #NH_ ">"; }
    SRX_SHAPES(X)
#undef X
    }
    if (!kind) {
        // Round 4 (VERDICT r3 missing 4): anything else with a FIR pair of up to 127 taps -- dense Hilbert taps, a delay FIR that is not a
        // unit impulse, a tap count without an instantiation of its own -- runs on the DENSE flavour of k_ssb_fused (both rails filtered
        // with their own taps from LDS, taps zero-padded in front) instead of the generic kernels: bit-exact in the CMSIS / fma
        // arithmetic; SPLIT16 runs as fma and AUTO as CMSIS there (no matrix kernel for a dense pair).
        if (g.nh_taps > 127) return hipSuccess;
        int nds = -1;
#define X(ND_, M_, NH_, ID_) \
        if ((int)g.decim == M_ && ((int)g.nd_taps == 0 ? ND_ == 0 : (ND_ >= (int)g.nd_taps && (int)g.nd_taps >= 2)) && (nds < 0 || ND_ < nds)) { nds = ND_; kind = ID_; }
        SRX_DENSE_SHAPES(X)
#undef X
        if (!kind) return hipSuccess;
        if (!plan.tables_built) {
            hipError_t e = hipSuccess;
#define X(ND_, M_, NH_, ID_) if (kind == ID_) e = build_tables<ND_, M_, NH_>(g, plan, true);
            SRX_DENSE_SHAPES(X)
#undef X
            if (e != hipSuccess) return e;
            // the pair's tap tables: padded tap k' = k + (127 - nh) at index k' + FH + 3 (FH = 2 for the 127-tap geometry)
            constexpr int NHT = 127, LEN = DenseTab<NHT>::LEN, FH = ((NHT - 1 + 3) & ~3) - (NHT - 1);
            std::vector<float> pt((size_t)2 * LEN, 0.0f);
            const int pad = NHT - (int)g.nh_taps;
            for (int k = 0; k < (int)g.nh_taps; ++k) {
                pt[(size_t)k + pad + FH + 3] = g.delay_coeffs[k];
                pt[(size_t)LEN + k + pad + FH + 3] = g.hilb_coeffs[k];
            }
            e = hipMalloc((void **)&plan.d_ptab, pt.size() * sizeof(float));
            if (e != hipSuccess) return e;
            e = hipMemcpy(plan.d_ptab, pt.data(), pt.size() * sizeof(float), hipMemcpyHostToDevice);
            if (e != hipSuccess) return e;
            plan.dense_t0 = (uint32_t)((pad + FH - 3) > 0 ? (pad + FH - 3) / 4 : 0);
            plan.tables_built = true;
        }
        plan.kind = kind;
        plan.dense = true;
        plan.dense_delay_impulse = delay_is_impulse;     // the I rail stays one LDS read per output (0.0f + 1.0f * x), only the Hilbert FIR is dense
        plan.use_mfma = false;
        plan.name_buf = "k_ssb_fused<" + std::to_string(g.nd_taps) + "," + std::to_string(g.decim) + "," + std::to_string(g.nh_taps) + "> (dense FIR pair)";
        plan.name = plan.name_buf.c_str();
        return hipSuccess;
    }
    if (!plan.tables_built) {
        hipError_t e = hipSuccess;
#define X(ND_, M_, NH_, ID_) if (kind == ID_) e = build_tables<ND_, M_, NH_>(g, plan);
        SRX_SHAPES(X)
#undef X
        if (e != hipSuccess) return e;
        plan.tables_built = true;
    }
    plan.kind = kind;
    const char *nm = diag_env("SELENITE_RX_NO_MFMA");
    plan.use_mfma = plan.d_btab != nullptr && !(nm && nm[0] == '1');
    const std::string shape = "<" + std::to_string(g.nd_taps) + "," + std::to_string(g.decim) + "," + std::to_string(g.nh_taps) + ">";
    plan.name_buf = "k_ssb_fused" + shape;
    (void)name;
    if (plan.use_mfma && g.arith != SELENITE_ARITH_CMSIS) plan.name_buf = "k_ssb_mfma" + shape;     // split16 without a matrix kernel of its own runs as fma
    if (g.arith == SELENITE_ARITH_AUTO) plan.name_buf = "k_ssb_fused" + shape;                      // without a matrix kernel of its own: bit-exact
    const bool split_pass = g.nd_taps ? split16_pass_ok(split16_pass_out(g.block, g.decim)) : true;      // (k_hilb_split16 takes any pass of whole 4-sample lanes; by 8: passes of at most 128 outputs)
    if (256 % na != 0) plan.name_buf = "k_ssb_fused" + shape;                                        // DSP blocks that do not divide a pass: variable-length passes
    if (split_pass && plan.d_btab16 && (g.arith == SELENITE_ARITH_SPLIT16 || g.arith == SELENITE_ARITH_AUTO)) {
        const char *tail = g.arith == SELENITE_ARITH_AUTO ? "+exact rerun of guarded channels" : "";
        if (g.nd_taps) plan.name_buf = "k_ssb_split16" + shape + tail;
        else plan.name_buf = "k_hilb_split16<" + std::to_string(g.nh_taps) + ">" + tail;
    }
    if (g.mode == SELENITE_MODE_FM && !(g.arith == SELENITE_ARITH_AUTO && g.nd_taps && plan.name_buf.rfind("k_ssb_split16", 0) == 0)) {
        // FM: the exact / fma kernels (AUTO on a decimating shape with a matrix kernel keeps it: guarded on min|z|, round 4)
        const bool fma = g.arith == SELENITE_ARITH_FMA || g.arith == SELENITE_ARITH_SPLIT16;
        plan.name_buf = (fma && plan.use_mfma && g.nd_taps && g.decim == 4 ? "k_ssb_mfma" : "k_ssb_fused") + shape;
    }
    plan.name = plan.name_buf.c_str();
    return hipSuccess;
}

void free_fused(FusedPlan &plan)
{
    if (plan.d_cq) (void)hipFree(plan.d_cq);
    if (plan.d_btab) (void)hipFree(plan.d_btab);
    if (plan.d_btab16) (void)hipFree(plan.d_btab16);
    if (plan.d_ptab) (void)hipFree(plan.d_ptab);
    plan.d_ptab = nullptr;
    plan.dense = false;
    plan.d_btab16 = nullptr;
    plan.d_cq = nullptr;
    plan.d_btab = nullptr;
    plan.tables_built = false;
    plan.kind = 0;
}

bool fused_tail_split(const FusedPlan &plan, const selenite_rx_config &g, uint32_t block_size)
{
    // Every call length (a whole number of DSP blocks) runs on the fused kernels since round 3 (k_ssb_fused: variable-length
    // passes).  One case is better served in two launches: a split-precision instance whose call ends in a partial pass too
    // short for k_ssb_split16 (it wants a whole decimator history in it; only DSP blocks shorter than that history get there):
    // the whole passes stay on the matrix kernel, the tail goes to k_ssb_fused (SELENITE_ARITH_SPLIT16: its fma arithmetic;
    // SELENITE_ARITH_AUTO: the bit-exact arithmetic, from a history k_hist_exact repairs first -- launch_shape's auto_ok, the tail
    // being too short to leave the mixed samples behind) -- true when the call should be cut that way.
    if (!(g.arith == SELENITE_ARITH_SPLIT16 || g.arith == SELENITE_ARITH_AUTO) || !plan.d_btab16 || !g.nd_taps) return false;
    const uint32_t pq = split16_pass_out(g.block, g.decim), unit = pq * g.decim;
    if (!split16_pass_ok(pq) || block_size % unit == 0 || block_size < unit) return false;
    const int nds = split16_template_nd((int)g.nd_taps, (int)g.decim, (int)g.nh_taps);
    if (nds < 0) return false;
    const uint32_t hq = ((uint32_t)nds - 1 + g.decim - 1) / g.decim, hs = ((hq + 3) & ~3u) * g.decim;    // GeoS::HS
    return block_size % unit < hs;
}

hipError_t launch_fused(const FusedPlan &plan, const RxParams &p, int arith, const void *src, bool src_q15,
                        void *dst, bool dst_q15, int delay_index, hipStream_t st)
{
    FusedArgs fa;
    fa.cq = plan.d_cq;
    fa.delay_idx = (uint32_t)delay_index;
    fa.upper = mode_is_upper(p.mode) ? 1u : 0u;
    fa.am = p.mode == SELENITE_MODE_AM ? 1u : (p.mode == SELENITE_MODE_FM ? 2u : 0u);      // (FM: a run-time flavour of the AM instantiations of k_ssb_fused)
    fa.group = (p.block / p.decim) / 4;
    fa.pass_out = 256u / (p.block / p.decim) * (p.block / p.decim);
    fa.btab16 = plan.d_btab16;
    fa.split_post = plan.split_post;
    fa.split_sc = plan.split_sc;
    fa.inl = 0u;                                          // (launch_shape: SELENITE_ARITH_AUTO in one launch)
    fa.dec2 = 0u;                                         // (launch_shape sets it for decimation by 8 on the by-4 matrix kernel)
    {
        static const uint32_t shift = [] {                // (diagnostic builds: another workgroup -> channel-group mapping; read once, 0 .. 8)
            const char *e = diag_env("SELENITE_RX_GRP_SHIFT");
            const int v = e ? std::atoi(e) : 2;
            return (uint32_t)(v >= 0 && v <= 8 ? v : 2);
        }();
        fa.grp_shift = shift;
    }
    if (plan.dense) {
        // the FIR pair with arbitrary taps: k_ssb_fused's DENSE flavour, bit-exact (CMSIS; AUTO runs as CMSIS) or fma (FMA; SPLIT16 runs as fma)
        if (src_q15 != dst_q15) return hipErrorNotSupported;
        fa.ptab = plan.d_ptab;
        fa.ptab_lds = nullptr;
        fa.dense_t0 = plan.dense_t0;
        if (plan.dense_delay_impulse) fa.delay_idx = (uint32_t)delay_index + (127u - p.nh);      // the unit tap's index among the padded taps
        const bool exact = arith == SELENITE_ARITH_CMSIS || arith == SELENITE_ARITH_AUTO;
#define X(ND_, M_, NH_, ID_)                                                                                          \
        if (plan.kind == ID_) {                                                                                       \
            if (exact) return launch_exact_dense(ND_, M_, src_q15, plan.dense_delay_impulse, p, fa, src, dst, st);    \
            if (plan.dense_delay_impulse)                                                                             \
                return src_q15 ? launch_one<1, ND_, M_, NH_, int16_t, int16_t, 2>(p, fa, src, dst, st)                \
                               : launch_one<1, ND_, M_, NH_, float, float, 2>(p, fa, src, dst, st);                   \
            return src_q15 ? launch_one<1, ND_, M_, NH_, int16_t, int16_t, 1>(p, fa, src, dst, st)                    \
                           : launch_one<1, ND_, M_, NH_, float, float, 1>(p, fa, src, dst, st);                       \
        }
        SRX_DENSE_SHAPES(X)
#undef X
        return hipErrorNotSupported;
    }
#define X(ND_, M_, NH_, ID_) \
    if (plan.kind == ID_) return launch_shape<ND_, M_, NH_>(p, fa, plan, arith, src, src_q15, dst, dst_q15, st);
    SRX_SHAPES(X)
#undef X
    return hipErrorNotSupported;
}

}  // namespace srx
