// rx_split16.hip -- SELENITE_ARITH_SPLIT16: the many-tap FIRs of the receive chain as split-precision
// matrix products on the 16-bit matrix cores of gfx950 (v_mfma_f32_16x16x32_f16).
//
//   k_ssb_split16<NCO, ND, 4, NH, TIn, TOut, AM, GROUP>   decimating shapes (BASELINE cfg3)
//   k_hilb_split16<NCO, NH, TIn, TOut, AM>                no-decimator shapes (cfg1 / cfg2 / cfg5)
//
// f32 MFMA executes on the FP32 vector ALUs (it ADDS to the VALU time, measured: DESIGN.md 5.1), so the
// only extra arithmetic throughput on the chip is the separate 16-bit matrix pipe.  The FIR
// (arm_fir_decimate_f32.c:193-284) is the banded-Toeplitz product
//     D[i][n] = sum_k A[i][k] B[k][n],   A[i][k] = x[64 i + k],   B[k][n] = cq[k - 4 n]
// (rows i = 16 blocks of 16 consecutive outputs of one rail); samples and taps are split EXACTLY into
// f16 hi + lo parts,  x*2^s = xh + xl + O(2^-22),  c*2^SC = ch + cl + O(2^-22),  and the product is
// xh*ch + (xh*cl + xl*ch) with f32 accumulation inside the MFMA (f16 x f16 is exact in f32); the
// dropped xl*cl term is 2^-22 relative.  NOT bit-reproducible by a CPU loop: parity for this mode is
// tolerance-based by construction (north star: 1e-5 relative per DSP block).
//
// Block floating point.  The sample scale 2^s is chosen per channel and per pass from the data: the
// largest |component| of the samples the pass's LDS image holds (the 1024 new mixed samples and the
// ND-1 history samples) lands in [2^14, 2^15), just under the f16 range, so hi AND lo parts stay
// normal f16 numbers for every sample within 100 dB of the pass maximum, for any input amplitude a
// float can hold.  s depends only on those samples, so a stream cut into calls at different pass
// boundaries gives identical bits.  When s changes between passes the history part of the image is
// re-split from an f32 copy kept in LDS (which is also what goes back to HBM as the CMSIS pState at
// the end of the call: the streaming state stays exact f32 in every mode).
#include "rx_split16_kernels.h"

#pragma clang fp contract(off)

namespace srx {

bool ssb_split16_has_shape(int nd, int m, int nh)
{
#define X(ND_, M_, NH_) if (nd == ND_ && m == M_ && nh == NH_) return true;
    SRX_SPLIT16_SHAPES(X)
#undef X
    return false;
}

bool ssb_split16_periodic_lo(int nd, int m, int nh)
{
    if (m == 8) m = 4;                                    // (decimation by 8 runs the by-4 kernel: FusedArgs::dec2)
#define X(ND_, M_, NH_) if (nd == ND_ && m == M_ && nh == NH_) return Geo<ND_, M_, NH_>::T % 256 == 0;
    SRX_SPLIT16_SHAPES(X)
#undef X
    return false;
}

hipError_t launch_ssb_split16(int nd, int m, int nh, const RxParams &p, const FusedArgs &fa, const void *src, bool q15,
                              void *dst, hipStream_t st)
{
    if (q15) return launch_ssb_split16_q15(nd, m, nh, p, fa, src, dst, st);              // rx_split16_q15.hip
#define X(ND_, M_, NH_) if (nd == ND_ && m == M_ && nh == NH_) return launch_nco<ND_, M_, NH_, float>(p, fa, src, dst, st);
    SRX_SPLIT16_SHAPES(X)
#undef X
    return hipErrorNotSupported;
}

template <int NH, typename TIn, typename TOut>
static hipError_t launch_hilb16(const RxParams &p, const FusedArgs &fa, const void *src, void *dst, hipStream_t st)
{
    using GH = GeoH<NH>;
    constexpr size_t lds = (size_t)(Geo<0, 1, NH>::total > GH::total ? Geo<0, 1, NH>::total : GH::total) * sizeof(float);     // (+ the bit-exact kernel's image: FusedArgs::inl)
    static_assert(lds <= 48 * 1024, "k_hilb_split16 LDS image");
    auto k = fa.am ? (p.nco == 2 ? k_hilb_split16<2, NH, TIn, TOut, 1>
                                 : (p.nco == 1 ? k_hilb_split16<1, NH, TIn, TOut, 1> : k_hilb_split16<0, NH, TIn, TOut, 1>))
                   : (p.nco == 2 ? k_hilb_split16<2, NH, TIn, TOut, 0>
                                 : (p.nco == 1 ? k_hilb_split16<1, NH, TIn, TOut, 0> : k_hilb_split16<0, NH, TIn, TOut, 0>));
    hipLaunchKernelGGL(k, dim3(p.channels), dim3(64), lds, st, p, fa, static_cast<const TIn *>(src),
                       static_cast<TOut *>(dst));
    return hipGetLastError();
}

hipError_t launch_hilb_split16(int nh, const RxParams &p, const FusedArgs &fa, const void *src, bool q15, void *dst,
                               hipStream_t st)
{
#define X(NH_) if (nh == NH_) return q15 ? launch_hilb16<NH_, int16_t, int16_t>(p, fa, src, dst, st) : launch_hilb16<NH_, float, float>(p, fa, src, dst, st);
    SRX_HILB16_SHAPES(X)
#undef X
    return hipErrorNotSupported;
}

}  // namespace srx
