// rx_fused_exact.hip -- the SELENITE_ARITH_CMSIS instantiations of k_ssb_fused (rx_fused_kernels.h): product rounded, then sum
// rounded, taps ascending from +0.0f -- bit-exact against the reference's arithmetic (arm_fir_decimate_f32.c:193-284,
// arm_fir_f32.c:640-936).  Also the rerun pass of SELENITE_ARITH_AUTO (RxParams::chan_flags).  A translation unit of its own so
// that it compiles beside rx_fused.hip (fma instantiations) instead of behind it.
#include "rx_fused_kernels.h"

#pragma clang fp contract(off)

namespace srx {

hipError_t launch_exact(int nd, int m, int nh, bool q15, const RxParams &p, const FusedArgs &fa, const void *src, void *dst,
                        hipStream_t st)
{
#define X(ND_, M_, NH_, ID_)                                                                             \
    if (nd == ND_ && m == M_ && nh == NH_)                                                               \
        return q15 ? launch_one<0, ND_, M_, NH_, int16_t, int16_t>(p, fa, src, dst, st)                   \
                   : launch_one<0, ND_, M_, NH_, float, float>(p, fa, src, dst, st);
    SRX_SHAPES(X)
#undef X
    return hipErrorNotSupported;
}

hipError_t launch_exact_dense(int nd, int m, bool q15, bool delay_impulse, const RxParams &p, const FusedArgs &fa, const void *src, void *dst, hipStream_t st)
{
#define X(ND_, M_, NH_, ID_)                                                                             \
    if (nd == ND_ && m == M_) {                                                                          \
        if (delay_impulse)                                                                               \
            return q15 ? launch_one<0, ND_, M_, NH_, int16_t, int16_t, 2>(p, fa, src, dst, st)            \
                       : launch_one<0, ND_, M_, NH_, float, float, 2>(p, fa, src, dst, st);               \
        return q15 ? launch_one<0, ND_, M_, NH_, int16_t, int16_t, 1>(p, fa, src, dst, st)                \
                   : launch_one<0, ND_, M_, NH_, float, float, 1>(p, fa, src, dst, st);                   \
    }
    SRX_DENSE_SHAPES(X)
#undef X
    return hipErrorNotSupported;
}

}  // namespace srx
