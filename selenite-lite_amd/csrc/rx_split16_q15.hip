// rx_split16_q15.hip -- the int16-slot instantiations of k_ssb_split16 (rx_split16_kernels.h): the firmware's wire format either
// side of the slot (dsp_if.c:286-289; arm_q15_to_float / arm_float_to_q15 fused into load and store).  A translation unit of its
// own so that it compiles beside rx_split16.hip (f32 slots) instead of behind it.
#include "rx_split16_kernels.h"

#pragma clang fp contract(off)

namespace srx {

hipError_t launch_ssb_split16_q15(int nd, int m, int nh, const RxParams &p, const FusedArgs &fa, const void *src, void *dst, hipStream_t st)
{
#define X(ND_, M_, NH_) if (nd == ND_ && m == M_ && nh == NH_) return launch_nco<ND_, M_, NH_, int16_t>(p, fa, src, dst, st);
    SRX_SPLIT16_SHAPES(X)
#undef X
    return hipErrorNotSupported;
}

}  // namespace srx
