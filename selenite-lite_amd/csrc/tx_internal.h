// tx_internal.h -- kernel parameter block and launch interfaces shared by tx.hip and tx_fused.hip.
// Not part of the C-ABI (include/selenite_tx.h is).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/selenite_tx.h"
#include "rx_device.h"

namespace srx {

struct TxParams {
    uint32_t channels, block, L, ni, P, nh, mode, nco, alc, block_size;
    uint32_t q15_round;    // int16 output: 1 = the ARM_MATH_ROUNDING build of arm_float_to_q15 (selenite_tx_config::q15_rounding)
    uint32_t lo_period;    // shared LO: 256 when it repeats every 256 output samples (NCO step a multiple of 2^24), else 0
    const float *ic, *hc, *dc, *sintab;
    const uint32_t *step;
    uint32_t *phase;
    float *fir_state;      // [C][2][nh-1]
    float *int_state;      // [C][2][P-1]
    float *gain;
    AgcParams alcp;
};

// tx_fused.hip: the BASELINE-like shape (ALC block 64, L = 4, 256-tap interpolator, 63-tap Hilbert
// pair with a unit-impulse delay and a type-III Hilbert).  `lo` = shared LO of the call as produced by
// launch_lo_table (cos, -sin) or NULL (per-channel NCO in the kernel).
bool tx_fused_ok(const selenite_tx_config &g, bool delay_is_impulse, bool hilb_odd_only, uint32_t block_size);
hipError_t launch_tx_fused(const TxParams &p, int arith, uint32_t delay_idx, const float2 *lo, const void *src, bool q15,
                           void *dst, hipStream_t st);

// SELENITE_ARITH_SPLIT16: the interpolator on the 16-bit matrix pipe (k_tx_split16), same shape as the fused kernel
hipError_t build_tx_split16_table(const float *interp_coeffs, void **d_table, int *tap_sc);
hipError_t launch_tx_split16(const TxParams &p, uint32_t delay_idx, const float2 *lo, const void *ttab16, int tap_sc,
                             const void *src, bool q15, void *dst, hipStream_t st);

}  // namespace srx
