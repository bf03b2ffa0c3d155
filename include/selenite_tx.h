/* selenite_tx.h -- C-ABI of the batched TX DSP block path: the mirror
 * image of selenite_rx.h for the other direction of the firmware's slot.
 *
 * In the reference the TX direction is, like RX, a pass-through of int16 frames: USB audio-out ->
 * DSP_Out_Buff_Write (Core/Src/dsp_if.c:116-180) -> DSP_Out_Buff_Read (dsp_if.c:204-219, "mix CW tone
 * to speaker signal here") -> I2S -> codec; no modulator exists.  What is
 * specified here is therefore [build-defined], composed -- like the RX chain -- only of CMSIS-DSP
 * 1.5.3 primitives vendored in the reference, each pinned bit-exactly against the real function:
 *
 *   audio -> arm_abs_f32 + arm_max_f32 -> gain law -> arm_scale_f32            (ALC = the RX AGC law on the input)
 *         -> arm_fir_f32 (delay taps) => I',  arm_fir_f32 (Hilbert taps) => Q'  (FilteringFunctions/arm_fir_f32.c)
 *         -> sideband: USB/DIG/CW z = (I', Q');  LSB/PKT/CWR z = (I', -Q')      (BasicMathFunctions/arm_negate_f32.c)
 *                      AM z = (0.5 + 0.5*I', 0)                                 (arm_scale_f32 + arm_offset_f32)
 *         -> arm_fir_interpolate_f32 by L on both rails                         (FilteringFunctions/arm_fir_interpolate_f32.c:136-563)
 *         -> NCO up-mix: out = z * (cos x, +sin x), x from the RX chain's integer phase rule
 *            (arm_sin_f32 / arm_cos_f32 + arm_cmplx_mult_cmplx_f32)
 *
 * Shapes: pSrcAudio [channels][blockSize] (float or q15), pDstIQ [channels][blockSize*interp][2].
 * blockSize counts AUDIO samples and must be a multiple of cfg.block (the ALC block).
 * Status codes, mode bytes and arithmetic modes are selenite_rx.h's: SELENITE_ARITH_CMSIS (bit-exact), _FMA (bit-exact vs the
 * fmaf restatement), _SPLIT16 (the interpolator as a split-precision matrix product for the BASELINE-like shape, <=1e-5 of the
 * output block's maximum + 1e-6 of the input level; other shapes run as _FMA).  SELENITE_ARITH_AUTO is an RX mode (its parity
 * guard compares the audio envelope the RX AGC computes anyway with the input level; the TX interpolator's output sits at its
 * input's level for in-band audio): selenite_tx_init returns SELENITE_RX_ARGUMENT_ERROR for it.
 * No CPU fallback: init fails with SELENITE_RX_DEVICE_ERROR without a HIP device.
 */
#ifndef SELENITE_TX_H
#define SELENITE_TX_H

#include <stdint.h>

#include "selenite_rx.h"

#ifdef __cplusplus
extern "C" {
#endif

typedef struct selenite_tx_instance selenite_tx_instance;

#define SELENITE_TX_CONFIG_SIZE_V1 96u

typedef struct {
    uint32_t struct_size;      /* sizeof(selenite_tx_config); a caller built against the version-1 header passes SELENITE_TX_CONFIG_SIZE_V1 (selenite_rx.h: ABI versions) */
    uint32_t channels;
    uint32_t block;            /* ALC block, audio samples */
    uint32_t interp;           /* L >= 1 (1: no interpolator, ni_taps must be 0) */
    uint32_t ni_taps;          /* interpolator taps, a multiple of interp (arm_fir_interpolate_init_f32.c:91-96:
                                  otherwise ARM_MATH_LENGTH_ERROR) */
    uint32_t nh_taps;          /* Hilbert / delay taps (0: Q' = 0, I' = audio) */
    uint8_t  arith;            /* SELENITE_ARITH_* */
    uint8_t  mode;             /* SELENITE_MODE_* (FM: ARGUMENT_ERROR) */
    uint8_t  nco_enable;
    uint8_t  alc_enable;
    uint32_t nco_step_all;     /* phase increment per OUTPUT sample when nco_step is NULL */
    const float *interp_coeffs;   /* [ni_taps]  CMSIS order (time reversed) */
    const float *hilb_coeffs;     /* [nh_taps] */
    const float *delay_coeffs;    /* [nh_taps] */
    const uint32_t *nco_step;     /* [channels] or NULL */
    float alc_target, alc_attack, alc_decay, alc_gain_min, alc_gain_max, alc_env_floor, alc_gain_init;
    uint32_t q15_rounding;     /* int16 I/Q output (arm_float_to_q15): 0 = truncate (the firmware's build, arm_float_to_q15.c:117), 1 = the ARM_MATH_ROUNDING
                                  build (arm_float_to_q15.c:90-101); other values: ARGUMENT_ERROR.  ABI version 2: read only when struct_size == sizeof (version 1 had padding here) */
    uint32_t abi_version;      /* ABI version 2: = SELENITE_RX_ABI_VERSION */
    uint32_t reserved;         /* 0 */
} selenite_tx_config;

typedef struct {
    float    *fir_state;       /* [channels][2][nh_taps-1]            arm_fir_f32 pState tails (delay, Hilbert) */
    float    *interp_state;    /* [channels][2][ni_taps/interp - 1]   arm_fir_interpolate_f32 pState tails (I, Q) */
    float    *alc_gain;        /* [channels] */
    uint32_t *nco_phase;       /* [channels] */
} selenite_tx_state_view;

int  selenite_tx_init(selenite_tx_instance **S, const selenite_tx_config *cfg);
void selenite_tx_free(selenite_tx_instance *S);
int  selenite_tx_set_mode(selenite_tx_instance *S, uint8_t mode);     /* DSP_Set_Mode, dsp_if.c:367-370 */
int  selenite_tx_status(const selenite_tx_instance *S);
/* which kernel serves calls whose blockSize is a multiple of 256: "k_tx_fused<4,256,63>" or "k_tx_generic" */
const char *selenite_tx_kernel_name(const selenite_tx_instance *S);
const char *selenite_tx_error_string(const selenite_tx_instance *S);

/* host buffers (copied in and out, synchronous) */
void selenite_tx_process_f32(selenite_tx_instance *S, const float *pSrcAudio, float *pDstIQ, uint32_t blockSize);
void selenite_tx_process_q15(selenite_tx_instance *S, const int16_t *pSrcAudio, int16_t *pDstIQ, uint32_t blockSize);
/* device buffers (asynchronous on the instance's stream) */
void selenite_tx_process_f32_device(selenite_tx_instance *S, const float *dSrcAudio, float *dDstIQ, uint32_t blockSize);
void selenite_tx_process_q15_device(selenite_tx_instance *S, const int16_t *dSrcAudio, int16_t *dDstIQ, uint32_t blockSize);

int  selenite_tx_set_stream(selenite_tx_instance *S, void *hip_stream);
int  selenite_tx_sync(selenite_tx_instance *S);
int  selenite_tx_get_state(selenite_tx_instance *S, const selenite_tx_state_view *view);
int  selenite_tx_set_state(selenite_tx_instance *S, const selenite_tx_state_view *view);
int  selenite_tx_reset(selenite_tx_instance *S);
/* mean milliseconds per process_f32_device call over `iters` calls (HIP events on the stream) */
int  selenite_tx_time_process_device(selenite_tx_instance *S, const float *dSrcAudio, float *dDstIQ,
                                     uint32_t blockSize, uint32_t iters, float *ms_per_call);

#ifdef __cplusplus
}
#endif
#endif /* SELENITE_TX_H */
