/*
 * rx_oracle.h -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.
 *
 * CPU restatement (plain C) of the CMSIS-DSP 1.5.3 f32 primitives the Selenite RX block path is
 * composed from, plus the chain composition of DESIGN.md.  Only tests/, __graft_entry__.smoke()
 * and bench.py's cpu_baseline leg may link or call this; the product (libselenite_rx.so) never
 * does.
 *
 * Parity status: the PRIMITIVES are pinned bit-exact against the real CMSIS-DSP sources compiled
 * from /root/reference (oracle/_ref/libcmsis_ref.so, tests/test_oracle_vs_ref.py, and the
 * committed fixtures under tests/golden/ generated from that library).  The CHAIN (which
 * primitives, in what order, with which taps and AGC law) does not exist in the reference
 * (SURVEY.md 0): it is build-defined, "parity unpinned" at chain level, and pinned only in the
 * sense that oracle/ref_chain.c composes the REAL CMSIS functions the same way and both agree
 * bit-for-bit.
 */
#ifndef RX_ORACLE_H
#define RX_ORACLE_H

#include <stdint.h>
#include "../include/selenite_rx.h"   /* config / state-view structs only */

#ifdef __cplusplus
extern "C" {
#endif

/* radians per NCO phase unit: x = (float)(phase >> 8) * ORC_NCO_K, 2*pi / 2^24 rounded to f32 */
#define ORC_NCO_K 0x1.921fb6p-22f

/* ---- primitives (arith: SELENITE_ARITH_CMSIS / SELENITE_ARITH_FMA) ---- */
const float *orc_sin_table(void);                      /* 513 entries */
float orc_sin_f32(float x, int arith);
float orc_cos_f32(float x, int arith);
void  orc_nco_lo(const uint32_t *phase, uint32_t n, float *lo);   /* lo[2n] = cos, lo[2n+1] = -sin of the chain's NCO */
void  orc_cmplx_mult_cmplx_f32(const float *a, const float *b, float *dst, uint32_t n, int arith);
void  orc_cmplx_mag_f32(const float *src, float *dst, uint32_t n, int arith);
void  orc_cmplx_conj_f32(const float *src, float *dst, uint32_t n);
float orc_fm_atan2_f32(float y, float x);             /* build-defined (fm_atan.h): CMSIS-DSP 1.5.3 has no arctangent */
void  orc_fir_decimate_f32(const float *coeffs, uint32_t num_taps, uint32_t M, float *state,
                           const float *src, float *dst, uint32_t block, int arith);
void  orc_fir_f32(const float *coeffs, uint32_t num_taps, float *state,
                  const float *src, float *dst, uint32_t block, int arith);
void  orc_biquad_cascade_df1_f32(const float *coeffs, uint32_t stages, float *state,
                                 const float *src, float *dst, uint32_t block, int arith);
void  orc_add_f32(const float *a, const float *b, float *dst, uint32_t n);
void  orc_sub_f32(const float *a, const float *b, float *dst, uint32_t n);
void  orc_abs_f32(const float *src, float *dst, uint32_t n);
void  orc_max_f32(const float *src, uint32_t n, float *result, uint32_t *index);
void  orc_scale_f32(const float *src, float scale, float *dst, uint32_t n);
void  orc_q15_to_float(const int16_t *src, float *dst, uint32_t n);
void  orc_float_to_q15(const float *src, int16_t *dst, uint32_t n);
void  orc_float_to_q15_rounding(const float *src, int16_t *dst, uint32_t n);   /* the ARM_MATH_ROUNDING build of the same function */

/* AGC gain law (build-defined): returns the new gain */
float orc_agc_update(const selenite_rx_config *cfg, float gain, float env, int arith);

/* ---- chain ---- */
typedef struct orc_rx orc_rx;
int   orc_rx_create(orc_rx **S, const selenite_rx_config *cfg);
void  orc_rx_destroy(orc_rx *S);
int   orc_rx_set_mode(orc_rx *S, uint8_t mode);
/* nthreads <= 1: scalar; > 1: channels split over pthreads (results identical) */
void  orc_rx_process_f32(orc_rx *S, const float *iq, float *audio, uint32_t block_size, int nthreads);
void  orc_rx_process_q15(orc_rx *S, const int16_t *iq, int16_t *audio, uint32_t block_size, int nthreads);
int   orc_rx_get_state(orc_rx *S, const selenite_rx_state_view *dst);
int   orc_rx_set_state(orc_rx *S, const selenite_rx_state_view *src);
/* global-gain variant pieces: envelope of one DSP block over all channels is taken inside
 * process when cfg.agc_global; for sharded tests the caller may inject the global envelopes:
 * env_override[block_size/cfg.block] (NULL = compute locally); env_out likewise receives them. */
void  orc_rx_process_f32_env(orc_rx *S, const float *iq, float *audio, uint32_t block_size,
                             const float *env_override, float *env_out);

/* ---- synthetic I/Q ---- */
void  orc_synth_iq(float *iq, uint32_t first_channel, uint32_t nch,
                   uint64_t first_sample, uint32_t nsamp, uint64_t seed);

#ifdef __cplusplus
}
#endif
#endif
