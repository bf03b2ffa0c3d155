/* tx_oracle.h -- TEST ORACLE ONLY: see tx_oracle.c. */
#ifndef TX_ORACLE_H_
#define TX_ORACLE_H_
#include <stdint.h>
#include "../include/selenite_tx.h"

typedef struct orc_tx orc_tx;
void orc_fir_interpolate_f32(const float *coeffs, uint32_t num_taps, uint32_t L, float *state,
                             const float *src, float *dst, uint32_t block, int arith);
void orc_negate_f32(const float *src, float *dst, uint32_t n);
void orc_offset_f32(const float *src, float offset, float *dst, uint32_t n);
int  orc_tx_create(orc_tx **S, const selenite_tx_config *cfg);
void orc_tx_destroy(orc_tx *S);
int  orc_tx_set_mode(orc_tx *S, uint8_t mode);
void orc_tx_process_f32(orc_tx *S, const float *audio, float *iq, uint32_t block_size);
void orc_tx_process_q15(orc_tx *S, const int16_t *audio, int16_t *iq, uint32_t block_size);
int  orc_tx_get_state(orc_tx *S, const selenite_tx_state_view *dst);
#endif
