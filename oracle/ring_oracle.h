/* ring_oracle.h -- TEST ORACLE ONLY (never linked into the product): CPU restatement of the
 * reference's DSP ring buffer, Core/Src/dsp_if.c:83-340 + Core/Inc/dsp_if.h:81-94, for a batch of
 * independent rings.  PINNED against the reference's own code: oracle/Makefile compiles Core/Src/dsp_if.c
 * (+ Core/Src/main.c) from /root/reference into oracle/_ref/libdsp_if_ref.so behind the harness
 * oracle/ref_ring.c; tests/test_ring_oracle_vs_ref.py compares every returned word and the whole ring
 * state on the fixture trace and on random traffic, and tests/golden/ring_trace.npz is generated from the
 * reference (tests/golden/make_ring_golden.py).  tests/test_ring_oracle.py additionally holds traces derived
 * by hand from the source. */
#ifndef RING_ORACLE_H_
#define RING_ORACLE_H_
#include <stdint.h>

typedef struct {
    uint32_t channels, frames;      /* frames = DSP_BUFF_SIZE */
    int16_t *i, *q;                 /* [channels][frames] */
    uint8_t *buff_enable;           /* [channels] */
    uint16_t *rd_ptr, *wr_ptr;      /* [channels] */
} orc_ring;

orc_ring *orc_ring_new(uint32_t channels, uint32_t frames);
void orc_ring_free(orc_ring *r);
void orc_ring_in_write(orc_ring *r, const int16_t *src, uint16_t size_words);   /* DSP_In_Buff_Write  */
void orc_ring_in_read(orc_ring *r, int16_t *dst, uint32_t size_bytes);          /* DSP_In_Buff_Read   */
void orc_ring_out_write(orc_ring *r, const int16_t *src, uint32_t size_bytes);  /* DSP_Out_Buff_Write */
void orc_ring_out_read(orc_ring *r, int16_t *dst, uint16_t size_words);         /* DSP_Out_Buff_Read  */
void orc_ring_mute(orc_ring *r);                                                /* DSP_Out_Buff_Mute  */
/* raw array access for the tests */
int16_t *orc_ring_i(orc_ring *r);
int16_t *orc_ring_q(orc_ring *r);
uint8_t *orc_ring_enable(orc_ring *r);
uint16_t *orc_ring_rd(orc_ring *r);
uint16_t *orc_ring_wr(orc_ring *r);
#endif
