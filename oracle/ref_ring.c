/* ref_ring.c -- TEST ORACLE ONLY: thin harness around the reference's OWN ring-buffer code.
 *
 * oracle/Makefile compiles Core/Src/dsp_if.c (and Core/Src/main.c, which defines the HAL handles dsp_if.c names)
 * straight from /root/reference into oracle/_ref/libdsp_if_ref.so together with this file -- no reference source
 * is copied, no header is stubbed; the HAL functions those two files call are never reached by the functions
 * driven here and stay unresolved lazy symbols.  This file only moves the reference's two global rings
 * (dsp_in_buff / dsp_out_buff, Core/Src/dsp_if.c:32-33) in and out, so that a batch of independent rings can be
 * run one after another through the real DSP_In_Buff_Write / DSP_In_Buff_Read / DSP_Out_Buff_Write /
 * DSP_Out_Buff_Read / DSP_Out_Buff_Mute (Core/Src/dsp_if.c:116-340).  Used to pin oracle/ring_oracle.c and to
 * generate tests/golden/ring_trace.npz. */
#include <string.h>
#include "main.h"
#include "dsp_if.h"

extern DSP_Buff_TypeDef dsp_out_buff;
extern DSP_Buff_TypeDef dsp_in_buff;

unsigned ref_ring_frames(void) { return DSP_BUFF_SIZE; }
unsigned ref_ring_audio_freq(void) { return USBD_AUDIO_FREQ; }

static DSP_Buff_TypeDef *pick(int out) { return out ? &dsp_out_buff : &dsp_in_buff; }

void ref_ring_set(int out, const int16_t *i, const int16_t *q, uint8_t enable, uint16_t rd, uint16_t wr)
{
    DSP_Buff_TypeDef *b = pick(out);
    memcpy(b->i, i, sizeof b->i);
    memcpy(b->q, q, sizeof b->q);
    b->buff_enable = enable; b->rd_ptr = rd; b->wr_ptr = wr;
}

void ref_ring_get(int out, int16_t *i, int16_t *q, uint8_t *enable, uint16_t *rd, uint16_t *wr)
{
    DSP_Buff_TypeDef *b = pick(out);
    memcpy(i, b->i, sizeof b->i);
    memcpy(q, b->q, sizeof b->q);
    *enable = b->buff_enable; *rd = b->rd_ptr; *wr = b->wr_ptr;
}

void ref_ring_in_write(const int16_t *src, uint16_t size_words) { DSP_In_Buff_Write((uint16_t *)src, size_words); }
void ref_ring_in_read(int16_t *dst, uint32_t size_bytes) { DSP_In_Buff_Read((uint8_t *)dst, size_bytes); }
void ref_ring_out_write(const int16_t *src, uint32_t size_bytes) { DSP_Out_Buff_Write((uint8_t *)src, size_bytes); }
void ref_ring_out_read(int16_t *dst, uint16_t size_words) { DSP_Out_Buff_Read((uint16_t *)dst, size_words); }
void ref_ring_out_mute(void) { DSP_Out_Buff_Mute(); }
