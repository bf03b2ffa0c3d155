/* ring_oracle.c -- TEST ORACLE ONLY.  See ring_oracle.h (pinned against the reference's dsp_if.c).
 *
 * One `ring` below is one DSP_Buff_TypeDef of the reference (dsp_if.h:87-94); every function walks
 * the batch and applies the reference's per-call logic to each ring on its own.  All pointer
 * arithmetic is done in uint16_t like the reference's fields and its local `gap`. */
#include "ring_oracle.h"

#include <stdlib.h>
#include <string.h>

typedef struct {
    int16_t *i, *q;
    uint8_t *enable;
    uint16_t *rd, *wr;
    uint16_t n;             /* DSP_BUFF_SIZE */
} ring;

static ring ring_of(orc_ring *r, uint32_t c)
{
    ring g = { r->i + (size_t)c * r->frames, r->q + (size_t)c * r->frames, r->buff_enable + c,
               r->rd_ptr + c, r->wr_ptr + c, (uint16_t)r->frames };
    return g;
}

/* dsp_in_buff_write / dsp_out_buff_write: dsp_if.c:229-240 and :91-102 -- store, advance, wrap */
static void push(ring *g, int16_t vi, int16_t vq)
{
    g->i[*g->wr] = vi;
    g->q[*g->wr] = vq;
    (*g->wr)++;
    if (*g->wr == g->n) *g->wr = 0;
}

/* distance writer - reader in frames, modulo the ring: dsp_if.c:137-144 and :256-265 */
static uint16_t gap_of(const ring *g)
{
    uint16_t gap = *g->wr;
    if (*g->rd > *g->wr) gap += g->n;
    gap -= *g->rd;
    return gap;
}

/* the common tail of both writers: slip / repeat, copy, repeated last frame, step back.
 * dsp_if.c:146-179 and :267-300 */
static void write_frames(ring *g, uint16_t gap, const int16_t *buf, uint32_t size_words)
{
    if (gap > (3U * g->n / 4U)) {            /* writer runs ahead: drop one frame position */
        if (*g->wr < 1U) *g->wr += g->n;
        (*g->wr)--;
    }
    if (gap < (g->n / 4U)) {                 /* reader runs ahead: skip one frame position */
        (*g->wr)++;
        if (*g->wr >= g->n) *g->wr -= g->n;
    }
    for (uint32_t k = 0; k < size_words; k += 2U) push(g, buf[k], buf[k + 1]);
    push(g, buf[size_words - 2], buf[size_words - 1]);   /* filler for a forward-shifted pointer */
    if (*g->wr < 1U) *g->wr += g->n;
    (*g->wr)--;
}

void orc_ring_in_write(orc_ring *r, const int16_t *src, uint16_t size_words)
{
    for (uint32_t c = 0; c < r->channels; ++c) {
        ring g = ring_of(r, c);
        uint16_t gap = 0U;                   /* dsp_if.c:252: stays 0 until a reader primed the ring */
        if (*g.enable) gap = gap_of(&g);
        write_frames(&g, gap, src + (size_t)c * size_words, size_words);
    }
}

void orc_ring_out_write(orc_ring *r, const int16_t *src, uint32_t size_bytes)
{
    const uint32_t size_words = size_bytes / 2U;         /* dsp_if.c:120 */
    for (uint32_t c = 0; c < r->channels; ++c) {
        ring g = ring_of(r, c);
        if (*g.enable == 0U) {                           /* dsp_if.c:124-134: park half a ring ahead */
            *g.wr = *g.rd + g.n / 2U;
            if (*g.wr >= g.n) *g.wr -= g.n;
            *g.enable = 1U;
        }
        write_frames(&g, gap_of(&g), src + (size_t)c * size_words, size_words);
    }
}

/* dsp_if.c:206-217 and :328-339 -- fetch, advance, wrap to 0 */
static void pop_frames(ring *g, int16_t *buf, uint32_t size_words)
{
    for (uint32_t k = 0; k < size_words; k += 2U) {
        buf[k] = g->i[*g->rd];
        buf[k + 1] = g->q[*g->rd];
        (*g->rd)++;
        if (*g->rd >= g->n) *g->rd = 0U;
    }
}

void orc_ring_out_read(orc_ring *r, int16_t *dst, uint16_t size_words)
{
    for (uint32_t c = 0; c < r->channels; ++c) {
        ring g = ring_of(r, c);
        pop_frames(&g, dst + (size_t)c * size_words, size_words);
    }
}

void orc_ring_in_read(orc_ring *r, int16_t *dst, uint32_t size_bytes)
{
    const uint32_t size_words = size_bytes / 2U;         /* dsp_if.c:314 */
    for (uint32_t c = 0; c < r->channels; ++c) {
        ring g = ring_of(r, c);
        if (*g.enable == 0U) {                           /* dsp_if.c:316-326 */
            *g.rd = *g.wr + g.n / 2U;
            if (*g.rd >= g.n) *g.rd = 0U;                /* sic: resets to 0, does not wrap (:320-323) */
            *g.enable = 1U;
        }
        pop_frames(&g, dst + (size_t)c * size_words, size_words);
    }
}

void orc_ring_mute(orc_ring *r)                          /* dsp_if.c:188-195: contents only */
{
    memset(r->i, 0, sizeof(int16_t) * (size_t)r->channels * r->frames);
    memset(r->q, 0, sizeof(int16_t) * (size_t)r->channels * r->frames);
}

orc_ring *orc_ring_new(uint32_t channels, uint32_t frames)
{
    orc_ring *r = calloc(1, sizeof *r);
    r->channels = channels;
    r->frames = frames;
    r->i = calloc((size_t)channels * frames, sizeof(int16_t));
    r->q = calloc((size_t)channels * frames, sizeof(int16_t));
    r->buff_enable = calloc(channels, 1);
    r->rd_ptr = calloc(channels, sizeof(uint16_t));
    r->wr_ptr = calloc(channels, sizeof(uint16_t));
    return r;
}

void orc_ring_free(orc_ring *r)
{
    if (!r) return;
    free(r->i); free(r->q); free(r->buff_enable); free(r->rd_ptr); free(r->wr_ptr); free(r);
}

int16_t *orc_ring_i(orc_ring *r) { return r->i; }
int16_t *orc_ring_q(orc_ring *r) { return r->q; }
uint8_t *orc_ring_enable(orc_ring *r) { return r->buff_enable; }
uint16_t *orc_ring_rd(orc_ring *r) { return r->rd_ptr; }
uint16_t *orc_ring_wr(orc_ring *r) { return r->wr_ptr; }
