/* selenite_ring.h -- C-ABI of the batched DSP ring buffer.
 *
 * The reference keeps two instances of one ring type either side of the DSP slot:
 *     DSP_Buff_TypeDef dsp_in_buff, dsp_out_buff        (Core/Src/dsp_if.c:33-34,
 *                                                        Core/Inc/dsp_if.h:87-94)
 * each DSP_BUFF_SIZE = 768 I/Q frames (the firmware build: USBD_AUDIO_FREQ 96000; `frames` is a parameter here) of int16 in separate i[] / q[] arrays, a read and a write
 * pointer and a "primed" flag.  Writers nudge the write pointer by one frame when the gap to the
 * reader leaves the middle half of the ring (slip / repeat drift compensation between the USB and
 * the codec clocks, dsp_if.c:116-180 and :250-301); the first reader / writer of the opposite side
 * parks its pointer half a ring away (dsp_if.c:125-135, :316-326).
 *
 * This library holds `channels` such rings in HBM and moves all of them with one kernel launch
 * per call; one ring == one reference DSP_Buff_TypeDef, bit for bit (pointers, flag and contents).
 * Function <-> reference mapping (same argument units as the reference: `words` counts uint16
 * words, `bytes` counts bytes; a frame is 2 words = 4 bytes, I then Q):
 *
 *   selenite_ring_in_write   DSP_In_Buff_Write  (uint16_t *pbuf, uint16_t size /+words+/)   dsp_if.c:250-301
 *   selenite_ring_in_read    DSP_In_Buff_Read   (uint8_t  *pbuf, uint32_t size /+bytes+/)   dsp_if.c:310-340
 *   selenite_ring_out_write  DSP_Out_Buff_Write (uint8_t  *pbuf, uint32_t size /+bytes+/)   dsp_if.c:116-180
 *   selenite_ring_out_read   DSP_Out_Buff_Read  (uint16_t *pbuf, uint16_t size /+words+/)   dsp_if.c:204-219
 *   selenite_ring_mute       DSP_Out_Buff_Mute  (void)                                      dsp_if.c:188-195
 *
 * Batch layout: pbuf of channel c starts at element c * words of the int16 array (channel-major,
 * each channel's packet contiguous) -- the same layout selenite_rx_process_q15 consumes, so
 * ring -> RX chain -> ring needs no repacking.
 *
 * Preconditions the reference leaves to its caller, checked here (sticky status, call ignored):
 * size is a non-zero multiple of one frame (the reference reads pbuf[size-2]); frames <= 32767
 * (the reference's uint16_t gap arithmetic, dsp_if.c:252-264, must not wrap).
 *
 * Parity: pinned against the reference's own code.  Core/Src/dsp_if.c compiles in the build image from the
 * reference tree alone (oracle/Makefile -> oracle/_ref/libdsp_if_ref.so; firmware defines STM32F411xE,
 * USE_HAL_DRIVER); the test oracle (oracle/ring_oracle.c) equals it word for word on the fixture trace and on
 * random traffic (tests/test_ring_oracle_vs_ref.py), tests/golden/ring_trace.npz is generated from it, and the
 * HIP kernels are bit-exact against both (tests/test_gpu_ring.py).
 */
#ifndef SELENITE_RING_H_
#define SELENITE_RING_H_

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct selenite_ring selenite_ring;

#define SELENITE_RING_FRAMES_DEFAULT 768u   /* DSP_BUFF_SIZE of the firmware build: dsp_if.h:81-84 with USBD_AUDIO_FREQ 96000 (usbd_audio.h:46), 8 packets */

/* Host-side view of every ring's state (arrays owned by the caller). */
typedef struct {
    int16_t  *i;            /* [channels][frames]  DSP_Buff_TypeDef.i */
    int16_t  *q;            /* [channels][frames]  DSP_Buff_TypeDef.q */
    uint8_t  *buff_enable;  /* [channels] */
    uint16_t *rd_ptr;       /* [channels] */
    uint16_t *wr_ptr;       /* [channels] */
} selenite_ring_state_view;

/* Status codes are selenite_rx.h's: 0 success, -1 argument, -2 length, -7 device. */
int  selenite_ring_init(selenite_ring **R, uint32_t channels, uint32_t frames);
void selenite_ring_free(selenite_ring *R);
int  selenite_ring_status(const selenite_ring *R);            /* sticky; 0 = ok */
const char *selenite_ring_error_string(const selenite_ring *R);
int  selenite_ring_set_stream(selenite_ring *R, void *hip_stream);
int  selenite_ring_sync(selenite_ring *R);

/* Device-pointer entry points (data already in HBM; asynchronous on the ring's stream). */
void selenite_ring_in_write_device (selenite_ring *R, const int16_t *dSrc, uint16_t size_words);
void selenite_ring_in_read_device  (selenite_ring *R, int16_t *dDst, uint32_t size_bytes);
void selenite_ring_out_write_device(selenite_ring *R, const int16_t *dSrc, uint32_t size_bytes);
void selenite_ring_out_read_device (selenite_ring *R, int16_t *dDst, uint16_t size_words);
void selenite_ring_mute(selenite_ring *R);

/* Host-pointer variants (copy in, run, copy out, synchronise). */
void selenite_ring_in_write (selenite_ring *R, const int16_t *src, uint16_t size_words);
void selenite_ring_in_read  (selenite_ring *R, int16_t *dst, uint32_t size_bytes);
void selenite_ring_out_write(selenite_ring *R, const int16_t *src, uint32_t size_bytes);
void selenite_ring_out_read (selenite_ring *R, int16_t *dst, uint16_t size_words);

int  selenite_ring_get_state(selenite_ring *R, selenite_ring_state_view *view);
int  selenite_ring_set_state(selenite_ring *R, const selenite_ring_state_view *view);

/* Mean milliseconds per (in_write + in_read) pair over `iters` pairs, HIP events on the ring's
 * stream; for bench / roofline use. */
int  selenite_ring_time_device(selenite_ring *R, const int16_t *dSrc, int16_t *dDst, uint16_t size_words,
                               uint32_t iters, float *ms_per_pair);

#ifdef __cplusplus
}
#endif
#endif /* SELENITE_RING_H_ */
